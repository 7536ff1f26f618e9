run() { for w in sign65 sign44 sign87; do env "$@" python bench.py --workload $w --no-extras --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '$w', round(d['value']/1e6,3), round(d['ms_per_step'],3))"; done; }
run MLDSA_SPEC_MAX=32; run MLDSA_SPEC_MAX=63; run MLDSA_SPEC_MAX=48; run MLDSA_SPEC_MAX=32; run MLDSA_SPEC_MAX=63
