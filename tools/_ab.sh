OUT=$PWD/gpurun_out/r3ah; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "not soak" 2>&1 | grep -E "passed|failed|rror|assert" | tail -n 5 > $OUT/test.txt
for w in verify65 sign65 keygen65 verify65 verify44 verify87; do python bench.py --workload $w --no-extras --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d.get('stage_ms_per_step',{}) or {}; print('$w', round(d['value']/1e6,3), round(d['ms_per_step'],3), s.get('expand_a'))"; done > $OUT/bench.txt 2>&1
cat $OUT/test.txt $OUT/bench.txt
