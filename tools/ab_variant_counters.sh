#!/bin/bash
# Round 6: the counters behind the memory-path A/Bs (tools/ab_variants.py gives the times).  For each library variant (`base` = the
# library in the tree, N = build/variants/libmldsa_hip_expN.so, field.h MLDSA_EXP) and workload: one L2 pass (TCC_HIT / TCC_MISS / TCC_REQ)
# and one SQ pass (SQ_WAIT_ANY, SQ_ACTIVE_INST_ANY, SQ_WAVE_CYCLES, SQ_INSTS_VALU, SQ_WAVES) -- separate --pmc runs with --kernel-trace
# only, python3 itself after `--`.  Summaries (tools/pmc_summary.py sq) -> gpurun_out/variant_counters/<round>_sq_{tcc,sq}_<workload>_<variant>.json
#     tools/ab_variant_counters.sh r06 "base 258 1 8" "verify65 verify_arith44 sign65"
set -u
R=${1:-r06}
VARS=${2:-"base 258 1 8"}
WLS=${3:-"verify65 verify_arith44 sign65"}
OUT=$PWD/gpurun_out/variant_counters
mkdir -p "$OUT"
export TMPDIR=/tmp
LIB=fips204_amd/csrc/libmldsa_hip.so
cp $LIB /tmp/keep_lib.so
for v in $VARS; do
  if [ $v = base ]; then cp /tmp/keep_lib.so $LIB; else cp build/variants/libmldsa_hip_exp$v.so $LIB || continue; fi
  for w in $WLS; do
    steps=3; [ $w = sign65 ] && steps=2; [ $w = verify_arith44 ] && steps=40
    for pass in "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "sq:SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAVES"; do
      tag=${pass%%:*}; ctrs=${pass#*:}
      name=${tag}_${w}_$v
      ( cd /tmp && rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$OUT/raw_$name" -o p -- python3 "$OLDPWD/bench.py" --workload $w --steps $steps --warmup 1 --no-cpu-baseline --no-extras --no-pmc --extras-file "" > "$OUT/$name.log" 2>&1 )
      python tools/pmc_summary.py sq $name "$OUT/raw_$name" $R > /dev/null 2>> "$OUT/errors.log"
      mv profiles/${R}_sq_$name.json "$OUT"/ 2>/dev/null
      rm -rf "$OUT/raw_$name" "$OUT/$name.log"
    done
  done
done
cp /tmp/keep_lib.so $LIB
python3 - "$OUT" "$R" <<'PY'
import glob, json, os, sys
out, rnd = sys.argv[1], sys.argv[2]
KEEP = ("k_verify_arith<", "k_verify_main<")
rows = []
for f in sorted(glob.glob(os.path.join(out, f"{rnd}_sq_*.json"))):
    d = json.load(open(f))
    for k, m in d["kernels"].items():
        if any(s in k for s in KEEP):
            short = k.split("(")[0].replace("void mldsa::", "")
            rows.append((d["name"], short, {a: (round(b, 4) if b < 100 else round(b)) for a, b in m.items()}))
with open(os.path.join(out, f"{rnd}_variant_counters_summary.txt"), "w") as fh:
    for name, k, m in rows:
        line = f"{name:<28} {k:<58} " + " ".join(f"{a}={b}" for a, b in m.items())
        print(line); fh.write(line + "\n")
PY
