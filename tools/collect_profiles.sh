#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   bench JSON lines, rocprofv3 kernel stats, and HBM-traffic PMC passes (separate --pmc runs, no other trace domains).
# Everything lands in gpurun_out/final/; copy what is to be judged into profiles/.
set -u
OUT=$PWD/gpurun_out/final
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python bench.py --workload sign65 > "$OUT/bench_sign65.json" 2>> "$OUT/bench_default.err"
python bench.py --workload verify_arith44 --steps 200 --warmup 10 > "$OUT/bench_verify_arith44.json" 2>> "$OUT/bench_default.err"
for w in verify44 verify87 sign44 sign87 keygen44 keygen65 keygen87 ntt inv_ntt mat_vec_mul65 expand_a65 expand_mask65 verify44_cached_a verify65_cached_a verify87_cached_a sign44_cached_a sign65_cached_a sign87_cached_a mixed; do
  python bench.py --workload $w --no-cpu-baseline 2>> "$OUT/bench_default.err" | tail -1 > "$OUT/bench_$w.json"
done
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_default" -o r -- python3 "$OLDPWD/bench.py" --no-cpu-baseline > "$OUT/prof_default.log" 2>&1 )
for wl in verify65:v65 verify_arith44:c2 sign65:s65; do
  w=${wl%%:*}; tag=${wl##*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_${tag}_$c" -o p -- python3 "$OLDPWD/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/pmc_${tag}_$c.log" 2>&1 )
  done
done
ls -la "$OUT"
