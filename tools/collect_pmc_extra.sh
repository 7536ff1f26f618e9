#!/bin/bash
# Round 4: L2 (TCC) hit / miss and LDS bank-conflict counters of the verify and sign kernels (separate --pmc passes with
# --kernel-trace only, the program itself after `--`).  Summaries -> gpurun_out/final_$R/ (copy the *.json you want judged to profiles/).
set -u
R=${1:-r04}
OUT=$PWD/gpurun_out/final_$R
mkdir -p "$OUT"
export TMPDIR=/tmp
for w in verify65 sign65; do
  steps=3; [ $w = sign65 ] && steps=2
  ( cd /tmp && rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d "$OUT/tcc_$w" -o p -- python3 "$OLDPWD/bench.py" --workload $w --steps $steps --warmup 1 --no-cpu-baseline --no-extras --no-pmc > "$OUT/tcc_$w.log" 2>&1 )
  python tools/pmc_summary.py sq tcc_$w "$OUT/tcc_$w" $R
  ( cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$OUT/lds_$w" -o p -- python3 "$OLDPWD/bench.py" --workload $w --steps $steps --warmup 1 --no-cpu-baseline --no-extras --no-pmc > "$OUT/lds_$w.log" 2>&1 )
  python tools/pmc_summary.py sq lds_$w "$OUT/lds_$w" $R
  cp profiles/${R}_sq_tcc_$w.json profiles/${R}_sq_lds_$w.json "$OUT"/ 2>/dev/null
  rm -rf "$OUT/tcc_$w" "$OUT/lds_$w"
done
ls -la "$OUT" | tail -8
