R=r02
OUT=$PWD/gpurun_out/sqv
mkdir -p $OUT
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/sq_verify65" -o p -- python3 "$OLDPWD/bench.py" --workload verify65 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/sq_verify65.log" 2>&1 )
python tools/pmc_summary.py sq verify65 "$OUT/sq_verify65" $R | cut -c1-600
( cd /tmp && rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d "$OUT/sq2_verify65" -o p -- python3 "$OLDPWD/bench.py" --workload verify65 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/sq2_verify65.log" 2>&1 )
python tools/pmc_summary.py sq verify65b "$OUT/sq2_verify65" $R | cut -c1-600
tail -3 "$OUT/sq2_verify65.log"
cp profiles/r02_sq_verify65*.json $OUT/
rm -rf $OUT/sq_verify65 $OUT/sq2_verify65
