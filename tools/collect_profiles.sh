#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   bench JSON lines, rocprofv3 kernel stats, HBM-traffic and SQ PMC passes (separate --pmc runs with --kernel-trace only,
#   the program itself directly after `--`).  Everything lands in gpurun_out/final_$R/ and, summarised, in profiles/.
set -u
R=${1:-r04}
OUT=$PWD/gpurun_out/final_$R
mkdir -p "$OUT"
export TMPDIR=/tmp
ERR="$OUT/bench.err"
python bench.py > "$OUT/bench_default.json" 2> "$ERR"
python bench.py --workload sign65 --no-extras > "$OUT/bench_sign65.json" 2>> "$ERR"
python bench.py --workload verify_arith44 --steps 500 --warmup 20 > "$OUT/bench_verify_arith44.json" 2>> "$ERR"
: > "$OUT/bench_other_workloads.jsonl"
for w in verify44 verify87 sign44 sign87 keygen44 keygen65 keygen87 mixed ntt inv_ntt mat_vec_mul65 expand_a65 expand_mask65 verify44_cached_a verify65_cached_a verify87_cached_a sign44_cached_a sign65_cached_a sign87_cached_a; do
  python bench.py --workload $w $( case $w in verify44|verify87|sign44|sign87|keygen65|mixed) ;; *) echo --no-cpu-baseline ;; esac ) 2>> "$ERR" | grep "^{" | tail -1 >> "$OUT/bench_other_workloads.jsonl"
done
# BASELINE config 4's per-GPU slice (131072 ML-DSA-87 verifies = two pipeline chunks) and config 5 at a larger step
python bench.py --workload verify87 --batch 131072 --no-cpu-baseline --steps 10 2>> "$ERR" | grep "^{" | tail -1 > "$OUT/bench_config4_slice.json"
python bench.py --workload mixed --batch 65536 --no-cpu-baseline --steps 5 --warmup 2 2>> "$ERR" | grep "^{" | tail -1 > "$OUT/bench_mixed_65536.json"
# the multi-rank launch path on this 1-GPU box: two ranks started by bench.py itself, sharing GPU 0, gloo rendezvous
python bench.py --gpus 2 --backend gloo --workload verify87 --batch 32768 --no-cpu-baseline --steps 5 2>> "$ERR" | grep "^{" | tail -1 > "$OUT/bench_gpus2_gloo_shared_gpu.json"
# and the RCCL code path with a world of one (torchrun-style environment)
MLDSA_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python bench.py --gpus 1 --workload verify87 --batch 32768 --no-cpu-baseline --steps 5 2>> "$ERR" | grep "^{" | tail -1 > "$OUT/bench_rccl_world1.json"
# the C ABI's in-library multi-GPU path (mldsa_group_*): two contexts on this box's one GPU, host-fed
python bench.py --inproc --gpus 2 --workload verify65 --batch 32768 --steps 5 --warmup 1 2>> "$ERR" | grep "^{" | tail -1 > "$OUT/bench_inproc_group2_verify65.json"
python bench.py --inproc --gpus 2 --workload sign65 --batch 32768 --steps 3 --warmup 1 2>> "$ERR" | grep "^{" | tail -1 > "$OUT/bench_inproc_group2_sign65.json"
# round 4: batch-size sweep, config 4 at full size over eight contexts, the device-resident in-process group, eight gloo ranks on the one GPU
python bench.py --workload sweep 2>> "$ERR" | grep "^{" | tail -1 > "$OUT/sweep_batch_sizes.json"
python tools/config4_full.py "$OUT/config4_full_2pow20_8ctx_on_1gpu_functional.json" > /dev/null 2>> "$ERR"
python bench.py --inproc --resident --gpus 8 --workload verify87 --batch 131072 --steps 5 --warmup 2 2>> "$ERR" | grep "^{" | tail -1 > "$OUT/inproc_resident_verify87_8ctx_on_1gpu_functional.json"
python bench.py --inproc --resident --gpus 4 --workload sign65 --batch 16384 --steps 5 --warmup 2 2>> "$ERR" | grep "^{" | tail -1 > "$OUT/inproc_resident_sign65_4ctx_on_1gpu_functional.json"
python bench.py --gpus 8 --backend gloo --workload verify87 --batch 131072 --steps 5 --warmup 2 --no-cpu-baseline 2>> "$ERR" | grep "^{" | tail -1 > "$OUT/gloo8ranks_verify87_on_1gpu_functional.json"
for w in verify65_wire sign65_wire verify65_corrupt1 verify44_wire verify87_wire; do
  python bench.py --workload $w --no-extras 2>> "$ERR" | grep "^{" | tail -1 >> "$OUT/bench_wire_and_corrupt.jsonl"
done
./tools/ubench_graph 90 > "$OUT/ubench_graph.txt" 2>&1
./tools/ubench_d2h2 > "$OUT/ubench_d2h2.txt" 2>&1
./tools/ubench_keccak_coop > "$OUT/ubench_keccak_coop.txt" 2>&1
python tools/ubench_overlap2.py > "$OUT/ubench_overlap2.txt" 2>&1
for n in 32768 65536 131072; do python tools/hostfed_sign.py $n 4 2>&1 | grep sign_host; MLDSA_HOST_DIRECT=0 python tools/hostfed_sign.py $n 4 2>&1 | grep sign_host; done > "$OUT/hostfed_sign.txt"
# rocprofv3 per-kernel summaries of the commands whose kernel times bench.py reports: the headline workload on its own (every
# k_expand_a / k_verify_main launch is a 65536-op launch: the averages must agree with roofline.kernel_ms), sign65 on its
# own, and the whole default run (which also contains the smaller launches of the host-fed passes)
for spec in "verify65:--no-extras" "sign65:--workload sign65 --no-extras --steps 30 --warmup 3" "default:"; do
  tag=${spec%%:*}; flags=${spec#*:}
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$tag" -o r -- python3 "$OLDPWD/bench.py" --no-cpu-baseline --no-pmc $flags > "$OUT/prof_$tag.log" 2>&1 )
  find "$OUT/prof_$tag" -name "*kernel_stats.csv" -exec cp {} "$OUT/rocprofv3_kernel_stats_$tag.csv" \;
  rm -rf "$OUT/prof_$tag"
done
for w in verify65 verify_arith44 sign65; do
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_${w}_$c" -o p -- python3 "$OLDPWD/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-pmc > "$OUT/pmc_${w}_$c.log" 2>&1 )
  done
  python tools/pmc_summary.py hbm $w "$OUT/pmc_${w}_FETCH_SIZE" "$OUT/pmc_${w}_WRITE_SIZE" $R
done
# is ExpandA VALU-issue-bound?  SQ counters of the kernel on its own (one pass: 5 of the 8 SQ slots)
( cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/sq_expand_a65" -o p -- python3 "$OLDPWD/bench.py" --workload expand_a65 --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > "$OUT/sq_expand_a65.log" 2>&1 )
python tools/pmc_summary.py sq expand_a65 "$OUT/sq_expand_a65" $R
( cd /tmp && rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d "$OUT/sq_sign65" -o p -- python3 "$OLDPWD/bench.py" --workload sign65 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-pmc > "$OUT/sq_sign65.log" 2>&1 )
python tools/pmc_summary.py sq sign65 "$OUT/sq_sign65" $R
# the verify65 kernels (k_verify_main: how much of it is VALU issue?), two passes: issue counters, instruction classes
( cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/sq_verify65" -o p -- python3 "$OLDPWD/bench.py" --workload verify65 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-pmc > "$OUT/sq_verify65.log" 2>&1 )
python tools/pmc_summary.py sq verify65 "$OUT/sq_verify65" $R
( cd /tmp && rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d "$OUT/sq2_verify65" -o p -- python3 "$OLDPWD/bench.py" --workload verify65 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-pmc > "$OUT/sq2_verify65.log" 2>&1 )
python tools/pmc_summary.py sq verify65b "$OUT/sq2_verify65" $R
rm -rf "$OUT"/sq_verify65 "$OUT"/sq2_verify65
cp profiles/${R}_pmc_*.json profiles/${R}_sq_*.json "$OUT"/ 2>/dev/null
rm -rf "$OUT"/pmc_*_FETCH_SIZE "$OUT"/pmc_*_WRITE_SIZE "$OUT"/sq_expand_a65 "$OUT"/sq_sign65  # raw traces: large
ls -la "$OUT"
