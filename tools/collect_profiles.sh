#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root, AFTER the last change under fips204_amd/csrc):
#
#     make -C fips204_amd/csrc -j8 all nolatearg variants VARIANTS="258 1 8 512"    (the variants: for the memory-path A/B below)
#     GIT_HEAD=$(git rev-parse HEAD) gpurun -- 'GIT_HEAD=... tools/collect_profiles.sh r06'
#
# bench lines (+ their side files), rocprofv3 kernel stats, HBM-traffic and SQ PMC passes (separate --pmc runs with --kernel-trace only,
# the program itself directly after `--`), the small-call kernel timelines, the micro-benchmarks.  Everything lands in
# gpurun_out/final_$R/ together with MANIFEST.jsonl: one line per file with the command that made it; tools/finish_profiles.py (run
# back in the container) copies the summaries into profiles/ and writes profiles/${R}_MANIFEST.json (command, HEAD, sha256 of the
# library, hash of the sources: tests/test_profiles_manifest_cpu.py).
set -u
R=${1:-r06}
OUT=$PWD/gpurun_out/final_$R
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
ERR="$OUT/stderr.log"
MAN="$OUT/MANIFEST.jsonl"
python3 - > "$OUT/BUILD.json" <<PY
import json, sys
sys.path.insert(0, "tools")
import csrc_hash, subprocess
ver = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.strip()
print(json.dumps({"head": "${GIT_HEAD:-unknown}", "lib_sha256": csrc_hash.lib_sha256(), "csrc_hash": csrc_hash.csrc_hash(), "hipcc_version": ver}))
PY
# note FILE "COMMAND": record what made FILE
note() { python3 -c 'import json,sys; print(json.dumps({"file": sys.argv[1], "command": sys.argv[2]}))' "$1" "$2" >> "$MAN"; }
# line FILE COMMAND...: run COMMAND, keep the last JSON line of its stdout as FILE
line() { f=$1; shift; "$@" 2>> "$ERR" | grep "^{" | tail -1 > "$OUT/$f"; note "$f" "$*"; }

# ---- the GPU suite on this box, on this library
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -2 > "$OUT/gpu_suite.txt"; note gpu_suite.txt "python -m pytest tests -m gpu -q  (summary line)"
# ---- bench lines
line bench_default.json python bench.py
cp bench_extras.json "$OUT/bench_default_extras.json" 2>/dev/null && note bench_default_extras.json "python bench.py  (the side file named in the line)"
line bench_full.json python bench.py --full
cp bench_extras.json "$OUT/bench_full_extras.json" 2>/dev/null && note bench_full_extras.json "python bench.py --full  (side file: corrupted / wire variants, host-fed legs, sweep)"
line bench_sign65.json python bench.py --workload sign65 --no-extras
line bench_verify_arith44.json python bench.py --workload verify_arith44 --steps 500 --warmup 20 --pmc
: > "$OUT/bench_other_workloads.jsonl"
for w in verify44 verify87 sign44 sign87 keygen44 keygen65 keygen87 mixed ntt inv_ntt mat_vec_mul65 expand_a65 expand_mask65 verify65_cached_a sign65_cached_a verify65_wire sign65_wire verify65_corrupt1; do
  python bench.py --workload $w --no-extras --no-pmc --extras-file "" $( case $w in verify44|verify87|sign44|sign87|keygen65|mixed) ;; *) echo --no-cpu-baseline ;; esac ) 2>> "$ERR" | grep "^{" | tail -1 >> "$OUT/bench_other_workloads.jsonl"
done
note bench_other_workloads.jsonl "for w in verify44 ... verify65_corrupt1: python bench.py --workload \$w --no-extras --no-pmc [--no-cpu-baseline]"
line bench_config4_slice.json python bench.py --workload verify87 --batch 131072 --no-cpu-baseline --steps 10 --no-extras --no-pmc --extras-file ""
line bench_mixed_65536.json python bench.py --workload mixed --batch 65536 --no-cpu-baseline --steps 5 --warmup 2 --no-extras --no-pmc --extras-file ""
line sweep_batch_sizes.json python bench.py --workload sweep --extras-file sweep_extras.json
cp sweep_extras.json "$OUT/sweep_batch_sizes_extras.json" 2>/dev/null && note sweep_batch_sizes_extras.json "python bench.py --workload sweep  (side file: every point, CPU and reference-published crossovers, single-op callers)"
# ---- the multi-rank and in-library group paths on this 1-GPU box (functional)
line bench_gpus2_gloo_shared_gpu.json python bench.py --gpus 2 --backend gloo --workload verify87 --batch 32768 --no-cpu-baseline --steps 5 --no-pmc --extras-file ""
MLDSA_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python bench.py --gpus 1 --workload verify87 --batch 32768 --no-cpu-baseline --steps 5 --no-pmc --no-extras --extras-file "" 2>> "$ERR" | grep "^{" | tail -1 > "$OUT/bench_rccl_world1.json"
note bench_rccl_world1.json "MLDSA_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 python bench.py --gpus 1 --workload verify87 --batch 32768 (RCCL code path, world of one)"
line inproc_resident_verify87_8ctx_on_1gpu_functional.json python bench.py --inproc --resident --gpus 8 --workload verify87 --batch 131072 --steps 5 --warmup 2
line inproc_group2_verify65_hostfed.json python bench.py --inproc --gpus 2 --workload verify65 --batch 32768 --steps 5 --warmup 1
python tools/config4_full.py "$OUT/config4_full_2pow20_8ctx_on_1gpu_functional.json" > /dev/null 2>> "$ERR"; note config4_full_2pow20_8ctx_on_1gpu_functional.json "python tools/config4_full.py"
python tools/scale_preflight.py 2>> "$ERR" | tail -1 > "$OUT/scale_preflight_1gpu.json"; note scale_preflight_1gpu.json "python tools/scale_preflight.py"
# ---- micro-benchmarks
./tools/ubench_keccak_coop > "$OUT/ubench_keccak_coop.txt" 2>&1; note ubench_keccak_coop.txt "./tools/ubench_keccak_coop"
# ---- rocprofv3 per-kernel summaries of the commands whose kernel times bench.py reports
for spec in "verify65:--no-extras" "sign65:--workload sign65 --no-extras --steps 30 --warmup 3" "default:"; do
  tag=${spec%%:*}; flags=${spec#*:}
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$tag" -o r -- python3 "$OLDPWD/bench.py" --no-cpu-baseline --no-pmc --extras-file "" $flags > "$OUT/prof_$tag.log" 2>&1 )
  find "$OUT/prof_$tag" -name "*kernel_stats.csv" -exec cp {} "$OUT/rocprofv3_kernel_stats_$tag.csv" \;
  rm -rf "$OUT/prof_$tag" "$OUT/prof_$tag.log"
  note rocprofv3_kernel_stats_$tag.csv "cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-pmc $flags"
done
# ---- HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes)
for w in verify65 verify_arith44 sign65; do
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_${w}_$c" -o p -- python3 "$OLDPWD/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-pmc --extras-file "" > /dev/null 2>&1 )
  done
  python tools/pmc_summary.py hbm $w "$OUT/pmc_${w}_FETCH_SIZE" "$OUT/pmc_${w}_WRITE_SIZE" $R >> "$ERR" 2>&1
  cp profiles/${R}_pmc_$w.json "$OUT/pmc_$w.json"; note pmc_$w.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace -- python3 bench.py --workload $w --steps 3 --no-extras --no-pmc; tools/pmc_summary.py hbm"
  rm -rf "$OUT"/pmc_${w}_FETCH_SIZE "$OUT"/pmc_${w}_WRITE_SIZE
done
# ---- SQ counters: the batch kernels, and the cooperative kernels of small calls (64-op calls: every launch is one of them)
sqpass() { # name, counters, program args...
  name=$1; ctrs=$2; shift 2
  ( cd /tmp && rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$OUT/sq_$name" -o p -- python3 "$@" > /dev/null 2>&1 )
  python tools/pmc_summary.py sq $name "$OUT/sq_$name" $R >> "$ERR" 2>&1
  cp profiles/${R}_sq_$name.json "$OUT/sq_$name.json"; note sq_$name.json "rocprofv3 --pmc $ctrs --kernel-trace -- python3 $*; tools/pmc_summary.py sq"
  rm -rf "$OUT/sq_$name"
}
sqpass expand_a65 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "$PWD/bench.py" --workload expand_a65 --steps 5 --warmup 1 --no-cpu-baseline --no-pmc --extras-file ""
sqpass sign65 "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAVES" "$PWD/bench.py" --workload sign65 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-pmc --extras-file ""
sqpass verify65 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "$PWD/bench.py" --workload verify65 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-pmc --extras-file ""
for op in verify sign keygen; do
  sqpass coop_${op}_n64 "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "$PWD/tools/latency_probe.py" $op 64 40
done
for op in verify sign keygen; do  # the same calls through the batch pipeline: the stand-alone cooperative kernels (k_expand_a_coop, k_shake256_2_coop, k_expand_s_coop, k_expand_mask_coop ...)
  MLDSA_TUNING_ENV=1 MLDSA_SMALL_FUSED=0 sqpass coop_${op}_pipeline_n64 "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "$PWD/tools/latency_probe.py" $op 64 40
done
# ---- small-call kernel timelines (one-op, 64-op, 1 024-op verify / sign / keygen; the batch pipeline's verify beside the single launch)
tools/collect_small_calls.sh $R > /dev/null 2>> "$ERR"
cp gpurun_out/small_$R/${R}_small_call_*.json gpurun_out/small_$R/${R}_small_call_kernel_stats_*.csv "$OUT"/ 2>/dev/null
for f in gpurun_out/small_$R/unprofiled_*.txt; do cat "$f"; done > "$OUT/small_call_unprofiled_wall.txt"
for f in "$OUT"/${R}_small_call_*; do b=$(basename "$f"); mv "$f" "$OUT/${b#${R}_}"; note "${b#${R}_}" "tools/collect_small_calls.sh $R  (rocprofv3 --kernel-trace --stats -- python3 tools/latency_probe.py <op> <n> 60; tools/small_call_timeline.py)"; done
note small_call_unprofiled_wall.txt "python3 tools/latency_probe.py <op> <n> 200 (no profiler): the wall time the timelines decompose"
# the batch pipeline the single-launch kernels replace (MLDSA_SMALL_FUSED=0): the same calls, timelines and kernel stats beside the others
MLDSA_TUNING_ENV=1 MLDSA_SMALL_FUSED=0 tools/collect_small_calls.sh ${R}pipe "verify sign keygen" "1 64" > /dev/null 2>> "$ERR"
for op in verify sign keygen; do for n in 1 64; do
  cp gpurun_out/small_${R}pipe/${R}pipe_small_call_timeline_${op}_n$n.json "$OUT/small_call_timeline_${op}_n${n}_batch_pipeline.json" 2>/dev/null
  note small_call_timeline_${op}_n${n}_batch_pipeline.json "MLDSA_SMALL_FUSED=0 tools/collect_small_calls.sh (the batch pipeline the single-launch kernels replace)"
  cp gpurun_out/small_${R}pipe/${R}pipe_small_call_kernel_stats_${op}_n$n.csv "$OUT/small_call_kernel_stats_${op}_n${n}_batch_pipeline.csv" 2>/dev/null
  note small_call_kernel_stats_${op}_n${n}_batch_pipeline.csv "MLDSA_SMALL_FUSED=0 tools/collect_small_calls.sh (rocprofv3 --kernel-trace --stats of the same calls through the batch pipeline)"
done; done
for f in gpurun_out/small_${R}pipe/unprofiled_*.txt; do cat "$f"; done > "$OUT/small_call_unprofiled_wall_batch_pipeline.txt"
note small_call_unprofiled_wall_batch_pipeline.txt "MLDSA_SMALL_FUSED=0 python3 tools/latency_probe.py <op> <n> 200"
./tools/batcher_bench_bin 65 1.5 0 1,8,64 1 > "$OUT/batcher_single_op_callers.json" 2>> "$ERR"; note batcher_single_op_callers.json "./tools/batcher_bench_bin 65 1.5 0 1,8,64 1"
# ---- same-box A/Bs behind the small-call defaults (AB=0 skips them)
if [ "${AB:-1}" != 0 ]; then
  # round 6: the cache policy of the streaming kernels (field.h MLDSA_EXP; build/variants/*.so built from THESE sources) and the counters behind it
  if [ -f build/variants/libmldsa_hip_exp258.so ]; then
    python tools/ab_variants.py --reps 3 --variants base,258,1,8 --out gpurun_out/ab_variants_final.json > "$OUT/ab_memory_path.txt" 2>> "$ERR"
    note ab_memory_path.txt "python tools/ab_variants.py --reps 3 --variants base,258,1,8  (base = the shipped library: nt loads of the verify side's read-once rows; 258 = that policy off; 1 = nt loads of the signer's A_hat; 8 = the signer's A_hat rows by LDS-DMA)"
    python tools/ab_variants.py --reps 3 --variants base,512 --workloads keygen65,keygen44,keygen87 --out gpurun_out/ab_variants_keygen.json > "$OUT/ab_memory_path_keygen.txt" 2>> "$ERR"
    note ab_memory_path_keygen.txt "python tools/ab_variants.py --reps 3 --variants base,512 --workloads keygen65,keygen44,keygen87  (512 = key generation's A_hat rows on the default cache policy instead of nontemporal)"
    tools/ab_variant_counters.sh $R "base 258 1 8" "verify65 verify_arith44 sign65" > /dev/null 2>> "$ERR"
    cp gpurun_out/variant_counters/${R}_variant_counters_summary.txt "$OUT/ab_memory_path_counters.txt" 2>/dev/null
    note ab_memory_path_counters.txt "tools/ab_variant_counters.sh $R 'base 258 1 8' 'verify65 verify_arith44 sign65'  (rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum / SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAVES, separate passes, mean per launch of k_verify_arith / k_verify_main)"
  fi
  SIZES="1 8 32 64 128 256" tools/ab_small_back.sh > "$OUT/ab_small_sign_back.txt" 2>> "$ERR"; note ab_small_sign_back.txt "tools/ab_small_back.sh  (MLDSA_SMALL_SIGN_BACK = 1, the default: the second half of a small signing round as one launch, against 0: k_sign_tail + k_resolve + k_compact_small)"
  tools/ab_small_limits.sh > "$OUT/ab_small_limits_per_set.txt" 2>> "$ERR"; note ab_small_limits_per_set.txt "tools/ab_small_limits.sh  (per parameter set: single-launch kernels forced up to 1 024 ops against the batch pipeline, verify / keygen / sign calls of 96 ... 512 ops: where the crossovers are)"
  tools/ab_small_sign.sh > "$OUT/ab_small_sign_switches.txt" 2>> "$ERR"; note ab_small_sign_switches.txt "tools/ab_small_sign.sh  (default against MLDSA_SMALL_SIGN_SPEC=0 / 1, MLDSA_SMALL_SIGN_FRONT=0, MLDSA_SMALL_FUSED=0: wall time per signing call, ML-DSA-44 / 65 / 87, 1 ... 256 ops)"
fi
# ---- the host's side of a one-op signing call (HIP API calls beside the kernels)
( cd /tmp && CALL_GAP_US=400 rocprofv3 --hip-trace --kernel-trace --output-format csv -d "$OUT/hat" -o t -- python3 "$OLDPWD/tools/latency_probe.py" sign 1 40 > /dev/null 2>&1 )
python3 tools/host_api_timeline.py "$OUT/hat" > "$OUT/host_api_sign_n1.txt" 2>> "$ERR"; rm -rf "$OUT/hat"
note host_api_sign_n1.txt "cd /tmp && CALL_GAP_US=400 rocprofv3 --hip-trace --kernel-trace --output-format csv -- python3 tools/latency_probe.py sign 1 40; python3 tools/host_api_timeline.py"
# ---- soaks (SOAK=0 skips them): the seeded randomised soak of tests/test_gpu_sign_schedule.py, once kept on the small-call kernels, once over every size
if [ "${SOAK:-1}" != 0 ]; then
  { MLDSA_SOAK_SECONDS=${SOAK_SMALL_S:-240} MLDSA_SOAK_SEED=77 MLDSA_SOAK_MAX_N=400 python -m pytest tests/test_gpu_sign_schedule.py -m gpu -k soak -s -q 2>&1 | grep -E "^soak:|passed|failed|Error" ;
    MLDSA_SOAK_SECONDS=${SOAK_LONG_S:-420} MLDSA_SOAK_SEED=5151 python -m pytest tests/test_gpu_sign_schedule.py -m gpu -k soak -s -q 2>&1 | grep -E "^soak:|passed|failed|Error" ; } > "$OUT/soak_long.txt"
  note soak_long.txt "MLDSA_SOAK_SECONDS=${SOAK_SMALL_S:-240} MLDSA_SOAK_SEED=77 MLDSA_SOAK_MAX_N=400 python -m pytest tests/test_gpu_sign_schedule.py -m gpu -k soak -s  (small calls only: every result of the keygen -> sign -> verify -> flip -> verify iterations against the oracle, ~80 iterations per second);  MLDSA_SOAK_SECONDS=${SOAK_LONG_S:-420} MLDSA_SOAK_SEED=5151 ... (sizes 1 .. 70 000)"
fi
ls -la "$OUT"
