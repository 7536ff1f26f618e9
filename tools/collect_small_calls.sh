#!/bin/bash
# Kernel timelines of small calls: one-op, 64-op and 1 024-op ML-DSA-65 verify / sign / keygen calls under
# rocprofv3 --kernel-trace --stats (the program itself after `--`), decomposed by tools/small_call_timeline.py into kernel time on the
# device, launch gaps and host + sync time.  usage: tools/collect_small_calls.sh <round> [ops] [sizes]   -> profiles/<round>_small_call_*.json
set -u
R=${1:-r06}
OPS=${2:-"verify sign keygen"}
SIZES=${3:-"1 64 1024"}
OUT=$PWD/gpurun_out/small_$R
mkdir -p "$OUT"
export TMPDIR=/tmp
for op in $OPS; do
  for n in $SIZES; do
    tag=${op}_n$n
    ( cd /tmp && WALL_JSON="$OUT/wall_$tag.json" CALL_GAP_US=400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$tag" -o t -- python3 "$OLDPWD/tools/latency_probe.py" $op $n 60 > "$OUT/probe_$tag.log" 2>&1 )
    kt=$(find "$OUT/trace_$tag" -name "*kernel_trace.csv" | head -1)
    ks=$(find "$OUT/trace_$tag" -name "*kernel_stats.csv" | head -1)
    [ -n "$ks" ] && cp "$ks" "$OUT/${R}_small_call_kernel_stats_$tag.csv"
    [ -n "$kt" ] && python3 tools/small_call_timeline.py "$kt" "$OUT/wall_$tag.json" > "$OUT/${R}_small_call_timeline_$tag.json"
    # the same call without the profiler: the wall time the decomposition has to explain
    python3 tools/latency_probe.py $op $n 200 2>/dev/null | tail -1 > "$OUT/unprofiled_$tag.txt"
    rm -rf "$OUT/trace_$tag"
  done
done
ls -la "$OUT"
