#!/bin/bash
# The bench lines that run live PMC passes, taken again on the SAME library after bench.py / benchlib changed (round 6: the PMC children now
# run before torch is loaded and the clock ramp enqueues deep bursts -- EXPERIMENTS.md).  Same file names as tools/collect_profiles.sh writes:
#     gpurun -- 'tools/refresh_bench_lines.sh r06';  python tools/finish_profiles.py r06 --extra
set -u
R=${1:-r06}
OUT=$PWD/gpurun_out/extra_$R
mkdir -p "$OUT"; : > "$OUT/MANIFEST.jsonl"
python3 - > "$OUT/BUILD.json" <<PY
import json, sys
sys.path.insert(0, "tools")
import csrc_hash
print(json.dumps({"lib_sha256": csrc_hash.lib_sha256(), "csrc_hash": csrc_hash.csrc_hash()}))
PY
note() { python3 -c 'import json,sys; print(json.dumps({"file": sys.argv[1], "command": sys.argv[2]}))' "$1" "$2" >> "$OUT/MANIFEST.jsonl"; }
line() { f=$1; shift; "$@" 2>> "$OUT/stderr.log" | grep "^{" | tail -1 > "$OUT/$f"; note "$f" "$*  (bench.py as of the round's last commit)"; }
line bench_default.json python bench.py
cp bench_extras.json "$OUT/bench_default_extras.json" && note bench_default_extras.json "python bench.py  (the side file named in the line; bench.py as of the round's last commit)"
line bench_full.json python bench.py --full
cp bench_extras.json "$OUT/bench_full_extras.json" && note bench_full_extras.json "python bench.py --full  (side file: corrupted / wire variants, host-fed legs, sweep; bench.py as of the round's last commit)"
line bench_verify_arith44.json python bench.py --workload verify_arith44 --steps 500 --warmup 20 --pmc
line bench_driver_shape_steps20.json python bench.py --steps 20 --warmup 5
ls -la "$OUT"
