#!/bin/bash
# one line per workload: value and the ExpandA stage (tools/ab_keygen.sh <tag>)
mkdir -p gpurun_out/r4
for w in keygen65 sign65 keygen87; do
  python bench.py --workload $w --steps 30 --warmup 3 --no-cpu-baseline --no-pmc > gpurun_out/r4/abk_$1_$w.json 2>/dev/null
  python - "$1" "$w" <<'PY'
import json, sys
tag, w = sys.argv[1], sys.argv[2]
d = json.loads(open(f"gpurun_out/r4/abk_{tag}_{w}.json").read().strip().splitlines()[-1])
print(tag, w, "%.3f M/s" % (d["value"] / 1e6), "ms/step %.3f" % d["ms_per_step"], "expand_a %s" % d.get("stage_ms_per_step", {}).get("expand_a"))
PY
done
