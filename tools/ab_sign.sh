#!/bin/bash
# same-box A/B helper: the signing workloads' value and the tail stages, one line each (tools/ab_sign.sh <tag>)
mkdir -p gpurun_out/r4
for w in sign65 sign44 sign87; do
  python bench.py --workload $w --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/r4/ab_$1_$w.json 2>/dev/null
  python - "$1" "$w" <<'PY'
import json, sys
tag, w = sys.argv[1], sys.argv[2]
d = json.loads(open(f"gpurun_out/r4/ab_{tag}_{w}.json").read().strip().splitlines()[-1])
st = d["stage_ms_per_step"]
print(tag, w, "%.3f M/s" % (d["value"] / 1e6), "ms/step %.3f" % d["ms_per_step"], "sign_tail %.4f resolve %.4f" % (st["sign_tail"], st["resolve"]),
      "profiled ms/step %.3f" % d["profiled_pass"]["ms_per_step"])
PY
done
