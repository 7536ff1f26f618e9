#!/bin/bash
# Which configuration of the default run starts its timed region slowly?  (first ten event marks of the headline workload; EXPERIMENTS.md round 6)
run() { tag=$1; shift; python bench.py "$@" > gpurun_out/mm_$tag.json 2>/dev/null; python - "$tag" <<'P'
import json, sys
x = json.load(open("bench_extras.json")); l = x["headline"]
print(sys.argv[1], "value", round(l["value"] / 1e6, 2), "ms", round(l["ms_per_step"], 4), "marks", l["step_ms_marks"][:10], "host_enqueue_ms", l.get("host_enqueue_ms"), flush=True)
P
}
run default_a
run nopmc_a --no-pmc
run default_b
run nopmc_b --no-pmc
run steps20 --steps 20 --warmup 5
