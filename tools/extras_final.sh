#!/bin/bash
# The later measurements on the round's final library, in one call (each through tools/collect_extra.sh -> manifest extras).
R=${1:-r06}
tools/collect_extra.sh $R soak_more_seeds.txt "tools/soak_more.sh  (the seeded soak of tests/test_gpu_sign_schedule.py on two more seeds kept on the small-call kernels and one over every size, 240 + 240 + 300 s; the batcher soak of tests/test_gpu_batcher.py for 120 s)" tools/soak_more.sh > /dev/null
tools/collect_extra.sh $R two_lane_stress.txt "MLDSA_LANES_CALLS=4000 python -m pytest tests/test_gpu_sign_schedule.py -m gpu -k two_signing_lanes -q  (4 000 two-lane ML-DSA-65 signing calls of 8 200 ops: every signature of every call equal to the first call's, which is checked against the oracle)" bash -c "MLDSA_LANES_CALLS=4000 python -m pytest tests/test_gpu_sign_schedule.py -m gpu -k two_signing_lanes -q 2>&1 | grep -E 'passed|failed|error'" > /dev/null
tools/collect_extra.sh $R ab_sign_lanes_default_large.txt "SETS='65 44 87' SIZES='131072 262144' LANES='0 1' tools/ab_sign_lanes.sh  (the shipped default -- two lanes at these sizes -- against one lane)" env SETS="65 44 87" SIZES="131072 262144" LANES="0 1" STEPS=20 tools/ab_sign_lanes.sh > /dev/null
tools/collect_extra.sh $R bench_driver_shape_steps20.json "python bench.py --steps 20 --warmup 5  (the driver's K)" bash -c "python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1" > /dev/null
cat gpurun_out/extra_$R/MANIFEST.jsonl | cut -c1-80
