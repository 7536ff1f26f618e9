#!/bin/bash
export MLDSA_TUNING_ENV=1
for rep in 1 2; do for po in 65536 32768 16384 8192 4096; do
  echo -n "MLDSA_PASS_OPS=$po: "
  MLDSA_PASS_OPS=$po python bench.py --workload verify65 --no-extras --no-pmc --no-cpu-baseline --extras-file x_extras.json 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); x=json.load(open('x_extras.json'))['headline']; print(round(j['value']/1e6,2), j['ms_per_step'], x['stage_ms_per_step'])"
done; done
