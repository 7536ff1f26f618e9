#!/bin/bash
# Two lanes at the headline size with the speculation target per LANE instead of per call: does halving MLDSA_SPEC_TARGET / MLDSA_SPEC_ROWS
# (each 32 768-op slice then speculates like a 65 536-op call does, candidates per signature back to ~6.4) make two lanes win at 65 536 ops?
export MLDSA_TUNING_ENV=1
for rep in 1 2; do
  for cfg in "1 65536 65536" "2 65536 65536" "2 32768 32768" "2 49152 49152" "2 32768 65536" "2 49152 32768"; do
    set -- $cfg
    echo -n "rep $rep sign65 65536 ops LANES=$1 SPEC_TARGET=$2 SPEC_ROWS=$3: "
    MLDSA_SIGN_LANES=$1 MLDSA_SPEC_TARGET=$2 MLDSA_SPEC_ROWS=$3 python bench.py --workload sign65 --no-extras --no-pmc --no-cpu-baseline --steps 40 --warmup 3 --extras-file x_extras.json 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); x=json.load(open('x_extras.json'))['headline']; print(round(j['value']/1e6,3), round(j['ms_per_step'],3), round(x.get('sign_iterations_per_signature',0),3))"
  done
done
