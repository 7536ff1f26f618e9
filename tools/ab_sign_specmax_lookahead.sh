#!/bin/bash
# sign65 (65 536 ops): MLDSA_SPEC_MAX (most speculative candidates per op and round) x MLDSA_LOOKAHEAD (two candidates generated at once in the
# early rounds) around the defaults (32 / 1), two passes, one box: M signs/s, ms per step, candidates per signature.
export MLDSA_TUNING_ENV=1
for rep in 1 2; do for sm in 16 32 64; do for la in 0 1 2; do
  echo -n "rep $rep SPEC_MAX=$sm LOOKAHEAD=$la: "
  MLDSA_SPEC_MAX=$sm MLDSA_LOOKAHEAD=$la python bench.py --workload sign65 --no-extras --no-pmc --no-cpu-baseline --steps 40 --warmup 3 --extras-file x_extras.json 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); x=json.load(open('x_extras.json'))['headline']; print(round(j['value']/1e6,3), round(j['ms_per_step'],3), round(x.get('sign_iterations_per_signature',0),3))"
done; done; done
