#!/bin/bash
# Is the signer's speculation rule still at its optimum with this round's kernels?  sign65 (65 536 ops) over a grid of MLDSA_SPEC_TARGET x
# MLDSA_SPEC_ROWS around the defaults (65536 / 65536), two passes, one box: M signs/s, ms per step, candidates per signature.
export MLDSA_TUNING_ENV=1
for rep in 1 2; do for t in 49152 65536 81920; do for r in 49152 65536 81920; do
  echo -n "rep $rep SPEC_TARGET=$t SPEC_ROWS=$r: "
  MLDSA_SPEC_TARGET=$t MLDSA_SPEC_ROWS=$r python bench.py --workload sign65 --no-extras --no-pmc --no-cpu-baseline --steps 40 --warmup 3 --extras-file x_extras.json 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); x=json.load(open('x_extras.json'))['headline']; print(round(j['value']/1e6,3), round(j['ms_per_step'],3), round(x.get('sign_iterations_per_signature',0),3))"
done; done; done
