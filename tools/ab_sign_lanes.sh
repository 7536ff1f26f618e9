#!/bin/bash
# One against two signing lanes (MLDSA_SIGN_LANES: a batch of >= 8 192 ops cut into two slices whose round chains run side by side on two
# streams), per parameter set and batch size, same box, interleaved: M signs/s, ms per step, candidates per signature.
export MLDSA_TUNING_ENV=1
for S in ${SETS:-65 44 87}; do for n in ${SIZES:-16384 32768 65536 131072}; do for rep in 1 2; do for l in ${LANES:-1 2}; do
  echo -n "sign$S n=$n rep $rep MLDSA_SIGN_LANES=$l: "
  MLDSA_SIGN_LANES=$l python bench.py --workload sign$S --batch $n --no-extras --no-pmc --no-cpu-baseline --steps ${STEPS:-30} --warmup 3 --extras-file x_extras.json 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); x=json.load(open('x_extras.json'))['headline']; print(round(j['value']/1e6,3), round(j['ms_per_step'],3), round(x.get('sign_iterations_per_signature',0),3))"
done; done; done; done
