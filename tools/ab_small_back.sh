#!/bin/bash
export MLDSA_TUNING_ENV=1  # the library reads its measurement knobs only when asked to (include/mldsa_hip.h "Environment")
# Same-box A/B of the fused second half of small signing rounds (k_sign_back_small, MLDSA_SMALL_SIGN_BACK = 1, default) against the three
# kernels it replaces (k_sign_tail + k_resolve + k_compact_small): wall time per signing call, ML-DSA-44 / 65 / 87, 1 ... 256 ops.
for S in ${SETS:-65 44 87}; do
  for n in ${SIZES:-1 8 32 64 128 256}; do
    for v in 1 0 1 0; do
      echo -n "ML-DSA-$S n=$n MLDSA_SMALL_SIGN_BACK=$v: "
      SET=$S MLDSA_SMALL_SIGN_BACK=$v python3 tools/latency_probe.py sign $n 150 2>/dev/null | tail -1
    done
  done
done
