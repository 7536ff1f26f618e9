#!/bin/bash
export MLDSA_TUNING_ENV=1  # the library reads its measurement knobs only when asked to (include/mldsa_hip.h "Environment")
# Same-box A/B of the small signing calls' switches (environment, read at context creation): the single-launch round front, the small
# calls' own speculation rule, the single-launch kernels as a whole.  ML-DSA-44 / 65 / 87, wall time per call (tools/latency_probe.py).
for S in 44 65 87; do
  for n in 1 8 26 32 48 64 128 256; do
    for v in "default:" "MLDSA_SMALL_SIGN_SPEC=0:MLDSA_SMALL_SIGN_SPEC=0" "MLDSA_SMALL_SIGN_SPEC=1:MLDSA_SMALL_SIGN_SPEC=1" "MLDSA_SMALL_SIGN_FRONT=0:MLDSA_SMALL_SIGN_FRONT=0" "MLDSA_SMALL_FUSED=0:MLDSA_SMALL_FUSED=0"; do
      tag=${v%%:*}; e=${v#*:}
      echo -n "ML-DSA-$S $tag: "
      if [ -n "$e" ]; then export "$e"; fi
      SET=$S python3 tools/latency_probe.py sign $n 150 2>/dev/null | tail -1
      if [ -n "$e" ]; then unset "${e%%=*}"; fi
    done
  done
done
