#!/bin/bash
export MLDSA_TUNING_ENV=1  # the library reads its measurement knobs only when asked to (include/mldsa_hip.h "Environment")
# Where do the single-launch kernels stop paying, per parameter set?  Same-box A/B of MLDSA_SMALL_FUSED (default 256) against 0 for
# verification, key generation and signing calls of 96 ... 512 ops (tools/latency_probe.py, wall time per call).
for S in 44 65 87; do
  for op in verify keygen sign; do
    for n in $( [ $op = sign ] && echo "96 128 192 256" || echo "96 128 192 256 384 512" ); do
      for v in 1024 0; do
        echo -n "ML-DSA-$S MLDSA_SMALL_FUSED=$v MLDSA_SMALL_KEYGEN_MAX=1024 MLDSA_SMALL_SIGN_MAX=256: "
        SET=$S MLDSA_SMALL_FUSED=$v MLDSA_SMALL_KEYGEN_MAX=1024 MLDSA_SMALL_SIGN_MAX=256 python3 tools/latency_probe.py $op $n 120 2>/dev/null | tail -1
      done
    done
  done
done
