#!/bin/bash
# Same-box A/B of two builds of the library on small signing calls: tools/ab_libs_small_sign.sh <libA.so> <libB.so>
# (wall time per call, tools/latency_probe.py, interleaved; SETS / SIZES as in tools/ab_small_back.sh).  The in-tree library is restored.
L=fips204_amd/csrc/libmldsa_hip.so
cp $L /tmp/ab_orig.so
for S in ${SETS:-65 44 87}; do
  for n in ${SIZES:-1 8 64 256}; do
    for v in A B A B; do
      if [ $v = A ]; then cp "$1" $L; else cp "$2" $L; fi
      echo -n "ML-DSA-$S n=$n lib $v: "
      SET=$S python3 tools/latency_probe.py sign $n 150 2>/dev/null | tail -1
    done
  done
done
cp /tmp/ab_orig.so $L
