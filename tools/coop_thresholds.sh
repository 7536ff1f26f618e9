#!/bin/bash
export MLDSA_TUNING_ENV=1  # the library reads its measurement knobs only when asked to (include/mldsa_hip.h "Environment")
# where the wave-cooperative sponges stop paying: tools/coop_thresholds.sh -- device-resident ML-DSA-65 calls of n ops with each
# cooperative kernel forced on (limit 2^20) or off (limit 0) at that size, the others at their defaults
for op in verify sign keygen; do
  for n in 64 128 256 512 1024 2048 4096 8192; do
    base=$(COOP=0 python tools/latency_probe.py $op $n 100 2>/dev/null | tail -1 | sed 's/.*median \([0-9.]*\) us.*/\1/')
    a_on=$(MLDSA_COOP_A_MAX=1048576 MLDSA_COOP_MASK_MAX=0 MLDSA_COOP_HASH_MAX=0 python tools/latency_probe.py $op $n 100 2>/dev/null | tail -1 | sed 's/.*median \([0-9.]*\) us.*/\1/')
    m_on=$(MLDSA_COOP_A_MAX=0 MLDSA_COOP_MASK_MAX=1048576 MLDSA_COOP_HASH_MAX=0 python tools/latency_probe.py $op $n 100 2>/dev/null | tail -1 | sed 's/.*median \([0-9.]*\) us.*/\1/')
    h_on=$(MLDSA_COOP_A_MAX=0 MLDSA_COOP_MASK_MAX=0 MLDSA_COOP_HASH_MAX=1048576 python tools/latency_probe.py $op $n 100 2>/dev/null | tail -1 | sed 's/.*median \([0-9.]*\) us.*/\1/')
    echo "$op n=$n: all off $base us; only ExpandA cooperative $a_on; only ExpandMask $m_on; only the hashes $h_on"
  done
done
