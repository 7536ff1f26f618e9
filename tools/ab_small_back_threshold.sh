export MLDSA_TUNING_ENV=1
for S in 65 44 87; do for n in 128 160 192 224 256; do for v in "1048576" "0" "1048576" "0"; do echo -n "ML-DSA-$S n=$n SLOTS_MAX=$v: "; SET=$S MLDSA_SMALL_BACK_SLOTS_MAX=$v python3 tools/latency_probe.py sign $n 150 2>/dev/null | tail -1; done; done; done
