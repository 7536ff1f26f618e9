#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_libs.sh <libA.so> <libB.so> [reps] -- alternates them under tools/ab_sign.sh
L=fips204_amd/csrc/libmldsa_hip.so
cp $L /tmp/orig.so
for rep in $(seq 1 ${3:-2}); do
  cp "$1" $L; bash tools/ab_sign.sh A$rep
  cp "$2" $L; bash tools/ab_sign.sh B$rep
done
cp /tmp/orig.so $L
