#!/bin/bash
# A/B on ONE box in ONE call: tools/variant_times.py (or any command) against two builds of the library, alternating
# (the pool's boxes differ by several percent on the long kernels).
# usage: tools/ab_times.sh libA.so libB.so [rounds=2] [command...]
A=$1; B=$2; N=${3:-2}; shift 3 2>/dev/null
CMD=${@:-python tools/variant_times.py}
for i in $(seq 1 $N); do
  for L in "$A" "$B"; do
    echo "=== $L (round $i)"
    GTOP_HIP_LIB=$(realpath $L) $CMD 2>&1 | grep -v amdgpu.ids
  done
done
