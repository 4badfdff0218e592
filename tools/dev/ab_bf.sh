#!/bin/bash
# the LIF + beamforming stage alone under several library builds, alternating, on ONE box:  bash tools/dev/ab_bf.sh default <lib> ...
for rep in 1 2; do
  for v in "$@"; do
    if [ $v = default ]; then unset MICLOC_DEV_LIB; else export MICLOC_DEV_LIB=$v; fi
    python tools/dev/bf_bench.py 2>&1 | tail -1
  done
done
