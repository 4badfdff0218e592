#!/bin/bash
# A/B of micloc_synth_awgn_f64 on ONE box (boxes differ by a few percent): alternates the in-tree library and variant builds
# usage: bash tools/dev/ab_synth.sh build_dev/libmicloc_x.so [more variants]
for rep in 1 2 3; do
  python tools/dev/synth_bench.py 2>&1 | tail -1
  for v in "$@"; do MICLOC_DEV_LIB=$v python tools/dev/synth_bench.py 2>&1 | tail -1; done
done
