#!/bin/bash
# A/B on ONE box: the in-tree library against tools/_variants/libmicloc_hip_prev_stht.so (the previous commit's stht.hip): STHT stage
# alone at the stress and at the sweep shape, then the headline and the stress step.
V=tools/_variants/libmicloc_hip_prev_stht.so
for rep in 1 2; do
  python tools/dev/stht_bench_stress.py
  MICLOC_DEV_LIB=$V python tools/dev/stht_bench_stress.py
  python tools/dev/stht_bench.py
  MICLOC_DEV_LIB=$V python tools/dev/stht_bench.py
done
bash tools/dev/ab_lib.sh $V "--steps 40 --warmup 5 --sustained-seconds 0"
bash tools/dev/ab_lib.sh $V "--config stress --steps 9 --warmup 3 --repeats 3 --sustained-seconds 0"
