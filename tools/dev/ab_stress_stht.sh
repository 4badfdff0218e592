#!/bin/bash
# A/B on ONE box: the 480-tap walking STHT with one time tile per wave (two workgroups per CU, the product) against two tiles per wave
# (one workgroup per CU: tools/_variants/libmicloc_hip_stht_wide2.so); the stage alone, then the stress step.
V=tools/_variants/libmicloc_hip_stht_wide2.so
for rep in 1 2; do
  python tools/dev/stht_bench_stress.py
  MICLOC_DEV_LIB=$V python tools/dev/stht_bench_stress.py
done
bash tools/dev/ab_lib.sh $V "--config stress --steps 9 --warmup 3 --repeats 3 --sustained-seconds 0"
