#!/bin/bash
# A/B of the in-tree library against a variant build on ONE box (the variant is swapped in on the box's scratch copy of the repo)
# usage: bash tools/dev/ab_lib.sh build_dev/libmicloc_x.so "<bench args>"
LIB=haghighatshoarmuir2024_amd/libmicloc_hip.so
cp $LIB /tmp/lib_intree.so
for rep in 1 2; do
  for which in intree variant; do
    if [ $which = intree ]; then cp /tmp/lib_intree.so $LIB; else cp $1 $LIB; fi
    python bench.py --no-other-configs --no-cpu-baseline $2 > gpurun_out/abl.log 2>&1
    python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/abl.log") if l.startswith("{")][-1])
    print("[$which]", "ms/step", round(d["ms_per_step"], 4), "e2e", round((d.get("e2e") or {}).get("ms_per_step") or 0, 4))
except Exception as e:
    print("[$which] failed", open("gpurun_out/abl.log").read()[-300:])
PY
  done
done
cp /tmp/lib_intree.so $LIB
