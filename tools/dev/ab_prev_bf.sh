#!/bin/bash
# A/B on ONE box: the in-tree library against tools/_variants/libmicloc_hip_prev_bf.so (the previous commit's beamform.hip): the stress step
# (stage times in the line: `beamform_kernel` = beamform_gen_kernel + power_argmax_kernel) and the 26-channel general-kernel shape.
V=tools/_variants/libmicloc_hip_prev_bf.so
LIB=haghighatshoarmuir2024_amd/libmicloc_hip.so
cp $LIB /tmp/lib_intree.so
for rep in 1 2; do
  for which in intree variant; do
    if [ $which = intree ]; then cp /tmp/lib_intree.so $LIB; else cp $V $LIB; fi
    python bench.py --config stress --steps 9 --warmup 3 --repeats 3 --sustained-seconds 0 --no-cpu-baseline --no-other-configs > gpurun_out/abl.log 2>&1
    python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/abl.log") if l.startswith("{")][-1])
print("[$which] stress ms/step", round(d["ms_per_step"], 3), "stages", {k: round(v, 3) for k, v in d["roofline"]["stages_ms"].items()}, "frac", round(d["roofline"]["frac"], 4))
PY
  done
done
cp /tmp/lib_intree.so $LIB
