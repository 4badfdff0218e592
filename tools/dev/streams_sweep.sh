#!/bin/bash
# headline step against the number of HIP streams consecutive steps are pipelined over (one box)
for s in 1 2 3 4 6; do
  python bench.py --no-other-configs --no-cpu-baseline --streams $s > gpurun_out/ss.log 2>&1
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/ss.log") if l.startswith("{")][-1])
print("streams $s ms/step", round(d["ms_per_step"], 4), "e2e", round(d["e2e"]["ms_per_step"], 4))
PY
done
