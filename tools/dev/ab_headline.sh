#!/bin/bash
# A/B of the default bench on ONE box (boxes differ by a few percent): alternates two environments
# usage: bash tools/dev/ab_headline.sh "VAR=1" ["bench args"]
for rep in 1 2; do
  for env in "" "$1"; do
    env $env python bench.py --no-other-configs --no-cpu-baseline $2 > gpurun_out/ab.log 2>&1
    python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/ab.log") if l.startswith("{")][-1])
print("[%s]" % "$env", "ms/step", round(d["ms_per_step"], 4), "e2e", round(d["e2e"]["ms_per_step"], 4), "launch", d["roofline"].get("launch_ms"))
PY
  done
done
