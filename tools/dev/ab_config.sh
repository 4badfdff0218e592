#!/bin/bash
# A/B of one of the other bench configs on ONE box: usage bash tools/dev/ab_config.sh "VAR=1" stress|speech|xylo
for env in "" "$1"; do
  env $env python bench.py --config $2 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/abc.log 2>&1
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/abc.log") if l.startswith("{")][-1])
print("$2 [%s]" % "$env", "ms/step", round(d["ms_per_step"], 3))
PY
done
