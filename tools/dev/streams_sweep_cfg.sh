#!/bin/bash
# step time of one bench config against the number of HIP streams: usage bash tools/dev/streams_sweep_cfg.sh speech "2 3 4 6"
for s in $2; do
  python bench.py --config $1 --steps ${3:-12} --warmup 3 --repeats 3 --no-cpu-baseline --no-other-configs --streams $s > gpurun_out/ssc.log 2>&1
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/ssc.log") if l.startswith("{")][-1])
    print("$1 streams $s ms/step", round(d["ms_per_step"], 3))
except Exception as e:
    print("$1 streams $s failed", e, open("gpurun_out/ssc.log").read()[-300:])
PY
done
