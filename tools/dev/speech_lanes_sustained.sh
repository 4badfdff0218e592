#!/bin/bash
# The speech step against the scan lane's width with a SUSTAINED region (steady state, no fill / drain): usage speech_lanes_sustained.sh 2 3 4
for cus in "$@"; do
  timeout -k 10 200 python bench.py --config speech --steps 16 --warmup 4 --repeats 2 --scan-lane-cus $cus --sustained-seconds 8 --no-cpu-baseline --no-other-configs > gpurun_out/speech_lane$cus.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("gpurun_out/speech_lane$cus.json").read().strip().splitlines()[-1])
print("lane $cus CUs/XCD: 16-step regions", round(d["ms_per_step"], 3), "ms; sustained", round(d["sustained"]["ms_per_step"], 3), "ms per step")
PY
done
