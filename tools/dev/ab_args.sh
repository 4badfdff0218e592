#!/bin/bash
# A/B of two bench argument sets on ONE box: usage bash tools/dev/ab_args.sh "<args A>" "<args B>" [common args]
for rep in 1 2; do
  for a in "$1" "$2"; do
    python bench.py --no-other-configs --no-cpu-baseline $3 $a > gpurun_out/aba.log 2>&1
    python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/aba.log") if l.startswith("{")][-1])
    print("[%s]" % "$a", "ms/step", round(d["ms_per_step"], 4), "e2e", round(d.get("e2e", {}).get("ms_per_step") or 0, 4))
except Exception as e:
    print("[%s] failed" % "$a", open("gpurun_out/aba.log").read()[-400:])
PY
  done
done
