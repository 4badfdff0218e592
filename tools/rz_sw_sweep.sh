#!/bin/bash
# Encoder streams-per-workgroup sweep (measurement only): MICLOC_RZ_SW = 64 / 32 / 16 on the noisy and xylo workloads.
# usage (GPU box, repo root): bash tools/rz_sw_sweep.sh <tag>
set -u
OUT=gpurun_out/${1:-rzsw}
mkdir -p $OUT
for sw in 64 32 16; do
  MICLOC_RZ_SW=$sw python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline 2> $OUT/noisy_$sw.err | tail -1 > $OUT/noisy_$sw.json
  MICLOC_RZ_SW=$sw python3 bench.py --config xylo --steps 4 --warmup 1 2> $OUT/xylo_$sw.err | tail -1 > $OUT/xylo_$sw.json
done
python3 - <<PY > $OUT/summary.txt
import json
for cfg in ("noisy", "xylo"):
    for sw in (64, 32, 16):
        try:
            d = json.load(open("$OUT/%s_%d.json" % (cfg, sw)))
            print(cfg, "SW", sw, "ms_per_step %.4f" % d["ms_per_step"], {k: round(v, 4) for k, v in d["roofline"]["stages_ms"].items()})
        except Exception as e:
            print(cfg, sw, "failed", e)
PY
cat $OUT/summary.txt
