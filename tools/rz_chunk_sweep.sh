#!/bin/bash
# Encoder time-chunk sweep on the xylo workload (measurement only).  usage: bash tools/rz_chunk_sweep.sh <tag>
set -u
OUT=gpurun_out/${1:-rzchunk}
mkdir -p $OUT
for sw in 64; do
for ch in -1 24000 12000 6000 3000; do
  python3 bench.py --config xylo --steps 4 --warmup 1 --encoder-chunk $ch 2> $OUT/xylo_${sw}_$ch.err | tail -1 > $OUT/xylo_${sw}_$ch.json
done
done
python3 - <<PY > $OUT/summary.txt
import json
for sw in (64,):
    for ch in (-1, 24000, 12000, 6000, 3000):
        try:
            d = json.load(open("$OUT/xylo_%d_%d.json" % (sw, ch)))
            print("xylo SW", sw, "chunk", ch, "ms_per_step %.4f" % d["ms_per_step"], {k: round(v, 4) for k, v in d["roofline"]["stages_ms"].items()})
        except Exception as e:
            print(sw, ch, "failed", e)
PY
cat $OUT/summary.txt
