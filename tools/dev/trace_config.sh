#!/bin/bash
# kernel averages of one bench config (one stream): usage bash tools/dev/trace_config.sh stress|speech|xylo|noisy
OUT=gpurun_out/trace_cfg
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o run -- python3 bench.py --config $1 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-other-configs --streams 1 > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(6), round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
