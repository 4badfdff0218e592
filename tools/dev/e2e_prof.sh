#!/bin/bash
# e2e line of the default bench + per-kernel averages of the input-side kernels (one rocprofv3 kernel trace).
# usage (on the GPU box, from the repo root): bash tools/dev/e2e_prof.sh
OUT=gpurun_out/e2e_prof
mkdir -p $OUT
python bench.py --no-other-configs --no-cpu-baseline > $OUT/bench.log 2>&1
python - <<PY
import json
l = [x for x in open("$OUT/bench.log") if x.startswith("{")][-1]
d = json.loads(l)
print("headline ms/step", round(d["ms_per_step"], 4), " e2e ms/step", round(d["e2e"]["ms_per_step"], 4))
PY
export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o run -- python3 bench.py --steps 10 --repeats 1 --no-cpu-baseline --no-other-configs --streams 1 > $OUT/trace.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(6), round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
