#!/bin/bash
# kernel timeline of a pipelined bench run (rocprofv3 kernel trace, 3 streams): start / end (ms) per launch in a steady-state window
# usage: bash tools/dev/timeline.sh speech [steps]
OUT=gpurun_out/timeline
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace -d $OUT/trace -o run -- python3 bench.py --config $1 --steps ${2:-6} --warmup 2 --repeats 1 --no-cpu-baseline --no-other-configs > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
big = [r for r in rows if (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) > int("${3:-200000}")]
# everything that takes longer than 0.2 ms -> long.csv (relative ms); print the pipelined region: the launches between the
# first and the last stht kernel of the timed steps
t0 = int(big[0]["Start_Timestamp"])
with open("$OUT/long.csv", "w") as fh:
    for r in big:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
        fh.write("%.3f,%.3f,%.3f,%s,%s\n" % (s, e, e - s, r.get("Queue_Id", "?"), r["Kernel_Name"][:50].replace(",", ";")))
print(len(big), "long launches ->", "$OUT/long.csv")
PY
