#!/bin/bash
# STHT stage alone: launch time and FETCH_SIZE per launch (one counter per pass: FETCH_SIZE + WRITE_SIZE together exceed the hardware's counters), in-tree library against variant builds on one box
# usage: bash tools/dev/stht_fetch_ab.sh [variant.so ...]
OUT=gpurun_out/stht_ab
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for lib in default "$@"; do
  if [ $lib = default ]; then unset MICLOC_DEV_LIB; else export MICLOC_DEV_LIB=$lib; fi
  tag=$(basename $lib .so)
  python3 tools/dev/stht_bench.py
  python3 tools/dev/stht_bench.py
  timeout -k 10 150 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/$tag -o run -- python3 tools/dev/stht_bench.py > $OUT/$tag.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/$tag/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen:
        seen.add((k, r["Dispatch_Id"])); n[k] += 1
for k in acc:
    if "stht" in k: print("$tag", k, "launches", n[k], {c: round(v / n[k]) for c, v in acc[k].items()})
PY
  rm -rf $OUT/$tag
done
