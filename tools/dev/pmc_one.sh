#!/bin/bash
# SQ counters of the default bench (one stream), summarised per kernel
# usage: bash tools/dev/pmc_one.sh
OUT=gpurun_out/pmc_one
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
extra="--steps 6 --warmup 1 --repeats 1 --no-cpu-baseline --no-other-configs --streams 1"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o run -- python3 bench.py $extra > $OUT/pmc_sq.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o run -- python3 bench.py $extra > $OUT/pmc_lds.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag in ("pmc_sq", "pmc_lds"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not f:
        print(tag, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); n[k] += 1
    for k in acc:
        if any(s in k for s in ("stht", "beamform_ws_kernel", "bandpass_rzcc_fast")):
            print(tag, k, "launches", n[k], {c: round(v / n[k]) for c, v in acc[k].items()})
PY
