"""Summarise rocprofv3 output directories into the two CSVs kept under profiles/rNN/.

python tools/summarize_profiles.py trace <dir> <out.csv>     kernel-trace grouped by kernel and launch shape
python tools/summarize_profiles.py pmc <out.csv> <name=dir>...  counter passes grouped by kernel, grid and counter
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def trace(d, out):
    rows = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"])
            rows[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "grid_x", "grid_y", "grid_z", "workgroup_x", "lds_bytes", "vgprs", "calls", "avg_us", "min_us", "max_us", "total_ms"])
        for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            w.writerow(list(k) + [len(v), f"{sum(v) / len(v):.2f}", f"{min(v):.2f}", f"{max(v):.2f}", f"{sum(v) / 1e3:.3f}"])


def pmc(out, pairs):
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["pass", "kernel", "grid_size", "counter", "dispatches", "mean_value"])
        for p in pairs:
            name, d = p.split("=", 1)
            acc = defaultdict(list)
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if not r["Kernel_Name"].startswith("micloc::") and "micloc" not in r["Kernel_Name"]:
                        continue
                    acc[(r["Kernel_Name"], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
            for (k, g, c), v in sorted(acc.items()):
                w.writerow([name, k, g, c, len(v), sum(v) / len(v)])


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        trace(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3:])
