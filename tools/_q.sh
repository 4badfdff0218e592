export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
rocprofv3 --output-format csv --kernel-trace -d gpurun_out/q -o run -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --streams 1 > gpurun_out/q.log 2>&1
python tools/summarize_profiles.py trace gpurun_out/q gpurun_out/q_summary.csv
rm -rf gpurun_out/q
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/q_summary.csv")):
    if ("stht" in r["kernel"] and int(r["calls"])>20) or ("rzcc_fast" in r["kernel"] and r["grid_x"]=="77120") or ("beamform_ws" in r["kernel"] and r["grid_y"]=="1100"):
        print(r["kernel"][:44], r["grid_x"], r["calls"], r["avg_us"], r["vgprs"])
PY
for i in 1 2 3; do python bench.py --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
python bench.py --steps 60 --warmup 6 --no-cpu-baseline --streams 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('serial', d['value'], d['ms_per_step'])"
