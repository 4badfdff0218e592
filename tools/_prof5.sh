export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace -d gpurun_out/q5 -o run -- python3 tools/stress_config5.py --check 0 > gpurun_out/q5.log 2>&1
python tools/summarize_profiles.py trace gpurun_out/q5 gpurun_out/q5_summary.csv
rm -rf gpurun_out/q5
tail -2 gpurun_out/q5.log
