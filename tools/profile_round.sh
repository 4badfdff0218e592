#!/bin/bash
# Round profiling recipe (run on the GPU box from the repo root: gpurun -- 'bash tools/profile_round.sh r1').
# Writes raw rocprofv3 output and bench JSONs under gpurun_out/<tag>/; copy the summaries into profiles/<tag>/.
set -u
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 40 --warmup 4 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python bench.py --steps 40 --warmup 4 --streams 1 --no-cpu-baseline > $OUT/bench_n1_streams1.json 2> $OUT/bench_n1_streams1.err
python bench.py --steps 40 --warmup 4 --grid 449 --no-cpu-baseline > $OUT/bench_n1_grid449.json 2> $OUT/bench_n1_grid449.err
# per-kernel times: serial steps so that every launch is timed alone
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o run -- python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline --streams 1 > $OUT/trace.log 2>&1
# counters: separate passes, nothing but --pmc (+ kernel trace)
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run -- python3 bench.py --steps 10 --warmup 1 --no-cpu-baseline --streams 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o run -- python3 bench.py --steps 10 --warmup 1 --no-cpu-baseline --streams 1 > $OUT/pmc_write.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o run -- python3 bench.py --steps 10 --warmup 1 --no-cpu-baseline --streams 1 > $OUT/pmc_sq.log 2>&1
python tools/summarize_profiles.py trace $OUT/trace $OUT/kernel_trace_summary_by_shape.csv
python tools/summarize_profiles.py pmc $OUT/pmc_summary.csv pmc_fetch=$OUT/pmc_fetch pmc_write=$OUT/pmc_write pmc_sq=$OUT/pmc_sq
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_bench_steps40_streams1.csv \;
# keep the merge-back small: drop the raw per-dispatch traces
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
ls -la $OUT
