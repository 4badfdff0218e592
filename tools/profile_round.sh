#!/bin/bash
# Round profiling recipe (run on the GPU box from the repo root: gpurun -- 'bash tools/profile_round.sh r4').
# Writes raw rocprofv3 output and bench JSONs under gpurun_out/<tag>/; copy the summaries into profiles/<tag>/.
# usage: bash tools/profile_round.sh <tag> [bench|extra|pmc|all]   (three gpurun calls of <= 20 minutes: `bench`, `extra`, `pmc`)
set -u
TAG=${1:-r6}
PART=${2:-all}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# records for profiles/<tag>/MANIFEST.json (tools/make_manifest.py): the box, the hashes of the sources AS RUN, and which command wrote
# which file (the files themselves are written by the lines below; COMMANDS.tsv names the command of every file that is kept)
python3 tools/make_manifest.py --record $OUT
cmd() { printf '%s\t%s\n' "$1" "$2" >> $OUT/COMMANDS.tsv; }
: > $OUT/COMMANDS.tsv
want() { [ $PART = all ] || [ $PART = $1 ]; }
cmd bench_n1.json "python3 bench.py --steps 20 --warmup 5"
cmd bench_n1_streams1.json "python3 bench.py --steps 40 --warmup 4 --streams 1 --no-cpu-baseline --no-other-configs"
cmd bench_n1_grid449.json "python3 bench.py --steps 40 --warmup 4 --grid 449 --no-cpu-baseline --no-other-configs"
cmd bench_n1_speech.json "python3 bench.py --config speech --steps 16 --warmup 4"
cmd bench_n1_speech_graphs.json "python3 bench.py --config speech --steps 16 --warmup 4 --schedule graphs --no-cpu-baseline"
cmd bench_n1_stress.json "python3 bench.py --config stress --steps 9 --warmup 3"
cmd bench_n1_xylo.json "python3 bench.py --config xylo --steps 12 --warmup 3"
cmd bench_n1_sustained60.json "python3 bench.py --steps 20 --warmup 5 --sustained-seconds 60 --no-cpu-baseline --no-other-configs"
cmd bench_n1_stress_sustained20.json "python3 bench.py --config stress --steps 9 --warmup 3 --sustained-seconds 20 --no-cpu-baseline --no-other-configs"
cmd kernel_trace_summary_by_shape.csv "rocprofv3 --output-format csv --kernel-trace --stats -- python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-other-configs --streams 1 | tools/summarize_profiles.py trace"
cmd kernel_stats_bench_steps40_streams1.csv "rocprofv3 --kernel-trace --stats (the same run): rocprofv3's own kernel_stats.csv"
cmd kernel_trace_summary_grid449.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 4 --grid 449 --no-cpu-baseline --no-other-configs --streams 1 | summarize_profiles.py trace"
cmd kernel_trace_summary_speech.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --config speech --steps 3 --warmup 1 --streams 1 | summarize_profiles.py trace"
cmd kernel_trace_summary_stress.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --config stress --steps 3 --warmup 1 --streams 1 | summarize_profiles.py trace"
cmd kernel_trace_summary_xylo.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --config xylo --steps 2 --warmup 1 --streams 1 | summarize_profiles.py trace"
cmd y_store.txt "python3 tools/k3_time.py {360,449} 1100 {1,0}"
cmd store_bw.txt "hipcc tools/store_bw.hip && ./store_bw"
cmd ablation_kstep.txt "bash tools/dev/ab_lib.sh tools/_variants/libmicloc_hip_ws_k4.so '--steps 40 --warmup 4'"
cmd ablation_stht.txt "bash tools/dev/ab_lib.sh tools/_variants/libmicloc_hip_stht_valu.so '--steps 40 --warmup 4'"
cmd ablation_stht_walk.txt "bash tools/dev/ab_lib.sh tools/_variants/libmicloc_hip_stht_one_tile.so '--steps 40 --warmup 4'"
cmd ablation_sparse_lif.txt "bash tools/dev/ab_bf.sh default tools/_variants/libmicloc_hip_ws_sparse_lif.so"
cmd c128_time.txt "python3 tools/c128_time.py {360,449,57} 1100 {0,1}"
cmd design_config5.txt "python3 tools/dev/design_cfg5_time.py 48 240"
cmd product_sweeps.txt "python3 tools/dev/speech_sweep_time.py 91; python3 tools/dev/xylo_sweep_time.py"
cmd speech_lanes.txt "bash tools/dev/speech_lanes.sh 2 4"
cmd stht_fetch.txt "bash tools/dev/stht_fetch_ab.sh; STHT_T=48000 bash tools/dev/stht_fetch_ab.sh"
cmd clock_sources.txt "python3 tools/dev/clock_sources.py"
cmd bench_n2_share_device.json "python3 bench.py --gpus 2 --share-device --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs  (REHEARSAL of the N > 1 launch: both ranks on device 0, gloo)"
cmd bench_share_stress2048.json "python3 bench.py --config stress --baseline-total --trials 2048 --steps 6 --warmup 2 --repeats 3 --sustained-seconds 10 --no-cpu-baseline --no-other-configs  (a rank's share of the 16 384-trial sweep at 8 ranks)"
cmd bench_share_speech125.json "python3 bench.py --config speech --baseline-total --trials 125 --steps 16 --warmup 4 --repeats 3 --sustained-seconds 10 --no-cpu-baseline --no-other-configs  (a rank's share of the 1000-trial sweep at 8 ranks)"
cmd ablation_stht_wide2.txt "bash tools/dev/ab_stress_stht.sh"
cmd track_time.txt "python3 tools/dev/track_time.py"
for c in "" _stress _speech _xylo; do cmd pmc_summary$c.csv "rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE | SQ_* ...> (one pass each) -- python3 bench.py --config <cfg> --steps <n> --warmup 1 --repeats 1 --no-cpu-baseline --no-other-configs --streams 1 --sustained-seconds 0 | summarize_profiles.py pmc"; done
if want bench; then
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err   # the driver's command: all blocks (other configs as child runs)
python3 bench.py --steps 40 --warmup 4 --streams 1 --no-cpu-baseline --no-other-configs > $OUT/bench_n1_streams1.json 2> $OUT/bench_n1_streams1.err
python3 bench.py --steps 40 --warmup 4 --grid 449 --no-cpu-baseline --no-other-configs > $OUT/bench_n1_grid449.json 2> $OUT/bench_n1_grid449.err
python3 bench.py --config speech --steps 16 --warmup 4 > $OUT/bench_n1_speech.json 2> $OUT/bench_n1_speech.err   # (scan-lane schedule, four batches in flight)
python3 bench.py --config speech --steps 16 --warmup 4 --schedule graphs --no-cpu-baseline > $OUT/bench_n1_speech_graphs.json 2> $OUT/bench_n1_speech_graphs.err
python3 bench.py --config stress --steps 9 --warmup 3 > $OUT/bench_n1_stress.json 2> $OUT/bench_n1_stress.err
python3 bench.py --config xylo --steps 12 --warmup 3 > $OUT/bench_n1_xylo.json 2> $OUT/bench_n1_xylo.err
# one minute of headline steps / twenty seconds of the stress step in one uninterrupted region each (clock, power, temperature per segment)
python3 bench.py --steps 20 --warmup 5 --sustained-seconds 60 --no-cpu-baseline --no-other-configs > $OUT/bench_n1_sustained60.json 2> $OUT/bench_n1_sustained60.err
python3 bench.py --config stress --steps 9 --warmup 3 --sustained-seconds 20 --no-cpu-baseline --no-other-configs > $OUT/bench_n1_stress_sustained20.json 2> $OUT/bench_n1_stress_sustained20.err
# per-kernel times: serial steps so that every launch is timed alone
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o run -- python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-other-configs --streams 1 --sustained-seconds 0 > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace_g449 -o run -- python3 bench.py --steps 20 --warmup 4 --grid 449 --no-cpu-baseline --no-other-configs --streams 1 --sustained-seconds 0 > $OUT/trace_g449.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace_speech -o run -- python3 bench.py --config speech --steps 3 --warmup 1 --streams 1 --sustained-seconds 0 --no-cpu-baseline --no-other-configs > $OUT/trace_speech.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace_stress -o run -- python3 bench.py --config stress --steps 3 --warmup 1 --streams 1 --sustained-seconds 0 --no-cpu-baseline --no-other-configs > $OUT/trace_stress.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace_xylo -o run -- python3 bench.py --config xylo --steps 2 --warmup 1 --streams 1 --sustained-seconds 0 --no-cpu-baseline --no-other-configs > $OUT/trace_xylo.log 2>&1
python3 tools/summarize_profiles.py trace $OUT/trace $OUT/kernel_trace_summary_by_shape.csv
python3 tools/summarize_profiles.py trace $OUT/trace_g449 $OUT/kernel_trace_summary_grid449.csv
python3 tools/summarize_profiles.py trace $OUT/trace_speech $OUT/kernel_trace_summary_speech.csv
python3 tools/summarize_profiles.py trace $OUT/trace_stress $OUT/kernel_trace_summary_stress.csv
python3 tools/summarize_profiles.py trace $OUT/trace_xylo $OUT/kernel_trace_summary_xylo.csv
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_bench_steps40_streams1.csv \;
rm -rf $OUT/trace $OUT/trace_g449 $OUT/trace_speech $OUT/trace_stress $OUT/trace_xylo
fi
if want extra; then
# API-faithful mode (y stored) and the store-pattern microbenchmark behind its design
(python3 tools/k3_time.py 360 1100 1; python3 tools/k3_time.py 449 1100 1; python3 tools/k3_time.py 360 1100 0; python3 tools/k3_time.py 449 1100 0) > $OUT/y_store.txt 2>/dev/null
hipcc -O3 --offload-arch=gfx950 -o /tmp/store_bw tools/store_bw.hip > /dev/null 2>&1 && /tmp/store_bw > $OUT/store_bw.txt 2>&1
# ablations on the same box, alternating the in-tree library with a variant build (tools/dev/make_variant.py; no run-time switches):
#   ws_k4      four MFMA k-steps for 14 channels instead of 3 + 2 channels on the vector ALU
#   stht_valu  the STHT on the vector ALU (round 2's kernel) instead of the matrix cores
bash tools/dev/ab_lib.sh tools/_variants/libmicloc_hip_ws_k4.so "--steps 40 --warmup 4" > $OUT/ablation_kstep.txt 2>&1
bash tools/dev/ab_lib.sh tools/_variants/libmicloc_hip_stht_valu.so "--steps 40 --warmup 4" > $OUT/ablation_stht.txt 2>&1
#   stht_one_tile  the matrix-core STHT with one time tile per workgroup (round 3) instead of the walk
bash tools/dev/ab_lib.sh tools/_variants/libmicloc_hip_stht_one_tile.so "--steps 40 --warmup 4" > $OUT/ablation_stht_walk.txt 2>&1
#   ws_sparse_lif  beamform_ws_kernel's LIF stage event by event (round 5) instead of the dense product; the stage alone, checksums of the power
bash tools/dev/ab_bf.sh default tools/_variants/libmicloc_hip_ws_sparse_lif.so > $OUT/ablation_sparse_lif.txt 2>&1
python3 tools/dev/clock_sources.py > $OUT/clock_sources.txt 2>&1
# round 6: the N > 1 launch rehearsed on this one GPU, the strong-scaling shares an 8-rank run gives every rank, the rejected two-tile form of
# the 480-tap STHT, the moving-target read-out at the script's shape
python3 bench.py --gpus 2 --share-device --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $OUT/bench_n2_share_device.json 2> $OUT/bench_n2_share_device.err
python3 bench.py --config stress --baseline-total --trials 2048 --steps 6 --warmup 2 --repeats 3 --sustained-seconds 10 --no-cpu-baseline --no-other-configs > $OUT/bench_share_stress2048.json 2> $OUT/bench_share_stress2048.err
python3 bench.py --config speech --baseline-total --trials 125 --steps 16 --warmup 4 --repeats 3 --sustained-seconds 10 --no-cpu-baseline --no-other-configs > $OUT/bench_share_speech125.json 2> $OUT/bench_share_speech125.err
bash tools/dev/ab_stress_stht.sh 2>&1 | grep -v amdgpu.ids > $OUT/ablation_stht_wide2.txt
python3 tools/dev/track_time.py 2>&1 | grep -v amdgpu.ids > $OUT/track_time.txt
# round 4: the complex Beamformer's contraction alone, the Xylo LIF launch forms, config 5's design
(python3 tools/c128_time.py 360 1100 0; python3 tools/c128_time.py 449 1100 0; python3 tools/c128_time.py 360 1100 1; python3 tools/c128_time.py 57 1100 1) > $OUT/c128_time.txt 2>/dev/null
python3 tools/dev/design_cfg5_time.py 48 240 > $OUT/design_config5.txt 2>/dev/null
# round 4, late: the product sweeps (whole call, one GPU), the speech step under its schedules (ONE masked configuration per process) and
# the STHT's input reads at two recording lengths
(python3 tools/dev/speech_sweep_time.py 91; python3 tools/dev/xylo_sweep_time.py) 2>/dev/null | grep -v amdgpu.ids > $OUT/product_sweeps.txt
bash tools/dev/speech_lanes.sh 2 4 > $OUT/speech_lanes.txt 2>&1
(bash tools/dev/stht_fetch_ab.sh; STHT_T=48000 bash tools/dev/stht_fetch_ab.sh) 2>/dev/null | grep -v amdgpu.ids > $OUT/stht_fetch.txt
fi
if want pmc; then
# counters: separate passes, nothing but --pmc (+ kernel trace)
for cfg in noisy stress speech xylo; do
  steps=6; [ $cfg = speech ] && steps=2; [ $cfg = xylo ] && steps=2
  extra="--config $cfg --steps $steps --warmup 1 --repeats 1 --no-cpu-baseline --no-other-configs --streams 1 --sustained-seconds 0"
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_$cfg -o run -- python3 bench.py $extra > $OUT/pmc_fetch_$cfg.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_$cfg -o run -- python3 bench.py $extra > $OUT/pmc_write_$cfg.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/pmc_sq_$cfg -o run -- python3 bench.py $extra > $OUT/pmc_sq_$cfg.log 2>&1
done
python3 tools/summarize_profiles.py pmc $OUT/pmc_summary.csv pmc_fetch=$OUT/pmc_fetch_noisy pmc_write=$OUT/pmc_write_noisy pmc_sq=$OUT/pmc_sq_noisy
python3 tools/summarize_profiles.py pmc $OUT/pmc_summary_stress.csv pmc_fetch=$OUT/pmc_fetch_stress pmc_write=$OUT/pmc_write_stress pmc_sq=$OUT/pmc_sq_stress
python3 tools/summarize_profiles.py pmc $OUT/pmc_summary_speech.csv pmc_fetch=$OUT/pmc_fetch_speech pmc_write=$OUT/pmc_write_speech pmc_sq=$OUT/pmc_sq_speech
python3 tools/summarize_profiles.py pmc $OUT/pmc_summary_xylo.csv pmc_fetch=$OUT/pmc_fetch_xylo pmc_write=$OUT/pmc_write_xylo pmc_sq=$OUT/pmc_sq_xylo
# keep the merge-back small: drop the raw per-dispatch traces
rm -rf $OUT/pmc_fetch_* $OUT/pmc_write_* $OUT/pmc_sq_*
fi
ls -la $OUT
