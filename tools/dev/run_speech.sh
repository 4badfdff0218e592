export TMPDIR=/tmp
python -m pytest tests/test_hip_chunked.py tests/test_speech_config.py tests/test_hip_fullsize.py -x -q -m gpu > gpurun_out/gputests_d.log 2>&1; tail -3 gpurun_out/gputests_d.log
python bench.py --config speech --steps 16 --warmup 4 --repeats 3 --no-cpu-baseline --no-other-configs > gpurun_out/bench_speech_a.json 2>gpurun_out/bench_speech_a.err
python -c "
import json;d=json.load(open('gpurun_out/bench_speech_a.json'));print('speech ms/step',d['ms_per_step'],d['ms_per_step_repeats'],d['roofline'].get('serial_stage'),d['roofline']['stages_ms'])"
rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/trace_speech -o run -- python3 bench.py --config speech --steps 3 --warmup 1 --streams 1 --repeats 1 --no-cpu-baseline --no-other-configs > gpurun_out/trace_speech.log 2>&1
python3 tools/summarize_profiles.py trace gpurun_out/trace_speech gpurun_out/kernel_trace_summary_speech_a.csv
rm -rf gpurun_out/trace_speech
head -8 gpurun_out/kernel_trace_summary_speech_a.csv | cut -c1-200
