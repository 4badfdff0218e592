#!/bin/bash
# the speech step under scan lanes of different widths (compute units per XCD), one process each:  bash tools/dev/speech_lanes.sh 2 3 4
for rep in 1 2; do
for c in "$@"; do
  python bench.py --config speech --steps 16 --warmup 4 --repeats 3 --no-cpu-baseline --no-other-configs --scan-lane-cus $c > gpurun_out/sl.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/sl.json'));print('lane $c CUs/XCD: speech ms/step %.3f' % d['ms_per_step'], ['%.2f' % v for v in d['ms_per_step_repeats']], 'e2e %.3f' % d['e2e']['ms_per_step'])"
done
done
