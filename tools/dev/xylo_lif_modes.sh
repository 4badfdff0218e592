#!/bin/bash
# the xylo step under the LIF launch forms (measurement): bash tools/dev/xylo_lif_modes.sh [mode ...]
for v in ${@:-static queue:4 static queue:4}; do
  timeout -k 5 200 python bench.py --config xylo --steps 12 --warmup 3 --repeats 3 --xylo-lif $v > gpurun_out/xylo_q.json 2> gpurun_out/xylo_q.err
  python -c "
import json
d=json.loads([l for l in open('gpurun_out/xylo_q.json') if l.startswith('{')][-1])
print('$v', round(d['ms_per_step'],3), [round(x,2) for x in d['ms_per_step_repeats']], {k: round(v, 3) for k, v in d['roofline']['stages_ms'].items()})
"
done
