#!/bin/bash
# bench.py over state counts (GPU box): bash tools/state_sweep.sh "64 128 ..." [batch] [frames]
STATES=${1:-"64 128 256 360 512 720 1024 1440 1441 2048 2052 3000 4096 4100 8192"}
B=${2:-512}
T=${3:-100}
for s in $STATES; do
  python bench.py --batch $B --frames $T --states $s --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > /tmp/line.json
  python - "$s" <<'PY'
import json, sys
d = json.load(open('/tmp/line.json'))
S = int(sys.argv[1])
print('S', S, d['config']['forward_path'], round(d['value'] / 1e6, 3), 'M ts/s', round(d['roofline']['launch_us'], 1), 'us/launch',
      'dense-equivalent Tcell/s', round(d['valu']['dense_equivalent_cells_per_s'] / 1e12, 1))
PY
done
