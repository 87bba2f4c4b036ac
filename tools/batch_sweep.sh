#!/bin/bash
# bench.py over batch sizes and forward paths (GPU box): bash tools/batch_sweep.sh "1 8 31 32 64" "auto dense"
BATCHES=${1:-"32 64 128 256 512 1024 2048"}
PATHS=${2:-"auto dense"}
for b in $BATCHES; do
  for f in $PATHS; do
    python bench.py --batch $b --forward $f --no-cpu-baseline --steps 5 2>/dev/null | tail -1 > /tmp/line.json
    python - "$b" "$f" <<'PY'
import json, sys
d = json.load(open('/tmp/line.json'))
print('B', sys.argv[1], sys.argv[2], d['config']['forward_path'], round(d['value'] / 1e6, 3), 'M ts/s',
      round(d['roofline']['launch_us'], 1), 'us/launch')
PY
  done
done
