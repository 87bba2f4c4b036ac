#!/bin/bash
# A/B of run-time selectable kernel forms on the GPU box: every "VAR=value,VAR=value" argument is one configuration,
# each run REPS times through tools/resident_probe.py (PROBE_ARGS, default "8 200"); prints the resident lines.
#   tools/env_probe.sh base TORBI_HIP_RESIDENT_VEC=0 TORBI_HIP_RESIDENT_KR=1,TORBI_HIP_RESIDENT_VEC=0
cd "$(dirname "$0")/.."
for rep in $(seq 1 ${REPS:-2}); do
  for cfg in "$@"; do
    if [ "$cfg" = base ]; then envs=""; else envs=$(echo "$cfg" | tr ',' ' '); fi
    line=$(env $envs python tools/resident_probe.py ${PROBE_ARGS:-8 200} 2>&1 | grep -E "^resident x${PROBE_N:-8}|Error|error" | tail -2)
    echo "$cfg: $line"
  done
done
