#!/bin/bash
# GPU box: memory-side traffic of the time-resident kernel with 3 seeds per item and with 1 (TORBI_HIP_RESIDENT_KR)
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/kr_traffic
mkdir -p $OUT
cd $R
CMD="python3 bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-secondary --pipeline 1"
for kr in 3 1; do
  export TORBI_HIP_RESIDENT_KR=$kr
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $set | tr ' ' '_')
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/kr${kr}/pmc_$n -o x --output-format csv -- $CMD > $OUT/kr${kr}_$n.json 2> $OUT/kr${kr}_$n.err || echo "pass kr=$kr $n failed"
  done
  python3 tools/summarise_pmc.py $OUT/kr${kr} | tail -12
done
