#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/collect_profiles.sh'): rocprofv3 kernel trace + separate PMC passes of the
# default bench command; summaries land in gpurun_out/profiles_new/ (copy the ones to keep into profiles/).
# The box has no .git: pass the commit as  gpurun -- "GIT_HASH=$(git rev-parse --short HEAD) bash tools/collect_profiles.sh"
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${PROFILE_OUT:-profiles_new}
mkdir -p $OUT
cd $R
# one warm-up launch group and one timed launch group of 8 batches on one stream, then the three profiled groups
# (PROFILE_CMD / PROFILE_OUT: another command under the same passes, e.g. "python3 tools/small_states_probe.py 512x500x64")
CMD=${PROFILE_CMD:-"python3 bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-secondary --no-single-call --pipeline 1"}
export PROFILE_CMD="$CMD"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o x --output-format csv -- $CMD > $OUT/bench_under_trace.json 2> $OUT/trace.err
# FETCH_SIZE and WRITE_SIZE do not fit one pass ("exceeds the capabilities of the hardware", after which rocprofv3
# hangs): one pass each, and every pass under its own timeout
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS TA_BUSY_avr"; do
  n=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$n -o x --output-format csv -- $CMD > /dev/null 2> $OUT/pmc_$n.err || echo "pass $n failed"
done
python3 tools/summarise_pmc.py $OUT
