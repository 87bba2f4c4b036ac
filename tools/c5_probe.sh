cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/c5b
python3 bench.py --batch 128 --frames 2000 --states 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --group 1 --pipeline 1 > gpurun_out/c5b/bench.json 2> gpurun_out/c5b/bench.err
CMD="python3 bench.py --batch 128 --frames 300 --states 4096 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --group 1 --pipeline 1"
for set in "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS TA_BUSY_avr"; do
  n=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/c5b/pmc_$n -o x --output-format csv -- $CMD > /dev/null 2> gpurun_out/c5b/pmc_$n.err || echo "pass $n failed"
done
python3 tools/summarise_pmc.py gpurun_out/c5b
cat gpurun_out/c5b/bench.json | head -c 1500
