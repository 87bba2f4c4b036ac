#!/bin/bash
# GPU box: SQ counters of the time-resident forward kernel, resident_forward_kernel against wide_forward_kernel
# (tools/resident_probe.py 8 200, one seed per item).   gpurun -- 'bash tools/pmc_ab.sh'
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_ab; mkdir -p $OUT; cd $R
export TORBI_HIP_RESIDENT_KR=1
for w in 0 1; do
  export TORBI_HIP_WIDE=$w
  for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD"; do
    n=$(echo $set | tr ' ' '_')
    timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/w${w}_$n -o x --output-format csv -- python3 tools/resident_probe.py 8 200 > /dev/null 2> $OUT/w${w}_$n.err || echo "pass $w $n failed"
  done
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/pmc_ab'
for d in sorted(glob.glob(out + '/w*')):
    if not os.path.isdir(d): continue
    agg = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].split('(')[0].replace('void ', '').strip()
            if 'forward_kernel' in name: agg[(name[:64], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()): print(os.path.basename(d)[:3], k[0], k[1], 'max', max(v), 'n', len(v))
PY
