#!/bin/bash
# Where do the split band kernel's LDS bank conflicts come from?  SQ_LDS_* counters of band_forward_kernel on the in-tree
# library and on -DBAND_ABL builds (tools/variants_probe.py build NAME "-DBAND_ABL=n"), + the address patterns alone.
# GPU box:  bash tools/band_lds_conflicts.sh base abl16 abl4 > gpurun_out/band_lds_conflicts.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
./tools/lds_pattern_probe
for name in "$@"; do
  if [ "$name" = base ]; then unset TORBI_HIP_LIBRARY; else export TORBI_HIP_LIBRARY=$R/tools/libtorbi_hip_$name.so; fi
  out=$R/gpurun_out/lds_$name
  rm -rf $out
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $out -o x --output-format csv -- python3 tools/band_time.py > $out.log 2>&1
  echo "== $name: $(tail -n 1 $out.log)"
  python3 - $out <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'band_forward_kernel' in r['Kernel_Name']:
            rows[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in rows.items():
    print('  ', k, {n: round(sum(v) / len(v) / 1e6, 1) for n, v in c.items()}, '(M per dispatch)')
PY
done
