"""Fold the rocprofv3 CSVs written by tools/collect_profiles.sh into pmc.json / kernel_stats.csv (copied to
profiles/rNN_pmc.json and profiles/rNN_bench_kernel_stats.csv by hand)."""
import collections, csv, glob, json, os, shutil, sys

out = sys.argv[1]
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '').strip()
        pmc[name][r['Counter_Name']].append(float(r['Counter_Value']))
summary = {k: {c: {'mean_per_dispatch': sum(v) / len(v), 'dispatches': len(v)} for c, v in cs.items()}
           for k, cs in pmc.items()}
# provenance of every summary: the commit of the box's snapshot, the profiled command; for the bench command also how many
# batches one forward launch covered (bench.py scales `traffic` by it) and the cells that launch examined
meta = {'units': 'FETCH_SIZE / WRITE_SIZE in KiB per dispatch', 'git': os.environ.get('GIT_HASH') or None,
        'command': os.environ.get('PROFILE_CMD') or 'python3 bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-secondary '
                                                    '--no-single-call --pipeline 1',
        'passes': 'one rocprofv3 --kernel-trace --pmc pass per counter group (tools/collect_profiles.sh)'}
try:
    line = json.load(open(os.path.join(out, 'bench_under_trace.json')))
    meta['batches_per_forward_launch'] = int(line['config']['launch_groups'][0])
    meta['executed_cells_per_launch'] = ((line.get('roofline') or {}).get('executed') or {}).get('cells_per_launch')
except (OSError, ValueError, KeyError, IndexError, TypeError):
    pass
summary['_meta'] = meta
json.dump(summary, open(os.path.join(out, 'pmc.json'), 'w'), indent=1)
for f in glob.glob(os.path.join(out, 'trace', '**', '*kernel_stats.csv'), recursive=True):
    shutil.copy(f, os.path.join(out, 'kernel_stats.csv'))
print(json.dumps({k: {c: v['mean_per_dispatch'] for c, v in cs.items()} for k, cs in summary.items()
                  if k != '_meta' and ('step' in k or 'resident' in k or 'small::' in k or 'band' in k)}, indent=1))
