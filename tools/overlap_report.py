"""Start/end times of the forward and backtrace kernels in a rocprofv3 kernel trace: do launch groups on different
streams overlap?    python tools/overlap_report.py <x_kernel_trace.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'resident' in r['Kernel_Name'] or 'step_pruned' in r['Kernel_Name']]
rows = [r for r in rows if 'order_items' not in r['Kernel_Name']]
t0 = min(int(r['Start_Timestamp']) for r in rows)
busy_until = 0
for r in sorted(rows, key=lambda r: int(r['Start_Timestamp'])):
    s, e = (int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - t0) / 1e6
    name = 'forward ' if 'forward' in r['Kernel_Name'] else 'backtrace'
    print(f"{name} queue {r.get('Queue_Id', '?'):>3} grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>8}  {s:9.2f} -> {e:9.2f} ms  ({e - s:7.2f})"
          + ('   overlaps the previous kernel' if s < busy_until else ''))
    busy_until = max(busy_until, e)
