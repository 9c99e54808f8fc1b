"""All kernels of the last Cholesky in a rocprofv3 --kernel-trace CSV that overlap [t0, t1] us (relative to the pass start):
start, duration, queue, name, grid size.   python tools/window_dump.py trace.csv 6000 7300"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
t0w, t1w = float(sys.argv[2]), float(sys.argv[3])
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['n'] = r['Kernel_Name'].split('(')[0].replace('void ', '')
rows.sort(key=lambda r: r['s'])
g = max(i for i, r in enumerate(rows) if r['n'].startswith('k_gram'))
last = rows[g:]
t0 = last[0]['s']
for r in last:
    s, e = (r['s'] - t0) / 1e3, (r['e'] - t0) / 1e3
    if e >= t0w and s <= t1w:
        grid = int(r.get('Grid_Size', 0) or 0) // max(int(r.get('Workgroup_Size', 1) or 1), 1)
        print(f"{s:10.1f} {e - s:8.1f}  q{r.get('Queue_Id', '?'):>3s}  wgs {grid:6d}  {r['n']}")
