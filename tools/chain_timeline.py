"""Timeline of the last Cholesky in a rocprofv3 --kernel-trace CSV: per kernel name the count / busy time, and for the diagonal
kernels the start-to-start interval (the chain step) with the gaps between the chain kernels.

    python tools/chain_timeline.py <kernel_trace.csv> [rows]
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for r in rows:
    r['s'] = int(r['Start_Timestamp'])
    r['e'] = int(r['End_Timestamp'])
    r['n'] = r['Kernel_Name'].split('(')[0].replace('void ', '')
rows.sort(key=lambda r: r['s'])
# last pass = everything after the last k_gram
g = max(i for i, r in enumerate(rows) if r['n'].startswith('k_gram'))
last = [r for r in rows[g:] if not r['n'].startswith('k_lml')]
t0 = last[0]['s']
span = (max(r['e'] for r in last) - t0) / 1e3
print(f'last pass: {len(last)} kernels, span {span:.1f} us')
agg = defaultdict(lambda: [0, 0.0])
for r in last:
    agg[r['n']][0] += 1
    agg[r['n']][1] += (r['e'] - r['s']) / 1e3
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'  {n:28s} {c:5d} launches {t:10.1f} us busy  avg {t / c:8.1f}')
diag = [r for r in last if r['n'].startswith('k_diag')]
steps = [(b['s'] - a['s']) / 1e3 for a, b in zip(diag, diag[1:])]
if steps:
    steps_sorted = sorted(steps)
    print(f'diag start-to-start: n={len(steps)} median {steps_sorted[len(steps) // 2]:.1f} us  mean {sum(steps) / len(steps):.1f}  '
          f'min {steps_sorted[0]:.1f} max {steps_sorted[-1]:.1f}')
if steps:
    print('steps in order (us), eight per line = one panel at NB = 1024:')
    for i in range(0, len(steps), 8):
        print('  ' + ' '.join(f'{x:7.1f}' for x in steps[i:i + 8]) + f'   | sum {sum(steps[i:i + 8]):8.1f}')
print('--- first kernels of the pass (start us, dur us, queue, name)')
for r in last[:nshow]:
    print(f"{(r['s'] - t0) / 1e3:10.1f} {(r['e'] - r['s']) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3s}  {r['n']}")
mid = len(last) // 2
print('--- middle')
for r in last[mid:mid + nshow]:
    print(f"{(r['s'] - t0) / 1e3:10.1f} {(r['e'] - r['s']) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3s}  {r['n']}")
print('--- tail')
for r in last[-nshow:]:
    print(f"{(r['s'] - t0) / 1e3:10.1f} {(r['e'] - r['s']) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3s}  {r['n']}")
