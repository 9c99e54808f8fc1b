"""For a kernel trace of a factorisation with the resident diagonal kernel (RCGP_DLOOP=1): the chain's step interval per outer panel (from
the k_prep2 start times) and the slowest chain kernels.   python tools/dloop_steps.py trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['n'] = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]
rows.sort(key=lambda r: r['s'])
g = max(i for i, r in enumerate(rows) if r['n'].startswith('k_gram'))
last = rows[g:]
t0 = last[0]['s']
p2 = [r for r in last if r['n'] == 'k_prep2']
print(f'span {(max(r["e"] for r in last) - t0) / 1e3:.0f} us, {len(p2)} k_prep2 launches')
for p in range(0, len(p2), 8):
    chunk = p2[p:p + 9]
    if len(chunk) > 1:
        steps = [(b['s'] - a['s']) / 1e3 for a, b in zip(chunk, chunk[1:])]
        print(f'panel {p // 8:2d}: starts {(chunk[0]["s"] - t0) / 1e3:8.0f} us; step intervals ' + ' '.join(f'{x:6.0f}' for x in steps))
for name in ('k_prep1g', 'k_prep1', 'k_prep2', 'k_trsm_panel', 'k_trsm_panel_pre', 'k_gemm_nt_sub_k128', '__amd_rocclr_streamOpsWrite', '__amd_rocclr_streamOpsWait'):
    d = sorted((r['e'] - r['s']) / 1e3 for r in last if r['n'] == name)
    if d:
        print(f'{name:30s} n={len(d):4d} median {d[len(d) // 2]:7.1f} p90 {d[int(len(d) * 0.9)]:7.1f} max {d[-1]:8.1f}  sum {sum(d):9.0f}')
