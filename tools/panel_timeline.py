"""Panel-level timeline of the last Cholesky in a rocprofv3 --kernel-trace CSV (workload: tools/potrf_once.py): for every outer
panel (NB / 128 diagonal kernels) when its chain started and ended, and when the outer updates that follow it (window pieces =
k_gemm_nt_sub launches longer than `long_us`, bulk = k_syrk_lower) ran. Shows how far the chain runs ahead of the bulk and
how much of the factorisation is the chain-bound tail.

    python tools/panel_timeline.py <kernel_trace.csv> [NB=1024] [long_us=60]
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
long_us = float(sys.argv[3]) if len(sys.argv) > 3 else 60.0
for r in rows:
    r['s'] = int(r['Start_Timestamp'])
    r['e'] = int(r['End_Timestamp'])
    r['n'] = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]
rows.sort(key=lambda r: r['s'])
g = max(i for i, r in enumerate(rows) if r['n'].startswith('k_gram'))
last = [r for r in rows[g + 1:] if not r['n'].startswith('k_lml')]
t0 = last[0]['s']
us = lambda t: (t - t0) / 1e3
end = max(r['e'] for r in last)
print(f'pass: {len(last)} kernels, span {us(end):.0f} us')
diag = [r for r in last if r['n'].startswith('k_diag')]
per = NB // 128
bulk = [r for r in last if r['n'].startswith('k_syrk_lower')]
piece = [r for r in last if r['n'].startswith('k_gemm_nt_sub') and (r['e'] - r['s']) / 1e3 > long_us]
print(f'{len(diag)} diagonal kernels, {len(bulk)} bulk kernels, {len(piece)} long k_gemm_nt_sub launches')
print(' panel  chain start      end    dur   wait-before |  bulk start      end    dur |  long gemm_nt_sub busy in [chain start, next chain start)')
prev_end = 0.0
for p in range(0, len(diag), per):
    d = diag[p:p + per]
    cs, ce = us(d[0]['s']), us(d[-1]['e'])
    nxt = us(diag[p + per]['s']) if p + per < len(diag) else us(end)
    b = bulk[p // per] if p // per < len(bulk) else None
    pb = sum((r['e'] - r['s']) / 1e3 for r in piece if cs <= us(r['s']) < nxt)
    bs = f"{us(b['s']):10.0f} {us(b['e']):8.0f} {(b['e'] - b['s']) / 1e3:6.0f}" if b else ' ' * 26
    print(f'{p // per:6d} {cs:12.0f} {ce:8.0f} {ce - cs:6.0f} {cs - prev_end:10.0f}    | {bs} | {pb:8.0f}')
    prev_end = ce
busy = sum((r['e'] - r['s']) / 1e3 for r in last if r['n'].startswith(('k_syrk_lower', 'k_gemm_nt_sub')))
print(f'sum of k_syrk_lower + k_gemm_nt_sub durations: {busy:.0f} us (they overlap each other)')
if bulk:
    lb = us(bulk[-1]['e'])
    print(f'last bulk kernel ends at {lb:.0f} us; chain-bound tail after it: {us(end) - lb:.0f} us')
