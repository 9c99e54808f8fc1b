"""Reduce the rocprofv3 --pmc passes of tools/one_eval.py (FETCH_SIZE, WRITE_SIZE, MFMA busy) to per-kernel HBM traffic and
MFMA utilisation, applying the gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 128-B read
requests as 64 B for wide coalesced streams (16 B per lane -- every bulk load in these kernels is a double2), so the read
side is doubled; WRITE_SIZE is exact for 16-B-per-lane stores. Units: FETCH_SIZE / WRITE_SIZE are KB.

    python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES profiles/r01_pmc_c2.json
"""
import collections
import csv
import glob
import os
import json
import sys

GEMM = ('k_syrk_lower', 'k_gemm_nt_sub', 'k_gemm_nt_sub_h64', 'k_trsm_subst', 'k_prep2', 'k_trtri_T', 'k_trtri_X', 'k_grad', 'k_predict_var')


def rows(folder, name):
    newest = max(glob.glob(f'{folder}/*/*_{name}.csv'), key=os.path.getmtime)       # (an earlier run's files may lie beside it)
    return list(csv.DictReader(open(newest)))


def short(kernel_name):
    return kernel_name.split('(')[0].replace('void ', '').split('<')[0]


def main(fetch_dir, write_dir, busy_dir, out):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for folder in (fetch_dir, write_dir, busy_dir):
        seen = collections.defaultdict(set)
        for r in rows(folder, 'counter_collection'):
            k = short(r['Kernel_Name'])
            if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and folder != busy_dir:
                continue                                         # the clock reference is the one of the MFMA-busy pass
            per[k][r['Counter_Name']] += float(r['Counter_Value'])
            seen[k].add(r['Dispatch_Id'])
        for k, ids in seen.items():
            per[k]['launches'] = len(ids)
    for r in rows(busy_dir, 'kernel_trace'):
        per[short(r['Kernel_Name'])]['duration_ms_under_pmc'] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    summary = {}
    for k, v in per.items():
        read = 2.0 * v.get('FETCH_SIZE', 0.0) * 1024.0          # gfx950 correction: double the read side
        write = v.get('WRITE_SIZE', 0.0) * 1024.0
        gui = v.get('GRBM_GUI_ACTIVE', 0.0) / 8.0               # reported as the sum over the 8 XCDs
        summary[k] = {'launches': int(v['launches']), 'hbm_read_bytes': read, 'hbm_write_bytes': write,
                      'hbm_bytes_per_launch': (read + write) / max(v['launches'], 1),
                      'mfma_busy_frac': v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui * 1024.0) if gui > 0 else None,
                      'duration_ms_under_pmc': v.get('duration_ms_under_pmc')}
    gemm = [summary[k] for k in summary if k in GEMM]
    launches = sum(g['launches'] for g in gemm)
    summary['_gemm_family'] = {'launches': launches,
                               'hbm_bytes_per_launch': sum(g['hbm_read_bytes'] + g['hbm_write_bytes'] for g in gemm) / max(launches, 1),
                               'note': 'one LML+gradient evaluation at C2 (N=16384, M=10): tools/one_eval.py'}
    json.dump(summary, open(out, 'w'), indent=1, sort_keys=True)
    for k in sorted(summary, key=lambda k: -summary[k].get('hbm_bytes_per_launch', 0) * summary[k]['launches']):
        s = summary[k]
        print(f"{k:18s} n={s['launches']:4d} GB/launch={s['hbm_bytes_per_launch'] / 1e9:8.3f}"
              + (f" mfma_busy={s['mfma_busy_frac']:.2f}" if s.get('mfma_busy_frac') else ''))


if __name__ == '__main__':
    main(*sys.argv[1:5])
