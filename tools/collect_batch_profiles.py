"""Copy the summaries of one tools/profile_batch.sh run (gpurun_out/prof_batch_<N>x<units>/) into profiles/r04_batch_eval_<N>x<units>_*:
the rocprofv3 kernel stats of three batched evaluations, and per kernel the MFMA-pipe busy fraction over three batched factorisations.

    python tools/collect_batch_profiles.py 8192 4"""
import collections
import csv
import glob
import json
import os
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
N, U = sys.argv[1], sys.argv[2]
src = ROOT / 'gpurun_out' / f'prof_batch_{N}x{U}'
dst = ROOT / 'profiles'
shutil.copy(max(glob.glob(str(src / 'eval/*/*_kernel_stats.csv')), key=os.path.getmtime), dst / f'r04_batch_eval_{N}x{U}_kernel_stats.csv')
per = collections.defaultdict(lambda: collections.defaultdict(float))
ids = collections.defaultdict(set)
for r in csv.DictReader(open(max(glob.glob(str(src / 'pmc_busy/*/*_counter_collection.csv')), key=os.path.getmtime))):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]
    per[k][r['Counter_Name']] += float(r['Counter_Value'])
    ids[k].add(r['Dispatch_Id'])
out = {}
for k, v in per.items():
    gui = v.get('GRBM_GUI_ACTIVE', 0.0) / 8.0                    # reported as the sum over the 8 XCDs
    if gui > 0 and v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) > 0:
        out[k] = {'launches': len(ids[k]), 'mfma_busy_frac': v['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024.0)}
json.dump({'N': int(N), 'units': int(U), 'workload': f'three batched factorisations: tools/batch_potrf_once.py {N} M {U} 1', 'kernels': out},
          open(dst / f'r04_batch_potrf_{N}x{U}_mfma_busy.json', 'w'), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]['launches']):
    print(f"{k:22s} n={v['launches']:5d} mfma_busy={v['mfma_busy_frac']:.2f}")
