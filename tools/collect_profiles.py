"""Copy the summaries of one tools/profile_c2.sh run (gpurun_out/prof_<tag>/) into profiles/ under the round's names.

    python tools/collect_profiles.py r01b r01
"""
import glob
import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag, name = sys.argv[1], sys.argv[2]
src = ROOT / 'gpurun_out' / f'prof_{tag}'
dst = ROOT / 'profiles'


def one(pattern):
    return max(glob.glob(str(src / pattern)), key=os.path.getmtime)      # (an earlier run's files may lie beside it)


shutil.copy(one('bench/*/*_kernel_stats.csv'), dst / f'{name}_bench_c2_kernel_stats.csv')
shutil.copy(one('eval/*/*_kernel_stats.csv'), dst / f'{name}_one_eval_c2_kernel_stats.csv')
shutil.copy(src / 'bench.json', dst / f'{name}_bench_c2.json')
shutil.copy(src / 'bench_under_rocprof.json', dst / f'{name}_bench_c2_under_rocprof.json')
subprocess.check_call([sys.executable, str(ROOT / 'tools' / 'pmc_summary.py'), str(src / 'pmc_FETCH_SIZE'), str(src / 'pmc_WRITE_SIZE'),
                       str(src / 'pmc_SQ_VALU_MFMA_BUSY_CYCLES'), str(dst / f'{name}_pmc_c2.json')])
