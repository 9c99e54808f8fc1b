"""Copy the summaries of one tools/profile_eval.sh run (gpurun_out/prof_<tag>/) into profiles/<tag>_*.

    python tools/collect_eval_profiles.py r02_c4
"""
import glob
import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1]
src = ROOT / 'gpurun_out' / f'prof_{tag}'
dst = ROOT / 'profiles'
shutil.copy(max(glob.glob(str(src / 'eval/*/*_kernel_stats.csv')), key=os.path.getmtime), dst / f'{tag}_one_eval_kernel_stats.csv')
subprocess.check_call([sys.executable, str(ROOT / 'tools' / 'pmc_summary.py'), str(src / 'pmc_FETCH_SIZE'), str(src / 'pmc_WRITE_SIZE'),
                       str(src / 'pmc_SQ_VALU_MFMA_BUSY_CYCLES'), str(dst / f'{tag}_pmc.json')])
