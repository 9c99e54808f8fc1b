"""One row per run of tools/bench_units.sh (gpurun_out/r04_bench_n<N>_units<U>.json) -> profiles/r04_units_per_gpu_bench.json; the C3-share
bench line and the config report are copied beside it."""
import glob
import json
import os
import shutil
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def line(path):
    return json.loads([ln for ln in open(path) if ln.startswith('{')][0])


rows = []
for path in glob.glob(str(ROOT / 'gpurun_out' / 'r04_bench_n*_units*.json')):
    d = line(path)
    c = d['config']
    rows.append({'N': c['N'], 'M': c['M'], 'units_per_gpu': c['units_per_gpu'], 'train_points_per_s': d['value'], 'ms_per_step': d['ms_per_step'],
                 'evaluations_timed_steps': c['lbfgs_evaluations_timed_steps'], 'rounds_timed_steps': c.get('evaluation_rounds_timed_steps'),
                 'ms_per_step_host_only': c['ms_per_step_host_only'], 'ms_per_step_inside_library_calls': c['ms_per_step_inside_library_calls'],
                 'command': f"python bench.py --no-host-api --no-cpu-baseline --steps {d['steps']} --warmup {d['warmup']} --shard outputs --rows {c['N']} "
                            f"--dims {c['M']} --units-per-gpu {c['units_per_gpu']}"})
rows.sort(key=lambda r: (r['N'], r['units_per_gpu']))
base = {r['N']: r['train_points_per_s'] for r in rows if r['units_per_gpu'] == 1}
for r in rows:
    r['gain_over_one_unit'] = r['train_points_per_s'] / base[r['N']] if r['N'] in base else None
    print(r['N'], r['M'], r['units_per_gpu'], round(r['train_points_per_s']), round(r['ms_per_step'], 1), 'host', round(r['ms_per_step_host_only'], 1),
          'gain', None if r['gain_over_one_unit'] is None else round(r['gain_over_one_unit'], 2))
json.dump(rows, open(ROOT / 'profiles' / 'r04_units_per_gpu_bench.json', 'w'), indent=1)
json.dump(line(ROOT / 'gpurun_out' / 'r04_bench_c3_share.json'), open(ROOT / 'profiles' / 'r04_bench_c3_share.json', 'w'))
if os.path.exists(ROOT / 'gpurun_out' / 'config_report.json'):
    shutil.copy(ROOT / 'gpurun_out' / 'config_report.json', ROOT / 'profiles' / 'r04_config_report.json')
