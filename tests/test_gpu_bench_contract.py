"""bench.py prints ONE JSON line with the fields the driver reads (the task's bench contract), here at a size that runs in seconds."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_bench_json_contract(gpu):
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--rows', '1024', '--dims', '3'],
                         capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
                'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'f64' and d['data'] == 'synthetic' and d['unit'] == 'train-points/s'
    assert 'workload' in d['config'] and 'model' not in d['config']
    assert d['value'] == pytest.approx(1024 * 2 / (d['ms_per_step'] * 2e-3), rel=1e-9)
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
    assert r['frac'] == pytest.approx(r['achieved'] / r['peak'], rel=1e-12) and r['achieved'] > 0
    assert 'traffic' in r                                    # None away from the profiled C2 size
    ch = d['stages']['cholesky']
    assert ch['ms'] > 0 and ch['frac'] == pytest.approx(ch['achieved_TFLOPs'] / r['peak'], rel=1e-12)
    c = d['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and c['unit'] == d['unit'] and c['sample']
