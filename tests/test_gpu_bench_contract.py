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
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and c['unit'] == d['unit'] and c['sample'] and c['blas']
    assert c['configs']['bench']['evaluation_s'] > 0 and c['configs']['bench']['N'] == 1024        # timed AT the configuration, not extrapolated
    assert d['roofline']['stages']['cholesky']['frac'] == ch['frac'] and d['roofline']['stages']['gram']['bound'] == 'hbm'
    assert d['config']['lbfgs_evaluations_timed_steps'] >= 2 * 5


def test_bench_under_torch_distributed_run_exercises_rccl(gpu):
    """The driver launches N > 1 as `python -m torch.distributed.run ... bench.py`; on the one-GPU box the same launcher with one rank and
    --force-dist initialises the RCCL process group and runs the real collectives (barrier, all-reduce MAX, all_gather_into_tensor)."""
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port', str(port),
           str(ROOT / 'bench.py'), '--gpus', '1', '--steps', '1', '--warmup', '1', '--rows', '1024', '--dims', '3', '--force-dist', '--no-cpu-baseline']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith('{')]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['value'] > 0 and 'cpu_baseline' not in d



def test_bench_two_ranks_rehearsal_on_one_gpu(gpu):
    """The N = 2 path of bench.py end to end -- torch.distributed.run with two ranks, fold (r + step) mod 8 per rank, barrier, MAX over
    ranks, the gather of both ranks' rows, value = 2 N steps / time -- rehearsed with both ranks on the one GPU of this box over gloo
    (on an 8-GPU node the driver launches the same command with the default backend: RCCL over xGMI, one rank per GPU)."""
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(port),
           str(ROOT / 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--rows', '1024', '--dims', '3', '--backend', 'gloo']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith('{')]
    assert len(lines) == 1, out.stdout                       # rank 0 alone prints
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and 'cpu_baseline' not in d
    assert d['value'] == pytest.approx(2 * 1024 * 2 / (d['ms_per_step'] * 2e-3), rel=1e-9)
    assert 'x2' in d['config']['parallelism']
