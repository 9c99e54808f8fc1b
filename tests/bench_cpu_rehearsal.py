"""bench.py's N > 1 bookkeeping (fold schedule per rank, barrier, MAX over ranks, the gather, value = ranks x N x steps / time) run with the
device calls answered by the oracle, so that EIGHT ranks can be rehearsed over gloo on a box without eight GPUs (tests/test_dist.py).
TEST INFRASTRUCTURE: launched as  python -m torch.distributed.run --nproc-per-node 8 tests/bench_cpu_rehearsal.py --gpus 8 ... --backend gloo"""
import runpy
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
import oracle_backend          # noqa: E402

oracle_backend.install()
sys.argv[0] = str(ROOT / 'bench.py')
runpy.run_path(str(ROOT / 'bench.py'), run_name='__main__')
