"""Host enqueue time of the Cholesky against its GPU time: stage_potrf() returns when everything is queued, sync() when it has run.
If the two are close the schedule is launch-bound. Usage: python tools/enqueue_time.py [N M]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib                                      # noqa: E402
from romcomma_amd.user.sample import bench_hyper, synthetic_fold   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
M = int(sys.argv[2]) if len(sys.argv) > 2 else 10
X, y = synthetic_fold(N, M)
gp = _lib.RcGP(X, y)
gp.set_hyper(*bench_hyper(M))
for i in range(6):
    gp.stage_gram()
    gp.sync()
    t0 = time.perf_counter()
    gp.stage_potrf()
    t1 = time.perf_counter()
    gp.sync()
    t2 = time.perf_counter()
    print(f'N={N}: enqueue {1e3 * (t1 - t0):7.2f} ms, done {1e3 * (t2 - t0):7.2f} ms', flush=True)
print('lml', gp.lml())
gp.close()
