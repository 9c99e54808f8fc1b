"""k_grad launch time (the library's own profiling brackets, HIP events riding on the dispatch) and the whole evaluation, by size:
    python tools/grad_time.py 8192 5 16384 10      (RCGP_DEV_LIB=tools/dev/<name>.so for a dev build)"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib                                      # noqa: E402
if os.environ.get('RCGP_DEV_LIB'):
    _lib.LIB_PATH = Path(os.environ['RCGP_DEV_LIB']).resolve()
from romcomma_amd.user.sample import bench_hyper, synthetic_fold   # noqa: E402

args = [int(a) for a in sys.argv[1:]] or [16384, 10]
for N, M in zip(args[::2], args[1::2]):
    X, y = synthetic_fold(N, M)
    gp = _lib.RcGP(X, y)
    gp.set_hyper(*bench_hyper(M))
    gp.lml_grad()
    gp.profile_sample(1)
    gp.profile_reset()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        gp.stage_gram()
        lml, g = gp.lml_grad()
    t1 = time.perf_counter()
    n, ms, work = gp.profile_get(5)
    print(f'N={N} M={M}: k_grad {ms / max(n, 1):8.3f} ms ({n} launches, {work / max(n, 1) / (ms / max(n, 1)) / 1e9:5.1f} TFLOP/s)  eval {1e3 * (t1 - t0) / reps:7.2f} ms'
          f'  lml {lml:.12e} grad {g[0]:.10e} {g[-1]:.10e}', flush=True)
    gp.close()
