"""L^-1 (rc_trtri) and whole-evaluation time at given sizes; RCGP_DEV_LIB selects a dev build (tools/build_dev_lib.sh), e.g. one with
-DRC_TRTRI_HALF_TILES=0 (no half-tile launches). Prints the LML and the first gradient entries so that builds can be compared.

    gpurun -- python tools/trtri_time.py 8192 5  16384 10
"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib                                      # noqa: E402
if os.environ.get('RCGP_DEV_LIB'):
    _lib.LIB_PATH = Path(os.environ['RCGP_DEV_LIB']).resolve()
from romcomma_amd.user.sample import bench_hyper, synthetic_fold   # noqa: E402


def run(N, M, reps=5):
    X, y = synthetic_fold(N, M)
    gp = _lib.RcGP(X, y)
    gp.set_hyper(*bench_hyper(M))
    gp.stage_gram()
    v, g = gp.lml_grad()                           # allocates L^-1 and its scratch
    gp.stage_gram(); gp.stage_potrf(); gp.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        gp.stage_trtri()
    gp.sync()
    tt = (time.perf_counter() - t0) / reps
    te = []
    for _ in range(reps):
        gp.stage_gram()
        t1 = time.perf_counter()
        v, g = gp.lml_grad()
        te.append(time.perf_counter() - t1)
    gp.close()
    print(f'N={N} M={M} lib={os.environ.get("RCGP_DEV_LIB", "product")}: L^-1 {tt * 1e3:7.3f} ms = {N ** 3 / 3 / tt / 1e12:5.1f} TF/s   '
          f'eval {1e3 * min(te):7.2f} ms   lml {v:.13e} grad {g[0]:.10e} {g[-1]:.10e}', flush=True)


if __name__ == '__main__':
    a = [int(x) for x in sys.argv[1:]] or [8192, 5]
    for N, M in zip(a[::2], a[1::2]):
        run(N, M)
