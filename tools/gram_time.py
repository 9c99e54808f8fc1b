"""Gram build alone (k_scale + k_gram) at one size: best host wall time around a sync over 20 launches, as GB/s of the algorithmic
bytes 8 (N (N + 1) / 2 + N M). Usage: python tools/gram_time.py [N M]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib                                      # noqa: E402
import os
if os.environ.get('RCGP_DEV_LIB'):
    _lib.LIB_PATH = Path(os.environ['RCGP_DEV_LIB']).resolve()
from romcomma_amd.user.sample import bench_hyper, synthetic_fold   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
M = int(sys.argv[2]) if len(sys.argv) > 2 else 10
X, y = synthetic_fold(N, M)
with _lib.RcGP(X, y) as gp:
    gp.set_hyper(*bench_hyper(M))
    gp.stage_gram()
    gp.sync()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        gp.stage_gram()
        gp.sync()
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    b = 8.0 * (N * (N + 1) / 2 + N * M)
    print(f'lib={os.environ.get("RCGP_DEV_LIB", "product")} N={N} M={M}: lml {gp.lml():.13e} gram {1e3 * t:.3f} ms = {b / t / 1e9:.0f} GB/s ({b / t / 8e12:.1%} of 8 TB/s)')
