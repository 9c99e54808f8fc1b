"""Time one LML + gradient evaluation of a covariant GP (L outputs, N rows each) next to an independent GP of the same system
size L * N, with the per-class HIP-event breakdown. Usage: python tools/mo_eval.py [N L M]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib                                      # noqa: E402
from romcomma_amd.user.sample import bench_hyper, synthetic_fold   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
M = int(sys.argv[3]) if len(sys.argv) > 3 else 10
X, y0 = synthetic_fold(N, M)
Y = np.stack([synthetic_fold(N, M, l=l)[1] if l else y0 for l in range(L)], axis=1)
ell, var, noise = bench_hyper(M)
ells = np.stack([ell * (1.0 + 0.1 * l) for l in range(L)])
C = np.tril(0.3 * np.ones((L, L)), -1) + np.eye(L)
F = var * C @ C.T
S = noise * (np.eye(L) + 0.1 * (np.ones((L, L)) - np.eye(L)))


def timed(gp, set_hyper, label, work):
    set_hyper(1.0)
    gp.lml_grad()
    ts = []
    for r in range(5):
        set_hyper(1.0 + 1e-9 * (r + 1))
        gp.sync()
        t0 = time.perf_counter()
        gp.lml_grad()
        ts.append(time.perf_counter() - t0)
    gp.set_profiling(True)
    gp.profile_reset()
    set_hyper(1.0 + 1e-8)
    gp.lml_grad()
    prof = {n: gp.profile_get(c) for c, n in enumerate(_lib.KERNEL_CLASS_NAMES)}
    gp.set_profiling(False)
    t = min(ts)
    print(f'{label}: evaluation {1e3 * t:7.2f} ms = {work / t / 1e12:5.1f} TFLOP/s of N^3 flops; '
          + ', '.join(f'{n} {v[1]:.1f} ms' for n, v in prof.items() if v[0]), flush=True)


with _lib.RcMOGP(X, Y) as gp:
    timed(gp, lambda s: gp.set_hyper(ells * s, F, S), f'covariant L={L} N={N} M={M}', float(L * N) ** 3)
Xb, yb = synthetic_fold(L * N, M)
with _lib.RcGP(Xb, yb) as gp:
    timed(gp, lambda s: gp.set_hyper(ell * s, var, noise), f'independent N={L * N} M={M}', float(L * N) ** 3)
