"""k_grad under the two tile orders (RCGP_GRAD_ORDER): time per launch (HIP events of the library's profiling bracket) and identical results."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib                                      # noqa: E402
from romcomma_amd.user.sample import bench_hyper, synthetic_fold   # noqa: E402

N, M = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16384, 10)
X, y = synthetic_fold(N, M)
out = {}
for order in ('0', '1', '0', '1'):
    os.environ['RCGP_GRAD_ORDER'] = order
    gp = _lib.RcGP(X, y)
    ell, var, noise = bench_hyper(M)
    gp.set_hyper(ell, var, noise)
    gp.lml_grad()
    gp.set_profiling(True)
    for i in range(4):
        gp.set_hyper(ell * (1 + 1e-3 * (i + 1)), var, noise)
        v, g = gp.lml_grad()
    n, ms, work = gp.profile_get(_lib.K_GRAD)
    print(f'order {order}: k_grad {ms / n:.3f} ms = {work / n / (ms / n) / 1e9:.1f} TFLOP/s   lml {v:.12e} g0 {g[0]:.12e}', flush=True)
    out.setdefault(order, (v, tuple(g)))
    assert out[order] == (v, tuple(g))
    gp.close()
assert out['0'] == out['1'], 'tile order changed the result'
print('bit-identical across orders')
