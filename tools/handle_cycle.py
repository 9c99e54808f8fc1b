"""Soak test of the handle life cycle on the shared per-device stream set: many handles created, used and closed in turn, some of
them alive together, covariant and independent, small sizes. Prints a progress line every 25 handles (a stall shows as silence)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib                                      # noqa: E402

rng = np.random.default_rng(0)
t0 = time.time()
n_cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ref = {}
for c in range(n_cycles):
    N = int(rng.choice([130, 260, 600, 1100]))
    M = 3
    L = int(rng.choice([1, 1, 2, 3]))
    key = (N, L)
    r2 = np.random.default_rng(hash(key) % 2 ** 32)
    X = r2.standard_normal((N, M))
    Y = np.sin(X @ r2.standard_normal((M, L))) + 0.1 * r2.standard_normal((N, L))
    ell = np.full((L, M), 1.3)
    F = np.eye(L) + 0.2 * (np.ones((L, L)) - np.eye(L))
    S = 0.05 * np.eye(L)
    Xs = r2.standard_normal((50, M))
    if L == 1:
        gp = _lib.RcGP(X, Y[:, 0])
        gp.set_hyper(ell[0], 1.0, 0.05)
        other = None
        if c % 5 == 0:                                   # a second handle alive beside it
            other = _lib.RcGP(X, Y[:, 0])
            other.set_hyper(ell[0] * 1.1, 1.0, 0.05)
            other.lml()
        v, g = gp.lml_grad()
        m, s = gp.predict(Xs)
        out = (v, float(np.sum(g)), float(np.sum(m)), float(np.sum(s)))
        if other is not None:
            other.close()
        gp.close()
    else:
        with _lib.RcMOGP(X, Y) as gp:
            gp.set_hyper(ell, F, S)
            v, gF, gell, gS = gp.lml_grad()
            m, s = gp.predict(Xs, c % 2 == 0)
            out = (v, float(np.sum(gell)), float(np.sum(m)), 0.0 if c % 2 else float(np.sum(s)))
            if c % 2:
                out = out[:3]
    k2 = key + (len(out),)
    if k2 in ref:
        assert ref[k2] == out, (c, k2, ref[k2], out)     # deterministic kernels: bit-identical from cycle to cycle
    else:
        ref[k2] = out
    if (c + 1) % 25 == 0:
        print(f'{c + 1} handles, {time.time() - t0:.1f} s', flush=True)
print('ok')
