"""Timing of the Sobol standard-error terms (rcgp_sobol_error_terms) at the BASELINE sizes."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib
from romcomma_amd.user.sample import bench_hyper, synthetic_fold
for N, M in ((8192, 5), (16384, 10)):
    X, y = synthetic_fold(N, M)
    gp = _lib.RcGP(X, y)
    gp.set_hyper(*bench_hyper(M))
    sl = [(m, m + 1) for m in range(M)] + [(0, m + 1) for m in range(M)] + [(m + 1, M) for m in range(M)]
    V = gp.sobol_closed(sl)
    gp.sobol_error_terms(sl[:1])
    t0 = time.perf_counter(); pd_, sd, pm, sm = gp.sobol_error_terms(sl); t1 = time.perf_counter()
    W = 4 * (pd_ - sd)
    T = np.sqrt(np.abs(W)) / V[2 * M - 1]
    print(f'N={N} M={M}: error terms {1e3 * (t1 - t0):.1f} ms; S_first={np.round(V[:M] / V[2 * M - 1], 4)}; T_first(partial)={np.round(T[:M], 5)}')
    gp.close()
