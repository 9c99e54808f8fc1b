"""Fit time and L-BFGS-B evaluation count of each of the 8 folds the bench hands round (synthetic_cv_fold(N, M, k)): how much of the
spread between bench steps is the fold, not the box.   gpurun -- python tools/fold_spread.py 16384 10"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib                                      # noqa: E402
from romcomma_amd.gpr.optimize import fit_lbfgsb                   # noqa: E402
from romcomma_amd.user.sample import synthetic_cv_fold             # noqa: E402

N, M = int(sys.argv[1]), int(sys.argv[2])
for k in range(8):
    X, y = synthetic_cv_fold(N, M, k)
    gp = _lib.RcGP(X, y)
    t0 = time.time()
    fit = fit_lbfgsb(gp, 5.0 * np.ones(M), 2.0, 0.02)
    print(f'fold {k}: X {X.shape}, {fit["nfev"]} evaluations, LML {fit["log_marginal"]:.3f}, {time.time() - t0:.2f} s', flush=True)
    gp.close()
