"""One LML+gradient evaluation and one Sobol pass at the C2 size with the fixed benchmark hyper-parameters: the workload
profiled by rocprofv3 (--kernel-trace --stats, and the --pmc passes for HBM traffic / MFMA busy cycles)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib                      # noqa: E402
from romcomma_amd.user.sample import bench_hyper, synthetic_fold   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
M = int(sys.argv[2]) if len(sys.argv) > 2 else 10
X, y = synthetic_fold(N, M)
gp = _lib.RcGP(X, y)
ell, var, noise = bench_hyper(M)
gp.set_hyper(ell, var, noise)
lml, grad = gp.lml_grad()
first = [(m, m + 1) for m in range(M)]
V = gp.sobol_closed(first + [(0, m + 1) for m in range(M)] + [(m + 1, M) for m in range(M)])
print('lml', lml, 'grad0', grad[0], 'S0', V[0] / V[2 * M - 1])
gp.close()
