"""Stand-alone stage times of a batch of units at one size, knobs from the environment (RCGP_*: read once per process, so one process per
knob set), and a bit-level fingerprint of the results (LML and gradient as hex floats: the factorisation's schedule must not change them).

    python tools/batch_potrf.py N M units [reps]"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib                                      # noqa: E402
from romcomma_amd.user.sample import bench_hyper, synthetic_outputs   # noqa: E402

N, M, U = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
REPS = int(sys.argv[4]) if len(sys.argv) > 4 else 5
X, Y = synthetic_outputs(N, M, U)
gps = [_lib.RcGP(X, Y[:, u]) for u in range(U)]
ell, var, noise = bench_hyper(M)
for u, gp in enumerate(gps):
    gp.set_hyper(ell * (1.0 + 0.05 * u), var * (1.0 + 0.1 * u), noise * (1.0 + 0.5 * u))
lml, grad, status = _lib.lml_grad_batch(gps)
out = {'N': N, 'M': M, 'units': U, 'knobs': {k: v for k, v in os.environ.items() if k.startswith('RCGP_')},
       'fingerprint': [float(lml[u]).hex() + ' ' + ' '.join(float(g).hex() for g in grad[u]) for u in range(U)]}
for stage, name in ((1, 'potrf_ms'), (2, 'trtri_alpha_ms')):
    ms = []
    for r in range(REPS):
        _lib.stage_batch(0, gps)
        if stage == 2:
            _lib.stage_batch(1, gps)
        gps[0].sync()
        t0 = time.perf_counter()
        _lib.stage_batch(stage, gps)
        gps[0].sync()
        ms.append(1e3 * (time.perf_counter() - t0))
    out[name] = min(ms)
ms = []
for r in range(REPS):
    for u, gp in enumerate(gps):
        gp.set_hyper(ell * (1.0 + 0.05 * u + 0.01 * (r % 2)), var * (1.0 + 0.1 * u), noise * (1.0 + 0.5 * u))
    gps[0].sync()
    t0 = time.perf_counter()
    _lib.lml_grad_batch(gps)
    ms.append(1e3 * (time.perf_counter() - t0))
out['evaluation_ms'] = min(ms)
out['potrf_TFLOPs'] = U * N ** 3 / 3.0 / (out['potrf_ms'] * 1e-3) / 1e12
print(json.dumps(out))
for gp in gps:
    gp.close()
