"""Batched units on one GPU (rcgp_lml_grad_batch): bit-identity against the single-handle call, and what a batch of 1, 2, 4 (8) units
costs per evaluation and per stage at one size.

    python tools/batch_check.py N M [max_units]

Prints one JSON object (also the source of profiles/r04_batch_*.json)."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib                                      # noqa: E402
from romcomma_amd.user.sample import bench_hyper, synthetic_outputs   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
M = int(sys.argv[2]) if len(sys.argv) > 2 else 10
U = int(sys.argv[3]) if len(sys.argv) > 3 else 4
REPS = 6

X, Y = synthetic_outputs(N, M, U)
gps = [_lib.RcGP(X, Y[:, u]) for u in range(U)]
ell, var, noise = bench_hyper(M)
thetas = [(ell * (1.0 + 0.05 * u), var * (1.0 + 0.1 * u), noise * (1.0 + 0.5 * u)) for u in range(U)]
other = [(e * 1.01, v, n) for e, v, n in thetas]


def set_all(ts):
    for gp, t in zip(gps, ts):
        gp.set_hyper(*t)


# 1. the single-handle call per unit, then the same points through batches of every size: equal to the last bit
set_all(thetas)
single = [gp.lml_grad() for gp in gps]
identical = {}
for nb in (1, 2, 3, 4, 8, 16):
    if nb > U:
        continue
    set_all(other)
    _lib.lml_grad_batch(gps[:nb])
    set_all(thetas)
    lml, grad, status = _lib.lml_grad_batch(gps[:nb])
    identical[nb] = bool(all(status[u] == 0 and lml[u] == single[u][0] and np.array_equal(grad[u], single[u][1]) for u in range(nb)))

# 2. wall time per batched evaluation (host included), alternating two points so that nothing is cached
timing = {}
for nb in (1, 2, 3, 4, 8, 16):
    if nb > U:
        continue
    ms = []
    for r in range(REPS):
        set_all(other if r % 2 == 0 else thetas)
        gps[0].sync()
        t0 = time.perf_counter()
        _lib.lml_grad_batch(gps[:nb])
        ms.append(1e3 * (time.perf_counter() - t0))
    stages = {}
    for stage, name in ((0, 'gram'), (1, 'potrf'), (2, 'trtri_alpha')):
        best = []
        for r in range(3):
            if stage >= 1:
                _lib.stage_batch(0, gps[:nb])
            if stage == 2:
                _lib.stage_batch(1, gps[:nb])
            gps[0].sync()
            t0 = time.perf_counter()
            _lib.stage_batch(stage, gps[:nb])
            gps[0].sync()
            best.append(1e3 * (time.perf_counter() - t0))
        stages[name] = min(best)
    timing[nb] = {'evaluation_ms': min(ms[1:]), 'evaluation_ms_all': ms, 'per_unit_ms': min(ms[1:]) / nb, **stages}
base = timing[1]['evaluation_ms']
for nb, t in timing.items():
    t['vs_one_unit'] = t['evaluation_ms'] / base
    t['TFLOPs'] = nb * float(N) ** 3 / (t['evaluation_ms'] * 1e-3) / 1e12
print(json.dumps({'N': N, 'M': M, 'bit_identical_to_single_call': identical, 'timing': timing}))
for gp in gps:
    gp.close()
