"""Three Gram + Cholesky passes of a batch of units at one size (knobs from the environment): the workload for a rocprofv3 --kernel-trace
timeline of the batched panel chain (tools/chain_timeline.py reads the trace).   python tools/batch_potrf_once.py N M units [stage]
stage: 1 = Cholesky only (default), 2 = + L^-1 / alpha, 3 = whole evaluations."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib                                      # noqa: E402
from romcomma_amd.user.sample import bench_hyper, synthetic_outputs   # noqa: E402

N, M, U = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
stage = int(sys.argv[4]) if len(sys.argv) > 4 else 1
X, Y = synthetic_outputs(N, M, U)
gps = [_lib.RcGP(X, Y[:, u]) for u in range(U)]
ell, var, noise = bench_hyper(M)
for r in range(3):
    for u, gp in enumerate(gps):
        gp.set_hyper(ell * (1.0 + 0.05 * u + 0.01 * r), var, noise)
    if stage == 3:
        print(_lib.lml_grad_batch(gps)[0])
        continue
    _lib.stage_batch(0, gps)
    _lib.stage_batch(1, gps)
    if stage == 2:
        _lib.stage_batch(2, gps)
    gps[0].sync()
for gp in gps:
    gp.close()
