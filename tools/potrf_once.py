"""Three Gram + Cholesky passes at one size (knobs from the environment): the workload for a rocprofv3 --kernel-trace timeline
of the panel chain (tools/chain_timeline.py reads the trace)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib                                      # noqa: E402
import os as _os
if _os.environ.get('RCGP_DEV_LIB'):
    _lib.LIB_PATH = Path(_os.environ['RCGP_DEV_LIB']).resolve()
from romcomma_amd.user.sample import bench_hyper, synthetic_fold   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
M = int(sys.argv[2]) if len(sys.argv) > 2 else 5
X, y = synthetic_fold(N, M)
gp = _lib.RcGP(X, y)
gp.set_hyper(*bench_hyper(M))
for _ in range(3):
    gp.stage_gram()
    gp.stage_potrf()
    gp.sync()
print('lml', gp.lml())
gp.close()
