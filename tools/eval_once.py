"""Three LML+gradient evaluations at one size (knobs from the environment): workload for a rocprofv3 --kernel-trace timeline."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib                                      # noqa: E402
from romcomma_amd.user.sample import bench_hyper, synthetic_fold   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
M = int(sys.argv[2]) if len(sys.argv) > 2 else 5
X, y = synthetic_fold(N, M)
gp = _lib.RcGP(X, y)
ell, var, noise = bench_hyper(M)
gp.set_hyper(ell, var, noise)
for _ in range(3):
    gp.stage_gram()                                # set_hyper with unchanged values keeps the factor: rebuild K instead
    v, g = gp.lml_grad()
print('lml', v, g[:2])
gp.close()
