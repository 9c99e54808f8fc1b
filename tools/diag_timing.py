"""Phase breakdown of the diagonal-block kernel (k_diag_factor) from in-kernel wall_clock64 stamps. Needs a dev build of the
library: make -C rom-comma_amd/csrc clean all EXTRA=-DRC_DIAG_TIMING (never the shipped build)."""
import ctypes
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib                                      # noqa: E402
import os
if os.environ.get('RCGP_DEV_LIB'):
    _lib.LIB_PATH = Path(os.environ['RCGP_DEV_LIB']).resolve()
from romcomma_amd.user.sample import bench_hyper, synthetic_fold   # noqa: E402

N, M = 1024, 5
X, y = synthetic_fold(N, M)
gp = _lib.RcGP(X, y)
gp.set_hyper(*bench_hyper(M))
gp.lml()
gp.stage_gram(); gp.stage_potrf(); gp.sync()
lib = _lib.load()
buf = (ctypes.c_longlong * 32)()
rc = lib.rcgp_debug_diag_times(buf)
t = np.array(buf[:], dtype=np.int64)
us = (t - t[0]) / 100.0            # wall_clock64 ticks at 100 MHz
names = {1: 'load', 2: 'pivot block 0'}
for c in range(8):
    names[3 + 2 * c] = f'panel {c}'
    names[4 + 2 * c] = f'trailing {c} + pivot {c + 1}'
names.update({19: 'store L, logdiag, diagonal-block inverses, w_j'})
print('pivot block 0 inside (pivot16): load rows %.2f, 16 pivots + scaling %.2f, publish L and its transpose %.2f, inverse %.2f, store it %.2f us'
      % tuple((t[b] - t[a]) / 100.0 for a, b in ((24, 25), (25, 26), (26, 27), (27, 28), (28, 30))))
prev = 0.0
for i in sorted(names):
    if t[i] == 0:
        continue
    print(f'{names[i]:28s} +{us[i] - prev:7.2f} us   (t = {us[i]:7.2f})')
    prev = us[i]
gp.close()
