"""In-kernel duration of every diagonal-block kernel of one C2 factorisation (wall_clock64 at entry and exit), to tell "the kernel waited for
a place on the chip" from "the kernel ran slowly beside the update kernels": compare with the durations of the same kernels in a rocprofv3
kernel trace (dispatch to completion). Dev build only: make -C rom-comma_amd/csrc clean all EXTRA=-DRC_DIAG_TIMING
    gpurun -- 'make -C rom-comma_amd/csrc clean all EXTRA=-DRC_DIAG_TIMING && python tools/diag_spans.py 16384 10'"""
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

N, M = int(sys.argv[1]), int(sys.argv[2])
X, y = synthetic_fold(N, M)
gp = _lib.RcGP(X, y)
gp.set_hyper(*bench_hyper(M))
gp.lml()
gp.stage_gram(); gp.stage_potrf(); gp.sync()
buf = (ctypes.c_longlong * 1024)()
_lib.load().rcgp_debug_diag_spans(buf)
t = np.array(buf[:], dtype=np.int64).reshape(512, 2)[: N // 128]
dur = (t[:, 1] - t[:, 0]) / 100.0
start = (t[:, 0] - t[0, 0]) / 100.0
step = np.diff(start)
for p in range(0, N // 128, 8):
    print(f'panel {p // 8:2d}: in-kernel us ' + ' '.join(f'{d:6.1f}' for d in dur[p:p + 8]) + '   | start-to-start us ' +
          ' '.join(f'{d:6.0f}' for d in step[p:p + 8]))
print(f'in-kernel: median {np.median(dur):.1f} max {dur.max():.1f} us; start-to-start: median {np.median(step):.1f} max {step.max():.1f} us; '
      f'first start to last exit {(t[-1, 1] - t[0, 0]) / 100.0:.0f} us')
gp.close()
