"""Cholesky timing sweep over the run-time knobs RCGP_LOOKAHEAD, RCGP_FINE, RCGP_NB, RCGP_EXT, RCGP_DEPTH. The library reads them once
per process, so every (knob set, size) runs in a child process of its own, one after the other.
Prints potrf time (Gram time subtracted) and the LML (must agree across knobs to ~1e-12 relative).

    gpurun -- python tools/potrf_sweep.py 8192 5  16384 10
    SWEEP_KNOBS='[{}, {"RCGP_NB": "512"}]' python tools/potrf_sweep.py 8192 5
"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib                                      # noqa: E402
import os as _os
if _os.environ.get('RCGP_DEV_LIB'):
    _lib.LIB_PATH = Path(_os.environ['RCGP_DEV_LIB']).resolve()
from romcomma_amd.user.sample import bench_hyper, synthetic_fold   # noqa: E402

KNOBS = [
    {},
    {'RCGP_FINE': '0'},
    {'RCGP_LOOKAHEAD': '0'},
    {'RCGP_DEPTH': '1'},
    {'RCGP_DEPTH': '4'},
    {'RCGP_EXT': '2'},
    {'RCGP_EXT': '6'},
    {'RCGP_NB': '512'},
    {'RCGP_NB': '2048'},
    {},
]


def run(N, M, knobs, reps=5):
    X, y = synthetic_fold(N, M)
    gp = _lib.RcGP(X, y)
    ell, var, noise = bench_hyper(M)
    gp.set_hyper(ell, var, noise)
    lml = gp.lml()
    gp.stage_gram(); gp.stage_potrf(); gp.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        gp.stage_gram()
    gp.sync()
    t1 = time.perf_counter()
    for _ in range(reps):
        gp.stage_gram()
        gp.stage_potrf()
    gp.sync()
    t2 = time.perf_counter()
    tp = (t2 - t1) / reps - (t1 - t0) / reps
    gp.stage_gram()                                # (set_hyper with unchanged values keeps the factor: rebuild K instead)
    gp.lml_grad()                                  # first call allocates L^-1 and its scratch
    gp.stage_gram()
    t3 = time.perf_counter()
    v, g = gp.lml_grad()
    t4 = time.perf_counter()
    gp.close()
    print(f'N={N} M={M} {knobs}: potrf {tp * 1e3:7.2f} ms = {N ** 3 / 3 / tp / 1e12:5.1f} TF/s ({N ** 3 / 3 / tp / 78.6e12:.1%})'
          f'  eval {1e3 * (t4 - t3):6.1f} ms  lml {lml:.12e} grad0 {g[0]:.10e}', flush=True)


if __name__ == '__main__':
    import json
    import subprocess
    if sys.argv[1] == '--one':                                      # child: the knobs are in the environment
        run(int(sys.argv[2]), int(sys.argv[3]), json.loads(sys.argv[4]))
        sys.exit(0)
    args = [int(a) for a in sys.argv[1:]] or [8192, 5, 16384, 10]
    if os.environ.get('SWEEP_KNOBS'):                               # e.g. SWEEP_KNOBS='[{}, {"RCGP_NB": "512"}, {}]'
        KNOBS = json.loads(os.environ['SWEEP_KNOBS'])
    for i in range(0, len(args), 2):
        for kn in KNOBS:
            subprocess.run([sys.executable, __file__, '--one', str(args[i]), str(args[i + 1]), json.dumps(kn)], env={**os.environ, **kn}, check=True)
