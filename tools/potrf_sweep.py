"""Cholesky timing sweep over the run-time knobs (read at rcgp_create): RCGP_FINE, RCGP_NB, RCGP_EXT, RCGP_RESERVE_CUS.
Prints potrf time (Gram time subtracted) and the LML (must agree across knobs to ~1e-12 relative).

    gpurun -- python tools/potrf_sweep.py 8192 5  16384 10
"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from romcomma_amd import _lib                                      # noqa: E402
from romcomma_amd.user.sample import bench_hyper, synthetic_fold   # noqa: E402

KNOBS = [
    {'RCGP_FINE': '0'},
    {'RCGP_SPLIT': '0', 'RCGP_EXT': '2', 'RCGP_DEPTH': '4'},
    {},
    {'RCGP_DEPTH': '1'},
    {'RCGP_DEPTH': '4'},
    {'RCGP_EXT': '2'},
    {'RCGP_EXT': '6'},
    {'RCGP_NB': '256'},
    {'RCGP_NB': '512'},
    {'RCGP_EXTEV': '0'},
    {'RCGP_PSPLIT': '0'},
    {'RCGP_PSPLIT': '1'},
    {'RCGP_DIAG': '1'},
    {},
]


def run(N, M, knobs, reps=5):
    saved = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        X, y = synthetic_fold(N, M)
        gp = _lib.RcGP(X, y)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
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
    args = [int(a) for a in sys.argv[1:]] or [8192, 5, 16384, 10]
    if os.environ.get('SWEEP_KNOBS'):                               # e.g. SWEEP_KNOBS='[{}, {"RCGP_PSPLIT": "2"}, {}]'
        import json
        KNOBS = json.loads(os.environ['SWEEP_KNOBS'])
    for i in range(0, len(args), 2):
        for kn in KNOBS:
            run(args[i], args[i + 1], kn)
