"""Per-config timings on one MI355X for the BASELINE.json configs (single-GPU share of C3 / C4): one LML+gradient evaluation,
full fit (evaluation count stated), predict, Sobol. Writes profiles/r01_config_report.json when run on the GPU box with
--save (the JSON is copied back through gpurun_out/)."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from romcomma_amd import _lib                                     # noqa: E402
from romcomma_amd.gpr.optimize import fit_lbfgsb                  # noqa: E402
from romcomma_amd.user.sample import bench_hyper, synthetic_fold  # noqa: E402

CONFIGS = [('C0', 256, 3, 64), ('C1', 8192, 5, 8192), ('C2', 16384, 10, 0), ('C3 (one output of 8)', 8192, 10, 0),
           ('C4 (one fold of 8)', 28672, 20, 4096)]


def timed(f):
    t0 = time.perf_counter()
    out = f()
    return out, time.perf_counter() - t0


def main():
    rows = []
    for name, N, M, n_pred in CONFIGS:
        X, y = synthetic_fold(N, M)
        gp = _lib.RcGP(X, y)
        ell, var, noise = bench_hyper(M)
        gp.set_hyper(ell, var, noise)
        gp.lml_grad()                                             # warm-up: allocations, code objects
        gp.stage_gram()                                           # same hyper-parameters: rebuild K so that the factor is recomputed
        (_, _), t_eval = timed(gp.lml_grad)
        fit, t_fit = timed(lambda: fit_lbfgsb(gp, 5.0 * np.ones(M), 2.0, 0.02))
        slices = [(m, m + 1) for m in range(M)] + [(0, m + 1) for m in range(M)] + [(m + 1, M) for m in range(M)] + [(0, M)]
        gp.sobol_closed(slices[:1])
        V, t_sobol = timed(lambda: gp.sobol_closed(slices))
        row = {'config': name, 'N': N, 'M': M, 'eval_ms': 1e3 * t_eval, 'eval_TFLOPs': N ** 3 / t_eval / 1e12, 'fit_s': t_fit,
               'fit_evaluations': fit['nfev'], 'log_marginal': fit['log_marginal'], 'sobol_ms': 1e3 * t_sobol,
               'S_first_sum': float(np.sum(V[:M]) / V[-1]), 'fit_plus_sobol_s': t_fit + t_sobol,
               'train_points_per_s': N / (t_fit + t_sobol)}
        if n_pred:
            Xs, _ = synthetic_fold(n_pred, M, k=99)
            gp.predict(Xs[:128])
            (mean, sd), t_pred = timed(lambda: gp.predict(Xs))
            row.update(predict_points=n_pred, predict_ms=1e3 * t_pred, predict_sd_mean=float(np.mean(sd)))
        rows.append(row)
        print(json.dumps(row), flush=True)
        gp.close()
    if '--save' in sys.argv:
        out = ROOT / 'gpurun_out' / 'config_report.json'
        out.parent.mkdir(exist_ok=True)
        json.dump(rows, open(out, 'w'), indent=1)


if __name__ == '__main__':
    main()
