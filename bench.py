"""Headline benchmark (BASELINE.json): GP fit + closed-form Sobol wall-time and train-points/s, fp64, one fold per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the hot path over one synthetic fold resident in HBM: L-BFGS-B hyper-parameter fit from the
reference's default start (ell 5.0, variance 2.0, noise 0.02; gpr/kernels.py:49-50, gpr/models.py:52) with the reference's
optimiser options (maxiter 5000, gtol 1e-16; gpr/models.py:327-330) to convergence, then the closed-form first-order /
closed / total Sobol indices (3M+1 quadratic forms), then the gather of every rank's indices (RCCL when N > 1).
Weak scaling: timed step s of rank r fits fold (r + s) mod 8 of an 8-fold split of one seeded dataset (every fold trains on N
rows, all folds a rank meets are resident in HBM beforehand); value = ranks * N *
steps / max-over-ranks wall time.
--units-per-gpu U (default 1 = the headline definition above): a step is U folds (or outputs) per GPU fitted AT ONCE -- lockstep L-BFGS-B,
one batched schedule per round of evaluations (rcgp_lml_grad_batch) -- and value counts U * N training rows per rank and step.
Outside the timed region, rank 0 of a one-GPU run also reports: the stand-alone Cholesky and L^-1 stages, a two-unit evaluation beside
a one-unit one (stages.units_at_once), the drop-in path from data.csv on disk through run.gpr + run.gsa (stages.host_api), and the CPU
oracle timed at the configuration with the parity of LML, gradient and Sobol variances at the fitted optimum (cpu_baseline,
parity_at_config).
Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: what RCCL needs between the ranks of one node on this driver

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X dense fp64 matrix peak (AMD spec). Sustained micro-benchmark: DESIGN.md.
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VALU_MEASURED_TFMAS = 35.0  # fp64 vector FMA instructions/s (x1e12, all lanes) sustained in a register-only loop: 70 TFLOP/s measured
SOBOL_VALU_OPS_PER_EXP = 24      # fp64 VALU instructions per exp of k_sobol_pairs: rc_exp (clamp 2, scale + rint 2, Cody-Waite 2, Horner 13,
                                 # ldexp + convert 2) + the exponent's fma / add and the weighted accumulation (3)
SOBOL_FRAC_SOURCE = ('valu_ops_per_exp is a hand count of the instructions of rc_exp + the accumulation in k_sobol_pairs; measured_valu_peak_Tops comes from '
                     'tools/fp64_latency.hip / tools/coissue.hip (register-only v_fma_f64 loop, run in round 3, NOT in this run): an estimate, unlike the '
                     'HIP-event based fractions beside it')


def all_slices(M):
    first = [(m, m + 1) for m in range(M)]
    closed = [(0, m + 1) for m in range(M)]
    complement = [(m + 1, M) for m in range(M)]
    return first + closed + complement + [(0, M)]


def sobol_indices(V, M):
    """First-order, closed and total indices from the 3M+1 conditional variances (gsa/models.py:207-214)."""
    full = V[3 * M]
    return np.concatenate([V[:M] / full, V[M:2 * M] / full, 1.0 - V[2 * M:3 * M] / full])


def fold_schedule(rank, warmup, steps, n_folds, units=1):
    """The fold(s) every step of one rank fits, warm-up steps first: timed step s takes fold (rank + s) mod n_folds, the warm-up steps
    the folds before `rank`. A K-fold cross-validation handed round the GPUs: one fold per GPU and step, and over the steps every
    rank meets the cheap and the expensive folds alike. With `units` > 1 a step is that many consecutive folds, fitted at once: a list
    of lists then."""
    if units == 1:
        return [(rank - warmup + s) % n_folds for s in range(warmup + steps)]
    return [[((rank - warmup + s) * units + j) % n_folds for j in range(units)] for s in range(warmup + steps)]


def pmc_traffic(N, M, kernel='k_grad'):
    """(HBM bytes per launch of `kernel`, the file they come from): the committed rocprofv3 --pmc passes of the same workload and build
    (profiles/r04_pmc_c2.json, falling back to earlier rounds; produced by tools/pmc_summary.py with the gfx950 FETCH_SIZE correction).
    Counters cannot be read inside this process, so the number is NOT measured in this run; (None, None) for any other size."""
    if (N, M) != (16384, 10):
        return None, None
    for name in ('r04_pmc_c2.json', 'r03_pmc_c2.json', 'r02_pmc_c2.json', 'r01_pmc_c2.json'):
        path = ROOT / 'profiles' / name
        if path.exists():
            try:
                return float(json.load(open(path))[kernel]['hbm_bytes_per_launch']), f'profiles/{name}'
            except Exception:
                continue
    return None, None


def _blas_info():
    try:
        from threadpoolctl import threadpool_info
        pools = [p for p in threadpool_info() if p.get('user_api') == 'blas'] or threadpool_info()
        cores = max([p.get('num_threads', 1) for p in pools] or [1])
        vendor = ', '.join(sorted({f"{p.get('internal_api', '?')} {p.get('version', '')}".strip() for p in pools})) or 'unknown'
        return int(cores), vendor
    except Exception:
        return int(os.cpu_count() or 1), 'unknown'


def _cpu_times(o, N, M, sobol_rows, fit=False, fold=None, theta=None, samples=1):
    """Wall times of the oracle on this host at one configuration: the LML+gradient evaluation fully timed at (N, M) (`samples` times: all
    reported, the fastest used); the 3M+1 Sobol quadratic forms on a stripe of `sobol_rows` rows against all N columns, scaled by
    N / sobol_rows (the work per row is uniform); with `fit`, a whole L-BFGS-B fit as well (small sizes only). `fold` = (X, y) and
    `theta` = (ell, var, noise) default to the seeded fold 0 and the fixed benchmark hyper-parameters."""
    X, y = o.synthetic_fold(N, M) if fold is None else fold
    ell, var, noise = (*o.bench_hyper(M)[:2], 1e-2) if theta is None else theta
    times, value = [], None
    for _ in range(max(samples, 1)):
        t0 = time.perf_counter()
        value = o.lml_and_grad_blas(X, y, ell, var, noise)
        times.append(time.perf_counter() - t0)
    out = {'N': N, 'M': M, 'evaluation_s': min(times), 'evaluation_samples_s': times}
    if fit:
        t0 = time.perf_counter()
        res = o.fit(X, y, 5.0 * np.ones(M))
        out['fit_s'] = time.perf_counter() - t0
        out['fit_evaluations'] = int(res['nfev'])
    alpha = o.k_inv_y(X, y, ell, var, noise)
    g, phi = o.sobol_prepare(X, alpha[None, :], np.array([var]), ell[None, :])
    rows = min(N, sobol_rows)
    t0 = time.perf_counter()
    o.sobol_V_pair(X, g[0], g[0], phi[0], phi[0], o.all_slices(M), rows=(0, rows))
    out['sobol_s'] = (time.perf_counter() - t0) * N / rows
    out['sobol_sample_rows'] = rows
    return out, value


def cpu_baseline(N, M, nfev_per_step, fold=None, theta=None, gpu_value=None, other_configs=False, gpu_sobol=None):
    """The oracle (NumPy/SciPy fp64, `oracle/gp_oracle.py`: LAPACK potrf + potri, BLAS-3 gradient sums) timed on this box's host
    cores AT the benchmark's configuration, on the fold of the last timed step and at the hyper-parameters the GPU fit ended on: one
    real LML+gradient evaluation at (N, M) (two samples at C2), times the evaluation count the GPU fit needed (the same SciPy driver
    would take the same path), plus the Sobol forms from a row stripe. The same call is the full-size parity check: its LML and
    gradient against the GPU's at that point (`parity_at_config`). SURVEY.md 8d's C0 and C1 are timed beside it. A reported baseline,
    not the target; conservative: GPflow autodiff does more work per evaluation."""
    from oracle import gp_oracle as o
    cores, vendor = _blas_info()
    o.lml_and_grad_blas(*o.synthetic_fold(512, M), *o.bench_hyper(M)[:2], 1e-2)      # BLAS threads up, pages touched: not timed
    configs = {}
    is_c2 = (N, M) == (16384, 10)
    if is_c2 and other_configs:                       # (--cpu-configs: SURVEY 8d's C0 and C1 beside C2; they keep the GPU idle for ~15 s)
        configs['C0'] = _cpu_times(o, 256, 3, 256, fit=True)[0]
        configs['C1'] = _cpu_times(o, 8192, 5, 256)[0]
    main, (lml_cpu, grad_cpu) = _cpu_times(o, N, M, 256 if N > 4096 else N, fold=fold, theta=theta, samples=2 if is_c2 else 1)
    configs['C2' if is_c2 else 'bench'] = main
    t_full = nfev_per_step * main['evaluation_s'] + main['sobol_s']
    out = {'value': N / t_full, 'unit': 'train-points/s', 'cores': cores, 'kind': 'port', 'blas': vendor,
           'sample': f'oracle.lml_and_grad_blas at the full N={N}, M={M} on the last timed fold at the GPU\'s fitted hyper-parameters, '
                     f'{len(main["evaluation_samples_s"])} sample(s) {["%.2f" % t for t in main["evaluation_samples_s"]]} s, fastest x {nfev_per_step:.1f} evaluations per '
                     f'fit (the GPU fit\'s count) + {3 * M + 1} Sobol quadratic forms timed on a {main["sobol_sample_rows"]}-row stripe x N/{main["sobol_sample_rows"]} '
                     f'({main["sobol_s"]:.1f} s); fit+Sobol wall time {t_full:.1f} s on {cores} threads ({vendor})',
           'configs': configs}
    if gpu_value is not None:
        lml_gpu, grad_gpu = gpu_value
        scale = float(np.max(np.abs(grad_cpu)))
        out['parity_at_config'] = {'lml_gpu': float(lml_gpu), 'lml_cpu': float(lml_cpu), 'rel': abs(float(lml_gpu) - float(lml_cpu)) / abs(float(lml_cpu)),
                                   'grad_max_rel': float(np.max(np.abs(np.asarray(grad_gpu) - grad_cpu)) / scale),
                                   'at': 'the last timed fold, the hyper-parameters its GPU fit converged to; gradient w.r.t. (lengthscales, variance, noise), '
                                         'error relative to the largest component'}
    if gpu_sobol is not None:
        # The second half of the metric at full size: four conditional variances (first-order 0, closed [0, M/2), complement [M/2, M), full;
        # gsa/calibrators.py:60-80) from the oracle's O(N^2) pair form over ALL rows, against the GPU's at the same hyper-parameters.
        slices, V_gpu = gpu_sobol
        X, y = fold
        t0 = time.perf_counter()
        alpha = o.k_inv_y(X, y, *theta)
        g, phi = o.sobol_prepare(X, alpha[None, :], np.array([theta[1]]), np.asarray(theta[0])[None, :])
        V_cpu = np.asarray(o.sobol_V_pair(X, g[0], g[0], phi[0], phi[0], slices))
        S_cpu, S_gpu = V_cpu[:-1] / V_cpu[-1], np.asarray(V_gpu)[:-1] / V_gpu[-1]
        out.setdefault('parity_at_config', {}).update({
            'sobol_slices': [list(map(int, sl)) for sl in slices], 'sobol_V_gpu': [float(v) for v in V_gpu], 'sobol_V_cpu': [float(v) for v in V_cpu],
            'sobol_rel': float(np.max(np.abs(np.asarray(V_gpu) - V_cpu) / np.abs(V_cpu))), 'index_abs': float(np.max(np.abs(S_gpu - S_cpu))),
            'sobol_cpu_s': time.perf_counter() - t0})
    return out


def host_api_leg(N, M, device=0):
    """The drop-in path a user of the reference runs, outside the timed headline: ONE fold from data.csv on disk through Repository -> Fold
    -> run.gpr(is_read=False, is_isotropic=False) -> run.gsa(kinds=ALL) (user/run.py:35-158), wall time split into the fit
    (HipGP.calibrate: L-BFGS-B + the parameter csv writes), test() (predict on the held-out rows + test.csv), the three GSA kinds, and
    everything else (reading the fold's csv files, building the model stores) -- beside the bare fit + Sobol on the same arrays through
    the C ABI, and with the library's own count of factorisations (the reference factors at least eight times besides its fit:
    gpr/models.py:365, 370, 439; gsa/calibrators.py:126-127 for each of three kinds)."""
    import shutil
    import tempfile
    import pandas as pd
    from romcomma_amd import _lib
    from romcomma_amd.data.storage import Fold, Repository
    from romcomma_amd.gpr.models import HipGP
    from romcomma_amd.gpr.optimize import fit_lbfgsb
    from romcomma_amd.gsa.models import GSA
    from romcomma_amd.user import run
    K = 8
    rows = int(round(N * K / (K - 1)))
    # the bench's own seeded dataset (already in the normalised units a Fold hands to the GP: inputs through the probit, output z-scored), so
    # that the fit behaves like the timed ones; the folds are therefore cut with is_normalization_applicable=False
    from romcomma_amd.user.sample import synthetic_fold
    Xall, yall = synthetic_fold(rows, M)
    columns = pd.MultiIndex.from_tuples([('X', f'X.{m}') for m in range(M)] + [('Y', 'Y.0')])
    root = Path(tempfile.mkdtemp(prefix='rcgp_bench_'))
    timers = {'fit_s': 0.0, 'test_s': 0.0}
    keep = (HipGP.calibrate, HipGP.test)

    def timed(name, method):
        def wrapper(self, *a, **kw):
            t = time.perf_counter()
            try:
                return method(self, *a, **kw)
            finally:
                timers[name] += time.perf_counter() - t
        return wrapper
    try:
        t0 = time.perf_counter()
        repo = Repository.from_df(root / 'repo', pd.DataFrame(np.concatenate([Xall, yall[:, None]], axis=1), columns=columns)).into_K_folds(
            -K, is_normalization_applicable=False, seed=1)
        setup_s = time.perf_counter() - t0
        HipGP.calibrate, HipGP.test = timed('fit_s', keep[0]), timed('test_s', keep[1])
        before = _lib.stat()
        t0 = time.perf_counter()
        fold = Fold(repo, 0)
        with contextlib.redirect_stdout(sys.stderr):            # (the Timer contexts of run.gpr / run.gsa print; stdout carries ONE JSON line)
            run.gpr('gpr', fold, is_read=False, is_covariant=False, is_isotropic=False)
            t_gpr = time.perf_counter()
            run.gsa('gpr', fold, is_covariant=False, is_isotropic=False, kinds=GSA.ALL_KINDS)
        total_s = time.perf_counter() - t0
        gsa_s = time.perf_counter() - t_gpr
        after = _lib.stat()
        Xf, yf = np.ascontiguousarray(fold.X.values, dtype=np.float64), np.ascontiguousarray(fold.Y.values[:, 0], dtype=np.float64)
        with _lib.RcGP(Xf, yf, device=device) as bare:
            t0 = time.perf_counter()
            fit = fit_lbfgsb(bare, 5.0 * np.ones(M), 2.0, 0.02)
            bare.sobol_closed(all_slices(M))
            bare_s = time.perf_counter() - t0
        S_file = pd.read_csv(fold.folder / 'gpr.v.a' / 'gsa' / 'total' / 'S.csv', index_col=[0, 1]).to_numpy()
    finally:
        HipGP.calibrate, HipGP.test = keep
        shutil.rmtree(root, ignore_errors=True)
    counts = {k: after[k] - before[k] for k in after}
    return {'N_train': int(Xf.shape[0]), 'N_test': int(rows - Xf.shape[0]), 'M': M, 'total_s': total_s, 'fit_s': timers['fit_s'], 'test_s': timers['test_s'],
            'gsa_s': gsa_s, 'io_s': total_s - timers['fit_s'] - timers['test_s'] - gsa_s, 'bare_fit_sobol_s': bare_s,
            'overhead_over_bare': total_s / bare_s - 1.0, 'bare_fit_evaluations': int(fit['nfev']), 'bare_fit_log_marginal': float(fit['log_marginal']),
            'bare_fit_message': str(getattr(fit['result'], 'message', '')), 'store_setup_s': setup_s,
            'factorisations': counts['factorisations'], 'inversions': counts['inversions'], 'gradient_evaluations': counts['gradients'],
            'factorisations_beyond_the_fit': counts['factorisations'] - counts['gradients'], 'S_total_from_csv': [float(v) for v in S_file[0]],
            'note': 'run.gpr + run.gsa on fold 0 of an 8-fold split stored on disk (csv read, model stores, calibrate, test on the held-out rows, '
                    'first-order / closed / total Sobol tables); io_s = everything outside calibrate / test / gsa; store_setup_s (building and '
                    'writing the 8-fold repository) is not part of the path per fold'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--rows', dest='n', type=int, default=16384, help='training rows per fold (BASELINE configs[2]: 16384)')
    ap.add_argument('--dims', dest='m', type=int, default=10, help='input dimensions (BASELINE configs[2]: 10)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-configs', action='store_true', help="time the CPU oracle at SURVEY 8d's C0 and C1 as well (C2 runs only; ~15 s of idle GPU)")
    ap.add_argument('--no-host-api', action='store_true', help='skip the drop-in leg (data.csv on disk -> run.gpr -> run.gsa), outside the timed region')
    ap.add_argument('--units-per-gpu', type=int, default=1,
                    help='(fold or output) units one GPU fits AT ONCE in a step: lockstep L-BFGS-B, one batched schedule per round (default 1: the headline)')
    ap.add_argument('--shard', choices=('folds', 'outputs'), default='folds',
                    help="what a rank owns: folds of an 8-fold split, (r + step) mod 8 (default, BASELINE configs[4] style), or output column r on a shared "
                         "design (configs[3] style)")
    ap.add_argument('--profile-steps', choices=('all', 'last', 'none'), default='last',
                    help='timed steps whose kernel launches carry HIP events (the roofline figures come from those launches)')
    ap.add_argument('--profile-every', type=int, default=6,
                    help='within a profiled step, the evaluations whose launches carry HIP events: every n-th one (1 = all)')
    ap.add_argument('--force-dist', action='store_true', help='initialise the process group even for one rank (exercises RCCL on a 1-GPU box)')
    ap.add_argument('--backend', choices=('nccl', 'gloo'), default=None,
                    help='torch.distributed backend (default: nccl = RCCL when a GPU is visible). gloo lets several ranks share ONE GPU: the way the '
                         'N > 1 path is rehearsed on a one-GPU box (RCCL refuses two ranks on one device)')
    args = ap.parse_args()

    from romcomma_amd import _lib, dist
    from romcomma_amd.gpr.optimize import fit_lbfgsb, fit_lbfgsb_batch
    from romcomma_amd.user.sample import synthetic_cv_fold, synthetic_outputs

    rank, world, local_rank = dist.env_rank_world()
    if world > 1 or args.force_dist:
        dist.init_process_group(args.backend)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    if _lib.device_count() <= 0:
        raise SystemExit('bench.py needs a GPU: librcgp has no CPU fallback')
    local_rank = local_rank % _lib.device_count()         # one rank per GPU on a full node; ranks share the device in the gloo rehearsal

    N, M, U = args.n, args.m, max(1, min(args.units_per_gpu, _lib.MAX_BATCH))
    K_folds = max(8, world * U)                      # folds of a K-fold split of one seeded dataset: every fold trains on N rows
    n_steps_total = args.warmup + args.steps
    if args.shard == 'outputs':                      # the same design on every rank, output column r (columns r U ... r U + U - 1 with U units)
        X, Y = synthetic_outputs(N, M, max(8, world * U))
        units = [rank if U == 1 else [rank * U + j for j in range(U)]] * n_steps_total
        handles = {l: _lib.RcGP(X, Y[:, l], device=local_rank) for l in ([rank] if U == 1 else units[0])}
    else:
        # Timed step s of rank r fits fold (r + s) mod K: a K-fold cross-validation handed round the GPUs, so that over the steps every
        # rank meets the cheap and the expensive folds alike (the folds' fits take 71-86 L-BFGS-B evaluations at C2) and one step
        # still is one fold per GPU. Every fold this rank will meet is resident in HBM before the timed region starts.
        # (Warm-up steps take the folds before r, so that timed step s is fold r + s.)
        units = fold_schedule(rank, args.warmup, args.steps, K_folds, U)
        handles = {}
        for k in dict.fromkeys(units if U == 1 else [k for step_units in units for k in step_units]):
            Xk, yk = synthetic_cv_fold(N, M, k=k, K=K_folds)
            handles[k] = _lib.RcGP(Xk, yk, device=local_rank)
            handles[k].set_hyper(4.0 * np.ones(M), 1.5, 0.03)          # one evaluation per handle, NOT at the fit's start point (an
            handles[k].lml_grad()                                       # unchanged point would let the first timed evaluation reuse the
                                                                        # factor): its work buffers (L^-1, scratch) exist before the timed region
    slices = all_slices(M)
    last = {}
    counter = {'s': 0, 'nfev': 0, 'lib_s': 0.0, 'sobol_s': 0.0}
    last_units = [units[-1]] if U == 1 else list(units[-1])
    gp = handles[last_units[0]]                      # the (first) handle of the last timed step: profiled launches, stand-alone stages

    class Timed:
        """The handle with the wall time spent inside the library calls of a fit (hyper-parameter upload + evaluation, the host's wait
        for the GPU included) added up: what is left of a step is the host's own share (SciPy's L-BFGS-B, the softplus chain rule)."""

        def __init__(self, gp):
            self._gp = gp
            self.M = gp.M

        def set_hyper(self, *a):
            t = time.perf_counter()
            self._gp.set_hyper(*a)
            counter['lib_s'] += time.perf_counter() - t

        def lml_grad(self):
            t = time.perf_counter()
            out = self._gp.lml_grad()
            counter['lib_s'] += time.perf_counter() - t
            return out

        def lml(self):
            t = time.perf_counter()
            out = self._gp.lml()
            counter['lib_s'] += time.perf_counter() - t
            return out

    def timed_batch(gps):
        t = time.perf_counter()
        out = _lib.lml_grad_batch(gps)
        counter['lib_s'] += time.perf_counter() - t
        counter['rounds'] = counter.get('rounds', 0) + 1
        return out

    def step(profiled=False):
        ids = [units[counter['s']]] if U == 1 else list(units[counter['s']])
        gps = [handles[i] for i in ids]
        counter['s'] += 1
        start = dict(lengthscales=5.0 * np.ones(M), variance=2.0, noise=0.02)
        if U == 1:
            fits = [fit_lbfgsb(Timed(gps[0]), **start)]
        else:                                        # the units' L-BFGS-B runs in lockstep: one batched schedule per round of evaluations
            fits = fit_lbfgsb_batch(gps, [start] * U, batch_lml_grad=timed_batch)
            for fit in fits:
                if isinstance(fit, Exception):
                    raise fit
        counter['nfev'] += sum(int(fit['nfev']) for fit in fits)
        rows = []
        for unit, g, fit in zip(ids, gps, fits):
            if profiled:
                g.set_profiling(True)
            t_sobol = time.perf_counter()
            V = g.sobol_closed(slices)
            counter['sobol_s'] += time.perf_counter() - t_sobol
            rows.append(np.concatenate([sobol_indices(V, M), fit['lengthscales'], [fit['variance'], fit['noise'], fit['log_marginal'], fit['nfev'], unit]]))
        table = dist.all_gather_rows(np.stack(rows), world * U, [rank * U + j for j in range(U)])      # the one collective: every rank's indices
        last.update(fit=fits[0], V=V, table=table)

    for _ in range(args.warmup):
        step()
    for h in handles.values():
        h.profile_reset()
        h.sync()
    dist.barrier()
    counter['nfev'], counter['lib_s'], counter['sobol_s'] = 0, 0.0, 0.0
    t0 = time.perf_counter()
    for i in range(args.steps):
        # Per-launch HIP events cost ~5 % (profiled dispatches, marker packets on the panel chain, harvesting), so by default
        # they are a SAMPLE of the timed region: the last step, every --profile-every-th evaluation of its fit, and its Sobol pass.
        profiled = args.profile_steps == 'all' or (args.profile_steps == 'last' and i == args.steps - 1)
        cur = handles[units[counter['s']] if U == 1 else units[counter['s']][0]]
        cur.profile_sample(args.profile_every if profiled else 0)
        cur.set_profiling(profiled and U > 1)        # (a batched call is profiled as a whole on its leading handle)
        step(profiled)
    for h in handles.values():
        h.sync()
    dist.barrier()
    elapsed = dist.max_over_ranks(time.perf_counter() - t0)
    prof = {}
    for h in handles.values():                       # launches carry HIP events on the handle of the profiled step(s) only
        h.profile_sample(0)
        for c, name in enumerate(_lib.KERNEL_CLASS_NAMES):
            got = h.profile_get(c)
            prof[name] = tuple(a + b for a, b in zip(prof.get(name, (0, 0.0, 0.0)), got))
        h.set_profiling(False)

    # Outside the timed region: the blocked Cholesky on its own at the fitted hyper-parameters (north_star quotes its MFMA fraction).
    # Inside a fit it cannot be timed separately: its kernels overlap on several streams.
    chol_ms, linv_ms, at_once = [], [], None
    if rank == 0:
        for _ in range(3):
            gp.stage_gram()
            gp.sync()
            t1 = time.perf_counter()
            gp.stage_potrf()
            gp.sync()
            chol_ms.append(1e3 * (time.perf_counter() - t1))
            t1 = time.perf_counter()
            gp.stage_trtri()                         # L^-1 by recursive doubling (+ alpha): the second N^3/3 of an evaluation
            gp.sync()
            linv_ms.append(1e3 * (time.perf_counter() - t1))
    if rank == 0 and world == 1:
        # Two units in one schedule beside one unit alone (rcgp_lml_grad_batch): whole evaluations, host included, alternating between two
        # points so that nothing is served from a cache.
        pair = [gp] + [h for h in handles.values() if h is not gp][:1]
        made = None
        if len(pair) == 1:
            made = _lib.RcGP(*(synthetic_cv_fold(N, M, k=1, K=K_folds) if args.shard == 'folds' else (X, Y[:, (rank * U + 1) % Y.shape[1]])), device=local_rank)
            pair.append(made)
        fit = last['fit']
        points = [(np.asarray(fit['lengthscales']), float(fit['variance']), float(fit['noise']))]
        ms, turn = {1: [], 2: []}, 0
        for nb in (2, 1, 2, 1):
            for r in range(3):
                turn += 1                            # (every evaluation at a point no handle has seen: nothing is served from a cache)
                for h in pair[:nb]:
                    h.set_hyper(points[0][0] * (1.0 + 1e-3 * turn), *points[0][1:])
                gp.sync()
                t1 = time.perf_counter()
                _lib.lml_grad_batch(pair[:nb])
                ms[nb].append(1e3 * (time.perf_counter() - t1))
        one, two = min(ms[1][1:]), min(ms[2][1:])
        at_once = {'units': 2, 'evaluation_ms_one_unit': one, 'evaluation_ms_two_units': two, 'ratio': two / one,
                   'TFLOPs_two_units': 2 * float(N) ** 3 / (two * 1e-3) / 1e12,
                   'note': 'whole LML + gradient evaluations through rcgp_lml_grad_batch, host included: two units in ONE schedule against one unit '
                           'alone (2.0 = no gain over one after the other); per-unit results are bit-identical either way (tests/test_gpu_batch.py)'}
        if made is not None:
            made.close()

    if rank == 0:
        n_gemm, ms_gemm, flops = prof['gemm']
        n_grad, ms_grad, grad_flops = prof['grad']
        n_gram, ms_gram, gram_bytes = prof['gram']
        n_sob, ms_sob, sob_exps = prof['sobol']
        n_diag, ms_diag, _ = prof['diag']
        # dominant kernel = k_grad (K^-1 = L^-T L^-1 fused with the gradient reduction): one launch per evaluation, N^3/3 flops,
        # the largest share of the GPU time of any kernel name (profiles/*_kernel_stats.csv)
        achieved = grad_flops / (ms_grad * 1e-3) / 1e12 if ms_grad > 0 else 0.0
        family = (flops + grad_flops) / ((ms_gemm + ms_grad) * 1e-3) / 1e12 if ms_gemm + ms_grad > 0 else 0.0
        nfev = int(last['fit']['nfev'])
        nfev_total = int(counter['nfev'])                    # rank 0's L-BFGS-B evaluations over the timed steps
        eval_ms = 1e3 * counter['lib_s'] / max(nfev_total, 1)
        traffic, traffic_file = pmc_traffic(N, M)
        sobol_texp = sob_exps / (ms_sob * 1e-3) / 1e12 if ms_sob > 0 else 0.0
        out = {
            'metric': 'GP-fit+Sobol train-points/s (wall-time per fit+Sobol in ms_per_step), fp64',
            'value': world * U * N * args.steps / elapsed,
            'unit': 'train-points/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'{"C2" if (N, M) == (16384, 10) else "custom"}: ARD-RBF GP fit (L-BFGS-B to convergence, reference defaults) + closed-form Sobol first/closed/'
                                   f'total indices, N={N}, M={M}, ' + (f'L=1, fold (r + step) mod {K_folds} of an {K_folds}-fold split on GPU r' if args.shard == 'folds' else
                                                          f'output r of {max(8, world * U)} independent outputs on one design per GPU') +
                                   (f'; {U} units per GPU fitted at once (lockstep L-BFGS-B, one batched schedule per round)' if U > 1 else ''),
                       'units_per_gpu': U,
                       'N': N, 'M': M, 'lbfgs_evaluations_last_step': nfev, 'lbfgs_evaluations_timed_steps': nfev_total,
                       'ms_per_evaluation_incl_host': 1e3 * elapsed / max(nfev_total, 1),
                       'ms_per_step_inside_library_calls': 1e3 * (counter['lib_s'] + counter['sobol_s']) / args.steps,
                       'ms_per_step_host_only': 1e3 * (elapsed - counter['lib_s'] - counter['sobol_s']) / args.steps, 'parallelism': f'{args.shard[:-1]}-per-gpu x{world}' + (f' x{U} at once' if U > 1 else ''),
                       'log_marginal': last['fit']['log_marginal'],
                       'gathered_rows_last_step': int(np.sum(~np.isnan(last['table']).any(axis=1))),
                       'units_last_step': [int(u) for u in last['table'][:, -1]]},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / FP64_MFMA_PEAK_TFLOPS, 'traffic': traffic,
                         'traffic_source': (f'{traffic_file}: rocprofv3 --pmc pass of the same workload and build, NOT measured in this run' if traffic_file else None),
                         'kernel': 'k_grad (K^-1 = L^-T L^-1 on fp64 MFMA fused with the LML-gradient reduction): the largest single launch, one per '
                                   'evaluation. By SUMMED time the K = NB update kernels of the Cholesky (k_gemm_nt_sub + k_syrk_lower, many launches on '
                                   'several streams) are the larger family: stages.mfma_gemm_family, profiles/r04_*_kernel_stats.csv',
                         'launches': int(n_grad), 'avg_launch_ms': ms_grad / max(n_grad, 1),
                         'algorithmic_flops_per_launch': grad_flops / max(n_grad, 1),
                         'evaluation': {'algorithmic_flops': float(N) ** 3, 'ms': eval_ms, 'achieved': float(N) ** 3 / (eval_ms * 1e-3) / 1e12,
                                        'frac': float(N) ** 3 / (eval_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                        'note': 'one LML + gradient evaluation = Gram + Cholesky + L^-1 + fused K^-1/gradient = N^3 flops; ms = wall time '
                                                'inside the library calls of the timed fits / evaluations (host wait included)'}},
            'stages': {
                'gram': {'bound': 'hbm', 'achieved_GBs': gram_bytes / (ms_gram * 1e-3) / 1e9 if ms_gram > 0 else 0.0, 'peak_GBs': HBM_PEAK_GBS,
                         'frac': (gram_bytes / (ms_gram * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_gram > 0 else 0.0,
                         'launches': int(n_gram), 'avg_launch_ms': ms_gram / max(n_gram, 1),
                         'timing': 'HIP events riding on the dispatches of the sampled evaluations in this run (hipExtLaunchKernelGGL start / stop events: '
                                   'kernel begin to kernel end); the rocprofv3 --kernel-trace duration of the same kernel is in profiles/*_one_eval_c2_kernel_stats.csv'},
                'cholesky': {'bound': 'mfma', 'algorithmic_flops': N ** 3 / 3.0, 'ms': min(chol_ms), 'achieved_TFLOPs': N ** 3 / 3.0 / (min(chol_ms) * 1e-3) / 1e12,
                             'frac': N ** 3 / 3.0 / (min(chol_ms) * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                             'note': 'stand-alone rcgp_stage_potrf (incl. w = L^-1 y) after the timed region, best of 3, host wall clock around a sync'},
                'linv': {'bound': 'mfma', 'algorithmic_flops': N ** 3 / 3.0, 'ms': min(linv_ms), 'achieved_TFLOPs': N ** 3 / 3.0 / (min(linv_ms) * 1e-3) / 1e12,
                         'frac': N ** 3 / 3.0 / (min(linv_ms) * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                         'note': 'stand-alone rcgp_stage_trtri (L^-1 by recursive doubling, 128^2 diagonal inverses + log2(N/128) levels, + alpha) after '
                                 'the timed region, best of 3, host wall clock around a sync'},
                'diag_blocks': {'launches': int(n_diag), 'total_ms': ms_diag},
                'sobol': {'bound': 'fp64 VALU (exp)', 'launches': int(n_sob), 'total_ms': ms_sob, 'Gexp_per_s': 1e3 * sobol_texp,
                          'valu_ops_per_exp': SOBOL_VALU_OPS_PER_EXP, 'measured_valu_peak_Tops': FP64_VALU_MEASURED_TFMAS,
                          'frac_of_measured_valu_peak': sobol_texp * SOBOL_VALU_OPS_PER_EXP / FP64_VALU_MEASURED_TFMAS,
                          'frac_source': SOBOL_FRAC_SOURCE},
                'mfma_gemm_family': {'launches': int(n_gemm + n_grad), 'summed_launch_ms': ms_gemm + ms_grad,
                                     'TFLOPs_over_summed_launch_time': family,
                                     'note': 'Cholesky / L^-1 / K^-1 kernels; the Cholesky runs them on several streams, so summed launch time exceeds wall time'},
                'timed_region_ms': 1e3 * elapsed, 'steps_with_hip_events': args.profile_steps,
                       'evaluations_with_hip_events': f'every {args.profile_every}-th of a profiled step'},
        }
        out['roofline']['stages'] = {k: {kk: vv for kk, vv in out['stages'][k].items() if kk in ('bound', 'frac', 'achieved_GBs', 'achieved_TFLOPs', 'ms', 'avg_launch_ms')}
                                     for k in ('gram', 'cholesky', 'linv')}
        if at_once is not None:
            out['stages']['units_at_once'] = at_once
        if U > 1:
            out['config']['evaluation_rounds_timed_steps'] = int(counter.get('rounds', 0))
        if world == 1 and not args.no_host_api:
            out['stages']['host_api'] = host_api_leg(N, M, device=local_rank)
            if (N, M) == (16384, 10):                # BASELINE configs[1] beside the headline configuration
                out['stages']['host_api_C1'] = host_api_leg(8192, 5, device=local_rank)
        if world == 1 and not args.no_cpu_baseline:
            fit = last['fit']
            theta = (np.asarray(fit['lengthscales'], dtype=float), float(fit['variance']), float(fit['noise']))
            gp.set_hyper(*theta)
            gpu_value = gp.lml_grad()
            parity_slices = [(0, 1), (0, max(M // 2, 1)), (max(M // 2, 1), M), (0, M)]
            gpu_sobol = (parity_slices, gp.sobol_closed(parity_slices))
            fold = synthetic_cv_fold(N, M, k=last_units[0], K=K_folds) if args.shard == 'folds' else (X, Y[:, last_units[0]])
            # (with several units per step the first unit of the last step carries `last`: its fit, its fold)
            out['cpu_baseline'] = cpu_baseline(N, M, nfev_total / max(args.steps * U, 1), fold=fold, theta=theta, gpu_value=gpu_value,
                                               other_configs=args.cpu_configs, gpu_sobol=gpu_sobol)
            out['parity_at_config'] = out['cpu_baseline'].get('parity_at_config')
        print(json.dumps(out), flush=True)
    for h in handles.values():
        h.close()
    if dist.is_distributed():
        import torch.distributed as td
        td.destroy_process_group()


if __name__ == '__main__':
    main()
