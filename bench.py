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
Prints ONE JSON line on rank 0.
"""
import argparse
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


def all_slices(M):
    first = [(m, m + 1) for m in range(M)]
    closed = [(0, m + 1) for m in range(M)]
    complement = [(m + 1, M) for m in range(M)]
    return first + closed + complement + [(0, M)]


def sobol_indices(V, M):
    """First-order, closed and total indices from the 3M+1 conditional variances (gsa/models.py:207-214)."""
    full = V[3 * M]
    return np.concatenate([V[:M] / full, V[M:2 * M] / full, 1.0 - V[2 * M:3 * M] / full])


def fold_schedule(rank, warmup, steps, n_folds):
    """The fold every step of one rank fits, warm-up steps first: timed step s takes fold (rank + s) mod n_folds, the warm-up steps
    the folds before `rank`. A K-fold cross-validation handed round the GPUs: one fold per GPU and step, and over the steps every
    rank meets the cheap and the expensive folds alike."""
    return [(rank - warmup + s) % n_folds for s in range(warmup + steps)]


def pmc_traffic(N, M, kernel='k_grad'):
    """(HBM bytes per launch of `kernel`, the file they come from): the committed rocprofv3 --pmc passes of the same workload and build
    (profiles/r03_pmc_c2.json, falling back to earlier rounds; produced by tools/pmc_summary.py with the gfx950 FETCH_SIZE correction).
    Counters cannot be read inside this process, so the number is NOT measured in this run; (None, None) for any other size."""
    if (N, M) != (16384, 10):
        return None, None
    for name in ('r03_pmc_c2.json', 'r02_pmc_c2.json', 'r01_pmc_c2.json'):
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


def cpu_baseline(N, M, nfev_per_step, fold=None, theta=None, gpu_value=None):
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
    if is_c2:
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
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--rows', dest='n', type=int, default=16384, help='training rows per fold (BASELINE configs[2]: 16384)')
    ap.add_argument('--dims', dest='m', type=int, default=10, help='input dimensions (BASELINE configs[2]: 10)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
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
    from romcomma_amd.gpr.optimize import fit_lbfgsb
    from romcomma_amd.user.sample import synthetic_cv_fold, synthetic_outputs

    rank, world, local_rank = dist.env_rank_world()
    if world > 1 or args.force_dist:
        dist.init_process_group(args.backend)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    if _lib.device_count() <= 0:
        raise SystemExit('bench.py needs a GPU: librcgp has no CPU fallback')
    local_rank = local_rank % _lib.device_count()         # one rank per GPU on a full node; ranks share the device in the gloo rehearsal

    N, M = args.n, args.m
    K_folds = max(8, world)                          # folds of a K-fold split of one seeded dataset: every fold trains on N rows
    n_steps_total = args.warmup + args.steps
    if args.shard == 'outputs':                      # the same design on every rank, output column r
        X, Y = synthetic_outputs(N, M, max(8, world))
        units = [rank] * n_steps_total
        handles = {rank: _lib.RcGP(X, Y[:, rank], device=local_rank)}
    else:
        # Timed step s of rank r fits fold (r + s) mod K: a K-fold cross-validation handed round the GPUs, so that over the steps every
        # rank meets the cheap and the expensive folds alike (the folds' fits take 71-86 L-BFGS-B evaluations at C2) and one step
        # still is one fold per GPU. Every fold this rank will meet is resident in HBM before the timed region starts.
        # (Warm-up steps take the folds before r, so that timed step s is fold r + s.)
        units = fold_schedule(rank, args.warmup, args.steps, K_folds)
        handles = {}
        for k in dict.fromkeys(units):
            Xk, yk = synthetic_cv_fold(N, M, k=k, K=K_folds)
            handles[k] = _lib.RcGP(Xk, yk, device=local_rank)
            handles[k].set_hyper(4.0 * np.ones(M), 1.5, 0.03)          # one evaluation per handle, NOT at the fit's start point (an
            handles[k].lml_grad()                                       # unchanged point would let the first timed evaluation reuse the
                                                                        # factor): its work buffers (L^-1, scratch) exist before the timed region
    slices = all_slices(M)
    last = {}
    counter = {'s': 0, 'nfev': 0, 'lib_s': 0.0, 'sobol_s': 0.0}
    gp = handles[units[-1]]                          # the handle of the last timed step: profiled launches, stand-alone stages

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

    def step(profiled=False):
        unit = units[counter['s']]
        gp = handles[unit]
        counter['s'] += 1
        fit = fit_lbfgsb(Timed(gp), 5.0 * np.ones(M), 2.0, 0.02)
        counter['nfev'] += int(fit['nfev'])
        if profiled:
            gp.set_profiling(True)
        t_sobol = time.perf_counter()
        V = gp.sobol_closed(slices)
        counter['sobol_s'] += time.perf_counter() - t_sobol
        row = np.concatenate([sobol_indices(V, M), fit['lengthscales'], [fit['variance'], fit['noise'], fit['log_marginal'], fit['nfev'], unit]])
        table = dist.all_gather_rows(row[None, :], world, [rank])      # the one collective: every rank's indices
        last.update(fit=fit, V=V, table=table)

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
        cur = handles[units[counter['s']]]
        cur.profile_sample(args.profile_every if profiled else 0)
        cur.set_profiling(False)
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
    chol_ms = []
    if rank == 0:
        for _ in range(3):
            gp.stage_gram()
            gp.sync()
            t1 = time.perf_counter()
            gp.stage_potrf()
            gp.sync()
            chol_ms.append(1e3 * (time.perf_counter() - t1))

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
            'value': world * N * args.steps / elapsed,
            'unit': 'train-points/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'{"C2" if (N, M) == (16384, 10) else "custom"}: ARD-RBF GP fit (L-BFGS-B to convergence, reference defaults) + closed-form Sobol first/closed/'
                                   f'total indices, N={N}, M={M}, ' + (f'L=1, fold (r + step) mod {K_folds} of an {K_folds}-fold split on GPU r' if args.shard == 'folds' else
                                                          f'output r of {max(8, world)} independent outputs on one design per GPU'),
                       'N': N, 'M': M, 'lbfgs_evaluations_last_step': nfev, 'lbfgs_evaluations_timed_steps': nfev_total,
                       'ms_per_evaluation_incl_host': 1e3 * elapsed / max(nfev_total, 1),
                       'ms_per_step_inside_library_calls': 1e3 * (counter['lib_s'] + counter['sobol_s']) / args.steps,
                       'ms_per_step_host_only': 1e3 * (elapsed - counter['lib_s'] - counter['sobol_s']) / args.steps, 'parallelism': f'{args.shard[:-1]}-per-gpu x{world}',
                       'log_marginal': last['fit']['log_marginal'],
                       'gathered_rows_last_step': int(np.sum(~np.isnan(last['table']).any(axis=1))),
                       'units_last_step': [int(u) for u in last['table'][:, -1]]},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / FP64_MFMA_PEAK_TFLOPS, 'traffic': traffic,
                         'traffic_source': (f'{traffic_file}: rocprofv3 --pmc pass of the same workload and build, NOT measured in this run' if traffic_file else None),
                         'kernel': 'k_grad (K^-1 = L^-T L^-1 on fp64 MFMA fused with the LML-gradient reduction): the largest single launch, one per '
                                   'evaluation. By SUMMED time the K = NB update kernels of the Cholesky (k_gemm_nt_sub + k_syrk_lower, many launches on '
                                   'several streams) are the larger family: stages.mfma_gemm_family, profiles/r03_*_kernel_stats.csv',
                         'launches': int(n_grad), 'avg_launch_ms': ms_grad / max(n_grad, 1),
                         'algorithmic_flops_per_launch': grad_flops / max(n_grad, 1),
                         'evaluation': {'algorithmic_flops': float(N) ** 3, 'ms': eval_ms, 'achieved': float(N) ** 3 / (eval_ms * 1e-3) / 1e12,
                                        'frac': float(N) ** 3 / (eval_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                        'note': 'one LML + gradient evaluation = Gram + Cholesky + L^-1 + fused K^-1/gradient = N^3 flops; ms = wall time '
                                                'inside the library calls of the timed fits / evaluations (host wait included)'}},
            'stages': {
                'gram': {'bound': 'hbm', 'achieved_GBs': gram_bytes / (ms_gram * 1e-3) / 1e9 if ms_gram > 0 else 0.0, 'peak_GBs': HBM_PEAK_GBS,
                         'frac': (gram_bytes / (ms_gram * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_gram > 0 else 0.0,
                         'launches': int(n_gram), 'avg_launch_ms': ms_gram / max(n_gram, 1)},
                'cholesky': {'bound': 'mfma', 'algorithmic_flops': N ** 3 / 3.0, 'ms': min(chol_ms), 'achieved_TFLOPs': N ** 3 / 3.0 / (min(chol_ms) * 1e-3) / 1e12,
                             'frac': N ** 3 / 3.0 / (min(chol_ms) * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                             'note': 'stand-alone rcgp_stage_potrf (incl. w = L^-1 y) after the timed region, best of 3, host wall clock around a sync'},
                'diag_blocks': {'launches': int(n_diag), 'total_ms': ms_diag},
                'sobol': {'bound': 'fp64 VALU (exp)', 'launches': int(n_sob), 'total_ms': ms_sob, 'Gexp_per_s': 1e3 * sobol_texp,
                          'valu_ops_per_exp': SOBOL_VALU_OPS_PER_EXP, 'measured_valu_peak_Tops': FP64_VALU_MEASURED_TFMAS,
                          'frac_of_measured_valu_peak': sobol_texp * SOBOL_VALU_OPS_PER_EXP / FP64_VALU_MEASURED_TFMAS},
                'mfma_gemm_family': {'launches': int(n_gemm + n_grad), 'summed_launch_ms': ms_gemm + ms_grad,
                                     'TFLOPs_over_summed_launch_time': family,
                                     'note': 'Cholesky / L^-1 / K^-1 kernels; the Cholesky runs them on several streams, so summed launch time exceeds wall time'},
                'timed_region_ms': 1e3 * elapsed, 'steps_with_hip_events': args.profile_steps,
                       'evaluations_with_hip_events': f'every {args.profile_every}-th of a profiled step'},
        }
        out['roofline']['stages'] = {k: {kk: vv for kk, vv in out['stages'][k].items() if kk in ('bound', 'frac', 'achieved_GBs', 'achieved_TFLOPs', 'ms', 'avg_launch_ms')}
                                     for k in ('gram', 'cholesky')}
        if world == 1 and not args.no_cpu_baseline:
            fit = last['fit']
            theta = (np.asarray(fit['lengthscales'], dtype=float), float(fit['variance']), float(fit['noise']))
            gp.set_hyper(*theta)
            gpu_value = gp.lml_grad()
            fold = synthetic_cv_fold(N, M, k=units[-1], K=K_folds) if args.shard == 'folds' else (X, Y[:, rank])
            out['cpu_baseline'] = cpu_baseline(N, M, nfev_total / max(args.steps, 1), fold=fold, theta=theta, gpu_value=gpu_value)
            out['parity_at_config'] = out['cpu_baseline'].get('parity_at_config')
        print(json.dumps(out), flush=True)
    for h in handles.values():
        h.close()
    if dist.is_distributed():
        import torch.distributed as td
        td.destroy_process_group()


if __name__ == '__main__':
    main()
