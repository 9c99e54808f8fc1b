"""``gpr`` / ``gsa`` over a Fold or over all folds of a Repository -- the two user-level entry points of the reference
(interface: user/run.py:35-158) -- plus the sharded forms this backend adds for a one-process-per-GPU job.

Organisation (own, not the reference's recursion): the ``None`` options ("do both") are expanded ONCE into an ordered plan of
``_Variant`` records (``_plan``); a fold is then simply walked through the plan. What the plan encodes:

    is_covariant None  ->  the independent model first, then the covariant one warm-started (start=None), anisotropic unless told otherwise
    is_isotropic None  ->  the isotropic model first, then the anisotropic one warm-started
    start None (warm)  ->  resume the model's own folder if it exists; else copy the nearest stored relative -- for a covariant model the
                           independent model of the same isotropy, otherwise / failing that the isotropic model of the same covariance -- and
                           resume from the copy; with no relative on disk, start fresh
    model folder name  ->  <name>.<c|v>.<i|a>

Folds are independent, so across a Repository they are dealt round-robin to the ranks of a ``torch.distributed`` job (no data-path
collective); ranks meet at a barrier and rank 0 concatenates the per-fold csv files exactly as a single process does
(``results.Collect``, user/results.py:98-114). Failure handling in a job: every rank reports after its share (``dist.agree_on_failure``)
and all raise together; only ``Exception`` is caught for that -- KeyboardInterrupt / SystemExit end the rank at once.
"""
import shutil
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Callable, Optional, Sequence

import numpy as np

from romcomma_amd import dist
from romcomma_amd.data.storage import Fold, Repository
from romcomma_amd.gpr.kernels import Kernel
from romcomma_amd.gpr.models import GPR, MOGP, default_units_per_gpu
from romcomma_amd.gsa.models import GSA, Sobol
from romcomma_amd.user import contexts, results


@dataclass(frozen=True)
class _Variant:
    """One model of the plan. ``start``: True = resume the stored model, False = fresh parameters, None = warm start (module docstring)."""
    covariant: bool
    isotropic: bool
    start: Optional[bool] = True

    def model_name(self, name: str) -> str:
        return f'{name}.{"c" if self.covariant else "v"}.{"i" if self.isotropic else "a"}'

    def relatives(self, name: str) -> list[str]:
        """Stored models a warm start may copy from, nearest first."""
        same_isotropy_independent = [_Variant(False, self.isotropic).model_name(name)] if self.covariant else []
        return same_isotropy_independent + [_Variant(self.covariant, True).model_name(name)]


def _plan(start: Optional[bool], is_covariant: Optional[bool], is_isotropic: Optional[bool]) -> list[_Variant]:
    if is_covariant is None:
        covariances = [(False, start, is_isotropic), (True, None, False if is_isotropic is None else is_isotropic)]
    else:
        covariances = [(is_covariant, start, is_isotropic)]
    plan = []
    for covariant, first_start, isotropic in covariances:
        if isotropic is None:
            plan += [_Variant(covariant, True, first_start), _Variant(covariant, False, None)]
        else:
            plan.append(_Variant(covariant, isotropic, first_start))
    return plan


def _resolve_warm_start(fold: Fold, name: str, variant: _Variant) -> bool:
    """Decide a ``start=None`` variant on this fold: True (resume, possibly from a relative just copied into place) or False (fresh)."""
    own = fold.folder / variant.model_name(name)
    if own.exists():
        return True
    for relative in variant.relatives(name):
        if (fold.folder / relative).exists() and fold.folder / relative != own:
            GPR.Data.copy(src_folder=fold.folder / relative, dst_folder=own)
            return True
    return False


def _my_folds(repo: Repository, shard_folds: bool = True) -> list[int]:
    rank, world, _ = dist.env_rank_world()
    every = list(repo.folds)
    return [every[i] for i in dist.shard_units(len(every), rank, world)] if (world > 1 and shard_folds) else every


def _each_fold(repo: Repository, shard_folds: bool, job: Callable[[Fold], list], group_job: Optional[Callable[[list], list]] = None,
               at_once: int = 1) -> list:
    """``job(Fold)`` on this rank's folds; returns the last result. With ``group_job`` and ``at_once`` > 1 the folds go to
    ``group_job([Fold, ...])`` in groups of ``at_once`` instead (several folds fitted at once on one GPU). Sharded: all ranks agree on
    success before anyone goes on to the barrier (a rank that failed would otherwise leave the rest waiting there until the RCCL timeout)."""
    outcome, caught = [], None
    try:
        mine = _my_folds(repo, shard_folds)
        if group_job is not None and at_once > 1 and len(mine) > 1:
            for first in range(0, len(mine), at_once):
                outcome = group_job([Fold(repo, k) for k in mine[first:first + at_once]])
        else:
            for k in mine:
                outcome = job(Fold(repo, k))
    except Exception as exception:
        caught = exception
    if shard_folds:
        dist.agree_on_failure(caught)
    elif caught is not None:
        raise caught
    return outcome


def _folds_at_once(folds: int, L: int, places: int) -> int:
    """How many folds are opened together when the GPU has ``places`` for (fold, output) units: the folds' units are dealt to the fewest
    groups that overfill the places by no more than a quarter, and the groups are of equal size -- a group that overfills has its surplus
    units start as places fall free (gpr/optimize.py::fit_lbfgsb_batch), which costs less than a small last group with the chip half
    empty (measured: tools/README.md, folds_rule). One fold at a time when a single fold's outputs fill the places."""
    import os
    rule = os.environ.get('RCGP_FOLDS_RULE', 'balanced')
    if L > places or folds < 2:
        return 1
    if rule == 'floor':
        return max(1, places // L)
    if rule == 'ceil':
        return -(-places // L)
    groups = max(1, -(-(4 * folds * L) // (5 * places)))
    return max(1, min(-(-folds // groups), (5 * places) // (4 * L)))


def _is_collector(shard_folds: bool) -> bool:
    """After a sharded pass: wait for every rank, then only rank 0 concatenates."""
    if dist.is_distributed() and shard_folds:
        dist.barrier()
        return dist.env_rank_world()[0] == 0
    return True


def gpr(name: str, repo: Repository, is_read: bool | None, is_covariant: bool | None, is_isotropic: bool | None,
        ignore_exceptions: bool = False, kernel_parameters: Kernel.Data | None = None, likelihood_variance: np.ndarray | None = None,
        is_calibrated: bool = True, is_tested: bool = True, shard_folds: bool = True, units_per_gpu: Optional[int] = None,
        **kwargs: Any) -> list[str]:
    """Fit (and test) GPs on a Fold, or on every fold of a Repository followed by the cross-fold csv collection. ``None`` for
    ``is_read`` / ``is_covariant`` / ``is_isotropic`` means warm start / both / both (module docstring). ``shard_folds=False`` keeps all
    folds on the calling rank (a rank that owns a whole ``Y.l`` repository, ``Y_splits_sharded``). Returns the model names, in plan order.

    ``units_per_gpu`` (not in the reference, which walks its folds one after the other, user/run.py:60-61): how many (fold, output) units
    this rank's GPU is given AT ONCE -- the independent GPs of that many folds are calibrated together, their L-BFGS-B runs in lockstep,
    every round of evaluations one batched schedule on the device (``HipGP.calibrate_group``). Each fold ends with exactly the files it
    gets alone. Default: RCGP_UNITS, else by the folds' size (``default_units_per_gpu``); 1 = one fold after the other.
    """
    plan = _plan(is_read, is_covariant, is_isotropic)
    names = [variant.model_name(name) for variant in plan]

    def on_fold(fold: Fold) -> list[str]:
        for variant, model_name in zip(plan, names):
            resume = _resolve_warm_start(fold, name, variant) if variant.start is None else variant.start
            with contexts.Timer(f'fold.{fold.meta["k"]} {model_name} GPR'):
                gp = None
                try:
                    fresh = () if resume else (kernel_parameters, likelihood_variance)
                    gp = MOGP(model_name, fold, resume, variant.covariant, variant.isotropic, *fresh, units_per_gpu=units_per_gpu)
                    if is_calibrated:
                        gp.calibrate(**kwargs)
                    if is_tested:
                        gp.test()
                except Exception:
                    if not ignore_exceptions:
                        raise
                finally:
                    if gp is not None:
                        gp.close()
        return names

    def on_fold_group(folds: list[Fold]) -> list[str]:
        """The plan, variant by variant, on several folds at once: construct the folds' GPs, calibrate them together, test each."""
        for variant, model_name in zip(plan, names):
            with contexts.Timer(f'folds {[fold.meta["k"] for fold in folds]} {model_name} GPR'):
                gps: list = []
                try:
                    for fold in folds:
                        try:
                            resume = _resolve_warm_start(fold, name, variant) if variant.start is None else variant.start
                            fresh = () if resume else (kernel_parameters, likelihood_variance)
                            gps.append(MOGP(model_name, fold, resume, variant.covariant, variant.isotropic, *fresh, units_per_gpu=units_per_gpu))
                        except Exception:
                            if not ignore_exceptions:
                                raise
                    ready = gps
                    if is_calibrated:
                        outcomes = MOGP.calibrate_group(gps, **kwargs)
                        failures = [outcome for outcome in outcomes if isinstance(outcome, Exception)]
                        if failures and not ignore_exceptions:
                            raise failures[0]
                        ready = [gp for gp, outcome in zip(gps, outcomes) if not isinstance(outcome, Exception)]
                    if is_tested:
                        for gp in ready:
                            try:
                                gp.test()
                            except Exception:
                                if not ignore_exceptions:
                                    raise
                finally:
                    for gp in gps:
                        gp.close()
        return names

    if isinstance(repo, Fold):
        return on_fold(repo)
    at_once = 1
    if is_calibrated and len(repo.folds) > 1:
        at_once = int(units_per_gpu) if units_per_gpu is not None else default_units_per_gpu(repo.N)
        at_once = _folds_at_once(len(_my_folds(repo, shard_folds)), max(int(repo.L), 1), at_once)
    _each_fold(repo, shard_folds, on_fold, on_fold_group, at_once)
    if _is_collector(shard_folds):
        per_model = {
            '': ({'test': {'header': [0, 1]}, 'test_summary': {'header': [0, 1], 'index_col': 0}} if is_tested else {}),
            '/likelihood': {'variance': {}, 'log_marginal': {}},
            '/kernel': {'variance': {}, 'lengthscales': {}},
        }
        for sub_folder, csvs in per_model.items():
            if csvs:
                results.Collect(csvs, {f'{n}{sub_folder}': {} for n in names}, ignore_exceptions).from_folds(repo, True)
    return names


def Y_splits_sharded(repo: Repository) -> list[Repository]:
    """The single-output ``Y.l`` repositories of ``repo`` (``Repository.Y_split``, written by rank 0) dealt round-robin to the ranks;
    returns this rank's share. Each is an ordinary repository: fold it, ``gpr(..., shard_folds=False)`` it (BASELINE configs[3]: one
    output per GPU), then ``gsa_outputs`` on the parent gives the full (L, L) Sobol matrices, cross-output entries included."""
    rank, world, _ = dist.env_rank_world()
    if rank == 0:
        repo.Y_split()
    if dist.is_distributed():
        dist.barrier()
    splits = sorted(repo.Y_splits)
    mine = dist.shard_units(len(splits), rank, world) if world > 1 else range(len(splits))
    return [Repository(splits[i][1]) for i in mine]


def _sobol_csvs(is_error_calculated: bool) -> dict:
    return {stem: {} for stem in (('S', 'V', 'T', 'W') if is_error_calculated else ('S', 'V'))}


def gsa(name: str, repo: Repository, is_covariant: Optional[bool], is_isotropic: Optional[bool],
        kinds: GSA.Kind | Sequence[GSA.Kind] = GSA.ALL_KINDS, m: int = -1, ignore_exceptions: bool = False,
        is_error_calculated: bool = False, shard_folds: bool = True, **kwargs: Any) -> list[Path]:
    """Closed-form Sobol indices of the GPs ``gpr`` stored (always resumed from disk), on a Fold or on every fold of a Repository followed
    by the cross-fold collection. Returns the calculation folders relative to the fold, in plan x kind order."""
    kinds = (kinds,) if isinstance(kinds, GSA.Kind) else tuple(kinds)
    plan = _plan(True, is_covariant, is_isotropic)

    def on_fold(fold: Fold) -> list[Path]:
        done: list[Path] = []
        for variant in plan:
            model_name = variant.model_name(name)
            with contexts.Timer(f'fold.{fold.meta["k"]} {model_name} GSA'):
                gp = None
                try:
                    gp = MOGP(model_name, fold, is_read=True, is_covariant=variant.covariant, is_isotropic=variant.isotropic)
                    for kind in kinds:
                        where = Sobol(gp, kind, m, is_error_calculated, **kwargs).calibrate().get('folder')
                        done.append(Path(where).relative_to(fold.folder))
                except Exception:
                    if not ignore_exceptions:
                        raise
                finally:
                    if gp is not None:
                        gp.close()
        return done

    if isinstance(repo, Fold):
        return on_fold(repo)
    done = _each_fold(repo, shard_folds, on_fold)
    if _is_collector(shard_folds) and done:
        results.Collect(_sobol_csvs(is_error_calculated), {n: {} for n in done}, ignore_exceptions).from_folds(repo, True)
        for n in done:
            shutil.copyfile(repo.fold_folder(repo.folds.start) / n / 'meta.json', repo.folder / n / 'meta.json')
    return done


def _agreed_fold_table(first_split: Path) -> tuple[list[int], dict[int, tuple[int, int]]]:
    """Folds of the ``Y.l`` repositories and the (N, M) of each, as rank 0 sees them after everybody's fits are on disk, handed to every
    rank -- so that all ranks walk the same fold list even if their view of the shared folder lags."""
    rank, world, _ = dist.env_rank_world()
    table, failure = None, None
    if rank == 0:
        try:                                               # (fallible disk reads: their failure travels WITH the broadcast, so nobody waits in it)
            first = Repository(first_split, meta_only=True)
            table = [(int(k), int(Repository(first.fold_folder(k), meta_only=True).N), int(first.M)) for k in first.folds]
        except Exception as exception:
            failure = exception
    if dist.is_distributed():
        table, message = dist.broadcast_object((table, None if failure is None else f'{type(failure).__name__}: {failure}'))
        if message is not None and failure is None:
            failure = RuntimeError(f'rank 0 could not read the fold table of {first_split}: {message}')
    if failure is not None:
        raise failure
    return [k for k, _, _ in table], {k: (n, mm) for k, n, mm in table}


def gsa_outputs(name: str, repo: Repository, is_isotropic: bool, kinds: GSA.Kind | Sequence[GSA.Kind] = GSA.ALL_KINDS, m: int = -1,
                is_error_calculated: bool = False, **kwargs: Any) -> list[Path]:
    """GSA of L independent outputs whose GPs live in the ``Y.l`` split repositories of ``repo``, one output (or a few) per rank
    (``Y_splits_sharded`` + ``gpr(..., shard_folds=False)`` before this). Produces what the single-process
    ``gsa(name, repo, is_covariant=False, ...)`` on the L-output repository produces -- ``fold.k/<name>.v.<i|a>/gsa/<kind>/S.csv,
    V.csv[, T.csv, W.csv]`` with all (l.0, l.1) rows, cross-output entries included (gsa/calibrators.py:79, gsa/models.py:66-75) --
    written by rank 0 into ``repo``'s fold folders. Per fold the ranks exchange (K_inv_Y, lengthscales, variance) of their outputs
    in one all-gather and the finished rows in another (``gpr.sharded.OutputShard``); every rank walks every fold.

    Ordering in a job: a barrier first (every rank's fits and folds are on disk before anyone reads another rank's folder); per fold
    the fallible LOCAL work -- reading this rank's stored GPs -- comes first and the ranks agree on its success BEFORE the first
    collective of that fold, so a missing model stops all ranks together instead of leaving some inside an all-gather. A failure after
    that point (inside the collectives) is not recoverable in step and propagates as it is."""
    from romcomma_amd.gpr.sharded import OutputShard
    kinds = (kinds,) if isinstance(kinds, GSA.Kind) else tuple(kinds)
    rank, world, _ = dist.env_rank_world()
    if dist.is_distributed():
        dist.barrier()
    splits = sorted(repo.Y_splits)
    L = len(splits)
    if L == 0:
        raise FileNotFoundError(f'{repo.folder} has no Y.l splits: call Y_splits_sharded(repo) and fit them first')
    owned = dist.shard_units(L, rank, world) if world > 1 else list(range(L))
    model_name = _Variant(False, is_isotropic).model_name(name)
    folds, shape_of = _agreed_fold_table(splits[0][1])
    done: list[Path] = []
    for k in folds:
        with contexts.Timer(f'fold.{k} {model_name} GSA over {L} outputs, {len(owned)} here'):
            gps, caught = {}, None
            try:
                for i in owned:
                    gps[splits[i][0]] = MOGP(model_name, Fold(Repository(splits[i][1]), k), is_read=True, is_covariant=False,
                                             is_isotropic=is_isotropic)
            except Exception as exception:
                caught = exception
            try:
                dist.agree_on_failure(caught)
            except Exception:
                for gp in gps.values():
                    gp.close()
                raise
            N, M = shape_of[k]
            shard = None
            try:
                shard = OutputShard(gps, L, repo.fold_folder(k) / model_name, N=N, M=M)
                done = []
                for kind in kinds:
                    where = Sobol(shard, kind, m, is_error_calculated, **kwargs).calibrate().get('folder')
                    done.append(Path(where).relative_to(shard.folder.parent))
            finally:
                for closable in ([shard] if shard is not None else gps.values()):
                    closable.close()
    if dist.is_distributed():
        dist.barrier()
    if rank == 0 and done and repo.K > 0 and all((repo.fold_folder(k) / 'meta.json').exists() for k in repo.folds):
        results.Collect(_sobol_csvs(is_error_calculated), {n: {} for n in done}).from_folds(repo, True)
    return done
