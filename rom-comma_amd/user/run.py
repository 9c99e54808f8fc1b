"""Orchestration of GPR and GSA over the folds of a Repository (reference user/run.py:35-158), with the one structural
change that suits 8 GPUs: the sequential ``for k in repo.folds`` loop (user/run.py:60-61,132-133) becomes a round-robin
shard over the ranks of a ``torch.distributed`` job (one process per GPU). Folds are independent, so there is no data-path
collective; ranks write their fold folders to the shared file system, meet at a barrier, and rank 0 runs the same
``results.Collect`` pass the reference runs. Single-process use is unchanged.

Naming, the isotropic -> anisotropic warm start by folder copy, per-fold Timer and ``ignore_exceptions`` follow the reference.
"""
from __future__ import annotations

import shutil
from pathlib import Path
from typing import Any, List, Optional, Sequence

import numpy as np

from romcomma_amd import dist
from romcomma_amd.data.storage import Fold, Repository
from romcomma_amd.gpr.kernels import Kernel
from romcomma_amd.gpr.models import GPR, MOGP
from romcomma_amd.gsa.models import GSA, Sobol
from romcomma_amd.user import contexts, results


def _my_folds(repo: Repository, shard_folds: bool = True) -> List[int]:
    rank, world, _ = dist.env_rank_world()
    folds = list(repo.folds)
    return [folds[i] for i in dist.shard_units(len(folds), rank, world)] if (world > 1 and shard_folds) else folds


def _over_folds(repo: Repository, shard_folds: bool, one_fold):
    """Run ``one_fold(Fold)`` over this rank's folds. In a distributed job a rank that fails must not leave the others waiting in the
    barrier that follows (they would sit there until the RCCL timeout): every rank reports, and all of them raise together."""
    result, error = [], None
    try:
        for k in _my_folds(repo, shard_folds):
            result = one_fold(Fold(repo, k))
    except BaseException as exception:                      # noqa: B902 -- re-raised on every rank below
        error = exception
    if shard_folds:
        dist.agree_on_failure(error)
    elif error is not None:
        raise error
    return result


def gpr(name: str, repo: Repository, is_read: bool | None, is_covariant: bool | None, is_isotropic: bool | None,
        ignore_exceptions: bool = False, kernel_parameters: Kernel.Data | None = None, likelihood_variance: np.ndarray | None = None,
        is_calibrated: bool = True, is_tested: bool = True, shard_folds: bool = True, **kwargs: Any) -> List[str]:
    """GPR on a Fold, or across the Folds of a Repository (sharded over ranks when distributed, unless ``shard_folds`` is False:
    a rank that owns a whole repository -- one output of ``Y_splits_sharded`` -- runs all of its folds itself).

    ``is_read`` None = warm start from the nearest calibrated ancestor (the independent '.v' model of the same isotropy before
    the '.i' model, user/run.py:75-84); ``is_isotropic`` None = run isotropic then anisotropic; ``is_covariant`` None = run the
    independent GPs, then the covariant GP warm-started from them (user/run.py:69-73). Returns the model names.
    """
    if not isinstance(repo, Fold):
        names: List[str] = _over_folds(repo, shard_folds, lambda fold: gpr(name, fold, is_read, is_covariant, is_isotropic, ignore_exceptions,
                                                                          kernel_parameters, likelihood_variance, is_calibrated, is_tested, **kwargs))
        if dist.is_distributed() and shard_folds:
            dist.barrier()
            if not names:                                   # a rank that owned no fold still needs the names for the return value
                names = _names(name, is_covariant, is_isotropic)
            if dist.env_rank_world()[0] != 0:
                return names
        if is_tested:
            results.Collect({'test': {'header': [0, 1]}, 'test_summary': {'header': [0, 1], 'index_col': 0}}, {n: {} for n in names},
                            ignore_exceptions).from_folds(repo, True)
        results.Collect({'variance': {}, 'log_marginal': {}}, {f'{n}/likelihood': {} for n in names}, ignore_exceptions).from_folds(repo, True)
        results.Collect({'variance': {}, 'lengthscales': {}}, {f'{n}/kernel': {} for n in names}, ignore_exceptions).from_folds(repo, True)
        return names
    if is_covariant is None:
        names = gpr(name, repo, is_read, False, is_isotropic, ignore_exceptions, kernel_parameters, likelihood_variance, is_calibrated, is_tested,
                    **kwargs)
        return names + gpr(name, repo, None, True, False if is_isotropic is None else is_isotropic, ignore_exceptions, kernel_parameters,
                           likelihood_variance, is_calibrated, is_tested, **kwargs)
    full_name = name + ('.c' if is_covariant else '.v')
    if is_isotropic is None:
        names = gpr(name, repo, is_read, is_covariant, True, ignore_exceptions, kernel_parameters, likelihood_variance, is_calibrated, is_tested, **kwargs)
        return names + gpr(name, repo, None, is_covariant, False, ignore_exceptions, kernel_parameters, likelihood_variance, is_calibrated,
                           is_tested, **kwargs)
    full_name = full_name + ('.i' if is_isotropic else '.a')
    if is_read is None:
        if not (repo.folder / full_name).exists():
            nearest_name = name + '.v' + full_name[-2:]
            if not (is_covariant and (repo.folder / nearest_name).exists()):
                nearest_name = full_name[:-2] + '.i'
            if not (repo.folder / nearest_name).exists():
                return gpr(name, repo, False, is_covariant, is_isotropic, ignore_exceptions, kernel_parameters, likelihood_variance,
                           is_calibrated, is_tested, **kwargs)
            GPR.Data.copy(src_folder=repo.folder / nearest_name, dst_folder=repo.folder / full_name)
        return gpr(name, repo, True, is_covariant, is_isotropic, ignore_exceptions, kernel_parameters, likelihood_variance, is_calibrated,
                   is_tested, **kwargs)
    with contexts.Timer(f'fold.{repo.meta["k"]} {full_name} GPR'):
        gp = None
        try:
            if is_read:
                gp = MOGP(full_name, repo, is_read, is_covariant, is_isotropic)
            else:
                gp = MOGP(full_name, repo, is_read, is_covariant, is_isotropic, kernel_parameters, likelihood_variance)
            if is_calibrated:
                gp.calibrate(**kwargs)
            if is_tested:
                gp.test()
        except BaseException as exception:
            if not ignore_exceptions:
                raise exception
        finally:
            if gp is not None:
                gp.close()
    return [full_name]


def Y_splits_sharded(repo: Repository) -> List[Repository]:
    """Independent outputs, one repository each (the reference's ``Repository.Y_split``, data/storage.py:226-243), dealt
    round-robin to the ranks of the job: rank 0 writes the ``Y.l`` folders, every rank returns ITS share after the barrier.
    Each ``Y.l`` is an ordinary single-output Repository: fold it and pass it to ``gpr`` with ``shard_folds=False`` (BASELINE
    configs[3]: one output per GPU), then call ``gsa_outputs`` on the parent for the full (L, L) Sobol matrices, cross-output entries
    included."""
    rank, world, _ = dist.env_rank_world()
    if rank == 0:
        repo.Y_split()
    if dist.is_distributed():
        dist.barrier()
    splits = sorted(repo.Y_splits)
    mine = dist.shard_units(len(splits), rank, world) if world > 1 else range(len(splits))
    return [Repository(splits[i][1]) for i in mine]


def _names(name: str, is_covariant: Optional[bool], is_isotropic: Optional[bool]) -> List[str]:
    if is_covariant is None:
        return _names(name, False, is_isotropic) + _names(name, True, False if is_isotropic is None else is_isotropic)
    base = name + ('.c' if is_covariant else '.v')
    return [base + '.i', base + '.a'] if is_isotropic is None else [base + ('.i' if is_isotropic else '.a')]


def gsa(name: str, repo: Repository, is_covariant: Optional[bool], is_isotropic: Optional[bool],
        kinds: GSA.Kind | Sequence[GSA.Kind] = GSA.ALL_KINDS, m: int = -1, ignore_exceptions: bool = False,
        is_error_calculated: bool = False, shard_folds: bool = True, **kwargs: Any) -> List[Path]:
    """GSA on a Fold, or across the Folds of a Repository (sharded over ranks when distributed and ``shard_folds``). Always resumes
    from the GP stored by ``gpr`` (is_read=True, user/run.py:151). Returns the calculation folders relative to the fold."""
    kinds = (kinds,) if isinstance(kinds, GSA.Kind) else kinds
    if not isinstance(repo, Fold):
        names: List[Path] = _over_folds(repo, shard_folds, lambda fold: gsa(name, fold, is_covariant, is_isotropic, kinds, m, ignore_exceptions,
                                                                           is_error_calculated, **kwargs))
        if dist.is_distributed() and shard_folds:
            dist.barrier()
            if dist.env_rank_world()[0] != 0:
                return names
            if not names:
                return names
        results.Collect({'S': {}, 'V': {}} | ({'T': {}, 'W': {}} if is_error_calculated else {}), {n: {} for n in names},
                        ignore_exceptions).from_folds(repo, True)
        for n in names:
            shutil.copyfile(repo.fold_folder(repo.folds.start) / n / 'meta.json', repo.folder / n / 'meta.json')
        return names
    if is_covariant is None:                                 # independent then covariant (user/run.py:137-140)
        names = gsa(name, repo, False, is_isotropic, kinds, m, ignore_exceptions, is_error_calculated, **kwargs)
        return names + gsa(name, repo, True, False if is_isotropic is None else is_isotropic, kinds, m, ignore_exceptions,
                           is_error_calculated, **kwargs)
    full_name = name + ('.c' if is_covariant else '.v')
    if is_isotropic is None:
        names = gsa(name, repo, is_covariant, True, kinds, m, ignore_exceptions, is_error_calculated, **kwargs)
        return names + gsa(name, repo, is_covariant, False, kinds, m, ignore_exceptions, is_error_calculated, **kwargs)
    full_name = full_name + ('.i' if is_isotropic else '.a')
    names = []
    with contexts.Timer(f'fold.{repo.meta["k"]} {full_name} GSA'):
        gp = None
        try:
            gp = MOGP(full_name, repo, is_read=True, is_covariant=is_covariant, is_isotropic=is_isotropic)
            for kind in kinds:
                folder = Sobol(gp, kind, m, is_error_calculated, **kwargs).calibrate().get('folder')
                names += [Path(folder).relative_to(repo.folder)]
        except BaseException as exception:
            if not ignore_exceptions:
                raise exception
        finally:
            if gp is not None:
                gp.close()
    return names


def gsa_outputs(name: str, repo: Repository, is_isotropic: bool, kinds: GSA.Kind | Sequence[GSA.Kind] = GSA.ALL_KINDS, m: int = -1,
                is_error_calculated: bool = False, **kwargs: Any) -> List[Path]:
    """GSA of L independent outputs whose GPs live in the ``Y.l`` split repositories of ``repo``, one output (or a few) per rank
    (``Y_splits_sharded`` + ``gpr(..., shard_folds=False)`` before this). Produces what the single-process
    ``gsa(name, repo, is_covariant=False, ...)`` on the L-output repository produces -- ``fold.k/<name>.v.<i|a>/gsa/<kind>/S.csv,
    V.csv[, T.csv, W.csv]`` with all (l.0, l.1) rows, cross-output entries included (gsa/calibrators.py:79, gsa/models.py:66-75) --
    written by rank 0 into ``repo``'s fold folders. Per fold the ranks exchange (K_inv_Y, lengthscales, variance) of their outputs
    in one all-gather and the finished rows in another (``gpr.sharded.OutputShard``); every rank walks every fold."""
    from romcomma_amd.gpr.sharded import OutputShard
    kinds = (kinds,) if isinstance(kinds, GSA.Kind) else kinds
    rank, world, _ = dist.env_rank_world()
    splits = sorted(repo.Y_splits)
    L = len(splits)
    if L == 0:
        raise FileNotFoundError(f'{repo.folder} has no Y.l splits: call Y_splits_sharded(repo) and fit them first')
    owned = dist.shard_units(L, rank, world) if world > 1 else list(range(L))
    full_name = name + '.v' + ('.i' if is_isotropic else '.a')
    first = Repository(splits[0][1])
    names: List[Path] = []
    error = None
    try:
        for k in first.folds:
            with contexts.Timer(f'fold.{k} {full_name} GSA over {L} outputs, {len(owned)} here'):
                shard = None
                gps = {}
                try:
                    for i in owned:
                        gps[splits[i][0]] = MOGP(full_name, Fold(Repository(splits[i][1]), k), is_read=True, is_covariant=False, is_isotropic=is_isotropic)
                    fold_meta = Fold(first, k).meta['data']
                    shard = OutputShard(gps, L, repo.fold_folder(k) / full_name, N=fold_meta['N'], M=fold_meta['M'])
                    names = []
                    for kind in kinds:
                        folder = Sobol(shard, kind, m, is_error_calculated, **kwargs).calibrate().get('folder')
                        names += [Path(folder).relative_to(shard.folder.parent)]
                finally:
                    if shard is not None:
                        shard.close()
                    else:
                        for gp in gps.values():
                            gp.close()
    except BaseException as exception:                      # noqa: B902
        error = exception
    dist.agree_on_failure(error)
    if dist.is_distributed():
        dist.barrier()
    if rank == 0 and names and repo.K > 0 and all((repo.fold_folder(k) / 'meta.json').exists() for k in repo.folds):
        results.Collect({'S': {}, 'V': {}} | ({'T': {}, 'W': {}} if is_error_calculated else {}), {n: {} for n in names}).from_folds(repo, True)
    return names
