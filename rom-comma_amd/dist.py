"""One process per GPU. (fold k, output l) units shard embarrassingly (the reference loops them sequentially:
user/run.py:60-61,132-133; gpr/models.py:340-342,360-361); there is no data-path collective. The single exchange is the
final gather of the per-unit result rows (Sobol indices, hyper-parameters, LML) -- the distributed analogue of
results.Collect.from_folds (user/results.py:98-114) -- plus, optionally, the all-gather of (alpha, lengthscales) that the
cross-output Sobol entries need.

Backend: RCCL ("nccl") over xGMI when a GPU is visible, gloo on CPU (tests).
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))


def prepare_environment():
    """Environment defaults that must be in place BEFORE the first GPU call of the process (ROCr reads HSA_* once, at hsa_init):
    ``HSA_ENABLE_IPC_MODE_LEGACY=0`` selects dmabuf IPC handles, the only kind the host driver of the target MI355X nodes supports --
    with the legacy mode RCCL's (and torch's) cross-process buffer sharing on one node fails with ``hipIpcGetMemHandle: invalid
    argument`` (stated by the platform notes of the build; the GPU boxes export the variable themselves). ``setdefault``: an explicit
    setting in the caller's environment wins. Called at ``romcomma_amd`` import and again at the top of ``init_process_group``."""
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def init_process_group(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    prepare_environment()                                     # before torch.cuda is touched below
    import torch
    import torch.distributed as dist
    rank, world, local_rank = env_rank_world()
    if not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def is_distributed() -> bool:
    """True once a process group exists (a world of one still runs the real collectives: that is how the RCCL path is
    exercised on a single-GPU box)."""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def shard_units(n_units: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment of independent (fold, output) units to ranks."""
    return list(range(rank, n_units, world))


def _device():
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def barrier():
    import torch
    import torch.distributed as dist
    if is_distributed():
        if dist.get_backend() == 'nccl':
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def all_gather_rows(local: np.ndarray, n_units: int, unit_ids: Sequence[int]) -> np.ndarray:
    """Every rank contributes its rows (one per owned unit, fixed width); every rank receives the (n_units, width) table in
    unit order. One all_gather of a fixed-size padded block per rank: latency-bound, a few KB."""
    import torch
    import torch.distributed as dist
    local = np.atleast_2d(np.asarray(local, dtype=np.float64))
    width = local.shape[1] if local.size else 0
    if not is_distributed():
        out = np.full((n_units, width), np.nan)
        out[list(unit_ids)] = local
        return out
    world = dist.get_world_size()
    per_rank = (n_units + world - 1) // world
    dev = _device()
    wt = torch.tensor([width], dtype=torch.int64, device=dev)
    dist.all_reduce(wt, op=dist.ReduceOp.MAX)
    width = int(wt.item())
    block = torch.full((per_rank, width + 1), float('nan'), dtype=torch.float64, device=dev)
    for r, (uid, row) in enumerate(zip(unit_ids, local)):
        block[r, 0] = float(uid)
        block[r, 1:] = torch.as_tensor(row, dtype=torch.float64)
    gathered = torch.empty((world * per_rank, width + 1), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(gathered, block)
    g = gathered.cpu().numpy()
    out = np.full((n_units, width), np.nan)
    for row in g:
        if not np.isnan(row[0]):
            out[int(row[0])] = row[1:]
    return out


def broadcast_object(obj, src: int = 0):
    """A small picklable object from rank ``src`` to every rank (host-side metadata: fold lists, shapes)."""
    import torch.distributed as dist
    if not is_distributed():
        return obj
    box = [obj if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src, device=_device())
    return box[0]


def max_over_ranks(value: float) -> float:
    import torch
    import torch.distributed as dist
    if not is_distributed():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())



def agree_on_failure(error: Exception | None) -> None:
    """Called by every rank after its share of a sharded loop, with the exception it caught (or None). If ANY rank failed, every
    rank raises -- the failing ones their own exception, the others a RuntimeError naming the ranks -- so that nobody is left
    waiting in the next collective for a rank that has gone (the reference's loop is sequential and simply propagates,
    user/run.py:60-61). One all-reduce (MAX) of a flag per rank."""
    if not is_distributed():
        if error is not None:
            raise error
        return
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    flags = torch.zeros(world, dtype=torch.int32, device=_device())
    if error is not None:
        flags[rank] = 1
    dist.all_reduce(flags, op=dist.ReduceOp.MAX)
    failed = [r for r in range(world) if int(flags[r].item())]
    if error is not None:
        raise error
    if failed:
        raise RuntimeError(f'rank(s) {failed} failed in a sharded loop; rank {rank} stops with them')
