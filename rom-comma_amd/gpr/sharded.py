"""Independent outputs of one design dealt to the ranks of a job (BASELINE configs[3]: one output per GPU).

The reference fits the L outputs of an independent MOGP one after the other in one process (gpr/models.py:340-342, 360-361)
and its Sobol calibrator then fills the whole (L, L) matrix of conditional variances, cross-output entries included
(gsa/calibrators.py:79: einsum '...->lj'; saved rows l.0 != l.1, gsa/models.py:66-75). With the outputs on different GPUs
entry (l, j) needs output j's weights next to output l's design: ONE all-gather of (K_inv_Y_j (N), lengthscales_j (M),
variance_j) per rank -- (N + M + 1) doubles, 65.6 KB at C3 -- before the pair kernels, and one of the finished rows after
them ("the one optional exchange step"). Nothing else crosses the ranks.

``OutputShard`` is what the Sobol calibrators see in place of a ``GPR``: the gathered parameters of all L outputs, device
handles for the outputs this rank owns, and the two gathers.
"""
from __future__ import annotations

import tempfile
from pathlib import Path
from typing import Any, Dict, List, Sequence

import numpy as np

from romcomma_amd import dist


class OutputShard:
    """L single-output GPs on the same inputs, ``gps[l]`` for the outputs this rank owns (each an ``HipGP`` with L = 1, e.g. read
    from fold k of the ``Y.l`` split repositories). Results are written under ``folder`` by rank 0 only; the other ranks get a
    private scratch folder so that the model stores they create collide with nobody."""

    def __init__(self, gps: Dict[int, Any], L: int, folder: Path | str, N: int | None = None, M: int | None = None):
        self.gps = dict(gps)
        self.owned_outputs: List[int] = sorted(self.gps)
        rank, world, _ = dist.env_rank_world()
        self.is_writer = rank == 0
        self._scratch = None
        if self.is_writer:
            self.folder = Path(folder)
        else:
            self._scratch = tempfile.TemporaryDirectory(prefix=f'rcgp_rank{rank}_')
            self.folder = Path(self._scratch.name) / Path(folder).name
        self.L = int(L)
        shapes = {(gp.N, gp.M) for gp in self.gps.values()}
        if N is not None and M is not None:
            shapes.add((int(N), int(M)))
        if len(shapes) != 1:
            raise ValueError(f'the outputs of an OutputShard must share one (N, M) design (a rank that owns no output passes N and M): {shapes}')
        self.N, self.M = next(iter(shapes))
        rows = []
        for l in self.owned_outputs:
            gp = self.gps[l]
            if gp.L != 1 or getattr(gp, '_is_covariant', False):
                raise ValueError('an OutputShard is made of independent single-output GPs')
            alpha = np.asarray(gp.K_inv_Y, dtype=np.float64).reshape(self.N)                              # gpr/models.py:441-444
            ell = np.broadcast_to(np.asarray(gp.kernel.data.frames.lengthscales.np, dtype=np.float64), (1, self.M))[0]
            variance = float(np.asarray(gp.kernel.data.frames.variance.np).reshape(-1)[0])
            rows.append(np.concatenate([alpha, ell, [variance]]))
        # the exchange: every rank learns (alpha_j, ell_j, F_j) of every output j
        table = dist.all_gather_rows(np.array(rows).reshape(len(rows), self.N + self.M + 1), self.L, self.owned_outputs)
        if np.isnan(table).any():
            raise ValueError(f'OutputShard: outputs {np.flatnonzero(np.isnan(table).any(axis=1)).tolist()} are owned by no rank')
        self.K_inv_Y = np.ascontiguousarray(table[:, :self.N])
        self.Lambda = np.ascontiguousarray(table[:, self.N:self.N + self.M])
        self.F = np.ascontiguousarray(table[:, -1])

    # ---- what the calibrators ask of a gp
    def _select(self, l: int):
        """The device handle of owned output ``l`` with its stored hyper-parameters current."""
        return self.gps[l]._select(0)

    def hyper_signature(self) -> tuple:
        return tuple(np.concatenate([self.Lambda.ravel(), self.F, self.K_inv_Y[:, :4].ravel()]))

    def gather_output_rows(self, block: np.ndarray) -> np.ndarray:
        """``block[l]`` is filled for the owned outputs l; returns the array with every row from its owner."""
        shape = block.shape
        local = block.reshape(shape[0], -1)[self.owned_outputs]
        return dist.all_gather_rows(local, self.L, self.owned_outputs).reshape(shape)

    def close(self):
        for gp in self.gps.values():
            gp.close()
        if self._scratch is not None:
            self._scratch.cleanup()
            self._scratch = None


def sobol_rows(shard: OutputShard, slices: Sequence[Sequence[int]]) -> np.ndarray:
    """(L, L, len(slices)) closed-form conditional variances of an OutputShard: the owner of output l evaluates entries (l, j >= l)
    -- ``rcgp_sobol_closed`` on the diagonal, ``rcgp_sobol_cross`` with the gathered (ell_j, F_j, alpha_j) off it -- the rows are
    gathered and the lower triangle mirrored, exactly the entries and the symmetry the single-process calibrator uses."""
    L = shard.L
    block = np.zeros((L, L, len(slices)))
    for l in shard.owned_outputs:
        handle = shard._select(l)
        for j in range(l, L):
            block[l, j] = handle.sobol_closed(slices) if j == l else handle.sobol_cross(shard.Lambda[j], shard.F[j], shard.K_inv_Y[j], slices)
    block = shard.gather_output_rows(block)
    for l in range(L):
        for j in range(l):
            block[l, j] = block[j, l]
    return block
