"""Closed-form Sobol calibrator on the GPU (stands where reference gsa/calibrators.py:31-143 ``ClosedSobol`` stands).

Reads from the gp exactly what the reference reads (gsa/calibrators.py:119-140): L, M, N, kernel variance, lengthscales,
K_inv_Y and X -- but never K_cho, which plain ClosedSobol pulls and does not use. For L outputs the
(L,L) matrices V, S hold the cross-output entries of the reference's 'lLN,lLNjJn,jJn->lj' einsum (gsa/calibrators.py:79).
``marginalize_all`` serves every slice of a kind (or of all three kinds) from one pass over the pair tiles.
"""
from __future__ import annotations

from typing import Any, Dict, List, Sequence, Tuple

import numpy as np

from romcomma_amd.gpr.models import GPR
from romcomma_amd.gsa.base import Calibrator


class ClosedSobol(Calibrator):
    """Closed Sobol conditional variances V and indices S = V / V[2] of a fitted independent-output GP."""

    @classmethod
    @property
    def META(cls) -> Dict[str, Any]:
        return {}

    def __init__(self, gp: GPR, **kwargs: Any):
        self.gp = gp
        self.meta = self.META | kwargs
        self.L, self.M, self.N = gp.L, gp.M, gp.N
        self.is_F_diagonal = self.meta.pop('is_F_diagonal', None)
        if self.is_F_diagonal is None:
            gp_options = gp.read_meta() if gp._meta_json.exists() else gp.META
            self.is_F_diagonal = not gp_options.get('kernel', {}).get('covariance', False)      # gsa/calibrators.py:129-132
        if not self.is_F_diagonal:
            raise NotImplementedError('non-diagonal kernel variance F (covariant GP) is outside this backend')
        F = np.asarray(gp.kernel.data.frames.variance.np, dtype=np.float64)
        self.F = (F if F.shape[0] == 1 else np.diag(F)).reshape(self.L)                          # :134-136
        self.Lambda = np.broadcast_to(np.asarray(gp.kernel.data.frames.lengthscales.np, dtype=np.float64), (self.L, self.M)).copy()
        self.K_inv_Y = np.asarray(gp.K_inv_Y, dtype=np.float64).reshape(self.L, self.N)          # also leaves each alpha cached on host
        self._cache: Dict[Tuple[int, int], np.ndarray] = {}
        self._calibrate()

    # ---- device calls
    def _V_many(self, slices: Sequence[Sequence[int]]) -> np.ndarray:
        """(L, L, len(slices)) conditional variances; results are memoised per slice."""
        slices = [(int(s[0]), int(s[1])) for s in slices]
        missing = [s for s in dict.fromkeys(slices) if s not in self._cache]
        if missing:
            block = np.empty((self.L, self.L, len(missing)))
            for l in range(self.L):
                handle = self.gp._select(l)
                for j in range(self.L):
                    if j == l:
                        block[l, j] = handle.sobol_closed(missing)
                    elif j > l or self.L == 1:
                        block[l, j] = handle.sobol_cross(self.Lambda[j], self.F[j], self.K_inv_Y[j], missing)
                    else:
                        block[l, j] = block[j, l]          # V is symmetric in (l, j) (gsa/calibrators.py:79 comment)
            for i, s in enumerate(missing):
                self._cache[s] = block[..., i]
        return np.stack([self._cache[s] for s in slices], axis=-1)

    def _calibrate(self):
        """V[0] = full-model V, V[1] = its diagonal, V[2] = sqrt(V1) outer sqrt(V1), S = V[0] / V[2] (gsa/calibrators.py:93-97).
        The first device pass already covers every first-order / closed / total slice, so later marginalize calls are lookups."""
        M = self.M
        canonical = [(m, m + 1) for m in range(M)] + [(0, m + 1) for m in range(M)] + [(m + 1, M) for m in range(M)]
        self._V_many(canonical)
        self.V = {0: self._cache[(0, M)]}
        self.V[1] = np.diagonal(self.V[0]).copy()
        root = np.sqrt(self.V[1])
        self.V[2] = np.einsum('l,i->li', root, root)
        self.S = self.V[0] / self.V[2]

    # ---- the Calibrator contract
    def marginalize(self, m) -> Dict[str, np.ndarray]:
        """{'V': (L,L), 'S': V / V[2]} for the slice [m[0], m[1]) (gsa/calibrators.py:49-58)."""
        V = self._V_many([m])[..., 0]
        return {'V': V, 'S': V / self.V[2]}

    def marginalize_all(self, slices: Sequence[Sequence[int]]) -> Dict[str, np.ndarray]:
        """The same for many slices at once, stacked on a new last axis."""
        V = self._V_many(slices)
        return {'V': V, 'S': V / self.V[2][..., None]}


class ClosedSobolWithError(ClosedSobol):
    """Standard errors T, W of the indices (reference gsa/calibrators.py:146-402): SURVEY.md 8f rank 2, not built yet."""

    @classmethod
    @property
    def META(cls) -> Dict[str, Any]:
        return {'is_T_partial': True}

    def __init__(self, gp: GPR, **kwargs: Any):
        raise NotImplementedError('ClosedSobolWithError (index standard errors T, W) is not implemented on this backend yet; '
                                  'run with is_error_calculated=False')
