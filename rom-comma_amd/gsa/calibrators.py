"""Closed-form Sobol calibrator on the GPU (stands where reference gsa/calibrators.py:31-143 ``ClosedSobol`` stands).

Reads from the gp exactly what the reference reads (gsa/calibrators.py:119-140): L, M, N, kernel variance, lengthscales,
K_inv_Y and X -- but never K_cho, which plain ClosedSobol pulls and does not use. For L outputs the
(L,L) matrices V, S hold the cross-output entries of the reference's 'lLN,lLNjJn,jJn->lj' einsum (gsa/calibrators.py:79).
``marginalize_all`` serves every slice of a kind (or of all three kinds) from one pass over the pair tiles.
"""
from __future__ import annotations

from typing import Any, Dict, List, Sequence, Tuple

import numpy as np

from romcomma_amd import _lib
from romcomma_amd.gpr.models import GPR
from romcomma_amd.gpr.sharded import OutputShard, sobol_rows
from romcomma_amd.gsa.base import Calibrator


def _virtual_outputs(K_inv_Y: np.ndarray, F: np.ndarray, lengthscales: np.ndarray, is_F_diagonal: bool):
    """The weight-vector recipes behind 'lLN, lLNjJn, jJn -> lj' (gsa/calibrators.py:79) as (phi, pre, alpha index) per (l, J):
    Lambda^2 = ell_l^2 with J = 0 only when F is diagonal (:105), ell_l ell_J otherwise (:107); phi = 1/(Lambda^2 + 1) (:92);
    pre = F sqrt(prod Lambda^2 phi) (:86). Returns phi (L, Lb, M), pre (L, Lb) and, for every (l, J), which alpha it multiplies."""
    L = K_inv_Y.shape[0]
    ell = np.asarray(lengthscales, dtype=np.float64)
    if is_F_diagonal:
        lam2 = (ell * ell)[:, None, :]
        Fm = np.asarray(F, dtype=np.float64).reshape(L, 1)
        which = np.arange(L)[:, None]                      # g0KY[l, 0] = g0[l, 0] * K_inv_Y[l]
    else:
        lam2 = ell[:, None, :] * ell[None, :, :]
        Fm = np.asarray(F, dtype=np.float64).reshape(L, L)
        which = np.broadcast_to(np.arange(L)[None, :], (L, L))   # K_inv_Y transposed to (1, L, N) (:138): g0KY[l, J] = g0[l, J] * K_inv_Y[J]
    phi = 1.0 / (lam2 + 1.0)
    pre = Fm * np.sqrt(np.prod(lam2 * phi, axis=-1))
    return phi, pre, which


def covariant_V(handle: '_lib.RcGP', K_inv_Y: np.ndarray, F: np.ndarray, lengthscales: np.ndarray, slices: Sequence[Sequence[int]],
                is_F_diagonal: bool = False) -> np.ndarray:
    """(len(slices), L, L) conditional variances from the generic pair entry of the library (``rcgp_sobol_pair``): every weight
    vector is built on the device from (phi, pre, alpha, shift), the shift being the mean over (J, N) for its l (:90). Any handle
    on the same X serves. V is symmetric in (l, j)."""
    alpha = np.ascontiguousarray(np.asarray(K_inv_Y, dtype=np.float64).reshape(-1, handle.N))
    L = alpha.shape[0]
    phi, pre, which = _virtual_outputs(alpha, F, np.broadcast_to(lengthscales, (L, handle.M)), is_F_diagonal)
    Lb = phi.shape[1]
    sums = np.array([[_lib.sobol_weight_sum(handle, phi[l, J], pre[l, J], alpha[which[l, J]]) for J in range(Lb)] for l in range(L)])
    shift = sums.sum(axis=1) / float(Lb * handle.N)
    V = np.zeros((len(slices), L, L))
    for l in range(L):
        for j in range(l + 1):
            for a in range(Lb):
                for b in range(Lb):
                    V[:, l, j] += _lib.sobol_pair(handle, phi[l, a], pre[l, a], alpha[which[l, a]], shift[l],
                                                  phi[j, b], pre[j, b], alpha[which[j, b]], shift[j], slices)
            V[:, j, l] = V[:, l, j]
    return V


class ClosedSobol(Calibrator):
    """Closed Sobol conditional variances V and indices S = V / V[2] of a fitted GP: independent outputs (one handle call per
    output pair), or a covariant GP -- with F reduced to its diagonal unless the kernel covariance was trained, as the reference
    decides from meta.json (gsa/calibrators.py:129-138)."""

    @classmethod
    @property
    def META(cls) -> Dict[str, Any]:
        return {}

    def __init__(self, gp: GPR | OutputShard, **kwargs: Any):
        self.gp = gp
        self.meta = self.META | kwargs
        self.L, self.M, self.N = gp.L, gp.M, gp.N
        self.is_F_diagonal = self.meta.pop('is_F_diagonal', None)
        self.is_sharded = isinstance(gp, OutputShard)
        self._cache: Dict[Tuple[int, int], np.ndarray] = {}
        if self.is_sharded:
            # independent outputs living on different ranks: the shard has already gathered (F, Lambda, K_inv_Y) of all of them
            self.is_F_diagonal, self.is_gp_covariant = True, False
            self.F, self.Lambda, self.K_inv_Y = gp.F, gp.Lambda, gp.K_inv_Y
            self.owned = list(gp.owned_outputs)
            self._share_cache()
            self._calibrate()
            return
        self.owned = list(range(self.L))
        if self.is_F_diagonal is None:
            gp_options = gp.read_meta() if gp._meta_json.exists() else gp.META
            self.is_F_diagonal = not gp_options.get('kernel', {}).get('covariance', False)      # gsa/calibrators.py:129-132
        self.is_gp_covariant = bool(getattr(gp, '_is_covariant', False))
        if not self.is_F_diagonal and not self.is_gp_covariant:
            raise NotImplementedError('a non-diagonal F needs the (L,L) kernel variance of a covariant GP')
        F = np.asarray(gp.kernel.data.frames.variance.np, dtype=np.float64)
        if self.is_F_diagonal:
            self.F = (F if F.shape[0] == 1 else np.diag(F)).reshape(self.L)                      # :134-136
        else:
            self.F = F.reshape(self.L, self.L)
        self.Lambda = np.broadcast_to(np.asarray(gp.kernel.data.frames.lengthscales.np, dtype=np.float64), (self.L, self.M)).copy()
        self.K_inv_Y = np.asarray(gp.K_inv_Y, dtype=np.float64).reshape(self.L, self.N)          # also leaves each alpha cached on host
        self._share_cache()
        self._calibrate()

    def _share_cache(self):
        """The conditional variances depend on (F, Lambda, K_inv_Y) only, and ONE device pass yields the slices of all three GSA kinds: the
        calibrators that ``run.gsa`` builds for 'first_order', 'closed' and 'total' on one gp (the reference builds -- and recomputes -- one
        per kind, gsa/models.py:205, user/run.py:141-147) share the memo through the gp, keyed by the values it was computed from."""
        signature = (self.is_F_diagonal, np.asarray(self.F).tobytes(), np.asarray(self.Lambda).tobytes(), np.asarray(self.K_inv_Y).tobytes())
        shared = getattr(self.gp, '_sobol_memo', None)
        if shared is None or shared[0] != signature:
            shared = (signature, {}, {}, {})
            try:
                self.gp._sobol_memo = shared
            except AttributeError:                          # (an object that takes no attributes: keep the memo to this calibrator)
                pass
        self._cache = shared[1]
        self._shared_W = (shared[2], shared[3])             # ClosedSobolWithError: the covariances W behind the standard errors, per slice

    # ---- device calls
    def _V_many(self, slices: Sequence[Sequence[int]]) -> np.ndarray:
        """(L, L, len(slices)) conditional variances; results are memoised per slice."""
        slices = [(int(s[0]), int(s[1])) for s in slices]
        missing = [s for s in dict.fromkeys(slices) if s not in self._cache]
        if missing and self.is_gp_covariant:
            # one (LN) system behind K_inv_Y: every V_lj is a sum of generic pair forms (covariant_V)
            block = covariant_V(self.gp._select_mo(), self.K_inv_Y, self.F, self.Lambda, missing, self.is_F_diagonal)
            for i, s in enumerate(missing):
                self._cache[s] = block[i]
        elif missing and self.is_sharded:
            block = sobol_rows(self.gp, missing)            # row l from the rank that owns output l (one all-gather)
            for i, s in enumerate(missing):
                self._cache[s] = block[..., i]
        elif missing:
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
    """Closed Sobol indices with their standard errors T and the covariances W behind them (reference
    gsa/calibrators.py:146-402). The reference builds rank-7 tensors under "rank equations"; algebraically ("Sobol
    error algebra"; checked against a literal transliteration in tests/test_oracle.py) every entry (a, b) of

        mu_phi_mu (:259-288)  = (1 + delta_ab) phi[a, b]        mu_psi_mu (:311-322) = (1 + delta_ab) psi[a, b]

    comes from O(N^2) pair sums and one |L_b^-1 f|^2, which ``rcgp_sobol_error_terms`` evaluates on the GPU for all
    first-order / closed / total slices at once. This class assembles W = (mu_phi_mu - mu_psi_mu) + transpose (:324-331) and
    T (:333-346) exactly as the reference does, for the DIAGONAL rank equations and, when ``is_T_partial`` is False, the MIXED one.
    """

    @classmethod
    @property
    def META(cls) -> Dict[str, Any]:
        return {'is_T_partial': True}

    def _error_matrices(self, slices: Sequence[Tuple[int, int]]) -> None:
        """Fill the caches W_diag[slice], W_mixed[slice] (L, L) for the given slices."""
        slices = [s for s in dict.fromkeys((int(a), int(b)) for a, b in slices) if s not in self._W_diag]
        if not slices:
            return
        L = self.L
        D = np.zeros((L, L, len(slices)))
        Mx = np.zeros((L, L, len(slices)))
        for b in self.owned:                                           # column b needs the Cholesky factor of output b: its owner's job
            handle = self.gp._select_mo() if self.is_gp_covariant else self.gp._select(b)
            for a in range(L):
                if self.is_gp_covariant:         # both outputs live in the one (LN) system; psi_factor embeds in block b (:304-308)
                    phi_d, psi_d, phi_m, psi_m = handle.sobol_error_terms(slices, a, b)
                elif a == b:
                    phi_d, psi_d, phi_m, psi_m = handle.sobol_error_terms(slices)
                else:
                    phi_d, psi_d, phi_m, psi_m = handle.sobol_error_terms(slices, self.Lambda[a], self.F[a], self.K_inv_Y[a])
                doubled = 2.0 if a == b else 1.0                       # set_diag(2 * diag): gsa/calibrators.py:281,284,322
                D[a, b] = doubled * (phi_d - psi_d)
                Mx[a, b] = doubled * (phi_m - psi_m)
        if self.is_sharded:                                            # columns from their owners
            both = self.gp.gather_output_rows(np.ascontiguousarray(np.stack([D, Mx], axis=2).transpose(1, 0, 2, 3)))   # [b][a][D|Mx][slice]
            D, Mx = both[:, :, 0, :].transpose(1, 0, 2), both[:, :, 1, :].transpose(1, 0, 2)
        for i, s in enumerate(slices):
            self._W_diag[s] = D[..., i] + D[..., i].T                  # _W: W += transpose(W)  (:324-331)
            self._W_mixed[s] = Mx[..., i] + Mx[..., i].T

    def _T(self, Wmm: np.ndarray, WMm: np.ndarray = None, Vm: np.ndarray = None) -> np.ndarray:
        """The index uncertainty (gsa/calibrators.py:333-346)."""
        if self.meta['is_T_partial']:
            Q = Wmm
        else:
            Q = Wmm - 2 * Vm * WMm / self.V[1] + Vm * Vm * self.Q
        return np.sqrt(np.abs(Q) / self.V[4])

    def _calibrate(self):
        super()._calibrate()
        if not self.is_F_diagonal:
            raise NotImplementedError('If the MOGP kernel covariance is not diagonal, the Sobol error calculation is unstable.')   # :380-381
        M = self.M
        # (shared like the conditional variances: one device pass per output pair yields the error terms of EVERY first-order / closed /
        # total slice, and the calibrators of the three kinds on one gp would otherwise each repeat all L^2 of them)
        self._W_diag, self._W_mixed = self._shared_W
        self.V[4] = self.V[2] * self.V[2]                              # :383
        canonical = [(m, m + 1) for m in range(M)] + [(0, m + 1) for m in range(M)] + [(m + 1, M) for m in range(M)]
        self._error_matrices(canonical)
        full = (0, M)
        if self.meta['is_T_partial']:
            self.W = self._W_diag[full]
            self.T = self._T(self.W)
        else:
            self.W = {'DIAGONAL': self._W_diag[full], 'MIXED': self._W_mixed[full]}
            q = np.diagonal(self._W_mixed[full]) / (4.0 * self.V[1] * self.V[1])                       # :400-401
            self.Q = q[None, :] + q[:, None] + 2.0 * np.diag(q)
            self.T = self._T(self._W_diag[full], self._W_mixed[full], self.V[0])

    def marginalize(self, m) -> Dict[str, np.ndarray]:
        """{'V', 'S', 'W', 'T'} for the slice [m[0], m[1]) (gsa/calibrators.py:348-373)."""
        result = super().marginalize(m)
        key = (int(m[0]), int(m[1]))
        self._error_matrices([key])
        Wmm = self._W_diag[key]
        result |= {'W': Wmm, 'T': self._T(Wmm, self._W_mixed[key], result['V'])}
        return result
