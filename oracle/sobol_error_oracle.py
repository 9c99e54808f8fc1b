"""CPU oracle for the Sobol-index standard errors (T, W): an op-by-op NumPy transliteration of
``romcomma.gsa.calibrators.ClosedSobolWithError`` (reference gsa/calibrators.py:146-402) on top of the literal
``LiteralClosedSobol`` of ``gp_oracle``. TEST INFRASTRUCTURE ONLY. PARITY UNPINNED: the reference cannot run here and holds no
fixtures for this path; the transliteration keeps the reference's tensor ranks, einsum strings and rank-equation machinery
line by line so that it can be audited against the source (tf.einsum -> np.einsum, tf.reshape -> np.reshape,
tf.linalg.triangular_solve -> scipy solve_triangular with the batch broadcast written out).

Memory is O(L^2 N^2 M): small N only. Axis letters follow the reference: l,i,j,k outer output indices; L,J inner (size 1 for
independent GPs); N,n samples; M inputs.
"""
from __future__ import annotations

from collections import namedtuple
from typing import Dict, List, Sequence, Tuple

import numpy as np
import scipy.linalg

from oracle.gp_oracle import Gaussian, LiteralClosedSobol

RankEquation = namedtuple('RankEquation', 'l i j k')
RankEquations = namedtuple('RankEquations', 'DIAGONAL MIXED')


def diag_det(tensor):                                                     # gsa/base.py:33-41
    return np.prod(tensor, axis=-1)


def _set_diag(matrix, diagonal):
    out = np.array(matrix, copy=True)
    idx = np.arange(out.shape[-1])
    out[..., idx, idx] = diagonal
    return out


class LiteralClosedSobolWithError(LiteralClosedSobol):
    """gsa/calibrators.py:146-402. ``K_cho`` (L,N,N) is needed here (psi_factor), unlike in plain ClosedSobol; a covariant GP
    passes its (LN,LN) factor (F still diagonal, :380-381) and takes the rank-2 branch of psi_factor."""

    RANK_EQUATIONS = RankEquations(DIAGONAL=(RankEquation(l='j', i='k', j='l', k='i'), RankEquation(l='k', i='j', j='i', k='l')),
                                   MIXED=(RankEquation(l='k', i='k', j='j', k='i'),))                       # :169-170

    def __init__(self, X, K_inv_Y, F, lengthscales, K_cho, is_T_partial: bool = True):
        self.K_cho = np.asarray(K_cho, dtype=np.float64)
        self.meta = {'is_T_partial': is_T_partial}
        self.Ms = (0, np.asarray(X).shape[1])
        super().__init__(X, K_inv_Y, F, lengthscales)

    # ---- :172-191
    def _equateRanks(self, liLNjkJM, rank_eq):
        shape = list(liLNjkJM.shape)
        eqRanks_j = 'j' if shape[4] == 1 else rank_eq.j
        eqRanks_k = 'k' if shape[5] == 1 else rank_eq.k
        liLNjkJM = np.reshape(liLNjkJM, shape[:-2] + [-1])
        if rank_eq in self.RANK_EQUATIONS.MIXED:
            result = np.einsum('iiLNjkS->LNjiS', liLNjkJM)
        else:
            result = np.einsum(f'liLN{eqRanks_j}{eqRanks_k}S->LN{rank_eq.j}{rank_eq.k}S', liLNjkJM)
        result = np.reshape(result, list(result.shape[:-1]) + shape[-2:])
        return np.einsum('LNjjJM->LNjJM', result)[..., None, :, :] if rank_eq.j == 'i' else result

    # ---- :193-212
    def _equatedRanksGaussian(self, mean, variance, ordinate, rank_eqs) -> List[Gaussian]:
        result = []
        N_axis = 3
        for rank_eq in rank_eqs:
            eq_ranks_variance = self._equateRanks(np.expand_dims(variance, N_axis), rank_eq)[..., None, :]
            eq_ranks_mean = self._equateRanks(mean, rank_eq)[..., None, :]
            shape = tuple(eq_ranks_mean.shape[:-2]) + tuple(ordinate.shape[-2:]) if np.ndim(ordinate) > 2 else None
            eq_ranks_mean = (eq_ranks_mean if shape is None else np.broadcast_to(eq_ranks_mean, shape)) - ordinate
            result += [Gaussian(mean=eq_ranks_mean, variance=eq_ranks_variance, LBunch=10000)]
        return result

    # ---- :214-242
    def _OmegaGaussian(self, mp, G, Phi, Upsilon, rank_eqs) -> List[Gaussian]:
        Gamma = 1 - Phi
        Gamma_inv = 1 / Gamma
        Pi = 1 + Phi + np.einsum('ikM,ikM,ikM->ikM', Phi, Gamma_inv, Phi)
        Pi = 1 / Pi
        B = np.einsum('jJM,jJM->jJM', Gamma, Phi)[None, :, None, ...]
        B = B + np.einsum('jJM,ikM,jJM->ijkJM', Phi, Pi, Phi)
        Gamma_reshape = Gamma[:, None, :, None, :]
        C = Gamma_reshape / (1 - np.einsum('lLM,ikM->liLkM', Phi, Upsilon))
        C = np.einsum('ikM,liLkM->liLkM', (1 - Upsilon), C)
        Omega = np.einsum('ikM,ikM,ikM->ikM', Pi, Phi, Gamma_inv)
        Omega = np.einsum('jJM,ikM->ijkJM', Phi, Omega)
        mean = np.einsum('ijkJM,liLkM,lLM,lLNM->liLNjkJM', Omega, C, Gamma_inv, G)
        variance = B[None, :, None, ...] + np.einsum('ijkJM,liLkM,ijkJM->liLjkJM', Omega, C, Omega)
        if tuple(mp) != tuple(self.Ms):
            variance = variance[..., mp[0]:mp[1]]
            mean = mean[..., mp[0]:mp[1]]
            G = G[..., mp[0]:mp[1]]
        return self._equatedRanksGaussian(mean, variance, G[:, None, ...], rank_eqs)

    # ---- :244-257
    def _UpsilonGaussian(self, G, Phi, Upsilon, rank_eqs) -> List[Gaussian]:
        Upsilon_cho = np.sqrt(Upsilon)
        mean = np.einsum('ikM,lLNM->liLNkM', Upsilon_cho, G)[..., None, :, None, :]
        variance = 1 - np.einsum('ikM,lLM,ikM->liLkM', Upsilon_cho, Phi, Upsilon_cho)[..., None, :, None, :]
        return self._equatedRanksGaussian(mean, variance, np.zeros(()), rank_eqs)

    # ---- :259-288
    def _mu_phi_mu(self, GGaussian, UpsilonGaussians, OmegaGaussians, rank_eqs):
        OmegaGaussians = list(OmegaGaussians)
        GGaussian = GGaussian.expand_dims([2])
        mu_phi_mu = 0.0
        for i, rank_eq in enumerate(rank_eqs):
            Om = OmegaGaussians[i] / GGaussian
            Om.exponent = Om.exponent + UpsilonGaussians[i].exponent
            if UpsilonGaussians[i].cho_diag.shape[-1] == GGaussian.cho_diag.shape[-1]:
                Om.cho_diag = Om.cho_diag * UpsilonGaussians[i].cho_diag
            else:
                Om.cho_diag = (diag_det(Om.cho_diag) * diag_det(UpsilonGaussians[i].cho_diag))[..., None]
            if rank_eq in self.RANK_EQUATIONS.MIXED:
                result = np.einsum('kLN,LNjkJn,jJn->jk', self.g0KY, Om.pdf, self.g0KY)
                mu_phi_mu = mu_phi_mu + np.einsum('k,jk->jk', self.mu_phi_mu_pre, result)
                mu_phi_mu = _set_diag(mu_phi_mu, 2 * np.diagonal(mu_phi_mu))
            elif rank_eq.l == 'k' and rank_eq.i == 'j':
                result = np.einsum('jLN,LNjkJn,jJn->j', self.g0KY, Om.pdf, self.g0KY)
                mu_phi_mu = mu_phi_mu + np.diag(np.einsum('j,j->j', self.mu_phi_mu_pre, result))
            else:
                result = np.einsum('jLN,LNjkJn,jJn->jk', self.g0KY, Om.pdf, self.g0KY)
                mu_phi_mu = mu_phi_mu + np.einsum('k,jk->jk', self.mu_phi_mu_pre, result)
        return mu_phi_mu

    # ---- :290-309
    def _psi_factor(self, G, Phi, GGaussian):
        D = Phi[..., None, None, :] - np.einsum('lLM,iIM,lLM->lLiIM', Phi, Phi, Phi)
        mean = np.einsum('lLM,iInM->lLiInM', Phi, G)
        mean = mean[:, :, None, ...] - G[..., None, None, None, :]
        gaussian = Gaussian(mean=mean, variance=D, LBunch=2)
        gaussian = gaussian / GGaussian.expand_dims([-1, -2, -3])
        factor = np.einsum('lLN,iIn,lLNiIn->liIn', self.g0KY, self.g0, gaussian.pdf)
        if self.K_cho.ndim == 2 and factor.shape[-2] == 1:                                           # :304-305, covariant GP with diagonal F:
            lNi = np.einsum('liIN->lNi', factor)                                                     # the vector goes into block i of (L N)
            diag = np.zeros(lNi.shape + (lNi.shape[-1],))
            idx = np.arange(lNi.shape[-1])
            diag[..., idx, idx] = lNi                                                                # tf.linalg.diag
            factor = np.einsum('lNiI->liIN', diag)
            factor = np.reshape(factor, list(factor.shape[:-2]) + [-1, 1])                           # (l, i, L N, 1)
            out = np.empty(factor.shape[:-1])
            for l in range(factor.shape[0]):
                for i in range(factor.shape[1]):
                    out[l, i] = scipy.linalg.solve_triangular(self.K_cho, factor[l, i, :, 0], lower=True, check_finite=False)
            return out
        # rank(K_cho) == 3 for independent GPs: the diag branch is not taken
        factor = np.reshape(factor, list(factor.shape[:-2]) + [-1, 1])                               # (l, i, N, 1)
        out = np.empty(factor.shape[:-1])
        for l in range(factor.shape[0]):
            for i in range(factor.shape[1]):                                                           # batch broadcast: K_cho[i]
                out[l, i] = scipy.linalg.solve_triangular(self.K_cho[i], factor[l, i, :, 0], lower=True, check_finite=False)
        return out

    # ---- :311-322
    def _mu_psi_mu(self, psi_factor, rank_eqs):
        first_psi_factor = self.psi_factor if rank_eqs is self.RANK_EQUATIONS.MIXED else psi_factor
        first_ein = 'liS' if rank_eqs is self.RANK_EQUATIONS.DIAGONAL else 'iiS'
        result = np.einsum(f'{first_ein},liS->li', first_psi_factor, psi_factor)
        return _set_diag(result, 2 * np.diagonal(result))

    # ---- :324-346
    def _W(self, mu_phi_mu, mu_psi_mu):
        W = mu_phi_mu - mu_psi_mu
        return W + W.T

    def _T(self, Wmm, WMm=None, Vm=None):
        if self.meta['is_T_partial']:
            Q = Wmm
        else:
            Q = Wmm - 2 * Vm * WMm / self.V[1] + Vm * Vm * self.Q
        return np.sqrt(np.abs(Q) / self.V[4])

    # ---- :348-373
    def marginalize(self, m: Sequence[int]) -> Dict[str, np.ndarray]:
        result = super().marginalize(m)
        G, Phi, Upsilon = tuple(tensor[..., m[0]:m[1]] for tensor in (self.G, self.Phi, self.Upsilon))
        GGaussian = Gaussian(G, Phi, LBunch=2)
        psi_factor = self._psi_factor(G, Phi, GGaussian)
        if self.meta['is_T_partial']:
            UpsilonGaussians = self._UpsilonGaussian(G, Phi, Upsilon, self.RANK_EQUATIONS.DIAGONAL)
            OmegaGaussians = self._OmegaGaussian(m, self.G, self.Phi, self.Upsilon, self.RANK_EQUATIONS.DIAGONAL)
            Wmm = self._W(self._mu_phi_mu(GGaussian, UpsilonGaussians, OmegaGaussians, self.RANK_EQUATIONS.DIAGONAL),
                          self._mu_psi_mu(psi_factor, self.RANK_EQUATIONS.DIAGONAL))
            result |= {'W': Wmm, 'T': self._T(Wmm)}
        else:
            UpsilonGaussians = RankEquations(*(self._UpsilonGaussian(G, Phi, Upsilon, rank_eqs) for rank_eqs in self.RANK_EQUATIONS))
            OmegaGaussians = RankEquations(*(self._OmegaGaussian(m, self.G, self.Phi, self.Upsilon, rank_eqs) for rank_eqs in self.RANK_EQUATIONS))
            Wmm = self._W(self._mu_phi_mu(GGaussian, UpsilonGaussians.DIAGONAL, OmegaGaussians.DIAGONAL, self.RANK_EQUATIONS.DIAGONAL),
                          self._mu_psi_mu(psi_factor, self.RANK_EQUATIONS.DIAGONAL))
            WMm = self._W(self._mu_phi_mu(GGaussian, self.UpsilonGaussians.MIXED, OmegaGaussians.MIXED, self.RANK_EQUATIONS.MIXED),
                          self._mu_psi_mu(psi_factor, self.RANK_EQUATIONS.MIXED))
            result |= {'W': Wmm, 'T': self._T(Wmm, WMm, result['V'])}
        return result

    # ---- :375-402
    def _calibrate(self):
        super()._calibrate()
        self.Upsilon = self.Lambda2[-1][2]
        self.V[4] = np.einsum('li,li->li', self.V[2], self.V[2])
        self.mu_phi_mu_pre = np.reshape(np.sqrt(np.prod(self.Lambda2[1][0] * self.Lambda2[-1][2], axis=-1)) * self.F, [-1])
        self.GGaussian = Gaussian(mean=self.G, variance=self.Phi, LBunch=2)
        self.psi_factor = self._psi_factor(self.G, self.Phi, self.GGaussian)
        if self.meta['is_T_partial']:
            self.UpsilonGaussians = self._UpsilonGaussian(self.G, self.Phi, self.Upsilon, self.RANK_EQUATIONS.DIAGONAL)
            self.OmegaGaussians = self._OmegaGaussian(self.Ms, self.G, self.Phi, self.Upsilon, self.RANK_EQUATIONS.DIAGONAL)
            self.W = self._W(self._mu_phi_mu(self.GGaussian, self.UpsilonGaussians, self.OmegaGaussians, self.RANK_EQUATIONS.DIAGONAL),
                             self._mu_psi_mu(self.psi_factor, self.RANK_EQUATIONS.DIAGONAL))
            self.T = self._T(self.W)
        else:
            self.UpsilonGaussians = RankEquations(*(self._UpsilonGaussian(self.G, self.Phi, self.Upsilon, rank_eq)
                                                    for rank_eq in self.RANK_EQUATIONS))
            self.OmegaGaussians = RankEquations(*(self._OmegaGaussian(self.Ms, self.G, self.Phi, self.Upsilon, rank_eq)
                                                  for rank_eq in self.RANK_EQUATIONS))
            self.W = RankEquations(*(self._W(self._mu_phi_mu(self.GGaussian, self.UpsilonGaussians[i], self.OmegaGaussians[i], rank_eq),
                                             self._mu_psi_mu(self.psi_factor, rank_eq)) for i, rank_eq in enumerate(self.RANK_EQUATIONS)))
            self.Q = np.diagonal(self.W.MIXED) / (4.0 * self.V[1] * self.V[1])
            self.Q = self.Q[None, ...] + self.Q[..., None] + 2.0 * np.diag(self.Q)
            self.T = self._T(self.W.DIAGONAL, self.W.MIXED, self.V[0])


# --------------------------------------------------------------------------------------------------------------------
# Reduced form (DESIGN.md "Sobol error algebra"): the same quantities from O(N^2) pair sums, for any output pair (a, b).
# Checked against the literal transliteration above in tests/test_oracle.py (<= 1e-8, L = 1 and 2, partial and full T).
# --------------------------------------------------------------------------------------------------------------------

def error_coefficients(phi_a, phi_b, ups_b, kind: str):
    """Per-dimension coefficients (c0, cN, cn, cx) of log q_m(N, n) = c0 + cN x_N^2 + cn x_n^2 + cx x_N x_n.

    kind 'H': the V-kernel H_ab (psi_factor, gsa/calibrators.py:299-303).
    kind 'D': Omega/Upsilon/G Gaussians under the DIAGONAL rank equations (:214-257, :269-277).
    kind 'M': the same under the MIXED rank equation, slice part only (the full-model Upsilon factor depends on N alone)."""
    gam_a, gam_b = 1 - phi_a, 1 - phi_b
    if kind == 'H':
        aa = phi_a * phi_b
        c2 = aa / (1 - aa)
        return -0.5 * np.log1p(-aa), -0.5 * c2 * phi_a, -0.5 * c2 * phi_b, c2
    Pi = 1 / (1 + phi_b + phi_b * phi_b / gam_b)
    B = gam_a * phi_a + phi_a * phi_a * Pi
    Om = phi_a * Pi * phi_b / gam_b
    if kind == 'D':
        C = gam_a * (1 - ups_b) / (1 - phi_a * ups_b)
        mu = Om * C * phi_a / gam_a
        Var = B + Om * Om * C
        den = 1 - ups_b * phi_a
        return (0.5 * np.log(phi_a / Var) - 0.5 * np.log(den), -0.5 * mu * mu / Var - 0.5 * ups_b * phi_a * phi_a / den,
                -0.5 * phi_a * phi_a / Var + 0.5 * phi_a, mu * phi_a / Var)
    if kind == 'M':
        C = gam_b * (1 - ups_b) / (1 - phi_b * ups_b)
        mu = Om * C * phi_b / gam_b
        Var = B + Om * Om * C
        return 0.5 * np.log(phi_a / Var), -0.5 * mu * mu / Var, -0.5 * phi_a * phi_a / Var + 0.5 * phi_a, mu * phi_a / Var
    raise ValueError(kind)


def _pair_matrix(X, coeff, sl):
    c0, cN, cn, cx = (c[sl[0]:sl[1]] for c in coeff)
    XS = X[:, sl[0]:sl[1]]
    t = c0[None, None, :] + cN * XS[:, None, :] ** 2 + cn * XS[None, :, :] ** 2 + cx * XS[:, None, :] * XS[None, :, :]
    return np.exp(t.sum(-1))                                               # [N, n]


def error_terms_pair(X, a, b, g0, g, phi, ups, pre, K_cho, sl):
    """(phi_d, psi_d, phi_m, psi_m) of the output pair (a, b) for the dimension slice ``sl``, WITHOUT the doubling of diagonal
    (a == b) entries: mu_phi_mu_DIAGONAL[a,b] = (1 + delta_ab) phi_d, etc."""
    M = X.shape[1]
    full = (0, M)
    QD = _pair_matrix(X, error_coefficients(phi[a], phi[b], ups[b], 'D'), sl)
    phi_d = pre[b] * (g[a] @ QD @ g[a])
    den = 1 - ups[b] * phi[b]
    g_tilde = g[b] * np.exp(np.sum(-0.5 * ups[b] * phi[b] ** 2 * X * X / den - 0.5 * np.log(den), axis=1))
    QM = _pair_matrix(X, error_coefficients(phi[a], phi[b], ups[b], 'M'), sl)
    phi_m = pre[b] * (g_tilde @ QM @ g[a])
    H = _pair_matrix(X, error_coefficients(phi[a], phi[b], ups[b], 'H'), sl)
    psi = scipy.linalg.solve_triangular(K_cho[b], g0[b] * (H.T @ g[a]), lower=True, check_finite=False)
    Hbb = _pair_matrix(X, error_coefficients(phi[b], phi[b], ups[b], 'H'), full)
    psi_full = scipy.linalg.solve_triangular(K_cho[b], g0[b] * (Hbb.T @ g[b]), lower=True, check_finite=False)
    return phi_d, psi @ psi, phi_m, psi_full @ psi


class ClosedSobolWithErrorOracle:
    """W, T for independent GPs from the reduced form; same results as LiteralClosedSobolWithError, O(N^2) memory."""

    def __init__(self, X, K_inv_Y, F, lengthscales, K_cho, is_T_partial: bool = True):
        from oracle.gp_oracle import ClosedSobolOracle
        self.base = ClosedSobolOracle(X, K_inv_Y, F, lengthscales)
        self.X = np.asarray(X, dtype=np.float64)
        self.N, self.M = self.X.shape
        self.L = self.base.L
        self.is_T_partial = is_T_partial
        alpha = np.asarray(K_inv_Y, dtype=np.float64).reshape(self.L, self.N)
        ell = np.broadcast_to(np.asarray(lengthscales, dtype=np.float64), (self.L, self.M))
        F = np.reshape(np.asarray(F, dtype=np.float64), (self.L,))
        self.phi = 1 / (ell * ell + 1)
        self.ups = 1 / (ell * ell + 2)
        self.g0 = (F * np.sqrt(np.prod(ell * ell * self.phi, axis=1)))[:, None] * np.exp(-0.5 * np.einsum('lm,nm->ln', self.phi, self.X ** 2))
        self.g = self.base.g
        self.pre = F * np.sqrt(np.prod(ell * ell * self.ups, axis=1))                                   # gsa/calibrators.py:384
        self.K_cho = np.asarray(K_cho, dtype=np.float64)
        self.V, self.S = self.base.V, self.base.S
        self.V[4] = self.V[2] * self.V[2]
        full = self._W((0, self.M))
        self.W = full[0]
        if not is_T_partial:
            self.W_mixed = full[1]
            q = np.diagonal(self.W_mixed) / (4.0 * self.V[1] * self.V[1])
            self.Q = q[None, :] + q[:, None] + 2.0 * np.diag(q)
        self.T = self._T(self.W, None if is_T_partial else full[1], self.V[0])

    def _W(self, sl):
        D = np.zeros((self.L, self.L))
        Mx = np.zeros((self.L, self.L))
        for a in range(self.L):
            for b in range(self.L):
                pd, sd, pm, sm = error_terms_pair(self.X, a, b, self.g0, self.g, self.phi, self.ups, self.pre, self.K_cho, sl)
                dbl = 2.0 if a == b else 1.0
                D[a, b] = dbl * (pd - sd)
                Mx[a, b] = dbl * (pm - sm)
        return D + D.T, Mx + Mx.T

    def _T(self, Wmm, WMm, Vm):
        Q = Wmm if self.is_T_partial else Wmm - 2 * Vm * WMm / self.V[1] + Vm * Vm * self.Q
        return np.sqrt(np.abs(Q) / self.V[4])

    def marginalize(self, m):
        result = self.base.marginalize(m)
        Wmm, WMm = self._W(m)
        result |= {'W': Wmm, 'T': self._T(Wmm, WMm, result['V'])}
        return result
