"""Second opinions on the NumPy oracle, written in torch-CPU (fp64) -- the closest thing to the reference's TensorFlow available here.

THIS IS TEST INFRASTRUCTURE (tests/test_oracle_pins.py only). PARITY STAYS UNPINNED BY THE REFERENCE: TensorFlow and GPflow cannot be
imported in this image, so nothing here is the reference's own output; these are independent re-statements that the oracle must agree with.

* ``lml_autograd``: the log marginal likelihood the way GPflow builds it at the reference's call site (gpr/models.py:359-361,
  ``gf.optimizers.Scipy().minimize(gp.training_loss, gp.trainable_variables)``): K via the ``-2ZZ' + s + s'`` expansion, ``cholesky``,
  ``triangular_solve``, ``multivariate_normal`` -- and its gradient by REVERSE-MODE AUTODIFF THROUGH THE CHOLESKY, which is the reference's
  mechanism (the oracle and the HIP path use the analytic trace formula instead).
* ``TorchGaussian`` / ``TorchClosedSobol``: gsa/base.py:92-126 and gsa/calibrators.py:60-143 re-typed op for op with the torch
  counterpart of each TF op (tf.einsum -> torch.einsum, tf.expand_dims -> unsqueeze, tf.newaxis -> None, tf.broadcast_to ->
  broadcast_to, tf.reduce_prod -> prod, tf.linalg.diag_part -> diagonal), independent outputs (is_F_diagonal=True).
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import numpy as np
import torch

F64 = torch.float64


def lml_autograd(X: np.ndarray, y: np.ndarray, ell: np.ndarray, var: float, noise: float):
    """(LML, dLML/d ell, dLML/d var, dLML/d noise) with the gradient from torch autograd through ``torch.linalg.cholesky``."""
    Xt, yt = torch.as_tensor(X, dtype=F64), torch.as_tensor(y, dtype=F64).reshape(-1, 1)
    ell_t = torch.tensor(np.asarray(ell, dtype=np.float64), requires_grad=True)
    var_t = torch.tensor(float(var), dtype=F64, requires_grad=True)
    noise_t = torch.tensor(float(noise), dtype=F64, requires_grad=True)
    Z = Xt / ell_t
    s = (Z * Z).sum(-1)
    D = -2.0 * Z @ Z.T + s[:, None] + s[None, :]                                # GPflow square_distance: no clamp
    K = var_t * torch.exp(-0.5 * D) + noise_t * torch.eye(Xt.shape[0], dtype=F64)
    L = torch.linalg.cholesky(K)
    a = torch.linalg.solve_triangular(L, yt, upper=False)                       # GPflow multivariate_normal
    N = Xt.shape[0]
    lml = -0.5 * (a * a).sum() - 0.5 * N * math.log(2.0 * math.pi) - torch.log(torch.diagonal(L)).sum()
    lml.backward()
    return float(lml.detach()), ell_t.grad.numpy().copy(), float(var_t.grad), float(noise_t.grad)


class TorchGaussian:
    """gsa/base.py:52-126, diagonal-variance branch, in torch."""

    def __init__(self, mean: torch.Tensor, variance: torch.Tensor, ordinate: torch.Tensor | None = None, LBunch: int = 2):
        variance_cho = torch.sqrt(variance)                                                            # :107
        ordinate = torch.zeros((), dtype=F64) if ordinate is None else ordinate
        if ordinate.shape == mean.shape:                                                               # :108-112
            shape = list(ordinate.shape)
            fill = [1, ] * (len(shape) - 1)
            ordinate = torch.reshape(ordinate, shape[:-1] + fill + [shape[-1]])
            mean = torch.reshape(mean, fill + shape)
        ordinate = ordinate - mean                                                                     # :113
        insertions = variance_cho.dim() - 1                                                            # :115
        insertions -= insertions % LBunch                                                              # :116
        for axis in range(insertions, 0, -LBunch):                                                     # :117-118
            variance_cho = torch.unsqueeze(variance_cho, axis)
        target = list(variance_cho.shape[:-2]) + list(ordinate.shape[-2:])
        exponent = ordinate / torch.broadcast_to(variance_cho, target)                                 # :121
        self.exponent = -0.5 * torch.einsum('...o,...o->...', exponent, exponent)                      # :124
        self.cho_diag = variance_cho                                                                   # :126

    @property
    def det(self) -> torch.Tensor:                                                                     # :58-61
        return torch.prod(self.cho_diag, dim=-1)

    @property
    def pdf(self) -> torch.Tensor:                                                                     # :63-66
        return torch.exp(self.exponent) / self.det

    def expand_dims(self, axes: Sequence[int]) -> 'TorchGaussian':                                     # :68-79
        result = TorchGaussian.__new__(TorchGaussian)
        result.exponent, result.cho_diag = self.exponent, self.cho_diag
        for axis in sorted(axes, reverse=True):
            # tf.expand_dims(x, -1) appends; torch.unsqueeze(x, -1) does too (both count a negative axis from rank + 1)
            result.exponent = torch.unsqueeze(result.exponent, axis)
            result.cho_diag = torch.unsqueeze(result.cho_diag, (axis - 1) if axis < 0 else axis)
        return result

    def __truediv__(self, other: 'TorchGaussian') -> 'TorchGaussian':                                  # :81-90
        result = TorchGaussian.__new__(TorchGaussian)
        result.exponent = self.exponent - other.exponent
        result.cho_diag = self.cho_diag / other.cho_diag
        return result


class TorchClosedSobol:
    """gsa/calibrators.py:60-143 for independent outputs: X (N,M), K_inv_Y (L,1,N), F (L,) or (1,L), lengthscales (L,M) or (1,M)."""

    def __init__(self, X: np.ndarray, K_inv_Y: np.ndarray, F: np.ndarray, lengthscales: np.ndarray):
        self.X = torch.as_tensor(np.asarray(X), dtype=F64)
        self.N, self.M = self.X.shape
        self.K_inv_Y = torch.as_tensor(np.asarray(K_inv_Y), dtype=F64)
        self.L = self.K_inv_Y.shape[0]
        self.F = torch.reshape(torch.as_tensor(np.asarray(F), dtype=F64), [self.L, 1])                 # :134-136
        self.Lambda = torch.broadcast_to(torch.as_tensor(np.asarray(lengthscales), dtype=F64), [self.L, self.M])   # :140
        result = torch.einsum('lM,lM->lM', self.Lambda, self.Lambda)[:, None, :]                        # :105
        result = tuple(result + j for j in range(3))                                                    # :108
        self.Lambda2 = {1: result, -1: tuple(value ** (-1) for value in result)}                        # :109
        self._calibrate()

    def _calibrate(self):                                                                               # :82-97
        pre_factor = torch.sqrt(torch.prod(self.Lambda2[1][0] * self.Lambda2[-1][1], dim=-1)) * self.F
        self.g0 = torch.exp(TorchGaussian(mean=self.X[None, None, ...], variance=self.Lambda2[1][1]).exponent)
        self.g0 = self.g0 * pre_factor[..., None]
        self.g0KY = self.g0 * self.K_inv_Y
        self.g0KY = self.g0KY - torch.einsum('lLN->l', self.g0KY)[..., None, None] / float(np.prod(self.g0KY.shape[1:]))
        self.G = torch.einsum('lLM,NM->lLNM', self.Lambda2[-1][1], self.X)
        self.Phi = self.Lambda2[-1][1]
        self.V = {0: self._V(self.G, self.Phi)}
        self.V[1] = torch.diagonal(self.V[0])
        V = torch.sqrt(self.V[1])
        self.V[2] = torch.einsum('l,i->li', V, V)
        self.S = self.V[0] / self.V[2]

    def _V(self, G: torch.Tensor, Phi: torch.Tensor) -> torch.Tensor:                                   # :60-80
        Gamma = 1 - Phi
        Psi = torch.unsqueeze(torch.unsqueeze(Gamma, 2), 2) + Gamma[None, None, ...]
        Psi = Psi - torch.einsum('lLM,jJM->lLjJM', Gamma, Gamma)
        PsiPhi = torch.einsum('lLjJM,lLM->lLjJM', Psi, Phi)
        PhiG = torch.unsqueeze(torch.einsum('lLM,jJnM->lLjJnM', Phi, G), 2)
        PhiGauss = TorchGaussian(mean=G, variance=Phi)
        H = TorchGaussian(mean=PhiG, variance=PsiPhi, ordinate=G[..., None, None, None, :])
        H = H / PhiGauss.expand_dims([-1, -2, -3])
        return torch.einsum('lLN,lLNjJn,jJn->lj', self.g0KY, H.pdf, self.g0KY)

    def marginalize(self, m: Sequence[int]) -> Dict[str, np.ndarray]:                                   # :49-58
        G, Phi = self.G[..., m[0]:m[1]], self.Phi[..., m[0]:m[1]]
        V = self._V(G, Phi)
        return {'V': V.numpy(), 'S': (V / self.V[2]).numpy()}
