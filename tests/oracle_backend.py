"""TEST INFRASTRUCTURE: a stand-in for ``romcomma_amd._lib.RcGP`` that answers from the CPU oracle, so that the HOST logic above the C
ABI -- sharding over ranks, the gathers, the csv stores, bench.py's N > 1 bookkeeping -- can be run with 8 ranks over gloo on a box
without a GPU. Never imported by the product; the product has no CPU path. Small N only (the oracle's Sobol form is O(N^2 M) NumPy).

    sys.path.insert(0, "<repo>/tests"); import oracle_backend; oracle_backend.install()      # patches _lib.RcGP / _lib.device_count in THIS process
"""
import numpy as np

from oracle import gp_oracle as o


class OracleGP:
    """The calls of ``_lib.RcGP`` that HipGP, the Sobol calibrators, OutputShard, fit_lbfgsb and bench.py make (independent outputs)."""

    def __init__(self, X, y, device=0):
        self.X = np.ascontiguousarray(X, dtype=np.float64)
        self.y = np.ascontiguousarray(y, dtype=np.float64)
        self.N, self.M, self.L = self.X.shape[0], self.X.shape[1], 1
        self._theta = None

    # ---- state
    def set_y(self, y):
        self.y = np.ascontiguousarray(y, dtype=np.float64)

    def set_hyper(self, ell, variance, noise):
        self._theta = (np.array(ell, dtype=np.float64), float(variance), float(noise))

    def close(self):
        pass

    def sync(self):
        pass

    # ---- evaluations
    def lml(self):
        return float(o.lml(self.X, self.y, *self._theta))

    def lml_grad(self):
        value, grad = o.lml_and_grad(self.X, self.y, *self._theta)
        return float(value), np.asarray(grad)

    def k_inv_y(self):
        return o.k_inv_y(self.X, self.y, *self._theta)

    def predict(self, Xnew, include_noise=True):
        return o.predict(self.X, self.y, *self._theta, np.asarray(Xnew, dtype=np.float64), include_noise)

    def _weights(self, ell, var, alpha):
        g, phi = o.sobol_prepare(self.X, np.asarray(alpha)[None, :], np.array([var]), np.asarray(ell)[None, :])
        return g[0], phi[0]

    def sobol_closed(self, slices):
        ell, var, _ = self._theta
        g, phi = self._weights(ell, var, self.k_inv_y())
        return o.sobol_V_pair(self.X, g, g, phi, phi, [tuple(int(v) for v in s) for s in slices])

    def sobol_cross(self, ell_j, var_j, alpha_j, slices):
        ell, var, _ = self._theta
        g_l, phi_l = self._weights(ell, var, self.k_inv_y())
        g_j, phi_j = self._weights(ell_j, var_j, alpha_j)
        return o.sobol_V_pair(self.X, g_l, g_j, phi_l, phi_j, [tuple(int(v) for v in s) for s in slices])

    # ---- profiling hooks bench.py drives (no kernels here: nothing recorded)
    def set_profiling(self, on):
        pass

    def profile_sample(self, every):
        pass

    def profile_reset(self):
        pass

    def profile_get(self, cls):
        return 0, 0.0, 0.0

    def stage_gram(self):
        pass

    def stage_potrf(self):
        pass

    def stage_trtri(self):
        pass


def lml_grad_batch(gps):
    """Stand-in for ``_lib.lml_grad_batch``: the units one after the other (a matrix that is not positive definite gives status 1)."""
    lml, grad, status = np.full(len(gps), np.nan), np.full((len(gps), gps[0].M + 2), np.nan), np.zeros(len(gps), dtype=np.int32)
    for u, gp in enumerate(gps):
        try:
            lml[u], grad[u] = gp.lml_grad()
        except np.linalg.LinAlgError:
            status[u] = 1
    return lml, grad, status


def factor_batch(gps):
    """Stand-in for ``_lib.factor_batch``: nothing to cache on the CPU; status 1 where the matrix is not positive definite."""
    status = np.zeros(len(gps), dtype=np.int32)
    for u, gp in enumerate(gps):
        try:
            gp.lml()
        except np.linalg.LinAlgError:
            status[u] = 1
    return status


def install():
    from romcomma_amd import _lib
    _lib.RcGP = OracleGP
    _lib.lml_grad_batch = lml_grad_batch
    _lib.factor_batch = factor_batch
    _lib.device_count = lambda: 1
