"""ctypes binding of librcgp.so (include/rcgp.h). There is no CPU fallback: if the HIP library is missing or no GPU is
visible the calls raise, loudly."""
from __future__ import annotations

import ctypes
import os
from pathlib import Path
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / 'librcgp.so'

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int32_p = ctypes.POINTER(ctypes.c_int32)
_c_int64_p = ctypes.POINTER(ctypes.c_int64)

# name -> (restype, argtypes); every symbol include/rcgp.h declares.
SIGNATURES = {
    'rcgp_version': (ctypes.c_int, []),
    'rcgp_device_count': (ctypes.c_int, []),
    'rcgp_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int64, ctypes.c_int, _c_double_p, _c_double_p]),
    'rcgp_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'rcgp_last_error': (ctypes.c_char_p, [ctypes.c_void_p]),
    'rcgp_set_y': (ctypes.c_int, [ctypes.c_void_p, _c_double_p]),
    'rcgp_set_hyper': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_double, ctypes.c_double]),
    'rcgp_lml': (ctypes.c_int, [ctypes.c_void_p, _c_double_p]),
    'rcgp_lml_grad': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, _c_double_p]),
    'rcgp_factor': (ctypes.c_int, [ctypes.c_void_p]),
    'rcgp_get_k_inv_y': (ctypes.c_int, [ctypes.c_void_p, _c_double_p]),
    'rcgp_get_k_cho': (ctypes.c_int, [ctypes.c_void_p, _c_double_p]),
    'rcgp_get_gram': (ctypes.c_int, [ctypes.c_void_p, _c_double_p]),
    'rcgp_predict': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, _c_double_p, ctypes.c_int, _c_double_p, _c_double_p]),
    'rcgp_predict_gradient': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, _c_double_p, _c_double_p, _c_double_p]),
    'rcgp_sobol_closed': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_int32_p, _c_double_p]),
    'rcgp_sobol_cross': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_double, _c_double_p, ctypes.c_int, _c_int32_p, _c_double_p]),
    'rcgp_sobol_error_terms': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_double, _c_double_p, ctypes.c_int, _c_int32_p, _c_double_p,
                                              _c_double_p, _c_double_p, _c_double_p]),
    'rcgp_create_mo': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _c_double_p,
                                      _c_double_p]),
    'rcgp_set_hyper_mo': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, _c_double_p, _c_double_p]),
    'rcgp_lml_grad_mo': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    'rcgp_predict_mo': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, _c_double_p, ctypes.c_int, _c_double_p, _c_double_p]),
    'rcgp_predict_gradient_mo': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, _c_double_p, _c_double_p, _c_double_p]),
    'rcgp_sobol_error_terms_mo': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_int32_p, _c_double_p, _c_double_p,
                                                 _c_double_p, _c_double_p]),
    'rcgp_sobol_weight_sum': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_double, _c_double_p, _c_double_p]),
    'rcgp_sobol_pair': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_double, _c_double_p, ctypes.c_double, _c_double_p, ctypes.c_double,
                                       _c_double_p, ctypes.c_double, ctypes.c_int, _c_int32_p, _c_double_p]),
    'rcgp_lml_grad_batch': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), _c_double_p, _c_double_p, ctypes.POINTER(ctypes.c_int)]),
    'rcgp_factor_batch': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]),
    'rcgp_stage_batch': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    'rcgp_stat': (ctypes.c_int64, [ctypes.c_int]),
    'rcgp_stage_gram': (ctypes.c_int, [ctypes.c_void_p]),
    'rcgp_stage_potrf': (ctypes.c_int, [ctypes.c_void_p]),
    'rcgp_stage_trtri': (ctypes.c_int, [ctypes.c_void_p]),
    'rcgp_sync': (ctypes.c_int, [ctypes.c_void_p]),
    'rcgp_set_profiling': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'rcgp_profile_reset': (ctypes.c_int, [ctypes.c_void_p]),
    'rcgp_profile_get': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_int64_p, _c_double_p, _c_double_p]),
}

K_GRAM, K_GEMM, K_DIAG, K_SOBOL, K_MISC, K_GRAD = range(6)
KERNEL_CLASS_NAMES = ('gram', 'gemm', 'diag', 'sobol', 'misc', 'grad')

_lib: Optional[ctypes.CDLL] = None


class RcgpError(RuntimeError):
    """A HIP / argument failure inside librcgp (negative status)."""


class NotPositiveDefiniteError(ValueError):
    """Cholesky failed: leading minor ``k`` is not positive definite (the reference raises tf InvalidArgumentError here)."""

    def __init__(self, k: int, message: str):
        super().__init__(message)
        self.k = k


def load() -> ctypes.CDLL:
    """Load librcgp.so from the package directory (built in-tree by ``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RcgpError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                            f'(or `make -C {_HERE / "csrc"}`). There is no CPU fallback.')
        lib = ctypes.CDLL(str(LIB_PATH))
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
        _lib = lib
    return _lib


def _dp(a: np.ndarray):
    return a.ctypes.data_as(_c_double_p)


def _f64(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f'expected shape {tuple(shape)}, got {a.shape}')
    return a


class RcGP:
    """One independent-output GP resident on one GPU: a thin object wrapper around an ``rcgp_handle``."""

    def __init__(self, X: np.ndarray, y: np.ndarray, device: int = 0):
        self._lib = load()
        self._h = ctypes.c_void_p()
        X = _f64(X)
        if X.ndim != 2:
            raise ValueError('X must be (N, M)')
        self.N, self.M = X.shape
        y = _f64(y).reshape(-1)
        if y.shape[0] != self.N:
            raise ValueError('y must have N entries')
        rc = self._lib.rcgp_create(ctypes.byref(self._h), int(device), self.N, self.M, _dp(X), _dp(y))
        if rc != 0:
            msg = self._lib.rcgp_last_error(None).decode()
            self._h = ctypes.c_void_p()
            raise RcgpError(f'rcgp_create failed ({rc}): {msg}')
        self.device = int(device)

    # -- plumbing
    def _check(self, rc: int, what: str):
        if rc == 0:
            return
        msg = self._lib.rcgp_last_error(self._h).decode()
        if rc > 0:
            raise NotPositiveDefiniteError(rc, f'{what}: {msg}')
        raise RcgpError(f'{what} failed ({rc}): {msg}')

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self._lib.rcgp_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- data and hyper-parameters
    def set_y(self, y):
        y = _f64(y).reshape(-1)
        if y.shape[0] != self.N:
            raise ValueError('y must have N entries')
        self._check(self._lib.rcgp_set_y(self._h, _dp(y)), 'rcgp_set_y')

    def set_hyper(self, ell, variance: float, noise: float):
        ell = np.ascontiguousarray(np.broadcast_to(np.asarray(ell, dtype=np.float64).reshape(-1), (self.M,)))
        self._check(self._lib.rcgp_set_hyper(self._h, _dp(ell), float(variance), float(noise)), 'rcgp_set_hyper')

    # -- the path
    def lml(self) -> float:
        out = ctypes.c_double()
        self._check(self._lib.rcgp_lml(self._h, ctypes.byref(out)), 'rcgp_lml')
        return out.value

    def lml_grad(self) -> Tuple[float, np.ndarray]:
        every = getattr(self, '_profile_every', 0)
        if every:                                        # bench: HIP events around the launches of every n-th evaluation only
            self.set_profiling(self._profile_count % every == 0)
            self._profile_count += 1
        out = ctypes.c_double()
        grad = np.empty(self.M + 2)
        self._check(self._lib.rcgp_lml_grad(self._h, ctypes.byref(out), _dp(grad)), 'rcgp_lml_grad')
        return out.value, grad

    def factor(self):
        self._check(self._lib.rcgp_factor(self._h), 'rcgp_factor')

    def k_inv_y(self) -> np.ndarray:
        out = np.empty(self.N)
        self._check(self._lib.rcgp_get_k_inv_y(self._h, _dp(out)), 'rcgp_get_k_inv_y')
        return out

    def k_cho(self) -> np.ndarray:
        out = np.empty((self.N, self.N))
        self._check(self._lib.rcgp_get_k_cho(self._h, _dp(out)), 'rcgp_get_k_cho')
        return out

    def gram(self) -> np.ndarray:
        out = np.empty((self.N, self.N))
        self._check(self._lib.rcgp_get_gram(self._h, _dp(out)), 'rcgp_get_gram')
        return out

    def predict(self, Xnew, include_noise: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        Xnew = _f64(Xnew)
        if Xnew.ndim != 2 or Xnew.shape[1] != self.M:
            raise ValueError('Xnew must be (n, M)')
        n = Xnew.shape[0]
        mean, sd = np.empty(n), np.empty(n)
        self._check(self._lib.rcgp_predict(self._h, n, _dp(Xnew), int(bool(include_noise)), _dp(mean), _dp(sd)), 'rcgp_predict')
        return mean, sd

    def predict_gradient(self, Xnew) -> Tuple[np.ndarray, np.ndarray]:
        """(mean (n, M), cov (n, M, n, M)) with cov = V^T V, V = L^-1 dK/dx (the caller applies the reference's sign and diagonal term)."""
        Xnew = _f64(Xnew)
        if Xnew.ndim != 2 or Xnew.shape[1] != self.M:
            raise ValueError('Xnew must be (n, M)')
        n = Xnew.shape[0]
        mean, cov = np.empty(n * self.M), np.empty((n * self.M, n * self.M))
        self._check(self._lib.rcgp_predict_gradient(self._h, n, _dp(Xnew), _dp(mean), _dp(cov)), 'rcgp_predict_gradient')
        return mean.reshape(n, self.M), cov.reshape(n, self.M, n, self.M)

    @staticmethod
    def _slices(slices: Sequence[Sequence[int]]) -> np.ndarray:
        s = np.ascontiguousarray(np.asarray(slices, dtype=np.int32).reshape(-1, 2))
        return s

    def sobol_closed(self, slices) -> np.ndarray:
        s = self._slices(slices)
        V = np.empty(s.shape[0])
        self._check(self._lib.rcgp_sobol_closed(self._h, s.shape[0], s.ctypes.data_as(_c_int32_p), _dp(V)), 'rcgp_sobol_closed')
        return V

    def sobol_cross(self, ell_j, var_j: float, alpha_j, slices) -> np.ndarray:
        s = self._slices(slices)
        ell_j = np.ascontiguousarray(np.broadcast_to(np.asarray(ell_j, dtype=np.float64).reshape(-1), (self.M,)))
        alpha_j = _f64(alpha_j).reshape(-1)
        if alpha_j.shape[0] != self.N:
            raise ValueError('alpha_j must have N entries')
        V = np.empty(s.shape[0])
        self._check(self._lib.rcgp_sobol_cross(self._h, _dp(ell_j), float(var_j), _dp(alpha_j), s.shape[0],
                                               s.ctypes.data_as(_c_int32_p), _dp(V)), 'rcgp_sobol_cross')
        return V

    def sobol_error_terms(self, slices, ell_a=None, var_a: float = 0.0, alpha_a=None):
        """(phi_d, psi_d, phi_m, psi_m), each (n_slices,), for the output pair (a, b = this handle); a = b when ``ell_a`` is None."""
        s = self._slices(slices)
        n = s.shape[0]
        out = [np.empty(n) for _ in range(4)]
        if ell_a is None:
            ell_p, alpha_p = None, None
        else:
            ell_a = np.ascontiguousarray(np.broadcast_to(np.asarray(ell_a, dtype=np.float64).reshape(-1), (self.M,)))
            alpha_a = _f64(alpha_a).reshape(-1)
            if alpha_a.shape[0] != self.N:
                raise ValueError('alpha_a must have N entries')
            ell_p, alpha_p = _dp(ell_a), _dp(alpha_a)
        self._check(self._lib.rcgp_sobol_error_terms(self._h, ell_p, float(var_a), alpha_p, n, s.ctypes.data_as(_c_int32_p),
                                                     *[_dp(o) for o in out]), 'rcgp_sobol_error_terms')
        return tuple(out)

    # -- stages and profiling (bench / kernel tests)
    def stage_gram(self):
        self._check(self._lib.rcgp_stage_gram(self._h), 'rcgp_stage_gram')

    def stage_potrf(self):
        self._check(self._lib.rcgp_stage_potrf(self._h), 'rcgp_stage_potrf')

    def stage_trtri(self):
        self._check(self._lib.rcgp_stage_trtri(self._h), 'rcgp_stage_trtri')

    def sync(self):
        self._check(self._lib.rcgp_sync(self._h), 'rcgp_sync')

    def set_profiling(self, on: bool):
        self._check(self._lib.rcgp_set_profiling(self._h, int(bool(on))), 'rcgp_set_profiling')

    def profile_sample(self, every: int):
        """Profile the kernel launches of every ``every``-th LML+gradient evaluation from now on (0: leave profiling as it is)."""
        self._profile_every, self._profile_count = int(every), 0

    def profile_reset(self):
        self._check(self._lib.rcgp_profile_reset(self._h), 'rcgp_profile_reset')

    def profile_get(self, cls: int) -> Tuple[int, float, float]:
        n, ms, work = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        self._check(self._lib.rcgp_profile_get(self._h, int(cls), ctypes.byref(n), ctypes.byref(ms), ctypes.byref(work)),
                    'rcgp_profile_get')
        return n.value, ms.value, work.value


class RcMOGP(RcGP):
    """One covariant GP over L outputs (the reference's romcomma.gpf.models.MOGPR) resident on one GPU. The (L N) axis of
    ``k_inv_y`` / ``k_cho`` / ``gram`` is output-major."""

    def __init__(self, X: np.ndarray, Y: np.ndarray, device: int = 0):
        self._lib = load()
        self._h = ctypes.c_void_p()
        X = _f64(X)
        Y = _f64(Y)
        if X.ndim != 2 or Y.ndim != 2 or Y.shape[0] != X.shape[0]:
            raise ValueError('X must be (N, M) and Y (N, L)')
        (self.N, self.M), self.L = X.shape, Y.shape[1]
        rc = self._lib.rcgp_create_mo(ctypes.byref(self._h), int(device), self.N, self.M, self.L, _dp(X), _dp(Y))
        if rc != 0:
            msg = self._lib.rcgp_last_error(None).decode()
            self._h = ctypes.c_void_p()
            raise RcgpError(f'rcgp_create_mo failed ({rc}): {msg}')
        self.device = int(device)

    def set_y(self, Y):
        Y = _f64(Y, (self.N, self.L))
        self._check(self._lib.rcgp_set_y(self._h, _dp(Y)), 'rcgp_set_y')

    def set_hyper(self, ell, F, Sigma):
        ell = np.ascontiguousarray(np.broadcast_to(np.asarray(ell, dtype=np.float64), (self.L, self.M)))
        F, Sigma = _f64(F, (self.L, self.L)), _f64(Sigma, (self.L, self.L))
        self._check(self._lib.rcgp_set_hyper_mo(self._h, _dp(ell), _dp(F), _dp(Sigma)), 'rcgp_set_hyper_mo')

    def lml_grad(self) -> Tuple[float, np.ndarray, np.ndarray, np.ndarray]:
        """(lml, d/dF (L, L), d/dell (L, M), d/dSigma (L, L)), every entry treated as an independent variable."""
        out = ctypes.c_double()
        g_ell, g_F, g_S = np.empty((self.L, self.M)), np.empty((self.L, self.L)), np.empty((self.L, self.L))
        self._check(self._lib.rcgp_lml_grad_mo(self._h, ctypes.byref(out), _dp(g_ell), _dp(g_F), _dp(g_S)), 'rcgp_lml_grad_mo')
        return out.value, g_F, g_ell, g_S

    def k_inv_y(self) -> np.ndarray:
        out = np.empty((self.L, 1, self.N))
        self._check(self._lib.rcgp_get_k_inv_y(self._h, _dp(out)), 'rcgp_get_k_inv_y')
        return out

    def k_cho(self) -> np.ndarray:
        out = np.empty((self.L * self.N, self.L * self.N))
        self._check(self._lib.rcgp_get_k_cho(self._h, _dp(out)), 'rcgp_get_k_cho')
        return out

    def gram(self) -> np.ndarray:
        out = np.empty((self.L * self.N, self.L * self.N))
        self._check(self._lib.rcgp_get_gram(self._h, _dp(out)), 'rcgp_get_gram')
        return out

    def predict(self, Xnew, include_noise: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        Xnew = _f64(Xnew)
        if Xnew.ndim != 2 or Xnew.shape[1] != self.M:
            raise ValueError('Xnew must be (n, M)')
        n = Xnew.shape[0]
        mean, sd = np.empty((n, self.L)), np.empty((n, self.L))
        self._check(self._lib.rcgp_predict_mo(self._h, n, _dp(Xnew), int(bool(include_noise)), _dp(mean), _dp(sd)), 'rcgp_predict_mo')
        return mean, sd

    def predict_gradient(self, Xnew) -> Tuple[np.ndarray, np.ndarray]:
        """(mean (L, n, M), cov (L_train, L, n, M, L, n, M)): cov[Lb] = V_Lb^T V_Lb over the rows of training output block Lb only."""
        Xnew = _f64(Xnew)
        if Xnew.ndim != 2 or Xnew.shape[1] != self.M:
            raise ValueError('Xnew must be (n, M)')
        n = Xnew.shape[0]
        R = self.L * n * self.M
        mean, cov = np.empty(R), np.empty((self.L, R, R))
        self._check(self._lib.rcgp_predict_gradient_mo(self._h, n, _dp(Xnew), _dp(mean), _dp(cov)), 'rcgp_predict_gradient_mo')
        return mean.reshape(self.L, n, self.M), cov.reshape(self.L, self.L, n, self.M, self.L, n, self.M)

    def sobol_closed(self, slices):
        raise NotImplementedError('use romcomma_amd.gsa.calibrators.covariant_V on a covariant GP')

    sobol_cross = sobol_closed

    def sobol_error_terms(self, slices, out_a: int, out_b: int):
        """(phi_d, psi_d, phi_m, psi_m), each (n_slices,), for the output pair (out_a, out_b) with F taken as diagonal."""
        s = self._slices(slices)
        n = s.shape[0]
        out = [np.empty(n) for _ in range(4)]
        self._check(self._lib.rcgp_sobol_error_terms_mo(self._h, int(out_a), int(out_b), n, s.ctypes.data_as(_c_int32_p),
                                                        *[_dp(o) for o in out]), 'rcgp_sobol_error_terms_mo')
        return tuple(out)


def sobol_weight_sum(gp: RcGP, phi, pre: float, alpha) -> float:
    phi, alpha = _f64(phi, (gp.M,)), _f64(alpha, (gp.N,))
    out = ctypes.c_double()
    gp._check(gp._lib.rcgp_sobol_weight_sum(gp._h, _dp(phi), float(pre), _dp(alpha), ctypes.byref(out)), 'rcgp_sobol_weight_sum')
    return out.value


def sobol_pair(gp: RcGP, phi_a, pre_a: float, alpha_a, shift_a: float, phi_b, pre_b: float, alpha_b, shift_b: float, slices) -> np.ndarray:
    s = RcGP._slices(slices)
    phi_a, phi_b, alpha_a, alpha_b = _f64(phi_a, (gp.M,)), _f64(phi_b, (gp.M,)), _f64(alpha_a, (gp.N,)), _f64(alpha_b, (gp.N,))
    V = np.empty(s.shape[0])
    gp._check(gp._lib.rcgp_sobol_pair(gp._h, _dp(phi_a), float(pre_a), _dp(alpha_a), float(shift_a), _dp(phi_b), float(pre_b), _dp(alpha_b),
                                      float(shift_b), s.shape[0], s.ctypes.data_as(_c_int32_p), _dp(V)), 'rcgp_sobol_pair')
    return V


MAX_BATCH = 16         # RC_MAX_BATCH of the library: units per batched call


def _handles(gps: Sequence[RcGP]):
    if not 1 <= len(gps) <= MAX_BATCH:
        raise ValueError(f'a batched call takes 1..{MAX_BATCH} units, got {len(gps)}')
    return (ctypes.c_void_p * len(gps))(*[gp._h for gp in gps])


def _batch_failure(gps: Sequence[RcGP], rc: int, what: str):
    raise RcgpError(f'{what} failed ({rc}): {gps[0]._lib.rcgp_last_error(gps[0]._h).decode()}')


def lml_grad_batch(gps: Sequence[RcGP]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """LML and gradient of several units in ONE schedule on the GPU (``rcgp_lml_grad_batch``): (lml (n,), grad (n, M + 2),
    status (n,)); status[u] = k > 0 where unit u's matrix is not positive definite at leading minor k (its numbers are NaN)."""
    handles = _handles(gps)                               # (checks the number of units)
    n, M = len(gps), gps[0].M
    lml, grad, status = np.empty(n), np.empty((n, M + 2)), np.zeros(n, dtype=np.int32)
    rc = gps[0]._lib.rcgp_lml_grad_batch(n, handles, _dp(lml), _dp(grad), status.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    if rc != 0:
        _batch_failure(gps, rc, 'rcgp_lml_grad_batch')
    return lml, grad, status


def factor_batch(gps: Sequence[RcGP]) -> np.ndarray:
    """``rcgp_factor`` for several units in one schedule; returns the status word of every unit."""
    handles = _handles(gps)
    status = np.zeros(len(gps), dtype=np.int32)
    rc = gps[0]._lib.rcgp_factor_batch(len(gps), handles, status.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    if rc != 0:
        _batch_failure(gps, rc, 'rcgp_factor_batch')
    return status


def stage_batch(stage: int, gps: Sequence[RcGP]):
    """Stage 0 (Gram), 1 (Cholesky) or 2 (L^-1 + alpha) on all units (bench / kernel tests)."""
    rc = gps[0]._lib.rcgp_stage_batch(int(stage), len(gps), _handles(gps))
    if rc != 0:
        _batch_failure(gps, rc, 'rcgp_stage_batch')


def stat() -> dict:
    """Process-wide work counters of the library (``rcgp_stat``), in units."""
    lib = load()
    return {name: int(lib.rcgp_stat(i)) for i, name in enumerate(('factorisations', 'inversions', 'gradients', 'batched_calls'))}


def device_count() -> int:
    return int(load().rcgp_device_count())
