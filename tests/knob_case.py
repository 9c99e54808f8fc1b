"""One knob set of the Cholesky schedule checked against the oracle in a process of its own (the knobs are read once per process).
Run by tests/test_gpu_parity.py::_knob_case with the RCGP_* variables in the environment; prints a line ending in 'ok'."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import gp_oracle as o              # noqa: E402
from romcomma_amd import _lib                  # noqa: E402


def relmax(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


case = sys.argv[1]
if case == 'evaluation':
    N, M = 1700, 4
    X, y = o.synthetic_fold(N, M, k=3)
    ell, var, noise = np.array([0.8, 1.3, 2.0, 2.9]), 1.1, 0.02
    gp = _lib.RcGP(X, y)
    gp.set_hyper(ell, var, noise)
    lml_ref, grad_ref = o.lml_and_grad(X, y, ell, var, noise)
    lml, grad = gp.lml_grad()
    assert abs(lml - lml_ref) <= 1e-10 * abs(lml_ref), (lml, lml_ref)
    np.testing.assert_allclose(grad, grad_ref, rtol=1e-7, atol=1e-9 * np.max(np.abs(grad_ref)))
    assert relmax(gp.k_inv_y(), o.k_inv_y(X, y, ell, var, noise)) < 1e-9
elif case == 'factor':
    N, M = 3400, 3
    X, y = o.synthetic_fold(N, M, k=11)
    ell, var, noise = np.array([0.7, 1.5, 2.4]), 0.9, 0.01
    gp = _lib.RcGP(X, y)
    gp.set_hyper(ell, var, noise)
    Lc = gp.k_cho()
    assert relmax(Lc, o.k_cho(X, ell, var, noise)) < 1e-11
    assert np.all(np.triu(Lc, 1) == 0.0)
    lml_ref = o.lml(X, y, ell, var, noise)
    assert abs(gp.lml() - lml_ref) <= 1e-11 * abs(lml_ref)
elif case == 'tall':
    # taller than the tail (64 blocks) and than the split far update's threshold (40 blocks): outer panels with window pieces and bulk updates in
    # front of the tail, far updates in two launches
    N, M = 9100, 3
    X, y = o.synthetic_fold(N, M, k=5)
    ell, var, noise = np.array([0.9, 1.6, 2.5]), 1.2, 0.015
    gp = _lib.RcGP(X, y)
    gp.set_hyper(ell, var, noise)
    lml_ref, grad_ref = o.lml_and_grad_blas(X, y, ell, var, noise)
    lml, grad = gp.lml_grad()
    assert abs(lml - lml_ref) <= 1e-10 * abs(lml_ref), (lml, lml_ref)
    assert np.max(np.abs(grad - grad_ref)) <= 1e-8 * np.max(np.abs(grad_ref))
else:
    raise SystemExit(f'unknown case {case}')
gp.close()
print(case, 'ok')
