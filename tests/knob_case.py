"""One knob set of the Cholesky schedule checked against the oracle in a process of its own (the knobs are read once per process).
Run by tests/test_gpu_parity.py::_knob_case with the RCGP_* variables in the environment; prints '<case> ok <fingerprint>', the
fingerprint being the GPU's numbers as hex floats (the parent holds every knob set of a case to the SAME bits: a tile adds up its
k-slabs in one fixed order whatever the schedule). The oracle's reference values are computed by the first child of a case and kept in
RCGP_KNOB_REF_DIR for the others (at N = 9100 they cost more than the GPU side)."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import gp_oracle as o              # noqa: E402
from romcomma_amd import _lib                  # noqa: E402


def relmax(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


def reference(case, compute):
    """The oracle's values for `case`: from the parent's cache directory if an earlier child left them there, else computed (and left)."""
    folder = os.environ.get('RCGP_KNOB_REF_DIR')
    path = Path(folder) / f'{case}.npz' if folder else None
    if path is not None and path.exists():
        with np.load(path) as z:
            return [z[f'a{i}'] for i in range(len(z.files))]
    values = [np.asarray(v) for v in compute()]
    if path is not None:
        tmp = path.with_suffix(f'.{os.getpid()}.tmp.npz')
        np.savez(tmp, **{f'a{i}': v for i, v in enumerate(values)})
        os.replace(tmp, path)
    return values


def fingerprint(*arrays):
    return ','.join(float(v).hex() for a in arrays for v in np.asarray(a, dtype=np.float64).ravel())


case = sys.argv[1]
if case == 'evaluation':
    N, M = 1700, 4
    X, y = o.synthetic_fold(N, M, k=3)
    ell, var, noise = np.array([0.8, 1.3, 2.0, 2.9]), 1.1, 0.02
    gp = _lib.RcGP(X, y)
    gp.set_hyper(ell, var, noise)
    lml_ref, grad_ref, alpha_ref = reference(case, lambda: (*o.lml_and_grad(X, y, ell, var, noise), o.k_inv_y(X, y, ell, var, noise)))
    lml, grad = gp.lml_grad()
    assert abs(lml - lml_ref) <= 1e-10 * abs(lml_ref), (lml, lml_ref)
    np.testing.assert_allclose(grad, grad_ref, rtol=1e-7, atol=1e-9 * np.max(np.abs(grad_ref)))
    alpha = gp.k_inv_y()
    assert relmax(alpha, alpha_ref) < 1e-9
    print_fp = fingerprint(lml, grad, alpha[::97])
elif case == 'factor':
    N, M = 3400, 3
    X, y = o.synthetic_fold(N, M, k=11)
    ell, var, noise = np.array([0.7, 1.5, 2.4]), 0.9, 0.01
    gp = _lib.RcGP(X, y)
    gp.set_hyper(ell, var, noise)
    Lc = gp.k_cho()
    Lref, lml_ref = reference(case, lambda: (o.k_cho(X, ell, var, noise), o.lml(X, y, ell, var, noise)))
    assert relmax(Lc, Lref) < 1e-11
    assert np.all(np.triu(Lc, 1) == 0.0)
    lml = gp.lml()
    assert abs(lml - lml_ref) <= 1e-11 * abs(lml_ref)
    print_fp = fingerprint(lml, Lc[::211, ::13], np.diag(Lc)[::7])
elif case == 'tall':
    # taller than the default tail (64 blocks) and than the split far update's threshold (40 blocks): outer panels with window pieces and bulk
    # updates in front of the tail, far updates in two launches
    N, M = 9100, 3
    X, y = o.synthetic_fold(N, M, k=5)
    ell, var, noise = np.array([0.9, 1.6, 2.5]), 1.2, 0.015
    gp = _lib.RcGP(X, y)
    gp.set_hyper(ell, var, noise)
    lml_ref, grad_ref = reference(case, lambda: o.lml_and_grad_blas(X, y, ell, var, noise))
    lml, grad = gp.lml_grad()
    assert abs(lml - lml_ref) <= 1e-10 * abs(lml_ref), (lml, lml_ref)
    assert np.max(np.abs(grad - grad_ref)) <= 1e-8 * np.max(np.abs(grad_ref))
    print_fp = fingerprint(lml, grad, gp.k_inv_y()[::401])
else:
    raise SystemExit(f'unknown case {case}')
gp.close()
print(case, 'ok', print_fp)
