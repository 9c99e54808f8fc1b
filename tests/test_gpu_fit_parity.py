"""End-to-end fit parity as SURVEY.md 8c defines it: the same SciPy L-BFGS-B (options maxiter 5000, gtol 1e-16, the reference's
gpr/models.py:327-330) drives the CPU oracle and the HIP backend from the same start; the optima must agree -- LML* within 1e-5
relative (north_star's tolerance), and, as the objective is evaluated to ~1e-12 on both sides, the whole trajectory in practice."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_independent_gp_fit_reaches_the_oracle_optimum(gpu):
    from oracle import gp_oracle as o
    from romcomma_amd import _lib
    from romcomma_amd.gpr.optimize import fit_lbfgsb
    X, y = o.synthetic_fold(400, 4, k=2)
    out = o.fit(X, y, 5.0 * np.ones(4), 2.0, 0.02)
    with _lib.RcGP(X, y) as gp:
        fit = fit_lbfgsb(gp, 5.0 * np.ones(4), 2.0, 0.02)
    ell_o, var_o, noise_o, lml_o = out['ell'], out['var'], out['noise'], out['lml']
    assert abs(fit['nfev'] - out['nfev']) <= max(3, out['nfev'] // 10)           # same driver, same path up to rounding
    assert fit['log_marginal'] == pytest.approx(lml_o, rel=1e-5)
    assert o.lml(X, y, fit['lengthscales'], fit['variance'], fit['noise']) == pytest.approx(fit['log_marginal'], rel=1e-9)
    np.testing.assert_allclose(fit['lengthscales'], ell_o, rtol=2e-3)
    assert fit['variance'] == pytest.approx(var_o, rel=2e-3) and fit['noise'] == pytest.approx(noise_o, rel=2e-3)


def test_covariant_gp_fit_reaches_the_oracle_optimum(gpu):
    from oracle import mogp_oracle as mo
    from oracle import gp_oracle as o
    from romcomma_amd import _lib
    from romcomma_amd.gpr.optimize import fit_lbfgsb_mo
    N, M, L = 220, 3, 2
    X, _ = o.synthetic_fold(N, M, k=4)
    Y = np.stack([o.synthetic_fold(N, M, k=4, l=l)[1] for l in range(L)], axis=1)
    ell0 = np.array([[1.2, 2.0, 3.0], [1.0, 2.5, 2.0]])
    F0, S0 = 2.0 * np.eye(L), mo.initial_noise(0.02, L)
    for trainable in ({}, {'kernel_covariance': True, 'lengthscales': True}):
        ell_o, F_o, S_o, lml_o, res_o, nfev_o = mo.fit(X, Y, ell0, F0, S0, trainable=trainable)
        with _lib.RcMOGP(X, Y) as gp:
            fit = fit_lbfgsb_mo(gp, ell0, F0, S0, train_kernel_covariance=bool(trainable.get('kernel_covariance', False)),
                                train_lengthscales=bool(trainable.get('lengthscales', False)))
        assert fit['log_marginal'] == pytest.approx(lml_o, rel=1e-5)
        assert mo.lml(X, Y, fit['lengthscales'], fit['variance'], fit['noise']) == pytest.approx(fit['log_marginal'], rel=1e-9)
        np.testing.assert_allclose(fit['variance'], F_o, rtol=5e-3, atol=5e-3 * np.abs(F_o).max())
        np.testing.assert_allclose(fit['noise'], S_o, rtol=5e-3, atol=5e-3 * np.abs(S_o).max())
        np.testing.assert_allclose(fit['lengthscales'], ell_o, rtol=5e-3)
