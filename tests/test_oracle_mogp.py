"""CPU checks that pin oracle/mogp_oracle.py internally (parity unpinned: the reference's gpf path needs TensorFlow/GPflow):
two formulations of the covariant kernel, analytic gradients against central differences (symmetric perturbations and the
Cholesky parametrisation), the L = 1 limit against gp_oracle, predict against the K_inv_Y identity, the literal Sobol
transliteration with a non-diagonal F against the pair form, and the fit's default trainable set."""
import numpy as np
import pytest

from oracle import gp_oracle as go
from oracle import mogp_oracle as mo


def _case(N=24, M=3, L=3, seed=1):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, M))
    Y = rng.standard_normal((N, L))
    ell = 0.6 + 2.0 * rng.random((L, M))
    C = np.tril(0.4 * rng.standard_normal((L, L)), -1) + np.diag(0.7 + rng.random(L))
    Cn = np.tril(0.05 * rng.standard_normal((L, L)), -1) + np.diag(0.1 + 0.1 * rng.random(L))
    return X, Y, ell, C @ C.T, Cn @ Cn.T


def test_literal_broadcast_kernel_equals_stacked_points_form():
    X, Y, ell, F, S = _case()
    L, N = F.shape[0], X.shape[0]
    assert np.abs(mo.unit_gram_literal(X, ell).reshape(L * N, L * N) - mo.unit_gram(X, ell)).max() < 1e-14
    Xs = np.random.default_rng(2).standard_normal((5, X.shape[1]))
    assert np.abs(mo.unit_gram_literal(X, ell, Xs).reshape(L * N, L * 5) - mo.unit_gram(X, ell, Xs)).max() < 1e-14
    K = mo.noisy_gram(X, ell, F, S)
    assert np.allclose(K, K.T) and np.all(np.linalg.eigvalsh(K) > 0)
    # block (l, j): F_lj E + Sigma_lj I
    assert K[0 * N + 3, 1 * N + 3] == pytest.approx(F[0, 1] * np.exp(-0.5 * np.sum((X[3] / ell[0] - X[3] / ell[1]) ** 2)) + S[0, 1])


def _fd_sym(f, x, eps=1e-6):
    g = np.zeros_like(x)
    for i in range(x.shape[0]):
        for j in range(i + 1):
            xp, xm = x.copy(), x.copy()
            xp[i, j] += eps
            xm[i, j] -= eps
            if i != j:
                xp[j, i] += eps
                xm[j, i] -= eps
            g[i, j] = g[j, i] = (f(xp) - f(xm)) / (2 * eps)
    return g


def test_gradient_against_central_differences():
    X, Y, ell, F, S = _case()
    v, dF, dell, dS = mo.lml_and_grad(X, Y, ell, F, S)
    assert v == pytest.approx(mo.lml(X, Y, ell, F, S), rel=1e-13)
    g = np.zeros_like(ell)
    for idx in np.ndindex(ell.shape):
        ep, em = ell.copy(), ell.copy()
        ep[idx] += 1e-6
        em[idx] -= 1e-6
        g[idx] = (mo.lml(X, Y, ep, F, S) - mo.lml(X, Y, em, F, S)) / 2e-6
    assert np.abs(g - dell).max() < 1e-6 * max(1.0, np.abs(dell).max())
    # a symmetric perturbation of an off-diagonal entry moves two independent entries
    for got, fd in ((dF, _fd_sym(lambda f: mo.lml(X, Y, ell, f, S), F)), (dS, _fd_sym(lambda s: mo.lml(X, Y, ell, F, s), S))):
        expect = got + got.T - np.diag(np.diag(got))
        assert np.abs(fd - expect).max() < 2e-6 * max(1.0, np.abs(expect).max())


def test_cholesky_parametrisation_round_trip_and_chain_rule():
    X, Y, ell, F, S = _case()
    kd, kl = mo.variance_to_params(F)
    C = mo.params_to_cholesky(kd, kl)
    assert np.allclose(C @ C.T, F, rtol=1e-13)
    assert len(kl) == 3 and np.allclose(kl, [C[1, 0], C[2, 0], C[2, 1]])          # row by row (gpf/base.py:92)
    _, dF, _, _ = mo.lml_and_grad(X, Y, ell, F, S)
    gd, gl = mo.cholesky_chain(dF, C, kd)
    for k in range(3):
        up, um = kd.copy(), kd.copy()
        up[k] += 1e-6
        um[k] -= 1e-6
        Cp, Cm = mo.params_to_cholesky(up, kl), mo.params_to_cholesky(um, kl)
        assert gd[k] == pytest.approx((mo.lml(X, Y, ell, Cp @ Cp.T, S) - mo.lml(X, Y, ell, Cm @ Cm.T, S)) / 2e-6, rel=1e-5, abs=1e-6)
        lp, lm = kl.copy(), kl.copy()
        lp[k] += 1e-6
        lm[k] -= 1e-6
        Cp, Cm = mo.params_to_cholesky(kd, lp), mo.params_to_cholesky(kd, lm)
        assert gl[k] == pytest.approx((mo.lml(X, Y, ell, Cp @ Cp.T, S) - mo.lml(X, Y, ell, Cm @ Cm.T, S)) / 2e-6, rel=1e-5, abs=1e-6)
    with pytest.raises(ValueError):
        mo.variance_to_params(np.diag([1.0, 1e-7]))                                # Cholesky diagonal <= 1e-3 (gpf/base.py:87-88)


def test_one_output_limit_is_the_independent_gp():
    X, Y, ell, F, S = _case(L=1)
    v1, g1 = go.lml_and_grad(X, Y[:, 0], ell[0], F[0, 0], S[0, 0])
    v2, dF, dell, dS = mo.lml_and_grad(X, Y, ell, F, S)
    assert v1 == pytest.approx(v2, rel=1e-14)
    assert np.allclose(g1, np.r_[dell[0], dF[0, 0], dS[0, 0]], rtol=1e-11, atol=1e-13)
    Xs = np.random.default_rng(3).standard_normal((6, 3))
    m1, s1 = go.predict(X, Y[:, 0], ell[0], F[0, 0], S[0, 0], Xs)
    m2, s2 = mo.predict(X, Y, ell, F, S, Xs)
    assert np.allclose(m1, m2[:, 0], rtol=1e-13) and np.allclose(s1, s2[:, 0], rtol=1e-13)


def test_block_diagonal_case_is_the_sum_of_independent_gps_and_predict_identity():
    X, Y, ell, F, S = _case()
    Fd, Sd = np.diag(np.diag(F)), np.diag(np.diag(S))
    total = sum(go.lml(X, Y[:, l], ell[l], Fd[l, l], Sd[l, l]) for l in range(3))
    assert mo.lml(X, Y, ell, Fd, Sd) == pytest.approx(total, rel=1e-13)
    Xs = np.random.default_rng(4).standard_normal((7, 3))
    assert np.max(mo.check_k_inv_y(X, Y, ell, F, S, Xs)) < 1e-12
    mean, sd = mo.predict(X, Y, ell, F, S, Xs, y_instead_of_f=False)
    mean_y, sd_y = mo.predict(X, Y, ell, F, S, Xs, y_instead_of_f=True)
    assert np.array_equal(mean, mean_y) and np.allclose(sd_y ** 2 - sd ** 2, np.diag(S)[None, :], rtol=1e-9)
    assert mo.k_inv_y(X, Y, ell, F, S).shape == (3, 1, X.shape[0])


def test_initial_noise_is_always_reduced_to_its_diagonal():
    assert np.array_equal(mo.initial_noise(0.02, 2), 0.02 * np.eye(2))
    assert np.array_equal(mo.initial_noise(np.array([[0.1, 0.2]]), 2), np.diag([0.1, 0.2]))
    assert np.array_equal(mo.initial_noise(np.array([[0.1, 0.05], [0.05, 0.2]]), 2), np.diag([0.1, 0.2]))


def test_sobol_with_a_non_diagonal_F_literal_against_pair_form():
    X, Y, ell, F, S = _case(N=20, M=3, L=2, seed=5)
    KiY = mo.k_inv_y(X, Y, ell, F, S)
    lit = mo.LiteralClosedSobolCovariant(X, KiY, F, ell)
    slices = [(0, 3), (0, 1), (1, 2), (0, 2), (2, 3), (1, 3), (3, 3)]
    V = mo.sobol_V_covariant(X, KiY, F, ell, slices)
    scale = np.abs(lit.V[0]).max()
    for s, sl in enumerate(slices):
        assert np.abs(lit.marginalize(sl)['V'] - V[s]).max() < 1e-10 * scale
    assert np.allclose(lit.V[0], lit.V[0].T) and np.allclose(np.diag(lit.S), 1.0)
    assert np.abs(V[-1]).max() < 1e-12 * scale                                    # empty slice: centred weights sum to zero


def test_fit_trains_the_reference_default_subset():
    X, Y, ell, F, S = _case(N=30, M=2, L=2, seed=6)
    F0, S0 = np.diag([2.0, 2.0]), mo.initial_noise(0.02, 2)
    start = mo.lml(X, Y, ell, F0, S0)
    ell1, F1, S1, value, res, nfev = mo.fit(X, Y, ell, F0, S0)
    assert np.array_equal(ell1, ell)                                               # lengthscales fixed (gpr/kernels.py:57)
    assert F1[0, 1] == 0.0 and not np.allclose(np.diag(F1), 2.0)                   # kernel: Cholesky diagonal only
    assert S1[0, 1] != 0.0                                                          # likelihood: full Cholesky factor
    assert value > start and nfev >= 2
    ell2, F2, S2, value2, _, _ = mo.fit(X, Y, ell, F0, S0, trainable={'kernel_covariance': True, 'lengthscales': True})
    assert value2 >= value - 1e-9 and F2[0, 1] != 0.0 and not np.array_equal(ell2, ell)


def test_predict_gradient_mean_is_the_gradient_of_the_posterior_mean():
    X, Y, ell, F, S = _case(N=25, M=3, L=2, seed=8)
    xs = np.random.default_rng(9).standard_normal((4, 3))
    mean, var = mo.predict_gradient(X, Y, ell, F, S, xs)
    assert mean.shape == (4, 2, 3) and var.shape == (4, 2, 4, 2, 3, 3)
    for m in range(3):
        xp, xm = xs.copy(), xs.copy()
        xp[:, m] += 1e-6
        xm[:, m] -= 1e-6
        fd = (mo.predict(X, Y, ell, F, S, xp, False)[0] - mo.predict(X, Y, ell, F, S, xm, False)[0]) / 2e-6
        assert np.abs(mean[:, :, m] - fd).max() < 1e-6 * max(1.0, np.abs(fd).max())
    # one output: the independent gradient GP (var[O, 0, o, 0] against the 'OoLMm' layout)
    X, Y, ell, F, S = _case(N=25, M=3, L=1, seed=8)
    m1, v1 = mo.predict_gradient(X, Y, ell, F, S, xs)
    m0, v0 = go.predict_gradient(X, Y[:, 0], ell[0], F[0, 0], S[0, 0], xs)
    assert np.allclose(m1[:, 0, :], m0, rtol=1e-11, atol=1e-13) and np.allclose(v1[:, 0, :, 0], v0, rtol=1e-9, atol=1e-12)


def test_sobol_errors_rank2_cholesky_branch_reduces_to_the_independent_case():
    """psi_factor with the (LN, LN) factor of a covariant GP (gsa/calibrators.py:304-308): for a block-diagonal system it must give
    what the (L, N, N) branch gives."""
    from oracle.sobol_error_oracle import LiteralClosedSobolWithError
    rng = np.random.default_rng(0)
    N, M, L = 14, 3, 2
    X, Y, ell = rng.standard_normal((N, M)), rng.standard_normal((N, L)), 0.8 + rng.random((L, M))
    F, S = np.diag([1.3, 0.7]), np.diag([0.05, 0.08])
    KiY, Kc = mo.k_inv_y(X, Y, ell, F, S), mo.k_cho(X, ell, F, S)
    Kc3 = np.stack([go.k_cho(X, ell[l], F[l, l], S[l, l]) for l in range(L)])
    a = LiteralClosedSobolWithError(X, KiY, np.diag(F)[None, :], ell, Kc, is_T_partial=False)
    b = LiteralClosedSobolWithError(X, KiY, np.diag(F)[None, :], ell, Kc3, is_T_partial=False)
    for sl in ((0, 2), (1, 2), (2, 3)):
        ra, rb = a.marginalize(sl), b.marginalize(sl)
        assert np.allclose(ra['W'], rb['W'], rtol=1e-9, atol=1e-14) and np.allclose(ra['T'], rb['T'], rtol=1e-6, atol=1e-12)


def test_oracle_reproduces_the_covariant_golden_fixture():
    """tests/golden/mogp_*.npz (made by tests/golden/make_golden.py from this oracle) pins the oracle against regressions."""
    from pathlib import Path
    g = np.load(Path(__file__).resolve().parent / 'golden' / 'mogp_N90_M3_L2.npz')
    X, Y, ell, F, S = g['X'], g['Y'], g['ell'], g['F'], g['Sigma']
    v, dF, dell, dS = mo.lml_and_grad(X, Y, ell, F, S)
    assert v == pytest.approx(float(g['lml']), rel=1e-12)
    assert np.allclose(dF, g['dF'], rtol=1e-9) and np.allclose(dell, g['dell'], rtol=1e-9) and np.allclose(dS, g['dSigma'], rtol=1e-9)
    assert np.allclose(mo.k_inv_y(X, Y, ell, F, S), g['K_inv_Y'], rtol=1e-9, atol=1e-12)
    mean, sd = mo.predict(X, Y, ell, F, S, g['Xs'])
    assert np.allclose(mean, g['mean_y'], rtol=1e-10, atol=1e-12) and np.allclose(sd, g['sd_y'], rtol=1e-10)
    assert np.allclose(mo.sobol_V_covariant(X, g['K_inv_Y'], F, ell, g['slices']), g['V_full'], rtol=1e-10, atol=1e-15)
