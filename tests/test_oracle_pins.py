"""Pins of the CPU oracle against independent implementations available in the build image (CPU only, no GPU).

PARITY REMAINS UNPINNED BY THE REFERENCE ITSELF: TensorFlow / GPflow cannot be imported here and the reference holds no fixtures for this
path. What these tests add is agreement with (a) scikit-learn's GaussianProcessRegressor -- an implementation nobody
in this project wrote -- for LML, its gradient, K^-1 y and the predictive mean / sd; (b) reverse-mode autodiff through
``torch.linalg.cholesky`` for the LML gradient, the reference's own mechanism (gpr/models.py:359-361); (c) the reference's Sobol code
re-typed with torch ops (gsa/calibrators.py:60-97, gsa/base.py:92-126) for the conditional variances, cross-output entries included.
"""
import numpy as np
import pytest

from oracle import gp_oracle as o


def _case(N, M, seed=0):
    X, y = o.synthetic_fold(N, M, k=seed)
    rng = np.random.default_rng(seed)
    return X, y, rng.uniform(0.6, 3.0, M), 1.3, 0.02


@pytest.mark.parametrize('N,M', [(200, 4), (333, 7)])
def test_oracle_agrees_with_scikit_learn(N, M):
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, WhiteKernel
    X, y, ell, var, noise = _case(N, M)
    kernel = ConstantKernel(var, 'fixed') * RBF(ell, 'fixed') + WhiteKernel(noise, 'fixed')
    sk = GaussianProcessRegressor(kernel=kernel, optimizer=None, alpha=0.0, normalize_y=False).fit(X, y)
    lml, grad = o.lml_and_grad(X, y, ell, var, noise)
    assert sk.log_marginal_likelihood_value_ == pytest.approx(lml, rel=1e-12)
    np.testing.assert_allclose(sk.alpha_.ravel(), o.k_inv_y(X, y, ell, var, noise), rtol=1e-9, atol=1e-11)
    # gradient: sklearn differentiates with respect to the LOG of (variance, lengthscales..., noise): d/d log(theta) = theta d/d theta
    free = ConstantKernel(var) * RBF(ell) + WhiteKernel(noise)
    sk_free = GaussianProcessRegressor(kernel=free, optimizer=None, alpha=0.0).fit(X, y)
    value, log_grad = sk_free.log_marginal_likelihood(sk_free.kernel_.theta, eval_gradient=True)
    assert value == pytest.approx(lml, rel=1e-12)
    ours = np.concatenate([[var * grad[M]], ell * grad[:M], [noise * grad[M + 1]]])          # oracle order: ell (M), variance, noise
    np.testing.assert_allclose(log_grad, ours, rtol=1e-8, atol=1e-9)
    Xs, _ = o.synthetic_fold(50, M, k=9)
    mean, sd = o.predict(X, y, ell, var, noise, Xs)                                          # predict_y: noise included, as WhiteKernel's diag
    sk_mean, sk_sd = sk.predict(Xs, return_std=True)
    np.testing.assert_allclose(sk_mean, mean, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(sk_sd, sd, rtol=1e-7)
    # the BLAS-3 arrangement that bench.py times is the same function
    lml_b, grad_b = o.lml_and_grad_blas(X, y, ell, var, noise)
    assert lml_b == pytest.approx(lml, rel=1e-12)
    np.testing.assert_allclose(grad_b, grad, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize('N,M', [(150, 3), (257, 6)])
def test_analytic_gradient_agrees_with_autodiff_through_cholesky(N, M):
    from oracle.torch_checks import lml_autograd
    X, y, ell, var, noise = _case(N, M, seed=1)
    lml, grad = o.lml_and_grad(X, y, ell, var, noise)
    t_lml, g_ell, g_var, g_noise = lml_autograd(X, y, ell, var, noise)
    assert t_lml == pytest.approx(lml, rel=1e-12)
    np.testing.assert_allclose(np.concatenate([g_ell, [g_var, g_noise]]), grad, rtol=1e-8, atol=1e-9)
    # and in the optimiser's space: softplus-unconstrained variables, noise with GPflow's 1e-6 lower bound
    u = o.pack_unconstrained(ell, var, noise)
    value, du = o.neg_lml_unconstrained(u, X, y)
    chain = -np.concatenate([g_ell, [g_var, g_noise]]) * o.sigmoid(u)
    assert value == pytest.approx(-lml, rel=1e-12)
    np.testing.assert_allclose(du, chain, rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize('L', [1, 3])
def test_sobol_oracle_agrees_with_torch_retyping_of_the_reference(L):
    from oracle.torch_checks import TorchClosedSobol
    N, M = 36, 4
    rng = np.random.default_rng(5)
    X, _ = o.synthetic_fold(N, M)
    K_inv_Y = rng.normal(size=(L, 1, N))
    F = rng.uniform(0.5, 2.0, L)
    ell = rng.uniform(0.5, 3.0, (L, M))
    torch_form = TorchClosedSobol(X, K_inv_Y, F, ell)
    numpy_form = o.LiteralClosedSobol(X, K_inv_Y, F, ell)
    pair_form = o.ClosedSobolOracle(X, K_inv_Y, F, ell)
    np.testing.assert_allclose(torch_form.V[0].numpy(), numpy_form.V[0], rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(np.diagonal(torch_form.S.numpy()), 1.0, rtol=1e-12)
    for pair in [(0, M), (0, 1), (1, 2), (0, 3), (2, M), (M, M)]:
        t, n, p = torch_form.marginalize(pair), numpy_form.marginalize(pair), pair_form.marginalize(pair)
        scale = np.abs(torch_form.V[0].numpy()).max()
        np.testing.assert_allclose(t['V'], n['V'], rtol=1e-10, atol=1e-13 * scale)
        np.testing.assert_allclose(p['V'], t['V'], rtol=1e-8, atol=1e-11 * scale)                # the O(NM)-memory form the GPU tests use
        assert np.allclose(t['V'], t['V'].T, rtol=1e-10, atol=1e-13 * scale)
