"""GPU parity of the covariant (dependent multi-output) GP path against oracle/mogp_oracle.py, through the C ABI
(rcgp_create_mo, rcgp_set_hyper_mo, rcgp_lml_grad_mo, rcgp_predict_mo, rcgp_sobol_pair). Parity unpinned (see the oracle)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(N, M, L, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, M))
    W = rng.standard_normal((M, L))
    Y = np.sin(X @ W) + 0.1 * rng.standard_normal((N, L))
    if N > 1:
        Y = (Y - Y.mean(0)) / Y.std(0)
    ell = 0.7 + 1.5 * rng.random((L, M))
    C = np.tril(0.4 * rng.standard_normal((L, L)), -1) + np.diag(0.8 + rng.random(L))
    Cn = np.tril(0.03 * rng.standard_normal((L, L)), -1) + np.diag(0.1 + 0.1 * rng.random(L))
    F, S = C @ C.T, Cn @ Cn.T
    return X, Y, ell, (F + F.T) / 2, (S + S.T) / 2


@pytest.mark.parametrize('N,M,L', [(200, 3, 2), (130, 4, 3), (128, 2, 2), (1, 2, 2)])
def test_gram_factor_and_alpha(N, M, L):
    from oracle import mogp_oracle as mo
    from romcomma_amd import _lib
    X, Y, ell, F, S = _case(N, M, L)
    with _lib.RcMOGP(X, Y) as gp:
        gp.set_hyper(ell, F, S)
        K = gp.gram()
        Kref = mo.noisy_gram(X, ell, F, S)
        assert np.abs(K - Kref).max() <= 1e-13 * np.abs(Kref).max()
        Lc = gp.k_cho()
        assert np.abs(Lc - mo.k_cho(X, ell, F, S)).max() <= 1e-10
        assert np.allclose(gp.k_inv_y(), mo.k_inv_y(X, Y, ell, F, S), rtol=1e-8, atol=1e-10)
        assert gp.lml() == pytest.approx(mo.lml(X, Y, ell, F, S), rel=1e-11)


@pytest.mark.parametrize('N,M,L', [(200, 3, 2), (130, 4, 3), (700, 5, 2), (60, 3, 20), (130, 2, 9)])      # (L = 9: the reference's sweep; 20 > round 3's 16)
def test_lml_gradient(N, M, L):
    from oracle import mogp_oracle as mo
    from romcomma_amd import _lib
    X, Y, ell, F, S = _case(N, M, L, seed=3)
    v, dF, dell, dS = mo.lml_and_grad(X, Y, ell, F, S)
    with _lib.RcMOGP(X, Y) as gp:
        gp.set_hyper(ell, F, S)
        lml, gF, gell, gS = gp.lml_grad()
    assert lml == pytest.approx(v, rel=1e-11)
    for got, ref in ((gF, dF), (gell, dell), (gS, dS)):
        assert np.abs(got - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max())


def test_one_output_limit_is_the_independent_gp():
    from romcomma_amd import _lib
    X, Y, ell, F, S = _case(300, 4, 1, seed=5)
    Xs = np.random.default_rng(9).standard_normal((40, 4))
    with _lib.RcGP(X, Y[:, 0]) as a, _lib.RcMOGP(X, Y) as b:
        a.set_hyper(ell[0], F[0, 0], S[0, 0])
        b.set_hyper(ell, F, S)
        la, ga = a.lml_grad()
        lb, gF, gell, gS = b.lml_grad()
        assert la == lb                                                  # same kernels, same order
        assert np.allclose(ga, np.r_[gell[0], gF[0, 0], gS[0, 0]], rtol=1e-11, atol=1e-12)
        ma, sa = a.predict(Xs)
        mb, sb = b.predict(Xs)
        assert np.array_equal(ma, mb[:, 0]) and np.array_equal(sa, sb[:, 0])


@pytest.mark.parametrize('y_instead_of_f', [True, False])
def test_predict(y_instead_of_f):
    from oracle import mogp_oracle as mo
    from romcomma_amd import _lib
    X, Y, ell, F, S = _case(260, 3, 3, seed=7)
    Xs = np.random.default_rng(11).standard_normal((150, 3))
    mean, sd = mo.predict(X, Y, ell, F, S, Xs, y_instead_of_f)
    with _lib.RcMOGP(X, Y) as gp:
        gp.set_hyper(ell, F, S)
        m, s = gp.predict(Xs, y_instead_of_f)
    assert np.allclose(m, mean, rtol=1e-9, atol=1e-10)
    assert np.allclose(s, sd, rtol=1e-8, atol=1e-10)


def test_many_panels_use_the_look_ahead_cholesky():
    """L N large enough for the fine-grained multi-stream factorisation (Np >= 512) with ragged output blocks."""
    from oracle import mogp_oracle as mo
    from romcomma_amd import _lib
    X, Y, ell, F, S = _case(1100, 6, 3, seed=13)
    v, dF, dell, dS = mo.lml_and_grad(X, Y, ell, F, S)
    with _lib.RcMOGP(X, Y) as gp:
        gp.set_hyper(ell, F, S)
        lml, gF, gell, gS = gp.lml_grad()
        alpha = gp.k_inv_y()
    assert lml == pytest.approx(v, rel=1e-10)
    assert np.abs(gell - dell).max() <= 1e-7 * max(1.0, np.abs(dell).max())
    assert np.abs(gF - dF).max() <= 1e-7 * max(1.0, np.abs(dF).max())
    assert np.abs(gS - dS).max() <= 1e-7 * max(1.0, np.abs(dS).max())
    assert np.allclose(alpha, mo.k_inv_y(X, Y, ell, F, S), rtol=1e-6, atol=1e-8)


def test_single_output_entries_refuse_a_covariant_handle():
    from romcomma_amd import _lib
    X, Y, ell, F, S = _case(64, 2, 2)
    with _lib.RcMOGP(X, Y) as gp:
        gp.set_hyper(ell, F, S)
        with pytest.raises(_lib.RcgpError):
            _lib.RcGP.lml_grad(gp)
        with pytest.raises(_lib.RcgpError):
            _lib.RcGP.set_hyper(gp, ell[0], 1.0, 0.1)
        with pytest.raises(_lib.RcgpError):
            gp.set_hyper(ell, np.array([[1.0, 0.2], [0.3, 1.0]]), S)     # not symmetric


def test_not_positive_definite_is_reported():
    from romcomma_amd import _lib
    X, Y, ell, F, S = _case(100, 2, 2)
    F = np.array([[1.0, 3.0], [3.0, 1.0]])                                 # indefinite output covariance
    with _lib.RcMOGP(X, Y) as gp:
        gp.set_hyper(ell, F, 1e-6 * np.eye(2))
        with pytest.raises(_lib.NotPositiveDefiniteError):
            gp.lml()


def test_sobol_pair_and_covariant_sum():
    from oracle import gp_oracle as go, mogp_oracle as mo
    from romcomma_amd import _lib
    from romcomma_amd.gsa.calibrators import covariant_V
    X, Y, ell, F, S = _case(300, 4, 2, seed=17)
    M = 4
    slices = go.all_slices(M) + [(1, 3), (M, M)]
    with _lib.RcMOGP(X, Y) as gp:
        gp.set_hyper(ell, F, S)
        KiY = gp.k_inv_y()
        g, phi = mo.sobol_prepare_covariant(X, KiY, F, ell)
        # one raw pair against the oracle's pair form
        lam = ell[0] * ell[1]
        pre = F[0, 1] * np.sqrt(np.prod(lam * phi[0, 1]))
        total = _lib.sobol_weight_sum(gp, phi[0, 1], pre, KiY[1, 0])
        g01 = pre * np.exp(-0.5 * (X * X) @ phi[0, 1]) * KiY[1, 0]
        assert total == pytest.approx(g01.sum(), rel=1e-11, abs=1e-13)
        shift = 0.123
        V = _lib.sobol_pair(gp, phi[0, 1], pre, KiY[1, 0], shift, phi[0, 1], pre, KiY[1, 0], shift, slices)
        ref = go.sobol_V_pair(X, g01 - shift, g01 - shift, phi[0, 1], phi[0, 1], slices)
        assert np.allclose(V, ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())
        # the full covariant V against the oracle
        Vc = covariant_V(gp, KiY, F, ell, slices)
    ref = mo.sobol_V_covariant(X, KiY, F, ell, slices)
    assert np.allclose(Vc, ref, rtol=1e-8, atol=1e-11 * np.abs(ref).max())


def test_predict_gradient():
    """rcgp_predict_gradient_mo through HipGP's covariant branch (gpr/models.py:392-406) against the oracle, ragged block sizes."""
    from oracle import mogp_oracle as mo
    from romcomma_amd import _lib
    X, Y, ell, F, S = _case(210, 3, 2, seed=19)
    xs = np.random.default_rng(21).standard_normal((5, 3))
    mean, var = mo.predict_gradient(X, Y, ell, F, S, xs)
    with _lib.RcMOGP(X, Y) as gp:
        gp.set_hyper(ell, F, S)
        m_lom, cov = gp.predict_gradient(xs)
    assert np.allclose(np.transpose(m_lom, (1, 0, 2)), mean, rtol=1e-8, atol=1e-10)
    got = -np.einsum('BlOMlom->OBolMm', cov)
    idx = np.arange(3)
    ref = var.copy()
    u = xs[None, :, :] / ell[:, None, :]
    d = u[:, :, None, None, :] - u[None, None, :, :, :]
    kxx = F[:, None, :, None] * np.exp(-0.5 * np.einsum('...M,...M->...', d, d))
    ref[..., idx, idx] -= np.einsum('LM,lM,LOlo->OLolM', 1.0 / ell, 1.0 / ell, kxx)       # leave -V^T V
    assert np.allclose(got, ref, rtol=1e-7, atol=1e-9 * np.abs(ref).max())
    # L n M > 4096 (one pass of derivative rows): 4200 rows go through in two chunks; points straddling the chunk boundary come out
    # exactly as when they are asked for on their own (row r = (l n + o) M + m: a sub-set of points keeps its own (l, o, m) order)
    big = np.random.default_rng(22).standard_normal((700, 3))
    with _lib.RcMOGP(X, Y) as gp:
        gp.set_hyper(ell, F, S)
        m_big, c_big = gp.predict_gradient(big)                           # (L, n, M), (Lb, L, n, M, L, n, M)
        pts = np.arange(660, 673)                                         # output 1's rows 4080..4118: across the chunk boundary at row 4096
        m_sub, c_sub = gp.predict_gradient(big[pts])
    assert m_big.shape == (2, 700, 3) and c_big.shape == (2, 2, 700, 3, 2, 700, 3)
    assert np.allclose(m_big[:, pts, :], m_sub, rtol=1e-13, atol=1e-15)
    sub = c_big[:, :, pts][:, :, :, :, :, pts]
    assert sub.shape == c_sub.shape and np.allclose(sub, c_sub, rtol=1e-12, atol=1e-14 * np.abs(c_sub).max())


def test_covariant_golden_fixture():
    """Every quantity of tests/golden/mogp_N90_M3_L2.npz through the C ABI (the oracle is not imported here)."""
    from pathlib import Path
    from romcomma_amd import _lib
    from romcomma_amd.gsa.calibrators import covariant_V
    g = np.load(Path(__file__).resolve().parent / 'golden' / 'mogp_N90_M3_L2.npz')
    X, Y, ell, F, S = g['X'], g['Y'], g['ell'], g['F'], g['Sigma']
    with _lib.RcMOGP(X, Y) as gp:
        gp.set_hyper(ell, F, S)
        lml, gF, gell, gS = gp.lml_grad()
        assert lml == pytest.approx(float(g['lml']), rel=1e-11)
        for got, ref in ((gF, g['dF']), (gell, g['dell']), (gS, g['dSigma'])):
            assert np.abs(got - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max())
        KiY = gp.k_inv_y()
        assert np.allclose(KiY, g['K_inv_Y'], rtol=1e-8, atol=1e-10)
        Lc = gp.k_cho()
        assert np.allclose(np.diag(Lc), g['K_cho_diag'], rtol=1e-10) and np.sum(Lc) == pytest.approx(float(g['K_cho_checksum']), rel=1e-10)
        for flag, mk, sk in ((True, 'mean_y', 'sd_y'), (False, 'mean_f', 'sd_f')):
            mean, sd = gp.predict(g['Xs'], flag)
            assert np.allclose(mean, g[mk], rtol=1e-9, atol=1e-10) and np.allclose(sd, g[sk], rtol=1e-8)
        m_lom, cov = gp.predict_gradient(g['Xs'][:3])
        assert np.allclose(np.transpose(m_lom, (1, 0, 2)), g['gmean'], rtol=1e-8, atol=1e-10)
        Vf = covariant_V(gp, g['K_inv_Y'], F, ell, g['slices'])
        Vd = covariant_V(gp, g['K_inv_Y'], np.diag(F), ell, g['slices'], is_F_diagonal=True)
    assert np.allclose(Vf, g['V_full'], rtol=1e-8, atol=1e-11 * np.abs(g['V_full']).max())
    assert np.allclose(Vd, g['V_diag'], rtol=1e-8, atol=1e-11 * np.abs(g['V_diag']).max())


def test_many_input_dimensions():
    """M > 32 takes the wide-LDS instantiation of the gradient kernel (k_grad_mo<65>)."""
    from oracle import mogp_oracle as mo
    from romcomma_amd import _lib
    X, Y, ell, F, S = _case(150, 40, 2, seed=23)
    ell = ell * 6.0
    v, dF, dell, dS = mo.lml_and_grad(X, Y, ell, F, S)
    with _lib.RcMOGP(X, Y) as gp:
        gp.set_hyper(ell, F, S)
        lml, gF, gell, gS = gp.lml_grad()
    assert lml == pytest.approx(v, rel=1e-10)
    for got, ref in ((gF, dF), (gell, dell), (gS, dS)):
        assert np.abs(got - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize('N,M,L', [(150, 70, 2), (131, 130, 3), (260, 65, 2)])
def test_wide_designs(N, M, L):
    """M > 64: k_gram<., WIDE> under the block structure and k_grad_mo<33, ., WIDE> (Z chunks of 32 dimensions per 16-row group, 2 M + 2
    sums per tile written in more than one pass of the workgroup at M = 130 with 512 threads ... 262 > 256), the joint prediction."""
    from oracle import mogp_oracle as mo
    from romcomma_amd import _lib
    X, Y, ell, F, S = _case(N, M, L, seed=29)
    ell = ell * np.sqrt(M)
    v, dF, dell, dS = mo.lml_and_grad(X, Y, ell, F, S)
    Xs = np.random.default_rng(2).standard_normal((9, M))
    with _lib.RcMOGP(X, Y) as gp:
        gp.set_hyper(ell, F, S)
        K = gp.gram()
        Kref = mo.noisy_gram(X, ell, F, S)
        assert np.abs(K - Kref).max() <= 1e-13 * np.abs(Kref).max()
        lml, gF, gell, gS = gp.lml_grad()
        mean, sd = gp.predict(Xs, True)
    assert lml == pytest.approx(v, rel=1e-10)
    for got, ref in ((gF, dF), (gell, dell), (gS, dS)):
        assert np.abs(got - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max())
    m_ref, s_ref = mo.predict(X, Y, ell, F, S, Xs, True)
    assert np.allclose(mean, m_ref, rtol=1e-8, atol=1e-10) and np.allclose(sd, s_ref, rtol=1e-7, atol=1e-10)


def test_wide_design_standard_error_terms():
    """rcgp_sobol_error_terms_mo at M = 70 (diagonal F): the pair forms and matvecs over chunked X panels, the psi vectors embedded in
    their output block of the (L N) system -- against the reduced-form oracle's pieces with the joint Cholesky factor."""
    import scipy.linalg
    from oracle import mogp_oracle as mo
    from oracle import sobol_error_oracle as e
    from romcomma_amd import _lib
    N, M, L = 150, 70, 2
    X, Y, ell, F, S = _case(N, M, L, seed=31)
    X = X / 3.0
    ell = ell * 4.0
    F = np.diag(np.diag(F))
    KiY = mo.k_inv_y(X, Y, ell, F, S)
    Kc = mo.k_cho(X, ell, F, S)
    ref = e.ClosedSobolWithErrorOracle(X, KiY.reshape(L, 1, N), np.diag(F)[None, :], ell, np.stack([np.eye(N)] * L), is_T_partial=False)     # (its own W is not used)
    slices = [(0, 1), (63, 64), (64, 65), (0, 65), (0, M), (64, M), (1, M), (30, 69), (5, 5)]

    def psi(a, b, sl):
        H = e._pair_matrix(X, e.error_coefficients(ref.phi[a], ref.phi[b], ref.ups[b], 'H'), sl)
        f = np.zeros(L * N)
        f[b * N:(b + 1) * N] = ref.g0[b] * (H.T @ ref.g[a])
        return scipy.linalg.solve_triangular(Kc, f, lower=True, check_finite=False)

    with _lib.RcMOGP(X, Y) as gp:
        gp.set_hyper(ell, F, S)
        for a in range(L):
            for b in range(L):
                got = gp.sobol_error_terms(slices, a, b)
                psi_full = psi(b, b, (0, M))
                for s, sl in enumerate(slices):
                    if sl[0] == sl[1]:
                        want = (0.0, 0.0, 0.0, 0.0)
                    else:
                        phi_d, _, phi_m, _ = e.error_terms_pair(X, a, b, ref.g0, ref.g, ref.phi, ref.ups, ref.pre,
                                                                [np.eye(N)] * L, sl)
                        p = psi(a, b, sl)
                        want = (phi_d, p @ p, phi_m, psi_full @ p)
                    for k in range(4):
                        assert got[k][s] == pytest.approx(want[k], rel=1e-6, abs=1e-12 * abs(want[0]) + 1e-15), (a, b, sl, k)


def test_full_size_properties():
    """L N = 16384 (two outputs of N = 8192, M = 10), where the oracle is out of reach: size-independent properties.
    (1) With diagonal F and Sigma the joint system is block diagonal: LML, gradients and K_inv_Y must be those of the two independent
    GPs (computed through the single-output entry points). (2) Swapping the outputs (with their lengthscales, and F, Sigma permuted)
    leaves the LML unchanged and permutes the gradients. (3) check_K_inv_Y: k(x, X) . K_inv_Y reproduces predict_f."""
    from romcomma_amd import _lib
    from romcomma_amd.user.sample import bench_hyper, synthetic_fold
    N, M, L = 8192, 10, 2
    X, y0 = synthetic_fold(N, M)
    Y = np.stack([y0, synthetic_fold(N, M, l=1)[1]], axis=1)
    ell0, var, noise = bench_hyper(M)
    ell = np.stack([ell0, 1.15 * ell0])
    Fd, Sd = np.diag([var, 0.8 * var]), np.diag([noise, 1.7 * noise])
    with _lib.RcMOGP(X, Y) as gp:
        gp.set_hyper(ell, Fd, Sd)
        lml, gF, gell, gS = gp.lml_grad()
        alpha = gp.k_inv_y()
        parts = []
        for l in range(L):
            with _lib.RcGP(X, Y[:, l]) as one:
                one.set_hyper(ell[l], Fd[l, l], Sd[l, l])
                v, g = one.lml_grad()
                parts.append((v, g, one.k_inv_y()))
        assert lml == pytest.approx(parts[0][0] + parts[1][0], rel=1e-11)
        for l in range(L):
            assert np.allclose(gell[l], parts[l][1][:M], rtol=1e-8, atol=1e-8 * np.abs(parts[l][1][:M]).max())
            assert gF[l, l] == pytest.approx(parts[l][1][M], rel=1e-8) and gS[l, l] == pytest.approx(parts[l][1][M + 1], rel=1e-8)
            assert np.allclose(alpha[l, 0], parts[l][2], rtol=1e-7, atol=1e-9 * np.abs(parts[l][2]).max())
        # (2) a genuinely covariant setting and its output-swapped twin
        C = np.array([[1.0, 0.0], [0.45, 0.8]])
        Cn = np.array([[0.04, 0.0], [0.01, 0.05]])
        F, S = var * C @ C.T, Cn @ Cn.T
        F, S = (F + F.T) / 2, (S + S.T) / 2
        gp.set_hyper(ell, F, S)
        lml_a, gF_a, gell_a, gS_a = gp.lml_grad()
        Xs = synthetic_fold(64, M, k=9)[0]
        mean_f, _ = gp.predict(Xs, False)
        KiY = gp.k_inv_y()[:, 0, :]
        u = Xs[None, :, :] / ell[:, None, :]
        U = X[None, :, :] / ell[:, None, :]
        recon = np.zeros((64, L))
        for l in range(L):
            for j in range(L):
                r2 = np.sum(u[l] ** 2, 1)[:, None] + np.sum(U[j] ** 2, 1)[None, :] - 2.0 * u[l] @ U[j].T
                recon[:, l] += F[l, j] * np.exp(-0.5 * r2) @ KiY[j]
        assert np.abs(recon - mean_f).max() <= 1e-7 * max(1.0, np.abs(mean_f).max())
    P = [1, 0]
    with _lib.RcMOGP(X, Y[:, P]) as gp:
        gp.set_hyper(ell[P], F[np.ix_(P, P)], S[np.ix_(P, P)])
        lml_b, gF_b, gell_b, gS_b = gp.lml_grad()
    assert lml_b == pytest.approx(lml_a, rel=1e-10)
    assert np.allclose(gell_b, gell_a[P], rtol=1e-6, atol=1e-7 * np.abs(gell_a).max())
    assert np.allclose(gF_b, gF_a[np.ix_(P, P)], rtol=1e-6, atol=1e-7 * np.abs(gF_a).max())
    assert np.allclose(gS_b, gS_a[np.ix_(P, P)], rtol=1e-6, atol=1e-7 * np.abs(gS_a).max())


def test_random_shapes_against_oracle():
    """A spread of sizes around the tile and panel boundaries (N just below / at / above multiples of 128, L N crossing 512 where
    the multi-stream factorisation starts, M from 1 to 12), each checked against the oracle: LML, all three gradients, predict."""
    from oracle import mogp_oracle as mo
    from romcomma_amd import _lib
    rng = np.random.default_rng(2024)
    cases = [(127, 1, 2), (128, 2, 2), (129, 3, 2), (255, 5, 2), (256, 2, 3), (257, 7, 2), (171, 12, 3), (383, 4, 4), (512, 3, 2),
             (640, 6, 2), (90, 2, 5), (333, 1, 3)]
    for i, (N, M, L) in enumerate(cases):
        X, Y, ell, F, S = _case(N, M, L, seed=100 + i)
        ell = ell * (1.0 + 0.5 * rng.random())
        Xs = rng.standard_normal((37, M))
        v, dF, dell, dS = mo.lml_and_grad(X, Y, ell, F, S)
        mean, sd = mo.predict(X, Y, ell, F, S, Xs)
        with _lib.RcMOGP(X, Y) as gp:
            gp.set_hyper(ell, F, S)
            lml, gF, gell, gS = gp.lml_grad()
            m, s = gp.predict(Xs)
        assert lml == pytest.approx(v, rel=1e-10), (N, M, L)
        for got, ref in ((gF, dF), (gell, dell), (gS, dS)):
            assert np.abs(got - ref).max() <= 1e-7 * max(1.0, np.abs(ref).max()), (N, M, L)
        assert np.allclose(m, mean, rtol=1e-8, atol=1e-9) and np.allclose(s, sd, rtol=1e-7, atol=1e-9), (N, M, L)
