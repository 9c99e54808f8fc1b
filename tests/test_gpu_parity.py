"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against the committed golden fixtures, against the
oracle on seeded inputs at sizes the oracle finishes in seconds, and -- at the BASELINE.json sizes -- through
size-independent properties. Tolerance: north_star asks 1e-5 relative in fp64; these tests hold the kernels to 1e-8 or
tighter so regressions show long before the contract is at risk.
"""
from pathlib import Path

import numpy as np
import pytest

from oracle import gp_oracle as o          # the checker, never the thing measured

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / 'golden'


def relmax(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


# --------------------------------------------------------------------------------------------------------------------
# golden fixtures (tests/golden/make_golden.py)
# --------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize('name', ['gp_N16_M1', 'gp_N64_M3', 'gp_N256_M10', 'gp_N300_M7'])
def test_golden_fixture(gpu, name):
    z = np.load(GOLDEN / f'{name}.npz')
    gp = gpu.RcGP(z['X'], z['y'])
    gp.set_hyper(z['ell'], float(z['var']), float(z['noise']))
    assert gp.lml() == pytest.approx(float(z['lml']), rel=1e-10)
    lml, grad = gp.lml_grad()
    assert lml == pytest.approx(float(z['lml']), rel=1e-10)
    np.testing.assert_allclose(grad, z['grad'], rtol=1e-8, atol=1e-9 * np.max(np.abs(z['grad'])))
    assert relmax(gp.k_inv_y(), z['alpha']) < 1e-9
    Lc = gp.k_cho()
    np.testing.assert_allclose(np.diag(Lc), z['K_cho_diag'], rtol=1e-10)
    np.testing.assert_allclose(Lc[:8, :8], z['K_cho_corner'], rtol=1e-10, atol=1e-14)
    assert np.sum(Lc) == pytest.approx(float(z['K_cho_checksum']), rel=1e-10)
    assert np.all(np.triu(Lc, 1) == 0.0)
    for flag, km, ks in ((True, 'mean_y', 'sd_y'), (False, 'mean_f', 'sd_f')):
        m, s = gp.predict(z['Xs'], flag)
        assert relmax(m, z[km]) < 1e-9
        np.testing.assert_allclose(s, z[ks], rtol=1e-8)
    V = gp.sobol_closed(z['slices'])
    full = abs(z['V'][-2])
    np.testing.assert_allclose(V, z['V'], rtol=1e-8, atol=1e-9 * full)    # empty slices: V = (sum g)^2 = 0 by centring
    assert abs(V[-1]) < 1e-9 * full
    gp.close()


def test_literal_multi_output_fixture(gpu):
    """(L, L) conditional-variance matrix incl. the cross-output entries of the reference's 'lLN,lLNjJn,jJn->lj' einsum
    (gsa/calibrators.py:79), expected values from the literal transliteration of the TF broadcasting."""
    z = np.load(GOLDEN / 'sobol_literal_N40_M4_L2.npz')
    X, Y, ell, F, noise, alpha = z['X'], z['Y'], z['ell'], z['F'], z['noise'], z['alpha']
    L = Y.shape[1]
    scale = np.max(np.abs(z['V0']))
    for l in range(L):
        gp = gpu.RcGP(X, Y[:, l])
        gp.set_hyper(ell[l], F[l], noise[l])
        assert relmax(gp.k_inv_y(), alpha[l]) < 1e-9
        for j in range(L):
            V = gp.sobol_closed(z['slices']) if j == l else gp.sobol_cross(ell[j], F[j], alpha[j], z['slices'])
            np.testing.assert_allclose(V, z['V'][l, j], rtol=1e-7, atol=1e-9 * scale)
        gp.close()


# --------------------------------------------------------------------------------------------------------------------
# oracle on seeded inputs, incl. ragged sizes around the 128 tile edge and the extremes of M
# --------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize('N,M', [(1, 1), (2, 3), (127, 2), (128, 5), (129, 4), (383, 6), (640, 3), (1500, 10), (1600, 4), (200, 64), (513, 33),
                                 (200, 96), (300, 65), (150, 200), (700, 130)])   # M > 64: the Gram / gradient / Sobol kernels stage their panels in chunks
def test_against_oracle(gpu, N, M):
    X, y = o.synthetic_fold(N, M, k=N % 7)
    rng = np.random.default_rng(N + M)
    if N < 8:                                             # the z-scoring in synthetic_fold degenerates for a handful of rows
        y = rng.standard_normal(N)
    ell = rng.uniform(0.6, 3.0, M) * (1.0 if M < 20 else np.sqrt(M / 4))       # keep K well away from the identity at large M
    var, noise = 1.3, 0.015
    gp = gpu.RcGP(X, y)
    gp.set_hyper(ell, var, noise)
    assert relmax(gp.gram(), o.noisy_gram(X, ell, var, noise)) < 1e-13
    assert relmax(gp.k_cho(), o.k_cho(X, ell, var, noise)) < 1e-11
    lml_ref, grad_ref = o.lml_and_grad(X, y, ell, var, noise)
    assert gp.lml() == pytest.approx(lml_ref, rel=1e-10, abs=1e-10)
    lml, grad = gp.lml_grad()
    assert lml == pytest.approx(lml_ref, rel=1e-10, abs=1e-10)
    np.testing.assert_allclose(grad, grad_ref, rtol=1e-7, atol=1e-9 * np.max(np.abs(grad_ref)))
    alpha_ref = o.k_inv_y(X, y, ell, var, noise)
    assert relmax(gp.k_inv_y(), alpha_ref) < 1e-9
    Xs, _ = o.synthetic_fold(37, M, k=90)
    m, s = gp.predict(Xs, True)
    mr, sr = o.predict(X, y, ell, var, noise, Xs, True)
    assert relmax(m, mr) < 1e-9
    np.testing.assert_allclose(s, sr, rtol=1e-8)
    slices = o.all_slices(M) + ([(1, M - 1)] if M > 3 else [])
    if M <= 10:
        V = gp.sobol_closed(slices)
        g, phi = o.sobol_prepare(X, alpha_ref[None, :], np.array([var]), ell[None, :])
        Vr = o.sobol_V_pair(X, g[0], g[0], phi[0], phi[0], slices)
        np.testing.assert_allclose(V, Vr, rtol=1e-7, atol=1e-10 * abs(Vr[-1 if M <= 3 else -2]))
    else:                                                   # large M: all 3M slices run on the GPU, the oracle checks a few
        few = [(0, M), (0, 1), (M - 1, M), (3, 7), (0, M // 2), (M // 2, M)]
        V = gp.sobol_closed(few)
        g, phi = o.sobol_prepare(X, alpha_ref[None, :], np.array([var]), ell[None, :])
        Vr = o.sobol_V_pair(X, g[0], g[0], phi[0], phi[0], few)
        np.testing.assert_allclose(V, Vr, rtol=1e-7, atol=1e-10 * abs(Vr[0]))
    gp.close()


def test_isotropic_kernel_via_repeated_lengthscale(gpu):
    """Isotropic GPs (the '.i' models, gpr/kernels.py:46-47) are the ARD kernel with one lengthscale repeated."""
    X, y = o.synthetic_fold(300, 5, k=4)
    gp = gpu.RcGP(X, y)
    gp.set_hyper(1.7, 0.9, 0.02)
    lml, grad = gp.lml_grad()
    lml_ref, grad_ref = o.lml_and_grad(X, y, np.array([1.7]), 0.9, 0.02)
    assert lml == pytest.approx(lml_ref, rel=1e-10)
    assert np.sum(grad[:5]) == pytest.approx(grad_ref[0], rel=1e-8)
    np.testing.assert_allclose(grad[5:], grad_ref[1:], rtol=1e-8)
    gp.close()


def test_set_y_switches_output_without_reupload(gpu):
    X, y0 = o.synthetic_fold(400, 4, k=1, l=0)
    _, y1 = o.synthetic_fold(400, 4, k=1, l=1)
    ell = np.array([0.9, 1.5, 2.5, 3.0])
    gp = gpu.RcGP(X, y0)
    gp.set_hyper(ell, 1.0, 0.01)
    a0 = gp.lml()
    gp.set_y(y1)
    a1 = gp.lml()
    assert a0 == pytest.approx(o.lml(X, y0, ell, 1.0, 0.01), rel=1e-10)
    assert a1 == pytest.approx(o.lml(X, y1, ell, 1.0, 0.01), rel=1e-10)
    gp.close()


def test_set_hyper_with_unchanged_values_keeps_the_factor(gpu):
    """The fit driver sets the optimum again after its last evaluation: bit-identical hyper-parameters must not cost another
    Cholesky, and anything else (another value, another y) must."""
    import time
    N, M = 2048, 3
    X, y = o.synthetic_fold(N, M, k=4)
    ell, var, noise = np.array([1.0, 1.5, 2.0]), 1.2, 0.01
    gp = gpu.RcGP(X, y)
    gp.set_hyper(ell, var, noise)
    lml, grad = gp.lml_grad()
    t0 = time.perf_counter()
    gp.set_hyper(ell.copy(), var, noise)
    again = gp.lml()
    dt_cached = time.perf_counter() - t0
    assert again == lml
    gp.set_hyper(ell, var, noise * (1 + 1e-9))                    # anything different invalidates
    assert gp.lml() != lml
    gp.set_hyper(ell, var, noise)
    t0 = time.perf_counter()
    fresh = gp.lml()
    dt_fresh = time.perf_counter() - t0
    assert fresh == pytest.approx(lml, rel=1e-13)
    assert dt_cached < 0.5 * dt_fresh                             # no Gram + Cholesky behind the cached call
    y2 = o.synthetic_fold(N, M, k=4, l=1)[1]
    gp.set_y(y2)
    gp.set_hyper(ell, var, noise)                                 # unchanged hyper-parameters, but y changed: recompute
    assert gp.lml() == pytest.approx(o.lml(X, y2, ell, var, noise), rel=1e-10)
    gp.close()


def test_predict_chunking_and_ragged_counts(gpu):
    X, y = o.synthetic_fold(500, 3, k=2)
    ell = np.array([0.8, 1.6, 2.4])
    gp = gpu.RcGP(X, y)
    gp.set_hyper(ell, 1.1, 0.02)
    for n in (1, 130, 4096 + 77):                       # last one crosses the internal 4096-point chunk
        Xs, _ = o.synthetic_fold(n, 3, k=33)
        m, s = gp.predict(Xs, False)
        mr, sr = o.predict(X, y, ell, 1.1, 0.02, Xs, False)
        assert relmax(m, mr) < 1e-9
        np.testing.assert_allclose(s, sr, rtol=1e-7)
    m, s = gp.predict(np.zeros((0, 3)))
    assert m.shape == (0,) and s.shape == (0,)
    gp.close()


def test_not_positive_definite_is_reported(gpu):
    """Duplicate rows with zero noise: K is singular. TensorFlow raises InvalidArgumentError from tf.linalg.cholesky
    (gpr/models.py:439); the C ABI returns the LAPACK-style index and the binding raises ValueError."""
    X, y = o.synthetic_fold(200, 2, k=5)
    X[150] = X[20]
    gp = gpu.RcGP(X, y)
    gp.set_hyper([1.0, 1.0], 1.0, 0.0)
    with pytest.raises(gpu.NotPositiveDefiniteError) as info:
        gp.lml()
    assert 1 <= info.value.k <= 200
    gp.set_hyper([1.0, 1.0], 1.0, 1e-2)                  # the handle stays usable
    assert np.isfinite(gp.lml())
    gp.close()


def test_fit_that_runs_into_a_singular_gram_matrix_stops_where_lapack_stops(gpu, tmp_path):
    """Found by the host tests of round 3: on this fold (seed 5, fold 1 of a 4-fold split of 200 smooth rows) the L-BFGS-B line search
    reaches hyper-parameters at which K + noise I is not positive definite in fp64. The reference would get TensorFlow's Cholesky error
    there; here the library must report it (not return a wrong LML) and name the SAME leading minor as LAPACK at the same point."""
    import pandas as pd
    import scipy.linalg
    from romcomma_amd.data.storage import Fold, Repository
    from romcomma_amd.gpr.optimize import fit_lbfgsb
    rng = np.random.default_rng(0)
    U = rng.random((200, 3))
    y = np.sin(2 * np.pi * U[:, 0]) + 0.7 * U[:, 1] ** 2 + 0.05 * U[:, 2] + 0.02 * rng.standard_normal(200)
    columns = pd.MultiIndex.from_tuples([('X', f'X.{m}') for m in range(3)] + [('Y', 'Y.0')])
    repo = Repository.from_df(tmp_path / 'repo', pd.DataFrame(np.concatenate([U, y[:, None]], axis=1), columns=columns)).into_K_folds(-4, seed=5)
    fold = Fold(repo, 1)
    X, Y = np.ascontiguousarray(fold.X.values), np.ascontiguousarray(fold.Y.values[:, 0])
    gp = gpu.RcGP(X, Y)
    seen = []
    set_hyper = gp.set_hyper
    gp.set_hyper = lambda ell, var, noise: (seen.append((np.array(ell), var, noise)), set_hyper(ell, var, noise))[1]
    with pytest.raises(gpu.NotPositiveDefiniteError) as failure:
        fit_lbfgsb(gp, 5.0 * np.ones(3), 2.0, 0.02)
    gp.close()
    ell, var, noise = seen[-1]
    with pytest.raises(np.linalg.LinAlgError) as lapack:
        scipy.linalg.cholesky(o.noisy_gram(X, ell, var, noise), lower=True)
    assert f'{failure.value.k}-th leading minor' in str(lapack.value), (failure.value.k, str(lapack.value))


def test_not_positive_definite_inside_the_multi_stream_cholesky(gpu):
    """The same failure deep inside the four-stream factorisation (4 outer panels): the duplicate of row 700 sits at row 1500, so
    the first non-positive pivot is found by a diagonal kernel of the third panel, long after the chain, the column work and the
    outer updates have been queued. The run must drain, report a leading minor in that neighbourhood and leave the handle usable."""
    N = 2000
    X, y = o.synthetic_fold(N, 3, k=8)
    X[1500] = X[700]
    gp = gpu.RcGP(X, y)
    ell = np.array([0.05, 0.06, 0.07])                   # short lengthscales: K is close to the identity apart from the duplicate
    gp.set_hyper(ell, 1.0, 0.0)
    with pytest.raises(gpu.NotPositiveDefiniteError) as info:
        gp.lml_grad()
    assert 1501 <= info.value.k <= N
    gp.set_hyper(ell, 1.0, 1e-2)
    lml, grad = gp.lml_grad()
    lml_ref, grad_ref = o.lml_and_grad(X, y, ell, 1.0, 1e-2)
    assert lml == pytest.approx(lml_ref, rel=1e-9)
    np.testing.assert_allclose(grad, grad_ref, rtol=1e-6, atol=1e-8 * np.max(np.abs(grad_ref)))
    gp.close()


def test_argument_errors(gpu):
    X, y = o.synthetic_fold(64, 2)
    gp = gpu.RcGP(X, y)
    with pytest.raises(gpu.RcgpError, match='not set'):
        gp.lml()
    with pytest.raises(gpu.RcgpError, match='positive'):
        gp.set_hyper([1.0, -1.0], 1.0, 0.1)
    gp.set_hyper([1.0, 2.0], 1.0, 0.1)
    with pytest.raises(gpu.RcgpError, match='bad slice'):
        gp.sobol_closed([(1, 5)])
    with pytest.raises(ValueError):
        gp.predict(np.zeros((3, 5)))
    gp.close()


def test_exp_accuracy_through_gram(gpu):
    """rc_exp against numpy. (a) 1-D points in [0, 39]: exponents cover [-760, 0], both sides form r^2 by the expansion
    |z_i|^2 + |z_j|^2 - 2 z_i z_j, whose rounding (~760 * 2.2e-16 absolute in the exponent) bounds the agreement; exact 0
    past the fp64 underflow. (b) points in [0, 4]: the expansion is benign and exp itself shows: a few ulp."""
    x = np.linspace(0.0, 39.0, 1024)[:, None]
    gp = gpu.RcGP(x, np.zeros(1024))
    gp.set_hyper([1.0], 1.0, 0.0)
    K = gp.gram()
    ref = o.gram(x, np.array([1.0]), 1.0)
    big = ref > 1e-300
    assert np.max(np.abs(K[big] - ref[big]) / ref[big]) < 1e-12
    assert np.all(K[ref == 0.0] == 0.0)
    gp.close()
    x = np.linspace(0.0, 4.0, 1024)[:, None]
    gp = gpu.RcGP(x, np.zeros(1024))
    gp.set_hyper([1.0], 1.0, 0.0)
    K = gp.gram()
    ref = np.exp(-0.5 * (x - x.T) ** 2)                    # difference form: the accurate value
    assert np.max(np.abs(K - ref) / ref) < 1e-14
    gp.close()


# --------------------------------------------------------------------------------------------------------------------
# BASELINE.json sizes: size-independent properties (the oracle cannot run these in seconds)
# --------------------------------------------------------------------------------------------------------------------

# BASELINE configs[1], [2] and one fold of [4]; N = 49152: beyond every configuration -- Np^2 > 2^31 elements, so every offset that is not
# 64-bit shows (58 GB of work matrices, 1.2e14 flops per evaluation: 8 s for the whole case)
@pytest.mark.parametrize('N,M', [(8192, 5), (16384, 10), (28672, 20), (49152, 6)])
def test_full_size_properties(gpu, N, M):
    X, y = o.synthetic_fold(N, M)
    ell, var, noise = o.bench_hyper(M)
    gp = gpu.RcGP(X, y)
    gp.set_hyper(ell, var, noise)
    lml, grad = gp.lml_grad()
    assert np.isfinite(lml) and np.all(np.isfinite(grad))
    # (1) K alpha = y - noise alpha  <=>  predict_f mean at training points (check_K_inv_Y, gpr/models.py:446-463)
    idx = np.random.default_rng(0).choice(N, 256, replace=False)
    alpha = gp.k_inv_y()
    mean, sd = gp.predict(X[idx], False)
    np.testing.assert_allclose(mean, y[idx] - noise * alpha[idx], rtol=0, atol=2e-8 * np.max(np.abs(y)))
    assert np.all(sd ** 2 <= noise * 1.000001) and np.all(sd >= 0)      # posterior f-variance at a training point < noise
    # (2) the gradient is the derivative of the LML: central difference along one lengthscale, the variance and the noise
    for p, h in ((0, 1e-5), (M, 1e-5), (M + 1, 1e-7)):
        def at(delta):
            e, v, n_ = ell.copy(), var, noise
            if p < M:
                e[p] += delta
            elif p == M:
                v += delta
            else:
                n_ += delta
            gp.set_hyper(e, v, n_)
            return gp.lml()
        fd = (at(h) - at(-h)) / (2 * h)
        assert grad[p] == pytest.approx(fd, rel=2e-4), f'parameter {p}'
    gp.set_hyper(ell, var, noise)
    # (3) Sobol invariants (gsa/calibrators.py:90,97; gsa/models.py:207-214)
    V = gp.sobol_closed(o.all_slices(M) + [(M, M)])
    first, closed, comp, full, empty = V[:M], V[M:2 * M], V[2 * M:3 * M], V[3 * M], V[3 * M + 1]
    assert full > 0 and abs(empty) < 1e-9 * full
    assert closed[M - 1] == pytest.approx(full, rel=1e-12)
    assert first[0] == pytest.approx(closed[0], rel=1e-12)
    assert np.all(np.diff(closed) >= -1e-9 * full)
    assert np.all(first <= closed + 1e-9 * full)
    total = 1.0 - comp / full
    assert np.all(total >= first / full - 1e-7)
    gp.close()


@pytest.mark.parametrize('L', [1, 2])
def test_sobol_error_terms(gpu, L):
    """rcgp_sobol_error_terms against the reduced-form oracle (itself checked against the literal transliteration of
    ClosedSobolWithError): the four ingredients per slice and output pair, at a ragged N."""
    from oracle import sobol_error_oracle as e
    N, M = 333, 4
    X, _ = o.synthetic_fold(N, M, k=2)
    rng = np.random.default_rng(11)
    ell = rng.uniform(0.7, 2.5, (L, M))
    F = rng.uniform(0.8, 1.5, L)
    noise = rng.uniform(0.01, 0.03, L)
    Y = np.stack([o.synthetic_fold(N, M, k=2, l=l)[1] for l in range(L)], 1)
    alpha = np.stack([o.k_inv_y(X, Y[:, l], ell[l], F[l], noise[l]) for l in range(L)])
    Kc = np.stack([o.k_cho(X, ell[l], F[l], noise[l]) for l in range(L)])
    ref = e.ClosedSobolWithErrorOracle(X, alpha[:, None, :], F[None, :], ell, Kc, is_T_partial=False)
    slices = o.all_slices(M)[:-1] + [(M, M), (1, 3)]               # (1, 3): neither first-order, closed nor a complement
    for b in range(L):
        gp = gpu.RcGP(X, Y[:, b])
        gp.set_hyper(ell[b], F[b], noise[b])
        for a in range(L):
            got = gp.sobol_error_terms(slices) if a == b else gp.sobol_error_terms(slices, ell[a], F[a], alpha[a])
            for s, sl in enumerate(slices):
                want = (0.0, 0.0, 0.0, 0.0) if sl[0] == sl[1] else e.error_terms_pair(X, a, b, ref.g0, ref.g, ref.phi, ref.ups, ref.pre, Kc, sl)
                for k in range(4):
                    assert got[k][s] == pytest.approx(want[k], rel=1e-7, abs=1e-12), (a, b, sl, k)
        with pytest.raises(gpu.RcgpError, match='bad slice'):
            gp.sobol_error_terms([(3, 1)])
        gp.close()


_KNOB_FINGERPRINTS = {}


@pytest.fixture(scope='module')
def knob_reference_dir(tmp_path_factory):
    """Where the first child process of a knob case leaves the oracle's reference values for the later ones (the oracle at
    N = 9100 costs more than the GPU side of a case; once per case, not once per knob set)."""
    return tmp_path_factory.mktemp('knob_reference')


def _knob_case(env, case, reference_dir):
    """The schedule knobs are read once per process (the first rcgp_create), so every knob set runs in a process of its own:
    tests/knob_case.py checks against the oracle there and prints one line with a bit-level fingerprint of the GPU's numbers, which
    must be THE SAME for every knob set of a case: the schedule decides when a tile is updated, never in which order its k-slabs are
    added up. One child at a time -- the GPU box limits concurrent processes."""
    import os
    import subprocess
    import sys
    full_env = {**os.environ, **env, 'RCGP_KNOB_REF_DIR': str(reference_dir)}
    done = subprocess.run([sys.executable, str(Path(__file__).resolve().parent / 'knob_case.py'), case], env=full_env, capture_output=True,
                          text=True, timeout=280)
    words = done.stdout.strip().split()
    assert done.returncode == 0 and len(words) == 3 and words[:2] == [case, 'ok'], (env, done.stdout[-2000:], done.stderr[-2000:])
    first_env, first = _KNOB_FINGERPRINTS.setdefault(case, (env, words[2]))
    assert words[2] == first, f'{case}: {env} and {first_env} differ in the bits of their results'


@pytest.mark.parametrize('env', [{}, {'RCGP_LOOKAHEAD': '0'}, {'RCGP_FINE': '0'}, {'RCGP_NB': '256', 'RCGP_EXT': '1', 'RCGP_DEPTH': '1', 'RCGP_TAIL': '0'},
                                 {'RCGP_NB': '128', 'RCGP_EXT': '3', 'RCGP_DEPTH': '8', 'RCGP_TAIL': '0'},
                                 {'RCGP_NB': '384', 'RCGP_DEPTH': '2', 'RCGP_TAIL': '4', 'RCGP_FARG': '3'},
                                 {'RCGP_EXT': '6', 'RCGP_DEPTH': '1', 'RCGP_NB': '256', 'RCGP_TAIL': '0', 'RCGP_LEAN': '0', 'RCGP_FARG': '1'},
                                 {'RCGP_EXT': '2', 'RCGP_FARG': '4', 'RCGP_LEAN': '5', 'RCGP_HALF_TILES': '0'}])
def test_tuning_knobs_do_not_change_results(gpu, env, knob_reference_dir):
    """Every run-time variant of the factorisation's schedule -- sequential Cholesky, coarse panel chain, and the fine-grained Cholesky
    with other panel widths / window depths / chain extensions / tail lengths / far-update groups / lean thresholds -- is the same
    arithmetic: LML, gradient and alpha against the oracle at N = 1700 (14 blocks; a ragged last outer panel), and the same BITS as the
    default schedule. With the default tail (64 blocks) a matrix this small is all tail and NB / EXT / DEPTH have nothing to act on
    (ADVICE r3): the sets that name them switch the tail off (RCGP_TAIL = 0: 7 and 14 outer panels) or shorten it to 4 blocks."""
    _knob_case(env, 'evaluation', knob_reference_dir)


@pytest.mark.parametrize('env', [{}, {'RCGP_NB': '256', 'RCGP_DEPTH': '3', 'RCGP_TAIL': '0'}, {'RCGP_EXT': '1', 'RCGP_DEPTH': '12', 'RCGP_NB': '384', 'RCGP_TAIL': '8'}])
def test_fine_grained_cholesky_factor_many_panels_other_schedules(gpu, env, knob_reference_dir):
    """The factor itself, entry by entry against LAPACK, at N = 3400 (27 blocks) over 14 outer panels without a tail (a window three
    panels deep: its last piece waits for the previous bulk kernel) and over 5 panels + an 8-block tail with a chain extension of one
    block (the chain's stream waits for the previous panel's first window piece)."""
    _knob_case(env, 'factor', knob_reference_dir)


@pytest.mark.parametrize('env', [{}, {'RCGP_NB': '512', 'RCGP_EXT': '1'}, {'RCGP_EXT': '2', 'RCGP_DEPTH': '1', 'RCGP_TAIL': '16', 'RCGP_FARG': '3'},
                                 {'RCGP_NB': '1536', 'RCGP_EXT': '6', 'RCGP_DEPTH': '3'}])
def test_tuning_knobs_on_a_matrix_taller_than_the_tail(gpu, env, knob_reference_dir):
    """The knob sets again at N = 9100 (72 blocks): outer panels with window pieces and a bulk update in front of the tail (64 blocks by
    default, 16 in one set: 4 outer panels of 1024 first), columns taller than the split far update's threshold, far updates in groups of
    three steps, a chain extension of one block (the next panel's first diagonal block then gets the last column's update from a window piece)."""
    _knob_case(env, 'tall', knob_reference_dir)


def test_fine_grained_cholesky_factor_many_panels(gpu):
    """The five-stream Cholesky (diagonal chain, column work, far updates, window pieces, bulk update) over 7+ outer panels with a
    ragged last one: the factor itself, entry by entry, against LAPACK on the oracle's Gram matrix, and w = L^-1 y through the LML."""
    N, M = 3400, 3
    X, y = o.synthetic_fold(N, M, k=11)
    ell, var, noise = np.array([0.7, 1.5, 2.4]), 0.9, 0.01
    gp = gpu.RcGP(X, y)
    gp.set_hyper(ell, var, noise)
    Lc = gp.k_cho()
    Lref = o.k_cho(X, ell, var, noise)
    assert relmax(Lc, Lref) < 1e-11
    assert np.all(np.triu(Lc, 1) == 0.0)
    assert gp.lml() == pytest.approx(o.lml(X, y, ell, var, noise), rel=1e-11)
    gp.close()


def test_knobs_are_read_once_per_process(gpu, monkeypatch, capfd):
    """One stream set and one schedule per process: a knob changed after the first rcgp_create is ignored (with a message), so two
    handles can never run different schedules on the shared streams."""
    N, M = 900, 3
    X, y = o.synthetic_fold(N, M, k=2)
    ell, var, noise = np.array([0.9, 1.4, 2.2]), 1.0, 0.02
    first = gpu.RcGP(X, y)
    first.set_hyper(ell, var, noise)
    before = first.lml_grad()
    monkeypatch.setenv('RCGP_NB', '128')
    monkeypatch.setenv('RCGP_FINE', '0')
    second = gpu.RcGP(X, y)
    second.set_hyper(ell, var, noise)
    after = second.lml_grad()
    assert after[0] == before[0] and np.array_equal(after[1], before[1])           # the same schedule: bit-identical
    assert 'ignored' in capfd.readouterr().err
    first.close()
    second.close()


@pytest.mark.parametrize('N,M', [(8192, 5), (16384, 10)])                  # BASELINE configs[1] and [2] (the headline configuration)
def test_full_size_oracle_comparison(gpu, N, M):
    """The headline sizes against the oracle itself, not only through properties: LML and its gradient at the fixed benchmark
    hyper-parameters from ``rcgp_lml_grad`` and from ``oracle.lml_and_grad_blas`` (LAPACK potrf + potri and BLAS-3 sums
    on the host cores: about 10 s and 25 s on the GPU box). 1e-8 on the LML, 1e-6 of the largest component on the gradient; K_inv_Y, four
    Sobol conditional variances and (at configs[1]) 512 predictions against the oracle as well."""
    X, y = o.synthetic_fold(N, M)
    ell, var, noise = o.bench_hyper(M)
    gp = gpu.RcGP(X, y)
    gp.set_hyper(ell, var, noise)
    lml, grad = gp.lml_grad()
    # the second half of the headline metric at full size (round 4): the conditional variances of four slices -- first-order 0, closed
    # [0, M/2), complement [M/2, M), full (gsa/calibrators.py:60-80, gsa/models.py:77-90) -- and, at BASELINE configs[1], predict on 512 of
    # its o = 8192 new points (gpr/models.py:375-384)
    slices = [(0, 1), (0, M // 2), (M // 2, M), (0, M)]
    V = gp.sobol_closed(slices)
    alpha_gpu = gp.k_inv_y()
    if N == 8192:
        Xs = o.synthetic_fold(8192, M, k=1)[0][::16]
        mean, sd = gp.predict(Xs)
        mean_f, sd_f = gp.predict(Xs, include_noise=False)
    gp.close()
    lml_ref, grad_ref = o.lml_and_grad_blas(X, y, ell, var, noise)
    assert lml == pytest.approx(lml_ref, rel=1e-8)
    assert np.max(np.abs(grad - grad_ref)) <= 1e-6 * np.max(np.abs(grad_ref))
    alpha = o.k_inv_y(X, y, ell, var, noise)
    assert relmax(alpha_gpu, alpha) < 1e-8
    g, phi = o.sobol_prepare(X, alpha[None, :], np.array([var]), ell[None, :])
    V_ref = o.sobol_V_pair(X, g[0], g[0], phi[0], phi[0], slices)             # the oracle's O(N^2) pair form over all rows: ~2 s per slice
    np.testing.assert_allclose(V, V_ref, rtol=1e-8)
    np.testing.assert_allclose(V[:3] / V[3], np.asarray(V_ref)[:3] / V_ref[3], rtol=0, atol=2e-8)      # the indices themselves (a ratio of two 1e-8 values)
    if N == 8192:
        rmean, rsd = o.predict(X, y, ell, var, noise, Xs)
        rmean_f, rsd_f = o.predict(X, y, ell, var, noise, Xs, False)
        np.testing.assert_allclose(mean, rmean, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(mean_f, rmean_f, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(sd, rsd, rtol=1e-6)
        np.testing.assert_allclose(sd_f, rsd_f, rtol=1e-5, atol=1e-9)         # (var_f = k** - |L^-1 k*|^2 cancels to ~1e-3 of k**)


def test_sobol_error_terms_many_dimensions(gpu):
    """M = 20 (BASELINE configs[4]): the matvec kernel's LDS footprint grows with M; values against the reduced-form oracle."""
    from oracle import sobol_error_oracle as e
    N, M = 150, 20
    X, y = o.synthetic_fold(N, M, k=6)
    ell = np.random.default_rng(3).uniform(2.0, 6.0, M)
    F, noise = 1.2, 0.03
    alpha = o.k_inv_y(X, y, ell, F, noise)
    Kc = o.k_cho(X, ell, F, noise)[None]
    ref = e.ClosedSobolWithErrorOracle(X, alpha[None, None, :], np.array([[F]]), ell[None, :], Kc, is_T_partial=False)
    slices = [(0, 1), (7, 8), (0, 5), (0, M), (12, M), (M - 1, M), (3, 11), (5, 7)]      # the last two: arbitrary slices
    gp = gpu.RcGP(X, y)
    gp.set_hyper(ell, F, noise)
    got = gp.sobol_error_terms(slices)
    for s, sl in enumerate(slices):
        want = e.error_terms_pair(X, 0, 0, ref.g0, ref.g, ref.phi, ref.ups, ref.pre, Kc, sl)
        for k in range(4):
            assert got[k][s] == pytest.approx(want[k], rel=1e-6, abs=1e-12 * abs(want[0]) + 1e-15), (sl, k)
    gp.close()
    # M > 29: the column accumulators of the matvec kernel no longer fit in LDS together and the canonical slices take several passes
    # (window of 3M indices walked by the host); same ingredients, same oracle, at a ragged N
    N2, M2 = 139, 33
    X2, y2 = o.synthetic_fold(N2, M2, k=2)
    ell2 = np.random.default_rng(5).uniform(3.0, 9.0, M2)
    alpha2 = o.k_inv_y(X2, y2, ell2, 0.9, 0.04)
    Kc2 = o.k_cho(X2, ell2, 0.9, 0.04)[None]
    ref2 = e.ClosedSobolWithErrorOracle(X2, alpha2[None, None, :], np.array([[0.9]]), ell2[None, :], Kc2, is_T_partial=False)
    slices2 = [(0, 1), (32, 33), (0, 17), (0, M2), (20, M2), (1, M2), (4, 30)]
    gp = gpu.RcGP(X2, y2)
    gp.set_hyper(ell2, 0.9, 0.04)
    got2 = gp.sobol_error_terms(slices2)
    for s, sl in enumerate(slices2):
        want = e.error_terms_pair(X2, 0, 0, ref2.g0, ref2.g, ref2.phi, ref2.ups, ref2.pre, Kc2, sl)
        for k in range(4):
            assert got2[k][s] == pytest.approx(want[k], rel=1e-6, abs=1e-12 * abs(want[0]) + 1e-15), (sl, k)
    gp.close()


def test_sobol_error_terms_at_size(gpu):
    """The standard-error ingredients at N = 4096 (32 x 32 pair tiles per form, |L^-1 f|^2 through the full-size L^-1): the output itself
    and a second output, a first-order, a closed, a complement and an arbitrary slice, against the reduced-form oracle (four N x N pair
    matrices per slice on the host: the reason this is not run at N = 16384)."""
    from oracle import sobol_error_oracle as e
    N, M = 4096, 5
    X, y = o.synthetic_fold(N, M)
    ell, var, noise = o.bench_hyper(M)
    ell_a, var_a, noise_a = ell * 1.3, 0.8, 0.02
    y_a = np.cos(X[:, 0]) + 0.5 * X[:, 1] * X[:, 2] + 0.05 * np.random.default_rng(4).standard_normal(N)      # a second output on the same design
    y_a = (y_a - y_a.mean()) / y_a.std()
    alpha = np.stack([o.k_inv_y(X, y_a, ell_a, var_a, noise_a), o.k_inv_y(X, y, ell, var, noise)])
    Kc = [None, o.k_cho(X, ell, var, noise)]                                   # (only output b's factor is ever used)
    ells, F = np.stack([ell_a, ell]), np.array([var_a, var])
    phi, ups = 1 / (ells * ells + 1), 1 / (ells * ells + 2)
    g0 = (F * np.sqrt(np.prod(ells * ells * phi, axis=1)))[:, None] * np.exp(-0.5 * np.einsum('lm,nm->ln', phi, X ** 2))
    g = g0 * alpha
    g = g - g.mean(axis=1, keepdims=True)                                      # gsa/calibrators.py:90
    pre = F * np.sqrt(np.prod(ells * ells * ups, axis=1))                      # :384
    slices = [(2, 3), (0, 3), (3, M), (1, 4)]
    gp = gpu.RcGP(X, y)
    gp.set_hyper(ell, var, noise)
    got_self = gp.sobol_error_terms(slices)
    got_pair = gp.sobol_error_terms(slices, ell_a, var_a, alpha[0])
    gp.close()
    for a, got in ((1, got_self), (0, got_pair)):
        want = np.array([e.error_terms_pair(X, a, 1, g0, g, phi, ups, pre, Kc, sl) for sl in slices])          # [slice][phi_d, psi_d, phi_m, psi_m]
        # every ingredient is a sum of N^2 terms of both signs (g is centred): for a slice the output hardly depends on -- here x_2 alone
        # for the second output, 4e-8 where sum |terms| = 6e6 -- the value carries the rounding of that sum: numpy's fp64 gives 3.98796e-8, the
        # same sum in np.longdouble 3.98792e-8, the GPU's tile-wise sums 3.98791e-8 (the oracle is the noisier side here). Held to 1e-6 of
        # the ingredient's largest value over the slices, i.e. on the scale W = phi - psi and T are formed on (tests/test_gpu_host_api.py)
        for k in range(4):
            scale = np.max(np.abs(want[:, k]))
            for s, sl in enumerate(slices):
                assert got[k][s] == pytest.approx(want[s, k], rel=1e-6, abs=1e-6 * scale), (a, sl, k, scale)


def test_multi_stream_cholesky_is_bitwise_reproducible(gpu):
    """The look-ahead factorisation runs on six streams ordered by events only. Every tile is written by kernels in one fixed order,
    so repeated factorisations must agree BIT FOR BIT; a missing dependency would show up here as run-to-run differences."""
    from romcomma_amd import _lib
    for N, M in ((3000, 4), (5200, 6)):
        X, y = o.synthetic_fold(N, M, k=7)
        ell, var, noise = o.bench_hyper(M)
        with _lib.RcGP(X, y) as gp:
            gp.set_hyper(ell, var, noise)
            first = None
            for rep in range(12):
                gp.stage_gram()
                gp.stage_potrf()
                gp.sync()
                Lc = gp.k_cho()                      # the factor is cached: no refactorisation here
                lml = gp.lml()
                if first is None:
                    first = (Lc, lml)
                else:
                    assert lml == first[1]
                    assert np.array_equal(Lc, first[0])
            _, g0 = gp.lml_grad()
            gp.stage_gram()
            _, g1 = gp.lml_grad()
            assert np.array_equal(g0, g1)


def test_full_size_evaluation_is_bitwise_reproducible(gpu):
    """The same at C2 size through quantities that depend on every entry of the factor: LML, gradient, K_inv_Y."""
    from romcomma_amd import _lib
    N, M = 16384, 10
    X, y = o.synthetic_fold(N, M)
    ell, var, noise = o.bench_hyper(M)
    with _lib.RcGP(X, y) as gp:
        gp.set_hyper(ell, var, noise)
        ref = None
        for rep in range(5):
            gp.stage_gram()                                 # invalidates the cached factor
            lml, grad = gp.lml_grad()
            alpha = gp.k_inv_y()
            if ref is None:
                ref = (lml, grad, alpha)
            else:
                assert lml == ref[0] and np.array_equal(grad, ref[1]) and np.array_equal(alpha, ref[2])


def test_random_shapes_against_oracle(gpu):
    """Independent GP over a spread of sizes around the tile (128), fine-chain (512) and outer-panel (1024) boundaries and M from 1
    to 33 (both LDS layouts of the gradient kernel): LML, gradient, K_inv_Y, predict and all Sobol slices against the oracle."""
    rng = np.random.default_rng(7)
    cases = [(127, 1), (128, 2), (129, 3), (255, 9), (511, 4), (512, 5), (513, 2), (1023, 3), (1024, 6), (1025, 33), (1151, 8),
             (1400, 12), (2049, 5)]
    for i, (N, M) in enumerate(cases):
        X, y = o.synthetic_fold(N, M, k=20 + i)
        ell = 0.6 + 2.5 * rng.random(M)
        var, noise = 0.5 + rng.random(), 0.005 + 0.03 * rng.random()
        Xs = o.synthetic_fold(41, M, k=70 + i)[0]
        v, g = o.lml_and_grad(X, y, ell, var, noise)
        mean, sd = o.predict(X, y, ell, var, noise, Xs)
        alpha = o.k_inv_y(X, y, ell, var, noise)
        slices = o.all_slices(M)[:12] + [(0, M)]
        with gpu.RcGP(X, y) as gp:
            gp.set_hyper(ell, var, noise)
            lml, grad = gp.lml_grad()
            a = gp.k_inv_y()
            m, s = gp.predict(Xs)
            V = gp.sobol_closed(slices)
        ref = o.ClosedSobolOracle(X, alpha[None, None, :], np.array([[var]]), ell[None, :])
        Vref = np.array([ref.marginalize(sl)['V'][0, 0] for sl in slices])
        assert lml == pytest.approx(v, rel=1e-10), (N, M)
        assert np.abs(grad - g).max() <= 1e-7 * max(1.0, np.abs(g).max()), (N, M)
        assert np.allclose(a, alpha, rtol=1e-6, atol=1e-8 * np.abs(alpha).max()), (N, M)
        assert np.allclose(m, mean, rtol=1e-8, atol=1e-9) and np.allclose(s, sd, rtol=1e-7, atol=1e-9), (N, M)
        assert np.allclose(V, Vref, rtol=1e-8, atol=1e-10 * np.abs(Vref).max()), (N, M)      # (the empty slice is 0 up to rounding)


def test_stage_potrf_needs_a_fresh_gram_matrix(gpu):
    """rcgp_stage_potrf factors what rcgp_stage_gram left in A; called twice (or after any factorising call) it would factor L as
    if it were K and every later result on the handle would be garbage: refused with status -5."""
    X, y = o.synthetic_fold(300, 3)
    gp = gpu.RcGP(X, y)
    with pytest.raises(gpu.RcgpError, match='hyper-parameters not set'):
        gp.stage_potrf()
    gp.set_hyper(*o.bench_hyper(3))
    with pytest.raises(gpu.RcgpError, match='no fresh Gram'):
        gp.stage_potrf()
    gp.stage_gram()
    gp.stage_potrf()
    with pytest.raises(gpu.RcgpError, match='no fresh Gram'):
        gp.stage_potrf()
    ref = o.lml(X, y, *o.bench_hyper(3))
    assert gp.lml() == pytest.approx(ref, rel=1e-10)            # the handle is still good
    gp.lml_grad()
    with pytest.raises(gpu.RcgpError, match='no fresh Gram'):   # A holds L again
        gp.stage_potrf()
    gp.close()


@pytest.mark.parametrize('flags', [dict(train_lengthscales=False), dict(train_variance=False), dict(train_noise=False),
                                   dict(train_lengthscales=False, train_noise=False), dict(train_variance=False, train_noise=False, is_isotropic=True)])
def test_fit_with_parameters_held_fixed(gpu, flags):
    """Kernel.calibrate / Likelihood.calibrate switch hyper-parameters off (gf.set_trainable, gpr/kernels.py:59-70,
    gpr/models.py:71-80): the SciPy driver then runs over the remaining ones. Same driver, same start, same mask on the oracle and
    on the GPU: same optimum (LML* within 1e-5 relative), the held parameters untouched, bit for bit."""
    import scipy.optimize
    from romcomma_amd.gpr.optimize import fit_lbfgsb, inv_softplus, softplus, sigmoid, LIKELIHOOD_LOWER
    N, M = 400, 3
    X, y = o.synthetic_fold(N, M, k=4)
    iso = flags.get('is_isotropic', False)
    ell0, var0, noise0 = (np.array([1.7]) if iso else np.array([1.5, 2.5, 4.0])), 1.3, 0.05
    gp = gpu.RcGP(X, y)
    fit = fit_lbfgsb(gp, ell0, var0, noise0, **flags)
    gp.close()
    tl, tv, tn = flags.get('train_lengthscales', True), flags.get('train_variance', True), flags.get('train_noise', True)
    n_ell = ell0.shape[0]
    u_all = np.concatenate([inv_softplus(ell0), [inv_softplus(var0)], [inv_softplus(noise0 - LIKELIHOOD_LOWER)]])
    mask = np.array([tl] * n_ell + [tv, tn])

    def objective(u_train):                                       # the oracle under the same parametrisation and mask
        u = u_all.copy()
        u[mask] = u_train
        ell = np.broadcast_to(softplus(u[:n_ell]), (M,))
        lml, g = o.lml_and_grad(X, y, ell, float(softplus(u[n_ell])), float(LIKELIHOOD_LOWER + softplus(u[n_ell + 1])))
        g_ell = np.array([np.sum(g[:M])]) if iso else g[:M]
        return -lml, -(np.concatenate([g_ell, g[M:]]) * sigmoid(u))[mask]
    ref = scipy.optimize.minimize(objective, u_all[mask], jac=True, method='L-BFGS-B', options={'maxiter': 5000, 'gtol': 1e-16})
    assert fit['log_marginal'] == pytest.approx(-ref.fun, rel=1e-5)
    if not tl:
        np.testing.assert_array_equal(fit['lengthscales'], np.broadcast_to(softplus(inv_softplus(ell0)), (M,)))
    if not tv:
        assert fit['variance'] == float(softplus(inv_softplus(var0)))
    if not tn:
        assert fit['noise'] == float(LIKELIHOOD_LOWER + softplus(inv_softplus(noise0 - LIKELIHOOD_LOWER)))
    if iso:
        assert np.all(fit['lengthscales'] == fit['lengthscales'][0])
    u_ref = u_all.copy()
    u_ref[mask] = ref.x
    np.testing.assert_allclose(fit['lengthscales'], np.broadcast_to(softplus(u_ref[:n_ell]), (M,)), rtol=5e-3)


@pytest.mark.parametrize('N,M', [(4097, 3), (5249, 2), (6527, 5), (8321, 6), (9300, 3), (11111, 4)])
def test_sizes_between_the_panel_boundaries(gpu, N, M):
    """Sizes that are no multiple of the outer panel (1024), of the tail (64 blocks) or of the far-update groups: the places where the chain's
    schedule switches form (tail panel, lean steps, split far updates, window pieces) fall differently for each. LML and gradient against the
    oracle (LAPACK potrf / potri), and bit-identical on a second evaluation."""
    X, y = o.synthetic_fold(N, M, k=N % 5)
    ell, var, noise = o.bench_hyper(M)
    ref, gref = o.lml_and_grad_blas(X, y, ell, var, noise)
    with gpu.RcGP(X, y) as gp:
        gp.set_hyper(ell, var, noise)
        lml, grad = gp.lml_grad()
        gp.stage_gram()
        lml2, grad2 = gp.lml_grad()
    assert lml == pytest.approx(ref, rel=1e-10)
    assert np.max(np.abs(grad - gref)) <= 1e-8 * np.max(np.abs(gref))
    assert lml == lml2 and np.array_equal(grad, grad2)


def test_wide_design_other_entry_points(gpu):
    """M = 70 > 64 beyond test_against_oracle: the gradient GP (its derivative rows loop over the dimensions), a whole fit against the
    oracle's, all 3 M + 1 canonical Sobol slices + an arbitrary one in one pass over chunked panels, a cross-output term, the
    ingredients of the standard errors for the output itself and for an output pair."""
    from romcomma_amd.gpr.optimize import fit_lbfgsb
    N, M = 260, 70
    X, y = o.synthetic_fold(N, M, k=4)
    rng = np.random.default_rng(7)
    ell, var, noise = rng.uniform(4.0, 12.0, M), 1.2, 0.03
    gp = gpu.RcGP(X, y)
    gp.set_hyper(ell, var, noise)
    xs = o.synthetic_fold(5, M, k=33)[0]
    mean, cov = gp.predict_gradient(xs)                      # (o, M), (o, M, o, M): V^T V, the caller applies sign and diagonal term
    m_ref, c_ref = o.predict_gradient(X, y, ell, var, noise, xs)
    np.testing.assert_allclose(mean, m_ref, rtol=1e-8, atol=1e-10)
    z = xs / ell
    kxx = var * np.exp(-0.5 * (np.sum(z * z, 1)[:, None] + np.sum(z * z, 1)[None, :] - 2.0 * z @ z.T))
    full = -np.transpose(cov, (0, 2, 1, 3))
    idx = np.arange(M)
    full[:, :, idx, idx] += kxx[:, :, None] / ell[None, None, :] ** 2
    np.testing.assert_allclose(full, c_ref, rtol=1e-7, atol=1e-9 * np.max(np.abs(c_ref)))
    slices = o.all_slices(M) + [(9, 66)]
    V = gp.sobol_closed(slices)
    alpha = o.k_inv_y(X, y, ell, var, noise)
    g, phi = o.sobol_prepare(X, alpha[None, :], np.array([var]), ell[None, :])
    check = [0, 63, 64, 69, M + 63, M + 64, 2 * M, 2 * M + 5, 2 * M + 64, 3 * M - 1, 3 * M, 3 * M + 1]      # either side of the chunk boundary
    Vr = o.sobol_V_pair(X, g[0], g[0], phi[0], phi[0], [slices[i] for i in check])
    np.testing.assert_allclose(V[check], Vr, rtol=1e-7, atol=1e-10 * abs(Vr[-2]))
    ell_j = rng.uniform(4.0, 12.0, M)
    alpha_j = o.k_inv_y(X, np.roll(y, 3), ell_j, 0.9, 0.05)
    gj, phij = o.sobol_prepare(X, alpha_j[None, :], np.array([0.9]), ell_j[None, :])
    few = [(0, M), (60, 68), (0, 1)]
    np.testing.assert_allclose(gp.sobol_cross(ell_j, 0.9, alpha_j, few), o.sobol_V_pair(X, g[0], gj[0], phi[0], phij[0], few), rtol=1e-7, atol=1e-12)
    # the standard errors' ingredients over chunked panels (k_sobol_pairs<false, WIDE>, k_sobol_matvec<WIDE>: 18 column accumulators per
    # pass beside two 64-dimension panels): slices either side of the chunk boundary, an arbitrary one across it, and a second output
    from oracle import sobol_error_oracle as e
    ell2 = np.stack([ell, ell_j])
    F2, noise2 = np.array([var, 0.9]), np.array([noise, 0.05])
    Y2 = np.stack([y, np.roll(y, 3)], 1)
    alpha2 = np.stack([alpha, alpha_j])
    Kc2 = np.stack([o.k_cho(X, ell2[l], F2[l], noise2[l]) for l in range(2)])
    ref = e.ClosedSobolWithErrorOracle(X, alpha2[:, None, :], F2[None, :], ell2, Kc2, is_T_partial=False)
    err_slices = [(0, 1), (63, 64), (64, 65), (0, 64), (0, 65), (0, M), (64, M), (63, M), (M - 1, M), (60, 68), (M, M)]
    for a, got in ((0, gp.sobol_error_terms(err_slices)), (1, gp.sobol_error_terms(err_slices, ell_j, 0.9, alpha_j))):
        for s, sl in enumerate(err_slices):
            want = (0.0, 0.0, 0.0, 0.0) if sl[0] == sl[1] else e.error_terms_pair(X, a, 0, ref.g0, ref.g, ref.phi, ref.ups, ref.pre, Kc2, sl)
            for k in range(4):
                assert got[k][s] == pytest.approx(want[k], rel=1e-6, abs=1e-12 * abs(want[0]) + 1e-15), (a, sl, k)
    # a whole fit in 72 parameters: the optimum is an optimum of the ORACLE's objective too (its LML there equals the GPU's, its gradient
    # in the unconstrained space vanishes to what L-BFGS-B's ftol leaves) -- two independent runs over so flat a surface need not meet
    start = o.lml(X, y, 5.0 * np.ones(M), 2.0, 0.02)
    fit = fit_lbfgsb(gp, 5.0 * np.ones(M), 2.0, 0.02)
    assert fit['log_marginal'] > start + 10.0 and fit['nfev'] > 20
    at_optimum, grad = o.lml_and_grad(X, y, fit['lengthscales'], fit['variance'], fit['noise'])
    assert fit['log_marginal'] == pytest.approx(at_optimum, rel=1e-9)
    lml_gpu, grad_gpu = gp.lml_grad()
    np.testing.assert_allclose(grad_gpu, grad, rtol=1e-6, atol=1e-8 * np.max(np.abs(grad)))
    gp.close()
