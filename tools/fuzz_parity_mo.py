"""Randomised parity sweep of the covariant GP and of the standard-error ingredients, not a test (GPU box; minutes):
    python tools/fuzz_parity_mo.py [cases] [seed]
Per case: a covariant GP of L <= 5 outputs on random N <= 420, M <= 70 (LML, its three gradients, joint predictions against
oracle/mogp_oracle.py), and an independent output pair on random N <= 600, M <= 40 (the four error ingredients of random slices against
oracle/sobol_error_oracle.py, held to 1e-6 of each ingredient's largest value over the slices: they are sums of both signs)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import gp_oracle as o                                  # noqa: E402  (the checkers)
from oracle import mogp_oracle as mo                               # noqa: E402
from oracle import sobol_error_oracle as e                         # noqa: E402
from romcomma_amd import _lib                                      # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
failures, worst = 0, dict(lml=0.0, grad=0.0, mean=0.0, sd=0.0, err=0.0)
t0 = time.perf_counter()
for case in range(cases):
    # ---- covariant GP
    L, M = int(rng.integers(2, 6)), int(rng.choice([1, 2, 3, 6, 11, 32, 33, 64, 65, 70]))
    N = int(rng.choice([1, 127, 128, 129, 255, 257])) if rng.random() < 0.3 else int(rng.integers(2, 421))
    X = rng.standard_normal((N, M))
    Y = np.sin(X @ rng.standard_normal((M, L)) / np.sqrt(M)) + 0.1 * rng.standard_normal((N, L))
    ell = (0.7 + 1.5 * rng.random((L, M))) * np.sqrt(M)
    C = np.tril(0.4 * rng.standard_normal((L, L)), -1) + np.diag(0.8 + rng.random(L))
    Cn = np.tril(0.03 * rng.standard_normal((L, L)), -1) + np.diag(0.1 + 0.1 * rng.random(L))
    F, S = C @ C.T, Cn @ Cn.T
    F, S = (F + F.T) / 2, (S + S.T) / 2
    Xs = rng.standard_normal((7, M))
    v, dF, dell, dS = mo.lml_and_grad(X, Y, ell, F, S)
    m_ref, s_ref = mo.predict(X, Y, ell, F, S, Xs, True)
    with _lib.RcMOGP(X, Y) as gp:
        gp.set_hyper(ell, F, S)
        lml, gF, gell, gS = gp.lml_grad()
        mean, sd = gp.predict(Xs, True)
    err = dict(lml=abs(lml - v) / max(abs(v), 1.0),
               grad=max(np.abs(a - b).max() / max(1.0, np.abs(b).max()) for a, b in ((gF, dF), (gell, dell), (gS, dS))),
               mean=np.abs(mean - m_ref).max() / max(np.abs(m_ref).max(), 1e-9), sd=np.abs(sd - s_ref).max() / np.abs(s_ref).max())
    ok = err['lml'] < 1e-10 and err['grad'] < 1e-7 and err['mean'] < 1e-7 and err['sd'] < 1e-6
    # ---- standard-error ingredients of an output pair
    M2 = int(rng.choice([1, 2, 4, 9, 20, 30, 33, 40]))
    N2 = int(rng.integers(20, 601))
    X2, y_b = o.synthetic_fold(N2, M2, k=3000 + case)
    y_a = np.cos(X2[:, 0]) + 0.3 * X2[:, -1] ** 2 + 0.05 * rng.standard_normal(N2)
    y_a = (y_a - y_a.mean()) / y_a.std()
    ells = (1.0 + 2.0 * rng.random((2, M2))) * np.sqrt(max(M2 / 4.0, 1.0))
    Fs, noises = 0.6 + rng.random(2), 0.01 + 0.04 * rng.random(2)
    alpha = np.stack([o.k_inv_y(X2, y, ells[l], Fs[l], noises[l]) for l, y in enumerate((y_a, y_b))])
    Kc = np.stack([o.k_cho(X2, ells[l], Fs[l], noises[l]) for l in range(2)])
    ref = e.ClosedSobolWithErrorOracle(X2, alpha[:, None, :], Fs[None, :], ells, Kc, is_T_partial=False)
    picks = [tuple(sorted(int(x) for x in rng.integers(0, M2 + 1, 2))) for _ in range(3)]
    slices = [(0, M2), (0, 1), (M2 - 1, M2)] + [p for p in picks if p[0] < p[1]]
    with _lib.RcGP(X2, y_b) as gp:
        gp.set_hyper(ells[1], Fs[1], noises[1])
        got = {1: gp.sobol_error_terms(slices), 0: gp.sobol_error_terms(slices, ells[0], Fs[0], alpha[0])}
    worst_err = 0.0
    for a in (0, 1):
        want = np.array([e.error_terms_pair(X2, a, 1, ref.g0, ref.g, ref.phi, ref.ups, ref.pre, Kc, sl) for sl in slices])
        for k in range(4):
            scale = np.abs(want[:, k]).max()
            worst_err = max(worst_err, float(np.abs(np.asarray(got[a][k]) - want[:, k]).max() / scale))
    err['err'] = worst_err
    ok &= worst_err < 1e-6
    for k, x in err.items():
        worst[k] = max(worst[k], float(x))
    failures += not ok
    print(f'case {case:3d} covariant N={N} M={M} L={L}; error terms N={N2} M={M2} slices={slices} {"ok" if ok else "FAILED " + str(err)}', flush=True)
print(f'{cases} cases, {failures} failed, worst {worst}, {time.perf_counter() - t0:.0f} s')
sys.exit(1 if failures else 0)
