"""Generates tests/golden/*.npz from the CPU oracle (oracle/gp_oracle.py).

PROVENANCE: these vectors come from the build's own restatement, NOT from the reference: the reference cannot be run in
this image (TensorFlow/GPflow absent) and ships no fixtures for this path. They pin the oracle
against regressions and give the GPU tests inputs that do not depend on the oracle being importable -- "parity unpinned"
with respect to the reference itself.

    python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import gp_oracle as o  # noqa: E402

HERE = Path(__file__).resolve().parent


def case(N, M, seed, n_star=24):
    X, y = o.synthetic_fold(N, M, k=seed)
    rng = np.random.Generator(np.random.PCG64(1234 + seed))
    ell = rng.uniform(0.5, 3.0, M)
    var = float(rng.uniform(0.5, 2.0))
    noise = float(rng.uniform(0.5e-2, 3e-2))
    Xs, _ = o.synthetic_fold(n_star, M, k=seed + 50)
    lml, grad = o.lml_and_grad(X, y, ell, var, noise)
    alpha = o.k_inv_y(X, y, ell, var, noise)
    Lc = o.k_cho(X, ell, var, noise)
    mean_y, sd_y = o.predict(X, y, ell, var, noise, Xs, True)
    mean_f, sd_f = o.predict(X, y, ell, var, noise, Xs, False)
    slices = np.array(o.all_slices(M) + [(M, M)], dtype=np.int32)
    cal = o.ClosedSobolOracle(X, alpha[None, None, :], np.array([[var]]), ell[None, :])
    V = np.array([cal.marginalize(s)['V'][0, 0] for s in slices])
    res = {kind: o.gsa_calibrate(cal, kind, M) for kind in (o.FIRST_ORDER, o.CLOSED, o.TOTAL)}
    np.savez(HERE / f'gp_N{N}_M{M}.npz', X=X, y=y, ell=ell, var=var, noise=noise, Xs=Xs, lml=lml, grad=grad, alpha=alpha,
             K_cho_diag=np.diag(Lc).copy(), K_cho_corner=Lc[:8, :8].copy(), K_cho_checksum=float(np.sum(Lc)),
             mean_y=mean_y, sd_y=sd_y, mean_f=mean_f, sd_f=sd_f, slices=slices, V=V,
             S_first=res[o.FIRST_ORDER]['S'][0, 0], S_closed=res[o.CLOSED]['S'][0, 0], S_total=res[o.TOTAL]['S'][0, 0])


def multi_output_case(N=40, M=4, L=2):
    """Two independent outputs, literal transliteration of the TF broadcasting code (cross-output V_lj included)."""
    X, _ = o.synthetic_fold(N, M, k=3)
    rng = np.random.Generator(np.random.PCG64(99))
    ell = rng.uniform(0.5, 3.0, (L, M))
    F = rng.uniform(0.5, 2.0, L)
    noise = rng.uniform(1e-2, 2e-2, L)
    Y = np.stack([o.synthetic_fold(N, M, k=3, l=l)[1] for l in range(L)], axis=1)
    alpha = np.stack([o.k_inv_y(X, Y[:, l], ell[l], F[l], noise[l]) for l in range(L)])
    lit = o.LiteralClosedSobol(X, alpha[:, None, :], F[None, :], ell)
    slices = np.array(o.all_slices(M) + [(M, M), (1, 3)], dtype=np.int32)
    V = np.stack([lit.marginalize(s)['V'] for s in slices], axis=-1)       # (L, L, n_slices)
    np.savez(HERE / f'sobol_literal_N{N}_M{M}_L{L}.npz', X=X, Y=Y, ell=ell, F=F, noise=noise, alpha=alpha, slices=slices, V=V,
             V0=lit.V[0], S=lit.S)


def covariant_case(N=90, M=3, L=2):
    """One covariant GP (romcomma.gpf): LML, gradients, K_inv_Y, predict, gradient GP, Sobol V with the full and the diagonal F."""
    from oracle import mogp_oracle as mo
    X, _ = o.synthetic_fold(N, M, k=5)
    Y = np.stack([o.synthetic_fold(N, M, k=5, l=l)[1] for l in range(L)], axis=1)
    rng = np.random.Generator(np.random.PCG64(7))
    ell = rng.uniform(0.6, 2.5, (L, M))
    C = np.tril(rng.uniform(-0.4, 0.4, (L, L)), -1) + np.diag(rng.uniform(0.8, 1.4, L))
    Cn = np.tril(rng.uniform(-0.03, 0.03, (L, L)), -1) + np.diag(rng.uniform(0.1, 0.2, L))
    F, Sigma = C @ C.T, Cn @ Cn.T
    F, Sigma = (F + F.T) / 2, (Sigma + Sigma.T) / 2
    Xs, _ = o.synthetic_fold(12, M, k=55)
    lml, dF, dell, dS = mo.lml_and_grad(X, Y, ell, F, Sigma)
    KiY = mo.k_inv_y(X, Y, ell, F, Sigma)
    mean_y, sd_y = mo.predict(X, Y, ell, F, Sigma, Xs, True)
    mean_f, sd_f = mo.predict(X, Y, ell, F, Sigma, Xs, False)
    gmean, gvar = mo.predict_gradient(X, Y, ell, F, Sigma, Xs[:3])
    slices = np.array(o.all_slices(M) + [(M, M)], dtype=np.int32)
    V_full = mo.sobol_V_covariant(X, KiY, F, ell, slices)
    cal = o.ClosedSobolOracle(X, KiY, np.diag(F)[None, :], ell)
    V_diag = np.stack([cal.marginalize(s)['V'] for s in slices])
    Lc = mo.k_cho(X, ell, F, Sigma)
    np.savez(HERE / f'mogp_N{N}_M{M}_L{L}.npz', X=X, Y=Y, ell=ell, F=F, Sigma=Sigma, Xs=Xs, lml=lml, dF=dF, dell=dell, dSigma=dS, K_inv_Y=KiY,
             K_cho_diag=np.diag(Lc).copy(), K_cho_checksum=float(np.sum(Lc)), mean_y=mean_y, sd_y=sd_y, mean_f=mean_f, sd_f=sd_f,
             gmean=gmean, gvar=gvar, slices=slices, V_full=V_full, V_diag=V_diag)


if __name__ == '__main__':
    for N, M, seed in [(16, 1, 0), (64, 3, 1), (256, 10, 2), (300, 7, 3)]:
        case(N, M, seed)
    multi_output_case()
    covariant_case()
    print('wrote', sorted(p.name for p in HERE.glob('*.npz')))
