"""Developer check run on the GPU box: parity of every C-ABI entry point against the oracle at small sizes, then stage
timings at the BASELINE sizes. Not a test (tests/ holds those) -- a quick look while iterating on kernels.

    gpurun -- python tools/dev_check.py [--big]
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import gp_oracle as o          # noqa: E402  (checker only)
from romcomma_amd import _lib as L         # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def relmax(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def parity(N, M, seed=0):
    X, y = o.synthetic_fold(N, M, k=seed)
    ell, var, noise = o.bench_hyper(M)
    noise = 1e-2
    gp = L.RcGP(X, y)
    gp.set_hyper(ell, var, noise)
    K = gp.gram()
    Kref = o.noisy_gram(X, ell, var, noise)
    print(f'N={N} M={M}: gram        {relmax(K, Kref):.2e}')
    Lc = gp.k_cho()
    Lref = o.k_cho(X, ell, var, noise)
    print(f'             k_cho       {relmax(Lc, Lref):.2e}')
    v = gp.lml()
    vref = o.lml(X, y, ell, var, noise)
    print(f'             lml         {abs(v - vref) / abs(vref):.2e}  ({v:.6f} vs {vref:.6f})')
    v2, g = gp.lml_grad()
    vref2, gref = o.lml_and_grad(X, y, ell, var, noise)
    print(f'             lml_grad    {rel(g, gref):.2e}  lml {abs(v2 - vref2) / abs(vref2):.2e}')
    a = gp.k_inv_y()
    aref = o.k_inv_y(X, y, ell, var, noise)
    print(f'             k_inv_y     {relmax(a, aref):.2e}')
    Xs, _ = o.synthetic_fold(200, M, k=seed + 7)
    for yn in (True, False):
        m, s = gp.predict(Xs, yn)
        mr, sr = o.predict(X, y, ell, var, noise, Xs, yn)
        print(f'             predict(y={int(yn)}) mean {relmax(m, mr):.2e} sd {rel(s, sr):.2e}')
    slices = o.all_slices(M) + [(M, M)] + ([(1, M - 1)] if M > 3 else [])
    V = gp.sobol_closed(slices)
    g_, phi = o.sobol_prepare(X, aref[None, :], np.array([var]), ell[None, :])
    Vref = o.sobol_V_pair(X, g_[0], g_[0], phi[0], phi[0], slices)
    nz = np.abs(Vref) > 1e-12 * np.max(np.abs(Vref))
    print(f'             sobol       {rel(V[nz], Vref[nz]):.2e}  empty {V[~nz]} vs {Vref[~nz]}')
    # cross term against a second output
    X2, y2 = o.synthetic_fold(N, M, k=seed, l=1)
    ell2 = ell[::-1].copy()
    a2 = o.k_inv_y(X, y2, ell2, 0.7, 2e-2)
    Vx = gp.sobol_cross(ell2, 0.7, a2, slices)
    g2, phi2 = o.sobol_prepare(X, a2[None, :], np.array([0.7]), ell2[None, :])
    Vxr = o.sobol_V_pair(X, g_[0], g2[0], phi[0], phi2[0], slices)
    print(f'             sobol_cross {relmax(Vx, Vxr):.2e}')
    gp.close()


def timings(N, M, reps=3, grad=True, sobol=True):
    X, y = o.synthetic_fold(N, M)
    ell, var, noise = o.bench_hyper(M)
    gp = L.RcGP(X, y)
    gp.set_hyper(ell, var, noise)
    gp.set_profiling(True)
    for _ in range(reps):
        gp.stage_gram()
        gp.stage_potrf()
    gp.sync()
    names = L.KERNEL_CLASS_NAMES
    for c in range(len(names)):
        n, ms, work = gp.profile_get(c)
        if n:
            print(f'  [{names[c]:5s}] launches {n:6d} total {ms:9.3f} ms  work/time {work / (ms * 1e-3) / 1e12:8.3f} T(unit)/s')
    gp.profile_reset()
    gp.set_profiling(False)
    t0 = time.perf_counter()
    for _ in range(reps):
        gp.stage_gram()
    gp.sync()
    t1 = time.perf_counter()
    for _ in range(reps):
        gp.stage_gram()
        gp.stage_potrf()
    gp.sync()
    t2 = time.perf_counter()
    tg = (t1 - t0) / reps
    tp = (t2 - t1) / reps - tg
    gb = 8 * (N * (N + 1) / 2 + N * M) / 1e9
    print(f'N={N} M={M}: gram {tg * 1e3:.3f} ms = {gb / tg:.0f} GB/s ({gb / tg / 8000:.1%} of 8 TB/s);'
          f' potrf {tp * 1e3:.2f} ms = {N ** 3 / 3 / tp / 1e12:.2f} TFLOP/s ({N ** 3 / 3 / tp / 78.6e12:.1%} of 78.6)')
    print(f'             lml {gp.lml():.6f}')
    if grad:
        gp.stage_gram()                                # (set_hyper with unchanged values keeps the factor: rebuild K instead)
        t0 = time.perf_counter()
        v, g = gp.lml_grad()
        t1 = time.perf_counter()
        print(f'             lml+grad eval {1e3 * (t1 - t0):.1f} ms = {N ** 3 / (t1 - t0) / 1e12:.2f} TFLOP/s  grad[0..2]={g[:3]}')
    if sobol:
        sl = o.all_slices(M)
        gp.sobol_closed(sl[:1])
        t0 = time.perf_counter()
        V = gp.sobol_closed(sl)
        t1 = time.perf_counter()
        print(f'             sobol {1e3 * (t1 - t0):.1f} ms  S_first={V[:M] / V[-1]}')
    gp.close()


if __name__ == '__main__':
    print('devices', L.device_count())
    parity(300, 3)
    parity(1000, 7, seed=1)
    parity(256, 1, seed=2)
    timings(2048, 5)
    timings(8192, 5)
    if '--big' in sys.argv:
        timings(16384, 10)
