// Blocked right-looking Cholesky, in place on the lower triangle of A (K + noise I), fused with the forward substitution
// w = L^-1 y. Replaces tf.linalg.cholesky / triangular_solve inside GPflow's GPR.log_marginal_likelihood
// (reference call sites gpr/models.py:360, 427-439).
//
// Two-level blocking: outer panels of RC_NB_OUTER columns (K of the big MFMA trailing update), inner blocks of 128:
//   k_diag       : one workgroup factors the 128x128 diagonal block in registers, inverts it, emits w_j and log L_ii
//   k_trsm_panel : rows below <- rows below * inv(L_jj)^T  (a GEMM, gemm.hip), rhs update fused
//   k_gemm_nt_sub: update of the remaining columns of the current outer panel (K = 128)
//   k_syrk_lower : trailing update with the whole outer panel (K = RC_NB_OUTER)
#include "common.h"

// Newton-refined reciprocal square root (v_rsq_f64 seed): relative error ~1 ulp.
__device__ __forceinline__ double rc_rsqrt(double d) {
  double y = __builtin_amdgcn_rsq(d);
  y = y * __builtin_fma(-0.5 * d * y, y, 1.5);
  y = y * __builtin_fma(-0.5 * d * y, y, 1.5);
  return y;
}

// The 128x128 block is distributed over the registers of 512 threads: thread (tx = t & 15, ty = t >> 4) owns rows
// ty + 32a (a < 4) and columns tx + 16b (b < 8). Both L and the working copy of X = L^-1 live in registers (32 doubles each), so
// the kernel needs only ~6 KB of LDS and can share a CU with a trailing-update workgroup (look-ahead). Per pivot the only
// communication is one 128-entry column (and, for the inverse, one row) through LDS and ONE barrier.
// KB = k >> 4 is a template parameter so that every register index is static; blocks strictly above the diagonal
// (b >= 2a + 2) are never touched.
template <int KB>
__device__ __forceinline__ void chol_steps(double (&Lr)[4][8], double* colbuf, double* rsd, int* info, int64_t j0, const int tx,
                                           const int ty) {
#pragma unroll 1
  for (int kx = 0; kx < 16; ++kx) {
    const int k = KB * 16 + kx;
    double* buf = colbuf + (k & 1) * 128;
    if (tx == kx) {
#pragma unroll
      for (int a = 0; a < 4; ++a) buf[ty + 32 * a] = Lr[a][KB];
    }
    __syncthreads();
    double d = buf[k];
    if (!(d > 0.0)) {                       // not positive definite (or NaN): flag the leading minor, keep going finite
      if (threadIdx.x == 0) atomicCAS(info, 0, (int)(j0 + k + 1));
      d = 1.0;
    }
    const double rs = rc_rsqrt(d);
    if (threadIdx.x == 0) rsd[k] = rs;
    double li[4], lj[8];
#pragma unroll
    for (int a = KB >> 1; a < 4; ++a) li[a] = buf[ty + 32 * a] * rs;
#pragma unroll
    for (int b = KB; b < 8; ++b) lj[b] = buf[tx + 16 * b] * rs;
    if (tx == kx) {
#pragma unroll
      for (int a = KB >> 1; a < 4; ++a) Lr[a][KB] = li[a];   // final L[i][k], i >= k (row k gets d*rs = sqrt(d))
    }
#pragma unroll
    for (int b = KB; b < 8; ++b) {
      if (b == KB && tx <= kx) continue;                      // only columns j > k
#pragma unroll
      for (int a = b >> 1; a < 4; ++a) Lr[a][b] = __builtin_fma(-li[a], lj[b], Lr[a][b]);
    }
  }
}

// Forward elimination on the identity: at step k row k of X = L^-1 becomes final, then rows i > k: X[i][j] -= L[i][k] X[k][j].
template <int KB>
__device__ __forceinline__ void inv_steps(double (&Xr)[4][8], const double (&Lr)[4][8], double* colbuf, double* rowbuf, const double* rsd,
                                          const int tx, const int ty) {
  constexpr int AK = KB >> 1;                                 // register row-group holding row k
#pragma unroll 1
  for (int kx = 0; kx < 16; ++kx) {
    const int k = KB * 16 + kx;
    const int kty = 16 * (KB & 1) + kx;                       // ty of the threads holding row k
    double* cb = colbuf + (k & 1) * 128;
    double* rb = rowbuf + (k & 1) * 128;
    if (ty == kty) {
      const double rk = rsd[k];
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int j = tx + 16 * b;
        const double x = (j < k) ? Xr[AK][b] * rk : (j == k ? rk : 0.0);
        Xr[AK][b] = x;
        rb[j] = x;
      }
    }
    if (tx == kx) {
#pragma unroll
      for (int a = AK; a < 4; ++a) cb[ty + 32 * a] = Lr[a][KB];
    }
    __syncthreads();
    double xj[8];
#pragma unroll
    for (int b = 0; b <= KB; ++b) xj[b] = rb[tx + 16 * b];    // X[k][j] = 0 for j > k: column groups b > KB untouched
#pragma unroll
    for (int a = AK; a < 4; ++a) {
      if (a == AK && ty <= kty) continue;                     // only rows i > k
      const double lik = cb[ty + 32 * a];
#pragma unroll
      for (int b = 0; b <= KB; ++b) Xr[a][b] = __builtin_fma(-lik, xj[b], Xr[a][b]);
    }
  }
}

__global__ void __launch_bounds__(512, 2) k_diag(double* __restrict__ A, int64_t ld, double* __restrict__ invL, double* __restrict__ rhs,
                                                 double* __restrict__ logdiag, int* __restrict__ info, int64_t j0) {
  __shared__ double colbuf[2 * 128], rowbuf[2 * 128], rsd[128], rv[128];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  double* At = A + j0 * ld + j0;
  double Lr[4][8], Xr[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int i = ty + 32 * a, j = tx + 16 * b;
      Lr[a][b] = (j <= i) ? At[(int64_t)i * ld + j] : 0.0;
      Xr[a][b] = (i == j) ? 1.0 : 0.0;
    }
  if (t < 128) rv[t] = rhs[j0 + t];

  chol_steps<0>(Lr, colbuf, rsd, info, j0, tx, ty);
  chol_steps<1>(Lr, colbuf, rsd, info, j0, tx, ty);
  chol_steps<2>(Lr, colbuf, rsd, info, j0, tx, ty);
  chol_steps<3>(Lr, colbuf, rsd, info, j0, tx, ty);
  chol_steps<4>(Lr, colbuf, rsd, info, j0, tx, ty);
  chol_steps<5>(Lr, colbuf, rsd, info, j0, tx, ty);
  chol_steps<6>(Lr, colbuf, rsd, info, j0, tx, ty);
  chol_steps<7>(Lr, colbuf, rsd, info, j0, tx, ty);
  __syncthreads();

  // L back to global (lower + diagonal, zeros above) and log-diagonal
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int i = ty + 32 * a, j = tx + 16 * b;
      At[(int64_t)i * ld + j] = (j <= i) ? Lr[a][b] : 0.0;
    }
  if (t < 128) logdiag[j0 + t] = -log(rsd[t]);              // log L_ii = log sqrt(d) = -log(1/sqrt(d))

  inv_steps<0>(Xr, Lr, colbuf, rowbuf, rsd, tx, ty);
  inv_steps<1>(Xr, Lr, colbuf, rowbuf, rsd, tx, ty);
  inv_steps<2>(Xr, Lr, colbuf, rowbuf, rsd, tx, ty);
  inv_steps<3>(Xr, Lr, colbuf, rowbuf, rsd, tx, ty);
  inv_steps<4>(Xr, Lr, colbuf, rowbuf, rsd, tx, ty);
  inv_steps<5>(Xr, Lr, colbuf, rowbuf, rsd, tx, ty);
  inv_steps<6>(Xr, Lr, colbuf, rowbuf, rsd, tx, ty);
  inv_steps<7>(Xr, Lr, colbuf, rowbuf, rsd, tx, ty);

  // X = L^-1 out (row-major, zeros above the diagonal) and w_j = X * rhs_j (16 lanes share a row)
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int i = ty + 32 * a;
    double s = 0.0;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int j = tx + 16 * b;
      const double v = (j <= i) ? Xr[a][b] : 0.0;
      invL[i * 128 + j] = v;
      s = __builtin_fma(v, rv[j], s);
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    s += __shfl_xor(s, 8);
    if (tx == 0) rhs[j0 + i] = s;
  }
}

int rc_launch_diag(rcgp_handle_s* h, int64_t j) {
  RcProfScope ps(h, RC_K_DIAG, 128.0 * 128.0 * 128.0 / 3.0);
  hipLaunchKernelGGL(k_diag, dim3(1), dim3(512), 0, h->stream, h->A, h->Np, h->invdiag + (j / 128) * 128 * 128, h->w, h->logdiag, h->info, j);
  RC_HIP(hipGetLastError());
  return 0;
}

int rc_potrf(rcgp_handle_s* h) {
  const int64_t Np = h->Np;
  int rc;
  RC_HIP(hipMemcpyAsync(h->w, h->y, (size_t)Np * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  RC_HIP(hipMemsetAsync(h->info, 0, sizeof(int), h->stream));
  for (int64_t J = 0; J < Np; J += RC_NB_OUTER) {
    const int64_t Jend = (J + RC_NB_OUTER < Np) ? J + RC_NB_OUTER : Np;
    for (int64_t j = J; j < Jend; j += 128) {
      if ((rc = rc_launch_diag(h, j))) return rc;
      const int64_t below = Np - (j + 128);
      if (below <= 0) continue;
      double* P = h->A + (j + 128) * Np + j;                       // rows below the diagonal block, 128 columns
      if ((rc = rc_launch_trsm_panel(h, P, Np, h->invdiag + (j / 128) * 128 * 128, below, h->w + j + 128, h->w + j))) return rc;
      const int64_t rest = Jend - (j + 128);                       // remaining columns inside the outer panel
      if (rest > 0) {
        double* C = h->A + (j + 128) * Np + (j + 128);
        if ((rc = rc_launch_gemm_nt_sub(h, C, Np, P, Np, P, Np, below, rest, 128, j + 128, j + 128))) return rc;
      }
    }
    const int64_t n = Np - Jend;
    if (n > 0) {
      if ((rc = rc_launch_syrk_lower(h, h->A + Jend * Np + Jend, Np, h->A + Jend * Np + J, Np, n, Jend - J))) return rc;
    }
  }
  h->factored = true;
  h->inverted = false;
  return 0;
}
