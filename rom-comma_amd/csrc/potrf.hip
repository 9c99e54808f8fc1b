// Blocked right-looking Cholesky, in place on the lower triangle of A (K + noise I), fused with the forward substitution
// w = L^-1 y. Replaces tf.linalg.cholesky / triangular_solve inside GPflow's GPR.log_marginal_likelihood
// (reference call sites gpr/models.py:360, 427-439).
//
// Two-level blocking: outer panels of RC_NB_OUTER columns (K of the big MFMA trailing update), inner blocks of 128:
//   k_diag       : one workgroup factors the 128x128 diagonal block in LDS, inverts it, emits w_j and log L_ii
//   k_trsm_panel : rows below <- rows below * inv(L_jj)^T  (a GEMM, gemm.hip), rhs update fused
//   k_gemm_nt_sub: update of the remaining columns of the current outer panel (K = 128)
//   k_syrk_lower : trailing update with the whole outer panel (K = RC_NB_OUTER)
#include "common.h"

#define DS 129   // LDS row stride (doubles) of the diagonal block

// Newton-refined reciprocal square root (v_rsq_f64 seed): relative error ~1 ulp.
__device__ __forceinline__ double rc_rsqrt(double d) {
  double y = __builtin_amdgcn_rsq(d);
  y = y * __builtin_fma(-0.5 * d * y, y, 1.5);
  y = y * __builtin_fma(-0.5 * d * y, y, 1.5);
  return y;
}

// The 128x128 block is distributed over the 256 threads' registers, 8x8 elements per thread in a 16-cyclic layout
// (thread (tx, ty) owns rows ty + 16a, columns tx + 16b), so the rank-1 update of a column step is 64 independent
// register FMAs and the only communication per step is one 128-entry column through LDS and ONE barrier.
// KB (the 16-column group of the pivot) is a template parameter so that every register index is static.
template <int KB>
__device__ __forceinline__ void chol_steps(double (&R)[8][8], double* colbuf, double* rsd, double* diag, int* info, int64_t j0,
                                           const int tx, const int ty) {
#pragma unroll 1
  for (int kx = 0; kx < 16; ++kx) {
    const int k = KB * 16 + kx;
    double* buf = colbuf + (k & 1) * 128;
    if (tx == kx) {
#pragma unroll
      for (int a = 0; a < 8; ++a) buf[ty + 16 * a] = R[a][KB];
    }
    __syncthreads();
    double d = buf[k];
    if (!(d > 0.0)) {                       // not positive definite (or NaN): flag the leading minor, keep going finite
      if (threadIdx.x == 0) atomicCAS(info, 0, (int)(j0 + k + 1));
      d = 1.0;
    }
    const double rs = rc_rsqrt(d);
    if (threadIdx.x == 0) { rsd[k] = rs; diag[k] = d * rs; }
    double li[8], lj[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) li[a] = buf[ty + 16 * a] * rs;
#pragma unroll
    for (int b = KB; b < 8; ++b) lj[b] = buf[tx + 16 * b] * rs;
    if (tx == kx) {
#pragma unroll
      for (int a = 0; a < 8; ++a) R[a][KB] = li[a];          // final L[i][k] for i > k (row k itself is fixed up from diag[])
    }
#pragma unroll
    for (int b = KB; b < 8; ++b) {
      if (b == KB && tx <= kx) continue;                      // only columns j > k
#pragma unroll
      for (int a = 0; a < 8; ++a) R[a][b] = __builtin_fma(-li[a], lj[b], R[a][b]);
    }
  }
}

// Forward elimination on the identity: after step k, row k of X = L^-1 is final. Rows i > k: R[i][j] -= L[i][k] X[k][j].
template <int KB>
__device__ __forceinline__ void inv_steps(double (&R)[8][8], const double* S, double* rowbuf, const double* rsd, const int tx,
                                          const int ty) {
#pragma unroll 1
  for (int kx = 0; kx < 16; ++kx) {
    const int k = KB * 16 + kx;
    double* buf = rowbuf + (k & 1) * 128;
    if (ty == kx) {                                           // owners of row k
      const double rk = rsd[k];
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int j = tx + 16 * b;
        const double x = (j < k) ? R[KB][b] * rk : (j == k ? rk : 0.0);
        R[KB][b] = x;
        buf[j] = x;
      }
    }
    __syncthreads();
    double xj[8];
#pragma unroll
    for (int b = 0; b <= KB; ++b) xj[b] = buf[tx + 16 * b];   // X[k][j] is zero for j > k: column groups b > KB untouched
#pragma unroll
    for (int a = KB; a < 8; ++a) {
      if (a == KB && ty <= kx) continue;                      // only rows i > k
      const double lik = S[(ty + 16 * a) * DS + k];
#pragma unroll
      for (int b = 0; b <= KB; ++b) R[a][b] = __builtin_fma(-lik, xj[b], R[a][b]);
    }
  }
}

__global__ void __launch_bounds__(256) k_diag(double* __restrict__ A, int64_t ld, double* __restrict__ invL, double* __restrict__ rhs,
                                              double* __restrict__ logdiag, int* __restrict__ info, int64_t j0) {
  extern __shared__ double S[];            // [128][DS] block, then diag[128], rsd[128], rv[128], buf[2][128]
  double* diag = S + 128 * DS;
  double* rsd = diag + 128;
  double* rv = rsd + 128;
  double* buf = rv + 128;
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  double* At = A + j0 * ld + j0;
  double R[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int i = ty + 16 * a, j = tx + 16 * b;
      R[a][b] = (j <= i) ? At[(int64_t)i * ld + j] : 0.0;
    }
  if (t < 128) rv[t] = rhs[j0 + t];

  chol_steps<0>(R, buf, rsd, diag, info, j0, tx, ty);
  chol_steps<1>(R, buf, rsd, diag, info, j0, tx, ty);
  chol_steps<2>(R, buf, rsd, diag, info, j0, tx, ty);
  chol_steps<3>(R, buf, rsd, diag, info, j0, tx, ty);
  chol_steps<4>(R, buf, rsd, diag, info, j0, tx, ty);
  chol_steps<5>(R, buf, rsd, diag, info, j0, tx, ty);
  chol_steps<6>(R, buf, rsd, diag, info, j0, tx, ty);
  chol_steps<7>(R, buf, rsd, diag, info, j0, tx, ty);
  __syncthreads();

  // L back to global (lower + diagonal, zeros above), a copy into LDS for the inverse sweep, log-diagonal
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int i = ty + 16 * a, j = tx + 16 * b;
      const double v = (j < i) ? R[a][b] : (j == i ? diag[i] : 0.0);
      At[(int64_t)i * ld + j] = v;
      S[i * DS + j] = v;
      R[a][b] = (i == j) ? 1.0 : 0.0;
    }
  if (t < 128) logdiag[j0 + t] = log(diag[t]);
  __syncthreads();

  inv_steps<0>(R, S, buf, rsd, tx, ty);
  inv_steps<1>(R, S, buf, rsd, tx, ty);
  inv_steps<2>(R, S, buf, rsd, tx, ty);
  inv_steps<3>(R, S, buf, rsd, tx, ty);
  inv_steps<4>(R, S, buf, rsd, tx, ty);
  inv_steps<5>(R, S, buf, rsd, tx, ty);
  inv_steps<6>(R, S, buf, rsd, tx, ty);
  inv_steps<7>(R, S, buf, rsd, tx, ty);
  __syncthreads();

  // X = L^-1 out (row-major, zeros above the diagonal) and into LDS for w_j = X * rhs_j
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int i = ty + 16 * a, j = tx + 16 * b;
      const double v = (j <= i) ? R[a][b] : 0.0;
      invL[i * 128 + j] = v;
      S[i * DS + j] = v;
    }
  __syncthreads();
  {
    // 2 threads per row, each summing half of the (lower-triangular) row
    const int i = t >> 1, h = t & 1;
    double s = 0.0;
    for (int j = h; j <= i; j += 2) s = __builtin_fma(S[i * DS + j], rv[j], s);
    s += __shfl_xor(s, 1);
    if (h == 0) rhs[j0 + i] = s;
  }
}

int rc_launch_diag(rcgp_handle_s* h, int64_t j) {
  const size_t lds = (size_t)(128 * DS + 5 * 128) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    RC_HIP(hipFuncSetAttribute((const void*)k_diag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  RcProfScope ps(h, RC_K_DIAG, 128.0 * 128.0 * 128.0 / 3.0);
  hipLaunchKernelGGL(k_diag, dim3(1), dim3(256), lds, h->stream, h->A, h->Np, h->invdiag + (j / 128) * 128 * 128, h->w, h->logdiag, h->info,
                     j);
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
