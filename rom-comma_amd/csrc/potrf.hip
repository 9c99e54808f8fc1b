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

__global__ void __launch_bounds__(256) k_diag(double* __restrict__ A, int64_t ld, double* __restrict__ invL, double* __restrict__ rhs,
                                              double* __restrict__ logdiag, int* __restrict__ info, int64_t j0) {
  extern __shared__ double S[];            // [128][DS] block, then diag[128], rsd[128], rv[128]
  double* diag = S + 128 * DS;
  double* rsd = diag + 128;
  double* rv = rsd + 128;
  const int t = threadIdx.x;
  double* At = A + j0 * ld + j0;
  for (int e = t; e < 128 * 128; e += 256) {
    const int r = e >> 7, c = e & 127;
    S[r * DS + c] = (c <= r) ? At[(int64_t)r * ld + c] : 0.0;
  }
  if (t < 128) rv[t] = rhs[j0 + t];
  __syncthreads();

  // ---- Cholesky, right-looking column sweep. S[k][k] keeps the running pivot; sqrt goes to diag[].
  const int ri = t >> 1, rh = t & 1;
  for (int k = 0; k < 128; ++k) {
    double d = S[k * DS + k];
    if (!(d > 0.0)) {                       // not positive definite (or NaN): flag the leading minor, keep going finite
      if (t == 0) atomicCAS(info, 0, (int)(j0 + k + 1));
      d = 1.0;
    }
    const double rs = rc_rsqrt(d);
    if (t < 128) {
      if (t > k) S[t * DS + k] *= rs;
      else if (t == k) { diag[k] = d * rs; rsd[k] = rs; }
    }
    __syncthreads();
    if (ri > k) {
      const double lik = S[ri * DS + k];
      for (int j = k + 1 + rh; j <= ri; j += 2) S[ri * DS + j] -= lik * S[j * DS + k];
    }
    __syncthreads();
  }

  // ---- write L back (lower + diagonal; zero above) and log-diagonal
  for (int e = t; e < 128 * 128; e += 256) {
    const int r = e >> 7, c = e & 127;
    At[(int64_t)r * ld + c] = (c < r) ? S[r * DS + c] : (c == r ? diag[r] : 0.0);
  }
  if (t < 128) logdiag[j0 + t] = log(diag[t]);

  // ---- inverse X = L^-1 by forward elimination on the identity. X[i][j] (i > j) lives transposed in the upper
  //      triangle: U(j,i) = S[j*DS + i]; X[i][i] = rsd[i]. (Upper triangle was zero-filled on load.)
  for (int k = 0; k < 127; ++k) {
    if (t < k) S[t * DS + k] *= rsd[k];                     // finalise row k of X: X[k][j], j < k
    __syncthreads();
    if (ri > k) {
      const double lik = S[ri * DS + k];                     // L[i][k]
      for (int j = rh; j <= k; j += 2) {
        const double xkj = (j == k) ? rsd[k] : S[j * DS + k];
        S[j * DS + ri] -= lik * xkj;                         // R[i][j] -= L[i][k] X[k][j]
      }
    }
    __syncthreads();
  }
  if (t < 127) S[t * DS + 127] *= rsd[127];
  __syncthreads();

  // ---- outputs: invL (row-major, lower, zeros above) and w_j = X * rhs_j
  for (int e = t; e < 128 * 128; e += 256) {
    const int r = e >> 7, c = e & 127;
    invL[e] = (c < r) ? S[c * DS + r] : (c == r ? rsd[r] : 0.0);
  }
  if (t < 128) {
    double s = rsd[t] * rv[t];
    for (int j = 0; j < t; ++j) s = fma(S[j * DS + t], rv[j], s);
    rhs[j0 + t] = s;
  }
}

int rc_launch_diag(rcgp_handle_s* h, int64_t j) {
  const size_t lds = (size_t)(128 * DS + 3 * 128) * sizeof(double);
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
