// Closed-form Sobol conditional variances V_lj(S) = sum_{n,n'} g_l[n] g_j[n'] prod_{m in S} h_m(n,n') for independent GPs.
// Restates gsa.calibrators.ClosedSobol._calibrate/_V/marginalize (gsa/calibrators.py:49-97) and the gsa.base.Gaussian
// arithmetic they call (gsa/base.py:92-126), reduced algebraically (DESIGN.md, section "Sobol algebra") to
//   log h_m(n,n') = c0_m + pl_m x_nm^2 + pj_m x_n'm^2 + c2_m x_nm x_n'm
//   a_m = phi_l,m phi_j,m ; c0 = -1/2 log(1-a) ; c2 = a/(1-a) ; pl = -1/2 c2 phi_l ; pj = -1/2 c2 phi_j ; phi = 1/(ell^2+1).
// The reference materialises (N,N,M) tensors per slice; here one pass over 128x128 pair tiles serves all first-order
// [m,m+1), closed [0,m+1) and total-complement [m+1,M) slices (gsa/models.py:77-90) with running sums in registers.
// VALU-bound (fp64 exp), traffic ~ 2 N M doubles per tile from L2: no HBM roofline applies.
#include "common.h"
#include "rc_math.h"
#include <algorithm>
#include <limits>

#define XST 130

// graw[i] = pre * exp(-1/2 sum_m phi_m x_im^2) * alpha_i   (gsa/calibrators.py:86-89); padded rows -> 0
__global__ void k_sobol_g0(const double* __restrict__ X, const double* __restrict__ alpha, const double* __restrict__ phi, double pre,
                           int64_t N, int64_t Np, int M, double* __restrict__ g, double* __restrict__ g0_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Np) return;
  double v = 0.0, v0 = 0.0;
  if (i < N) {
    double s = 0.0;
    for (int m = 0; m < M; ++m) {
      const double x = X[i * M + m];
      s = fma(phi[m] * x, x, s);
    }
    v0 = pre * rc_exp(-0.5 * s);
    v = v0 * alpha[i];
  }
  g[i] = v;
  if (g0_out) g0_out[i] = v0;
}

__global__ void __launch_bounds__(1024) k_sum1(const double* __restrict__ v, int64_t n, double* __restrict__ out) {
  __shared__ double sm[1024];
  double a = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) a += v[i];
  sm[threadIdx.x] = a;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sm[0];
}

// g -= mean (over the N valid rows, gsa/calibrators.py:90); out_sum[1] = sum of centred g (for the empty slice)
__global__ void k_sobol_center(double* __restrict__ g, int64_t N, const double* __restrict__ sum) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) g[i] -= sum[0] / (double)N;
}

// Pair-tile kernel. consts = [c0 | c2 | pl | pj] (4 x M). Output partial[blk][3*M]: first[m], closed[m], total[m] where
// total[m] is the slice [m+1, M) (total[M-1] = empty slice, left 0 here and filled by the caller).
// mode 0: canonical (all three kinds). mode 1: one arbitrary slice [ma, mb), result in column 0.
template <bool SYM, bool WIDE = false>
__global__ void __launch_bounds__(256) k_sobol_pairs(const double* __restrict__ X, const double* __restrict__ gl,
                                                     const double* __restrict__ gj, const double* __restrict__ consts, int M, int mode,
                                                     int ma, int mb, double* __restrict__ partial) {
  extern __shared__ double sm[];
  // The X panels of the tile pass through LDS in chunks of MC = min(M, RC_MAX_M) dimensions: ONE chunk up to M = 64 (staged once, the fast
  // path); beyond that the ascending pass stages the chunks in rising order and the descending pass in falling order -- the running
  // exponent sums live in registers, so a chunk is needed only while its dimensions are being added.
  const int MC = WIDE ? RC_MAX_M : M;
  constexpr bool chunked = WIDE;
  double* xi = sm;                     // [MC][XST]
  double* xj = sm + MC * XST;          // [MC][XST]
  double* wsum = xj + MC * XST;        // [4][3*M]
  int ti, tj;
  if (SYM) {
    int t = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((int64_t)(t + 1) * (t + 2) / 2 <= (int64_t)blockIdx.x) ++t;
    while ((int64_t)t * (t + 1) / 2 > (int64_t)blockIdx.x) --t;
    ti = t;
    tj = (int)(blockIdx.x - (int64_t)t * (t + 1) / 2);
  } else {
    tj = blockIdx.x;
    ti = blockIdx.y;
  }
  const int64_t blk = SYM ? (int64_t)blockIdx.x : (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int cb = 0;                                                 // first dimension of the chunk in LDS (stays 0 on the fast path)
  auto stage = [&](int c0) {                                  // dimensions [c0, c0 + MC) (every thread calls it at the same points of uniform loops)
    cb = c0;
    const int mc = (M - c0 < MC) ? M - c0 : MC;
    for (int e = t; e < 128 * mc; e += 256) {
      const int rr = e / mc, m = e - rr * mc;
      xi[m * XST + rr] = X[((int64_t)ti * 128 + rr) * M + c0 + m];
      xj[m * XST + rr] = X[((int64_t)tj * 128 + rr) * M + c0 + m];
    }
  };
  if constexpr (!WIDE) {
    for (int e = t; e < 128 * M; e += 256) {
      const int rr = e / M, m = e - rr * M;
      xi[m * XST + rr] = X[((int64_t)ti * 128 + rr) * M + m];
      xj[m * XST + rr] = X[((int64_t)tj * 128 + rr) * M + m];
    }
  }
  for (int e = t; e < 4 * 3 * M; e += 256) wsum[e] = 0.0;
  const int tx = t & 15, ty = t >> 4;
  double gi[8], gc[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) gi[a] = gl[(int64_t)ti * 128 + ty + 16 * a];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    gc[2 * b] = gj[(int64_t)tj * 128 + 2 * tx + 32 * b];
    gc[2 * b + 1] = gj[(int64_t)tj * 128 + 2 * tx + 32 * b + 1];
  }
  __syncthreads();
  const double* c0 = consts;
  const double* c2 = consts + M;
  const double* pl = consts + 2 * M;
  const double* pj = consts + 3 * M;
  auto wave_sum = [](double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
  };
  double e[8][8];
  // ---------------- ascending pass: first-order and closed (or the single generic slice)
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int c = 0; c < 8; ++c) e[a][c] = 0.0;
  const int m_lo = (mode == 0) ? 0 : ma, m_hi = (mode == 0) ? M : mb;
  for (int m = m_lo; m < m_hi; ++m) {
    if constexpr (chunked) {
      if (m == m_lo || m % MC == 0) {                         // the chunk that holds dimension m
        __syncthreads();
        stage((m / MC) * MC);
        __syncthreads();
      }
    }
    const double k0 = c0[m], k2 = c2[m], kl = pl[m], kj = pj[m];
    double ai[8], ui[8], bj[8], xc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const double x = xi[(m - cb) * XST + ty + 16 * a];
      ai[a] = fma(kl * x, x, k0);
      ui[a] = k2 * x;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const double2 x = *reinterpret_cast<const double2*>(xj + (m - cb) * XST + 2 * tx + 32 * b);
      xc[2 * b] = x.x;
      xc[2 * b + 1] = x.y;
      bj[2 * b] = kj * x.x * x.x;
      bj[2 * b + 1] = kj * x.y * x.y;
    }
    double sf = 0.0, sc = 0.0;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      double rf = 0.0, rc = 0.0;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const double tm = fma(ui[a], xc[c], ai[a] + bj[c]);
        e[a][c] += tm;
        if (mode == 0) {
          rf = fma(gc[c], rc_exp(tm), rf);
          rc = fma(gc[c], rc_exp(e[a][c]), rc);
        }
      }
      sf = fma(gi[a], rf, sf);
      sc = fma(gi[a], rc, sc);
    }
    if (mode == 0) {
      sf = wave_sum(sf);
      sc = wave_sum(sc);
      if (lane == 0) {
        wsum[wave * 3 * M + m] = sf;
        wsum[wave * 3 * M + M + m] = sc;
      }
    }
  }
  if (mode != 0) {
    double sc = 0.0;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      double rc = 0.0;
#pragma unroll
      for (int c = 0; c < 8; ++c) rc = fma(gc[c], rc_exp(e[a][c]), rc);
      sc = fma(gi[a], rc, sc);
    }
    sc = wave_sum(sc);
    if (lane == 0) wsum[wave * 3 * M] = sc;
  } else {
    // ---------------- descending pass: complements [m, M) for m = M-1 .. 1  -> total[m-1]
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int c = 0; c < 8; ++c) e[a][c] = 0.0;
    for (int m = M - 1; m >= 1; --m) {
      if constexpr (chunked) {
        if (m == M - 1 || m % MC == MC - 1) {
          __syncthreads();
          stage((m / MC) * MC);
          __syncthreads();
        }
      }
      const double k0 = c0[m], k2 = c2[m], kl = pl[m], kj = pj[m];
      double ai[8], ui[8], bj[8], xc[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const double x = xi[(m - cb) * XST + ty + 16 * a];
        ai[a] = fma(kl * x, x, k0);
        ui[a] = k2 * x;
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const double2 x = *reinterpret_cast<const double2*>(xj + (m - cb) * XST + 2 * tx + 32 * b);
        xc[2 * b] = x.x;
        xc[2 * b + 1] = x.y;
        bj[2 * b] = kj * x.x * x.x;
        bj[2 * b + 1] = kj * x.y * x.y;
      }
      double st = 0.0;
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        double rt = 0.0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          e[a][c] += fma(ui[a], xc[c], ai[a] + bj[c]);
          rt = fma(gc[c], rc_exp(e[a][c]), rt);
        }
        st = fma(gi[a], rt, st);
      }
      st = wave_sum(st);
      if (lane == 0) wsum[wave * 3 * M + 2 * M + (m - 1)] = st;
    }
  }
  __syncthreads();
  const double wgt = (SYM && ti != tj) ? 2.0 : 1.0;
  for (int c = t; c < 3 * M; c += 256)
    partial[blk * 3 * M + c] = wgt * ((wsum[c] + wsum[3 * M + c]) + (wsum[2 * 3 * M + c] + wsum[3 * 3 * M + c]));
}

__global__ void __launch_bounds__(256) k_rowreduce_s(const double* __restrict__ partial, int64_t rows, int cols, double* __restrict__ out) {
  __shared__ double sm[256];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int64_t r = threadIdx.x; r < rows; r += 256) s += partial[r * cols + c];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = sm[0];
}

static int sobol_make_g(rcgp_handle_s* h, const double* ell, double var, const double* alpha_d, double* phi_d, double* g_d,
                        double* sum_d, std::vector<double>& phi_host, double* g0_d = nullptr) {
  const int M = h->M;
  phi_host.resize(M);
  double pre = var;
  for (int m = 0; m < M; ++m) {
    const double l2 = ell[m] * ell[m];
    phi_host[m] = 1.0 / (l2 + 1.0);
    pre *= sqrt(l2 * phi_host[m]);                        // sqrt(prod ell^2/(ell^2+1)) * F  (gsa/calibrators.py:86)
  }
  RC_HIP(hipMemcpyAsync(phi_d, phi_host.data(), (size_t)M * sizeof(double), hipMemcpyHostToDevice, h->stream));
  RC_HIP(hipStreamSynchronize(h->stream));               // phi_host may be reused by the caller
  RcProfScope ps(h, RC_K_MISC, 0.0);
  const unsigned nb = (unsigned)((h->Nb + 255) / 256);
  hipLaunchKernelGGL(k_sobol_g0, dim3(nb), dim3(256), 0, h->stream, h->X, alpha_d, phi_d, pre, h->N, h->Nb, M, g_d, g0_d);
  hipLaunchKernelGGL(k_sum1, dim3(1), dim3(1024), 0, h->stream, g_d, h->N, sum_d);
  hipLaunchKernelGGL(k_sobol_center, dim3(nb), dim3(256), 0, h->stream, g_d, h->N, sum_d);
  hipLaunchKernelGGL(k_sum1, dim3(1), dim3(1024), 0, h->stream, g_d, h->N, sum_d + 1);
  RC_HIP(hipGetLastError());
  return 0;
}

// The pair quadratic forms V[s] = sum_{n,n'} g_l[n] g_j[n'] prod_{m in slice s} h_m(n,n') for weight vectors already on the device
// (Np_x rows, zero beyond the N valid ones) with their phi vectors on the host. sym: g_j is g_l, only the lower pair tiles run.
static int sobol_run_slices(rcgp_handle_s* h, int64_t Np, const double* g_l, const double* g_j, const std::vector<double>& phi_l,
                            const std::vector<double>& phi_j, bool sym, double empty_value, double* consts_d, double* out_d, int n_slices,
                            const int32_t* slices, double* V_host) {
  const int M = h->M;
  const int64_t T = Np / 128;
  int rc;
  std::vector<double> consts(4 * M);
  for (int m = 0; m < M; ++m) {
    const double a = phi_l[m] * phi_j[m];
    consts[m] = -0.5 * log1p(-a);
    consts[M + m] = a / (1.0 - a);
    consts[2 * M + m] = -0.5 * consts[M + m] * phi_l[m];
    consts[3 * M + m] = -0.5 * consts[M + m] * phi_j[m];
  }
  RC_HIP(hipMemcpyAsync(consts_d, consts.data(), (size_t)4 * M * sizeof(double), hipMemcpyHostToDevice, h->stream));
  RC_HIP(hipStreamSynchronize(h->stream));

  const int64_t nblk = sym ? T * (T + 1) / 2 : T * T;
  rc = rc_ensure_partial(h, (size_t)nblk * 3 * M);
  if (rc) return rc;
  const size_t lds = (size_t)(2 * (M < RC_MAX_M ? M : RC_MAX_M) * XST + 4 * 3 * M) * sizeof(double);      // (panels in chunks of <= 64 dimensions)
  const bool wide = M > RC_MAX_M;
  RC_HIP(hipFuncSetAttribute(wide ? (const void*)k_sobol_pairs<true, true> : (const void*)k_sobol_pairs<true, false>,
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  RC_HIP(hipFuncSetAttribute(wide ? (const void*)k_sobol_pairs<false, true> : (const void*)k_sobol_pairs<false, false>,
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const double pairs = sym ? (double)h->N * ((double)h->N + 1.0) / 2.0 : (double)h->N * (double)h->N;

  auto run = [&](int mode, int ma, int mb, std::vector<double>& out) -> int {
    {
      RcProfScope ps(h, RC_K_SOBOL, pairs * (mode == 0 ? (double)(3 * M - 1) : 1.0));     // exp evaluations
      if (sym && !wide)
        hipLaunchKernelGGL((k_sobol_pairs<true, false>), dim3((unsigned)nblk), dim3(256), lds, h->stream, h->X, g_l, g_j, consts_d, M, mode, ma, mb,
                           h->partial);
      else if (sym)
        hipLaunchKernelGGL((k_sobol_pairs<true, true>), dim3((unsigned)nblk), dim3(256), lds, h->stream, h->X, g_l, g_j, consts_d, M, mode, ma, mb,
                           h->partial);
      else if (!wide)
        hipLaunchKernelGGL((k_sobol_pairs<false, false>), dim3((unsigned)T, (unsigned)T), dim3(256), lds, h->stream, h->X, g_l, g_j, consts_d, M,
                           mode, ma, mb, h->partial);
      else
        hipLaunchKernelGGL((k_sobol_pairs<false, true>), dim3((unsigned)T, (unsigned)T), dim3(256), lds, h->stream, h->X, g_l, g_j, consts_d, M,
                           mode, ma, mb, h->partial);
      RC_HIP(hipGetLastError());
    }
    {
      RcProfScope ps(h, RC_K_MISC, 0.0);
      hipLaunchKernelGGL(k_rowreduce_s, dim3((unsigned)(3 * M)), dim3(256), 0, h->stream, h->partial, nblk, 3 * M, out_d);
      RC_HIP(hipGetLastError());
    }
    out.resize(3 * M);
    RC_HIP(hipMemcpyAsync(out.data(), out_d, (size_t)3 * M * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    RC_HIP(hipStreamSynchronize(h->stream));
    return 0;
  };

  std::vector<double> canon;
  bool have_canon = false;
  for (int s = 0; s < n_slices; ++s) {
    const int a = slices[2 * s], b = slices[2 * s + 1];
    if (a < 0 || b > M || a > b) { h->err = "sobol: bad slice"; return -2; }
    if (a == b) { V_host[s] = empty_value; continue; }
    const bool canonical = (b == a + 1) || (a == 0) || (b == M);
    if (canonical) {
      if (!have_canon) {
        if ((rc = run(0, 0, M, canon))) return rc;
        have_canon = true;
      }
      if (a == 0) V_host[s] = canon[M + (b - 1)];              // closed [0,b)
      else if (b == M) V_host[s] = canon[2 * M + (a - 1)];      // complement [a,M)
      else V_host[s] = canon[a];                                // first-order [a,a+1)
    } else {
      std::vector<double> one;
      if ((rc = run(1, a, b, one))) return rc;
      V_host[s] = one[0];
    }
  }
  return 0;
}

// scratch layout in h->sob: g_l[Np] g_j[Np] alpha_j[Np] phi_l[M] phi_j[M] consts[4M] sums[4] out[3M]
static int sobol_scratch(rcgp_handle_s* h, int64_t Np) {
  const int M = h->M;
  const size_t need = (size_t)3 * Np + 2 * M + 4 * M + 4 + 3 * M;
  if (h->sob_elems < need) {
    if (h->sob) { RC_HIP(hipStreamSynchronize(h->stream)); RC_HIP(hipFree(h->sob)); h->sob = nullptr; }
    RC_HIP(hipMalloc(&h->sob, need * sizeof(double)));
    h->sob_elems = need;
  }
  return 0;
}

int rc_sobol(rcgp_handle_s* h, const double* ell_j, double var_j, const double* alpha_j_host, int n_slices, const int32_t* slices,
             double* V_host) {
  const int M = h->M;
  const int64_t Np = h->Np;
  const bool sym = (ell_j == nullptr);
  int rc = sobol_scratch(h, Np);
  if (rc) return rc;
  double* g_l = h->sob;
  double* g_j = g_l + Np;
  double* a_j = g_j + Np;
  double* phi_l_d = a_j + Np;
  double* phi_j_d = phi_l_d + M;
  double* consts_d = phi_j_d + M;
  double* sums_d = consts_d + 4 * M;
  double* out_d = sums_d + 4;
  std::vector<double> phi_l, phi_j;
  rc = sobol_make_g(h, h->ell.data(), h->var, h->alpha, phi_l_d, g_l, sums_d, phi_l);
  if (rc) return rc;
  if (!sym) {
    RC_HIP(hipMemsetAsync(a_j, 0, (size_t)Np * sizeof(double), h->stream));
    RC_HIP(hipMemcpyAsync(a_j, alpha_j_host, (size_t)h->N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    rc = sobol_make_g(h, ell_j, var_j, a_j, phi_j_d, g_j, sums_d + 2, phi_j);
    if (rc) return rc;
  } else {
    phi_j = phi_l;
    g_j = g_l;
  }
  RC_HIP(hipStreamSynchronize(h->stream));
  double sums[4];
  RC_HIP(hipMemcpy(sums, sums_d, 4 * sizeof(double), hipMemcpyDeviceToHost));
  const double empty_value = sym ? sums[1] * sums[1] : sums[1] * sums[3];
  return sobol_run_slices(h, Np, g_l, g_j, phi_l, phi_j, sym, empty_value, consts_d, out_d, n_slices, slices, V_host);
}

// g[n] -= shift on the valid rows
__global__ void k_sobol_shift(double* __restrict__ g, int64_t N, double shift) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) g[i] -= shift;
}

// Weight vector of one "virtual output" (phi, pre, alpha) on the device: g[n] = pre exp(-1/2 sum_m phi_m x_nm^2) alpha[n] - shift;
// sums_d[0] = sum of g before the shift, sums_d[1] = after. The covariant Sobol calculation (gsa/calibrators.py:82-92 with a
// non-diagonal F) builds its (l, J) weight vectors this way: phi = 1/(ell_l ell_J + 1), pre = F_lJ sqrt(prod ell_l ell_J phi),
// alpha = K_inv_Y[J], shift = the mean over (J, N) for that l.
static int sobol_make_weights(rcgp_handle_s* h, int64_t Np, const double* phi, double pre, const double* alpha_host, double shift,
                              double* stage_d, double* phi_d, double* g_d, double* sums_d) {
  const int M = h->M;
  RC_HIP(hipMemsetAsync(stage_d, 0, (size_t)Np * sizeof(double), h->stream));
  RC_HIP(hipMemcpyAsync(stage_d, alpha_host, (size_t)h->N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  RC_HIP(hipMemcpyAsync(phi_d, phi, (size_t)M * sizeof(double), hipMemcpyHostToDevice, h->stream));
  RcProfScope ps(h, RC_K_MISC, 0.0);
  const unsigned nb = (unsigned)((Np + 255) / 256);
  hipLaunchKernelGGL(k_sobol_g0, dim3(nb), dim3(256), 0, h->stream, h->X, stage_d, phi_d, pre, h->N, Np, M, g_d, (double*)nullptr);
  hipLaunchKernelGGL(k_sum1, dim3(1), dim3(1024), 0, h->stream, g_d, h->N, sums_d);
  hipLaunchKernelGGL(k_sobol_shift, dim3(nb), dim3(256), 0, h->stream, g_d, h->N, shift);
  hipLaunchKernelGGL(k_sum1, dim3(1), dim3(1024), 0, h->stream, g_d, h->N, sums_d + 1);
  RC_HIP(hipGetLastError());
  RC_HIP(hipStreamSynchronize(h->stream));                 // alpha_host and phi may be reused by the caller
  return 0;
}

int rc_sobol_weight_sum(rcgp_handle_s* h, const double* phi, double pre, const double* alpha_host, double* sum) {
  const int M = h->M;
  const int64_t Np = h->Nb;                                // rows of one output block: X's first block is the N x M design matrix
  int rc = sobol_scratch(h, Np);
  if (rc) return rc;
  double* g_l = h->sob;
  double* a_j = g_l + 2 * Np;
  double* phi_l_d = a_j + Np;
  double* sums_d = phi_l_d + 2 * M + 4 * M;
  if ((rc = sobol_make_weights(h, Np, phi, pre, alpha_host, 0.0, a_j, phi_l_d, g_l, sums_d))) return rc;
  RC_HIP(hipMemcpy(sum, sums_d, sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int rc_sobol_pair(rcgp_handle_s* h, const double* phi_a, double pre_a, const double* alpha_a, double shift_a, const double* phi_b,
                  double pre_b, const double* alpha_b, double shift_b, int n_slices, const int32_t* slices, double* V_host) {
  const int M = h->M;
  const int64_t Np = h->Nb;
  int rc = sobol_scratch(h, Np);
  if (rc) return rc;
  double* g_l = h->sob;
  double* g_j = g_l + Np;
  double* a_j = g_j + Np;
  double* phi_l_d = a_j + Np;
  double* phi_j_d = phi_l_d + M;
  double* consts_d = phi_j_d + M;
  double* sums_d = consts_d + 4 * M;
  double* out_d = sums_d + 4;
  if ((rc = sobol_make_weights(h, Np, phi_a, pre_a, alpha_a, shift_a, a_j, phi_l_d, g_l, sums_d))) return rc;
  if ((rc = sobol_make_weights(h, Np, phi_b, pre_b, alpha_b, shift_b, a_j, phi_j_d, g_j, sums_d + 2))) return rc;
  double sums[4];
  RC_HIP(hipMemcpy(sums, sums_d, 4 * sizeof(double), hipMemcpyDeviceToHost));
  const std::vector<double> pa(phi_a, phi_a + M), pb(phi_b, phi_b + M);
  return sobol_run_slices(h, Np, g_l, g_j, pa, pb, false, sums[1] * sums[3], consts_d, out_d, n_slices, slices, V_host);
}

// =====================================================================================================================
// Standard errors of the Sobol indices (reference gsa/calibrators.py:146-402, ClosedSobolWithError). DESIGN.md
// "Sobol error algebra": for an output pair (a, b) every ingredient is either
//   (1) a pair quadratic form  sum_{N,n} gA[N] gB[n] exp(sum_{m in S} c0 + cN x_N^2 + cn x_n^2 + cx x_N x_n)   -> k_sobol_pairs
//   (2) a pair matrix-vector   u_S[n] = sum_N g_a[N] H^S_ab(N,n)                                                   -> k_sobol_matvec
//   (3) |L_b^-1 f|^2 for f_S = g0_b * u_S                                                                          -> k_predict_var
// with per-dimension coefficients computed on the host.
// =====================================================================================================================

// u partials: partial[(ti * nS + s) * Np + n] = sum over the rows of row-tile ti of gl[N] * prod_{m in slice s} h_m(N, n). Same
// tile/thread layout as k_sobol_pairs.
// mode 0: the canonical slices (index m: first-order [m,m+1); M + m: closed [0,m+1); 2M + m: complement [m+1,M)) with indices in the
//         window [s_lo, s_hi), nS = s_hi - s_lo: all 3M in one pass when the column accumulators fit in LDS (M <= 29), otherwise the host
//         walks the window over the 3M indices (the running exponent sums are rebuilt every pass, exponentials only where wanted);
// mode 1: ONE arbitrary slice [ma, mb) (ClosedSobolWithError.marginalize takes any, gsa/calibrators.py:348-373), nS = 1.
// WIDE (M > RC_MAX_M = 64): the X panels pass through LDS in chunks of 64 dimensions as in k_sobol_pairs<., WIDE> (the running exponent sums
// live in registers; a chunk is staged when the walk enters it). The fast instantiation is the kernel as it was.
template <bool WIDE = false>
__global__ void __launch_bounds__(256) k_sobol_matvec(const double* __restrict__ X, const double* __restrict__ gl,
                                                      const double* __restrict__ consts, int M, int64_t Np, double* __restrict__ partial,
                                                      int mode, int ma, int mb, int s_lo, int s_hi) {
  extern __shared__ double sm[];
  const int MC = WIDE ? RC_MAX_M : M;
  double* xi = sm;                        // [MC][XST] rows (N side)
  double* xj = sm + MC * XST;             // [MC][XST] columns (n side)
  double* slots = xj + MC * XST;          // [4 waves][2 kinds][128]
  double* colacc = slots + 4 * 2 * 128;   // [nS][128]
  const int nS = s_hi - s_lo;
  const int tj = blockIdx.x, ti = blockIdx.y;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int cb = 0;                             // first dimension of the chunk in LDS (stays 0 on the fast path)
  bool staged = !WIDE;
  auto enter = [&](int m) {               // WIDE: make the chunk of dimension m resident (every thread, at the same points of uniform loops)
    if constexpr (WIDE) {
      const int c0 = (m / MC) * MC;
      if (staged && c0 == cb) return;
      __syncthreads();                    // (everybody has read the chunk staged before)
      cb = c0;
      staged = true;
      const int mc = (M - c0 < MC) ? M - c0 : MC;
      for (int e = t; e < 128 * mc; e += 256) {
        const int rr = e / mc, mm = e - rr * mc;
        xi[mm * XST + rr] = X[((int64_t)ti * 128 + rr) * M + c0 + mm];
        xj[mm * XST + rr] = X[((int64_t)tj * 128 + rr) * M + c0 + mm];
      }
      __syncthreads();
    }
  };
  if constexpr (!WIDE) {
    for (int e = t; e < 128 * M; e += 256) {
      const int rr = e / M, m = e - rr * M;
      xi[m * XST + rr] = X[((int64_t)ti * 128 + rr) * M + m];
      xj[m * XST + rr] = X[((int64_t)tj * 128 + rr) * M + m];
    }
  }
  for (int e = t; e < nS * 128; e += 256) colacc[e] = 0.0;
  const int tx = t & 15, ty = t >> 4;
  double gi[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) gi[a] = gl[(int64_t)ti * 128 + ty + 16 * a];
  __syncthreads();
  const double* c0 = consts;
  const double* c2 = consts + M;
  const double* pl = consts + 2 * M;
  const double* pj = consts + 3 * M;
  double e[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int c = 0; c < 8; ++c) e[a][c] = 0.0;

  // per-column partial sums over this thread's rows -> column accumulators idx0 (kind 0, if want0) and idx1 (kind 1, if want1);
  // want0 / want1 are uniform over the workgroup
  auto publish = [&](const double (&cf)[8], const double (&cc)[8], bool want0, bool want1, int idx0, int idx1) {
    if (!want0 && !want1) return;
    // reduce over the 4 row-groups of the wave (lanes differing in lane >> 4), then one slot per wave
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      double vf = cf[c], vc = cc[c];
      vf += __shfl_xor(vf, 16);
      vf += __shfl_xor(vf, 32);
      vc += __shfl_xor(vc, 16);
      vc += __shfl_xor(vc, 32);
      if ((lane >> 4) == 0) {
        const int col = 2 * tx + 32 * (c >> 1) + (c & 1);
        slots[(wave * 2 + 0) * 128 + col] = vf;
        slots[(wave * 2 + 1) * 128 + col] = vc;
      }
    }
    __syncthreads();
    if (t < 256) {
      const int kind = t >> 7, col = t & 127;
      if (kind == 0 ? want0 : want1) {
        const double sum = (slots[(0 * 2 + kind) * 128 + col] + slots[(1 * 2 + kind) * 128 + col]) +
                           (slots[(2 * 2 + kind) * 128 + col] + slots[(3 * 2 + kind) * 128 + col]);
        colacc[(kind == 0 ? idx0 : idx1) * 128 + col] += sum;          // a single owner thread per (slice, column): fixed order
      }
    }
    __syncthreads();
  };
  auto in_window = [&](int idx) { return idx >= s_lo && idx < s_hi; };

  // one m-step of the running exponent sums; cf / cc receive sum_rows gi exp(t_m) / gi exp(e) where wanted
  auto step = [&](int m, bool want_f, bool want_c, double (&cf)[8], double (&cc)[8]) {
    enter(m);
    const double k0 = c0[m], k2 = c2[m], kl = pl[m], kj = pj[m];
    double ai[8], ui[8], bj[8], xc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const double x = xi[(m - cb) * XST + ty + 16 * a];
      ai[a] = fma(kl * x, x, k0);
      ui[a] = k2 * x;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const double2 x = *reinterpret_cast<const double2*>(xj + (m - cb) * XST + 2 * tx + 32 * b);
      xc[2 * b] = x.x;
      xc[2 * b + 1] = x.y;
      bj[2 * b] = kj * x.x * x.x;
      bj[2 * b + 1] = kj * x.y * x.y;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) { cf[c] = 0.0; cc[c] = 0.0; }
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const double tm = fma(ui[a], xc[c], ai[a] + bj[c]);
        e[a][c] += tm;
        if (want_f) cf[c] = fma(gi[a], rc_exp(tm), cf[c]);
        if (want_c) cc[c] = fma(gi[a], rc_exp(e[a][c]), cc[c]);
      }
  };

  double cf[8], cc[8];
  if (mode != 0) {                                                      // one generic slice: the exponent sum over [ma, mb), one exp
    for (int m = ma; m < mb; ++m) step(m, false, m == mb - 1, cf, cc);
    publish(cc, cc, true, false, 0, 0);
  } else {
    int last_asc = -1;                                                  // no need to walk past the last wanted ascending index
    for (int m = 0; m < M; ++m)
      if (in_window(m) || in_window(M + m)) last_asc = m;
    for (int m = 0; m <= last_asc; ++m) {                               // ascending: first-order [m,m+1) and closed [0,m+1)
      const bool wf = in_window(m), wc = in_window(M + m);
      step(m, wf, wc, cf, cc);
      publish(cf, cc, wf, wc, m - s_lo, M + m - s_lo);
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int c = 0; c < 8; ++c) e[a][c] = 0.0;
    int last_desc = M;                                                  // smallest m whose complement index is wanted
    for (int m = M - 1; m >= 1; --m)
      if (in_window(2 * M + m - 1)) last_desc = m;
    for (int m = M - 1; m >= last_desc; --m) {                          // descending: complements [m, M) -> index 2M + m - 1
      const bool wt = in_window(2 * M + m - 1);
      step(m, false, wt, cf, cc);
      publish(cc, cc, wt, false, 2 * M + m - 1 - s_lo, 0);
    }
  }
  for (int idx = t; idx < nS * 128; idx += 256) {
    const int sidx = idx >> 7, col = idx & 127;
    partial[((int64_t)ti * nS + sidx) * Np + (int64_t)tj * 128 + col] = colacc[idx];
  }
}

__global__ void k_reduce_rows(const double* __restrict__ partial, int64_t rows, int64_t n, double* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double s = 0.0;
  for (int64_t r = 0; r < rows; ++r) s += partial[r * n + j];
  out[j] = s;
}

// g~[N] = g[N] * exp(sum_m cu_m x_Nm^2 + c)   (the full-model Upsilon factor of the MIXED rank equation)
__global__ void k_sobol_gtilde(const double* __restrict__ X, const double* __restrict__ g, const double* __restrict__ cu, double cadd,
                               int64_t N, int64_t Np, int M, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Np) return;
  double v = 0.0;
  if (i < N) {
    double s = cadd;
    for (int m = 0; m < M; ++m) {
      const double x = X[i * M + m];
      s = fma(cu[m] * x, x, s);
    }
    v = g[i] * rc_exp(s);
  }
  out[i] = v;
}

// F rows for the psi terms (row-major [rows][Np] into KsT) from R vectors u_r: r < R: g0 * u_r ; R <= r < 2R: g0 * (u_{r-R} + u_full) ;
// r == 2R: g0 * u_full, where u_full is the (b,b) full-model vector ; zero beyond.
// F has leading dimension ldf and the row lands at column offset off: a covariant GP embeds the vector in its output block of the
// (L N) system, zeros elsewhere (the set_diag / reshape of gsa/calibrators.py:304-306 for a rank-2 K_cho).
__global__ void k_sobol_psi_rows(const double* __restrict__ U, const double* __restrict__ Ufull, const double* __restrict__ g0,
                                 int R, int64_t Np, int64_t rows_padded, double* __restrict__ F, int64_t ldf, int64_t off) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (n >= Np) return;
  double v = 0.0;
  if (r < R) v = g0[n] * U[(int64_t)r * Np + n];
  else if (r < 2 * R) v = g0[n] * (U[(int64_t)(r - R) * Np + n] + Ufull[n]);
  else if (r == 2 * R) v = g0[n] * Ufull[n];
  F[(int64_t)r * ldf + off + n] = v;
}

// out_b < 0: the handle's single output is b, and a is b itself (ell_a == nullptr) or described by (ell_a, var_a, alpha_a_host).
// out_b >= 0 (covariant handle, diagonal F): a = output block out_a, b = output block out_b of the handle, each with its own
// lengthscale row, F[l][l] and its N entries of the joint K_inv_Y; the psi vectors are embedded in block b of the (L N) system.
int rc_sobol_error_terms(rcgp_handle_s* h, const double* ell_a, double var_a, const double* alpha_a_host, int n_slices,
                         const int32_t* slices, double* phi_d_out, double* psi_d_out, double* phi_m_out, double* psi_m_out, int out_a,
                         int out_b) {
  const int M = h->M;
  const int64_t Np = h->Nb, T = Np / 128;                        // rows of one output block: the X-side kernels work on those
  const bool mo = (out_b >= 0);
  const bool self = mo ? (out_a == out_b) : (ell_a == nullptr);
  const double* ell_b = h->ell.data() + (mo ? (size_t)out_b * M : 0);
  const double var_b = mo ? h->Fm[(size_t)out_b * h->L + out_b] : h->var;
  const double* alpha_b_d = h->alpha + (mo ? (size_t)out_b * h->Nb : 0);
  if (mo) {
    ell_a = h->ell.data() + (size_t)out_a * M;
    var_a = h->Fm[(size_t)out_a * h->L + out_a];
  }
  auto is_canonical = [M](int a, int b) { return a == b || b == a + 1 || a == 0 || b == M; };
  for (int s = 0; s < n_slices; ++s) {
    const int a = slices[2 * s], b = slices[2 * s + 1];
    if (a < 0 || b > M || a > b) { h->err = "sobol errors: bad slice"; return -2; }
  }
  int rc;
  if ((rc = rc_ensure_pred(h))) return rc;
  // scratch: g_b g0_b g_a alpha_a g~_b [5 Np] | U [3M Np] | Ubb_full... we keep a second U for the (b,b) pass when a != b
  const size_t need = (size_t)5 * Np + (size_t)2 * 3 * M * Np + 16 * M + 16;
  if (h->sob_elems < need) {
    if (h->sob) { RC_HIP(hipStreamSynchronize(h->stream)); RC_HIP(hipFree(h->sob)); h->sob = nullptr; }
    RC_HIP(hipMalloc(&h->sob, need * sizeof(double)));
    h->sob_elems = need;
  }
  double* g_b = h->sob;
  double* g0_b = g_b + Np;
  double* g_a = g0_b + Np;
  double* al_a = g_a + Np;
  double* gt_b = al_a + Np;
  double* U = gt_b + Np;
  double* Ubb = U + (size_t)3 * M * Np;
  double* small = Ubb + (size_t)3 * M * Np;            // phi_b[M] phi_a[M] consts[4M] cu[M] sums[8]
  double* phi_b_d = small;
  double* phi_a_d = small + M;
  double* consts_d = small + 2 * M;
  double* cu_d = small + 6 * M;
  double* sums_d = small + 7 * M;
  std::vector<double> phi_b, phi_a;
  if ((rc = sobol_make_g(h, ell_b, var_b, alpha_b_d, phi_b_d, g_b, sums_d, phi_b, g0_b))) return rc;
  if (!self) {
    if (mo) {
      RC_HIP(hipMemcpyAsync(al_a, h->alpha + (size_t)out_a * h->Nb, (size_t)Np * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    } else {
      RC_HIP(hipMemsetAsync(al_a, 0, (size_t)Np * sizeof(double), h->stream));
      RC_HIP(hipMemcpyAsync(al_a, alpha_a_host, (size_t)h->N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    if ((rc = sobol_make_g(h, ell_a, var_a, al_a, phi_a_d, g_a, sums_d + 2, phi_a))) return rc;
  } else {
    phi_a = phi_b;
    g_a = g_b;
  }
  std::vector<double> ups_b(M);
  double pre_b = var_b;
  for (int m = 0; m < M; ++m) {
    const double l2 = ell_b[m] * ell_b[m];
    ups_b[m] = 1.0 / (l2 + 2.0);
    pre_b *= sqrt(l2 * ups_b[m]);                                  // gsa/calibrators.py:384
  }
  // host coefficient sets, kernel order [c0 | c2 (cross) | pl (row side, x_N^2) | pj (column side, x_n^2)]
  auto coeffs = [&](char kind, const std::vector<double>& pa, const std::vector<double>& pb, std::vector<double>& out) {
    out.resize(4 * M);
    for (int m = 0; m < M; ++m) {
      const double fa = pa[m], fb = pb[m], ub = ups_b[m], ga = 1.0 - fa, gb = 1.0 - fb;
      double k0, kN, kn, kx;
      if (kind == 'H') {
        const double aa = fa * fb, c2 = aa / (1.0 - aa);
        k0 = -0.5 * log1p(-aa); kN = -0.5 * c2 * fa; kn = -0.5 * c2 * fb; kx = c2;
      } else {
        const double Pi = 1.0 / (1.0 + fb + fb * fb / gb);
        const double B = ga * fa + fa * fa * Pi;
        const double Om = fa * Pi * fb / gb;
        if (kind == 'D') {
          const double C = ga * (1.0 - ub) / (1.0 - fa * ub), mu = Om * C * fa / ga, Var = B + Om * Om * C, den = 1.0 - ub * fa;
          k0 = 0.5 * log(fa / Var) - 0.5 * log(den);
          kN = -0.5 * mu * mu / Var - 0.5 * ub * fa * fa / den;
          kn = -0.5 * fa * fa / Var + 0.5 * fa;
          kx = mu * fa / Var;
        } else {
          const double C = gb * (1.0 - ub) / (1.0 - fb * ub), mu = Om * C * fb / gb, Var = B + Om * Om * C;
          k0 = 0.5 * log(fa / Var);
          kN = -0.5 * mu * mu / Var;
          kn = -0.5 * fa * fa / Var + 0.5 * fa;
          kx = mu * fa / Var;
        }
      }
      out[m] = k0; out[M + m] = kx; out[2 * M + m] = kN; out[3 * M + m] = kn;
    }
  };
  const int64_t nblk = T * T;
  const bool wide = M > RC_MAX_M;                                 // X panels in chunks of 64 dimensions (k_sobol_pairs / k_sobol_matvec <., WIDE>)
  const int Mc = wide ? RC_MAX_M : M;
  const size_t lds_pairs = (size_t)(2 * Mc * XST + 4 * 3 * M) * sizeof(double);
  // canonical slices per k_sobol_matvec pass: as many column accumulators (1 KB each) as fit beside the two X panels in 160 KB of LDS
  // -- all 3M for M <= 29, 18 at M = 64
  const size_t lds_mv_fixed = (size_t)(2 * Mc * XST + 4 * 2 * 128) * sizeof(double);
  const int mv_window = (int)std::min<size_t>((size_t)3 * M, (160 * 1024 - 512 - lds_mv_fixed) / (128 * sizeof(double)));
  const size_t lds_mv = lds_mv_fixed + (size_t)mv_window * 128 * sizeof(double);
  const auto pairs_kernel = wide ? k_sobol_pairs<false, true> : k_sobol_pairs<false, false>;
  const auto matvec_kernel = wide ? k_sobol_matvec<true> : k_sobol_matvec<false>;
  RC_HIP(hipFuncSetAttribute((const void*)pairs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pairs));
  RC_HIP(hipFuncSetAttribute((const void*)matvec_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mv));
  const double pairs = (double)h->N * (double)h->N;
  double* out_d = sums_d + 8;                                   // not used (reductions write into small scratch below)
  (void)out_d;

  auto pair_pass = [&](const std::vector<double>& c, const double* gl, const double* gj, std::vector<double>& out) -> int {
    int r = rc_ensure_partial(h, (size_t)nblk * 3 * M);
    if (r) return r;
    RC_HIP(hipMemcpyAsync(consts_d, c.data(), (size_t)4 * M * sizeof(double), hipMemcpyHostToDevice, h->stream));
    {
      RcProfScope ps(h, RC_K_SOBOL, pairs * (double)(3 * M - 1));
      hipLaunchKernelGGL(pairs_kernel, dim3((unsigned)T, (unsigned)T), dim3(256), lds_pairs, h->stream, h->X, gl, gj, consts_d, M, 0, 0,
                         M, h->partial);
      RC_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_rowreduce_s, dim3((unsigned)(3 * M)), dim3(256), 0, h->stream, h->partial, nblk, 3 * M, U);   // U reused as tiny out
    RC_HIP(hipGetLastError());
    out.resize(3 * M);
    RC_HIP(hipMemcpyAsync(out.data(), U, (size_t)3 * M * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    RC_HIP(hipStreamSynchronize(h->stream));
    return 0;
  };
  // one arbitrary slice [ma, mb) of the same quadratic form (k_sobol_pairs mode 1)
  auto pair_pass_one = [&](const std::vector<double>& c, const double* gl, const double* gj, int ma, int mb, double* value) -> int {
    int r = rc_ensure_partial(h, (size_t)nblk * 3 * M);
    if (r) return r;
    RC_HIP(hipMemcpyAsync(consts_d, c.data(), (size_t)4 * M * sizeof(double), hipMemcpyHostToDevice, h->stream));
    {
      RcProfScope ps(h, RC_K_SOBOL, pairs);
      hipLaunchKernelGGL(pairs_kernel, dim3((unsigned)T, (unsigned)T), dim3(256), lds_pairs, h->stream, h->X, gl, gj, consts_d, M, 1, ma,
                         mb, h->partial);
      RC_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_rowreduce_s, dim3(1), dim3(256), 0, h->stream, h->partial, nblk, 3 * M, sums_d + 8);
    RC_HIP(hipGetLastError());
    RC_HIP(hipMemcpyAsync(value, sums_d + 8, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    RC_HIP(hipStreamSynchronize(h->stream));
    return 0;
  };
  // u_S for all canonical slices -> Uout[3M][Np], the window of column accumulators walked over the 3M slice indices
  auto matvec_pass = [&](const std::vector<double>& c, const double* gl, double* Uout) -> int {
    int r = rc_ensure_partial(h, (size_t)T * mv_window * Np);
    if (r) return r;
    RC_HIP(hipMemcpyAsync(consts_d, c.data(), (size_t)4 * M * sizeof(double), hipMemcpyHostToDevice, h->stream));
    for (int s_lo = 0; s_lo < 3 * M; s_lo += mv_window) {
      const int s_hi = std::min(3 * M, s_lo + mv_window), nS = s_hi - s_lo;
      {
        RcProfScope ps(h, RC_K_SOBOL, pairs * (double)nS);
        hipLaunchKernelGGL(matvec_kernel, dim3((unsigned)T, (unsigned)T), dim3(256), lds_mv, h->stream, h->X, gl, consts_d, M, Np, h->partial, 0, 0,
                           M, s_lo, s_hi);
        RC_HIP(hipGetLastError());
      }
      const int64_t n = (int64_t)nS * Np;
      hipLaunchKernelGGL(k_reduce_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->partial, T, n, Uout + (size_t)s_lo * Np);
      RC_HIP(hipGetLastError());
    }
    return 0;
  };
  // u_S for one arbitrary slice -> Uout[Np]
  auto matvec_one = [&](const std::vector<double>& c, const double* gl, int ma, int mb, double* Uout) -> int {
    int r = rc_ensure_partial(h, (size_t)T * Np);
    if (r) return r;
    RC_HIP(hipMemcpyAsync(consts_d, c.data(), (size_t)4 * M * sizeof(double), hipMemcpyHostToDevice, h->stream));
    {
      RcProfScope ps(h, RC_K_SOBOL, pairs);
      hipLaunchKernelGGL(matvec_kernel, dim3((unsigned)T, (unsigned)T), dim3(256), lds_mv, h->stream, h->X, gl, consts_d, M, Np, h->partial, 1, ma,
                         mb, 0, 1);
      RC_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_reduce_rows, dim3((unsigned)((Np + 255) / 256)), dim3(256), 0, h->stream, h->partial, T, Np, Uout);
    RC_HIP(hipGetLastError());
    return 0;
  };

  std::vector<double> cD, cM, cH, canonD, canonM;
  coeffs('D', phi_a, phi_b, cD);
  coeffs('M', phi_a, phi_b, cM);
  coeffs('H', phi_a, phi_b, cH);
  // (1) DIAGONAL quadratic form: g_a on both sides
  if ((rc = pair_pass(cD, g_a, g_a, canonD))) return rc;
  // (2) MIXED quadratic form: g~_b on the row side, g_a on the column side
  {
    std::vector<double> cu(M);
    double cadd = 0.0;
    for (int m = 0; m < M; ++m) {
      const double den = 1.0 - ups_b[m] * phi_b[m];
      cu[m] = -0.5 * ups_b[m] * phi_b[m] * phi_b[m] / den;
      cadd -= 0.5 * log(den);
    }
    RC_HIP(hipMemcpyAsync(cu_d, cu.data(), (size_t)M * sizeof(double), hipMemcpyHostToDevice, h->stream));
    RC_HIP(hipStreamSynchronize(h->stream));
    hipLaunchKernelGGL(k_sobol_gtilde, dim3((unsigned)((Np + 255) / 256)), dim3(256), 0, h->stream, h->X, g_b, cu_d, cadd, h->N, Np, M, gt_b);
    RC_HIP(hipGetLastError());
  }
  if ((rc = pair_pass(cM, gt_b, g_a, canonM))) return rc;
  // (3) psi terms
  if ((rc = matvec_pass(cH, g_a, U))) return rc;
  const int full = 2 * M - 1;                                     // closed [0, M)
  const double* Ufull_bb = U + (size_t)full * Np;
  if (!self) {
    std::vector<double> cHbb;
    coeffs('H', phi_b, phi_b, cHbb);
    if ((rc = matvec_pass(cHbb, g_b, Ubb))) return rc;
    Ufull_bb = Ubb + (size_t)full * Np;
  }
  const int64_t rows = 6 * M + 1, rows_padded = ((rows + 127) / 128) * 128;
  if (mo) RC_HIP(hipMemsetAsync(h->KsT, 0, (size_t)rows_padded * h->Np * sizeof(double), h->stream));
  hipLaunchKernelGGL(k_sobol_psi_rows, dim3((unsigned)((Np + 255) / 256), (unsigned)rows_padded), dim3(256), 0, h->stream, U, Ufull_bb, g0_b,
                     3 * M, Np, rows_padded, h->KsT, h->Np, mo ? (int64_t)out_b * h->Nb : 0);
  RC_HIP(hipGetLastError());
  if ((rc = rc_launch_predict_var(h, rows_padded))) return rc;
  std::vector<double> pv(rows_padded);
  RC_HIP(hipMemcpyAsync(pv.data(), h->pvar, (size_t)rows_padded * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  RC_HIP(hipStreamSynchronize(h->stream));
  // NOTE rows 3M..6M-1 hold u_S + u_full of the (a,b) pass; the mixed term needs psi_full of (b,b): for a == b they coincide.
  for (int s = 0; s < n_slices; ++s) {
    const int a = slices[2 * s], b = slices[2 * s + 1];
    if (a == b) { phi_d_out[s] = psi_d_out[s] = phi_m_out[s] = psi_m_out[s] = 0.0; continue; }
    if (!is_canonical(a, b)) continue;                              // below
    const int idx = (a == 0) ? M + (b - 1) : ((b == M) ? 2 * M + (a - 1) : a);
    phi_d_out[s] = pre_b * canonD[idx];
    phi_m_out[s] = pre_b * canonM[idx];
    psi_d_out[s] = pv[idx];
    if (self) {
      psi_m_out[s] = 0.5 * (pv[3 * M + idx] - pv[idx] - pv[full]);
    } else {
      psi_m_out[s] = std::numeric_limits<double>::quiet_NaN();     // filled by the second GEMM below
    }
  }
  if (!self) {
    // psi_full(b,b) . psi_S(a,b) by polarisation with rows g0 * (u_S + u_full_bb): rebuild rows 3M.. with Ufull_bb (already the
    // case: k_sobol_psi_rows adds Ufull = Ufull_bb), and |psi_full_bb|^2 is row 6M.
    for (int s = 0; s < n_slices; ++s) {
      const int a = slices[2 * s], b = slices[2 * s + 1];
      if (a == b || !is_canonical(a, b)) continue;
      const int idx = (a == 0) ? M + (b - 1) : ((b == M) ? 2 * M + (a - 1) : a);
      psi_m_out[s] = 0.5 * (pv[3 * M + idx] - pv[idx] - pv[6 * M]);
    }
  }
  // ---- arbitrary slices (gsa/calibrators.py:348-373 takes any [m0, m1)): the same four ingredients, one slice at a time for the
  // quadratic forms and the u vector, the psi norms of up to 3M of them per GEMM. The canonical U rows are no longer needed (their
  // norms are on the host), only the (b,b) full-model vector, which moves to a scratch row of its own.
  std::vector<int> generic;
  for (int s = 0; s < n_slices; ++s)
    if (!is_canonical(slices[2 * s], slices[2 * s + 1])) generic.push_back(s);
  if (!generic.empty()) {
    RC_HIP(hipMemcpyAsync(al_a, Ufull_bb, (size_t)Np * sizeof(double), hipMemcpyDeviceToDevice, h->stream));      // alpha_a is spent: g_a exists
    const double* Ufull_keep = al_a;
    const int chunk = 3 * M;
    for (size_t g0i = 0; g0i < generic.size(); g0i += chunk) {
      const int G = (int)std::min<size_t>(chunk, generic.size() - g0i);
      for (int g = 0; g < G; ++g) {
        const int s = generic[g0i + g], a = slices[2 * s], b = slices[2 * s + 1];
        double qd = 0.0, qm = 0.0;
        if ((rc = pair_pass_one(cD, g_a, g_a, a, b, &qd)) || (rc = pair_pass_one(cM, gt_b, g_a, a, b, &qm))) return rc;
        phi_d_out[s] = pre_b * qd;
        phi_m_out[s] = pre_b * qm;
        if ((rc = matvec_one(cH, g_a, a, b, U + (size_t)g * Np))) return rc;
      }
      const int64_t rows_g = 2 * G + 1, rows_g_padded = ((rows_g + 127) / 128) * 128;
      if (mo) RC_HIP(hipMemsetAsync(h->KsT, 0, (size_t)rows_g_padded * h->Np * sizeof(double), h->stream));
      hipLaunchKernelGGL(k_sobol_psi_rows, dim3((unsigned)((Np + 255) / 256), (unsigned)rows_g_padded), dim3(256), 0, h->stream, U, Ufull_keep, g0_b,
                         G, Np, rows_g_padded, h->KsT, h->Np, mo ? (int64_t)out_b * h->Nb : 0);
      RC_HIP(hipGetLastError());
      if ((rc = rc_launch_predict_var(h, rows_g_padded))) return rc;
      std::vector<double> pg(rows_g_padded);
      RC_HIP(hipMemcpyAsync(pg.data(), h->pvar, (size_t)rows_g_padded * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      RC_HIP(hipStreamSynchronize(h->stream));
      for (int g = 0; g < G; ++g) {
        const int s = generic[g0i + g];
        psi_d_out[s] = pg[g];
        psi_m_out[s] = 0.5 * (pg[G + g] - pg[g] - pg[2 * G]);       // polarisation against the (b,b) full-model vector
      }
    }
  }
  return 0;
}
