// ARD-RBF Gram build (HBM-bound): K_ij = var * exp(-1/2 r_ij^2), r^2 = |z_i|^2 + |z_j|^2 - 2 z_i.z_j, Z = X / ell.
// This is GPflow's square_distance expansion (no clamp), the arithmetic behind gf.kernels.RBF.K at the reference call
// sites gpr/kernels.py:176 and gpr/models.py:435-437 (the +noise on the diagonal is fused here).
//
// One workgroup = one 128x128 tile, 512 threads; the two Z panels (128 x M each) are staged in LDS m-major, the dot products
// z_i . z_j run on the matrix cores (three v_mfma_f64_16x16x4 per 16 x 16 outputs at M = 10), the exponentials on the VALU; stores
// are 8 B per lane, 128 B contiguous per 16 lanes. Only tiles on or below the diagonal are written (the Cholesky reads the lower
// triangle).
#include "common.h"
#include "rc_math.h"

#define ZST 130   // LDS stride (doubles) between consecutive m of a Z panel: even (double2 alignment), bank-shift 4 per m

// rows_per_block: rows sharing one lengthscale vector (the output block of a covariant GP; every row when there is one output)
__global__ void k_scale(RcBP<const double> Xb, RcBP<const double> ellb, RcBP<double> Zb, RcBP<double> sqb, int64_t rows, int M,
                        int64_t rows_per_block) {
  const double* __restrict__ X = Xb.p[blockIdx.z];
  const double* __restrict__ ell = ellb.p[blockIdx.z];
  double* __restrict__ Z = Zb.p[blockIdx.z];
  double* __restrict__ sq = sqb.p[blockIdx.z];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const double* el = ell + (i / rows_per_block) * M;
  double s = 0.0;
  for (int m = 0; m < M; ++m) {
    const double z = X[i * M + m] / el[m];
    Z[i * M + m] = z;
    s = fma(z, z, s);
  }
  sq[i] = -0.5 * s;
}

// rows of new points scaled by the lengthscales of output `out`
int rc_launch_scale_rows(rcgp_handle_s* h, const double* X, double* Z, double* sq, int64_t rows, int out) {
  RcProfScope ps(h, RC_K_MISC, 0.0);
  RcBP<const double> Xb = {{X}}, eb = {{h->ell_d + (size_t)out * h->M}};       // (new points: always one unit)
  RcBP<double> Zb = {{Z}}, sb = {{sq}};
  hipLaunchKernelGGL(k_scale, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, h->launch, Xb, eb, Zb, sb, rows, h->M, rows > 0 ? rows : 1);
  RC_HIP(hipGetLastError());
  return 0;
}

int rc_launch_scale(rcgp_handle_s* h) {
  RC_BP(const double, Xb, h->X)
  RC_BP(const double, eb, h->ell_d)
  RC_BP(double, Zb, h->Z)
  RC_BP(double, sb, h->sq)
  RcProfScope ps(h, RC_K_MISC, 0.0);
  hipLaunchKernelGGL(k_scale, dim3((unsigned)((h->Np + 255) / 256), 1, (unsigned)h->nb), dim3(256), 0, h->launch, Xb, eb, Zb, sb, h->Np, h->M, h->Nb);
  RC_HIP(hipGetLastError());
  return 0;
}

__device__ __forceinline__ void tri_decode_g(int64_t id, int& ti, int& tj) {
  int t = (int)((sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
  while ((int64_t)(t + 1) * (t + 2) / 2 <= id) ++t;
  while ((int64_t)t * (t + 1) / 2 > id) --t;
  ti = t;
  tj = (int)(id - (int64_t)t * (t + 1) / 2);
}

// CROSS = false: square Gram, lower tiles, + noise on the diagonal, identity on the padding.
// CROSS = true : rectangular cross-Gram out[row = test point][col = training point], zero on the padding.
// Several outputs (covariant GP, gpf/kernels.py:93-104 and gpf/likelihoods.py:61-64): the rows/columns come in L blocks of
// tb tiles; block pair (bi, bj) has variance FS[bi * L + bj] and, where the in-block indices agree, noise FS[L * L + bi * L + bj];
// nr_valid / nc_valid count the valid rows of ONE block. The test points of a cross-Gram all belong to output rb.
// 512 threads per 128x128 tile, 32 outputs per thread (the C/D layout of 4 x 2 MFMA tiles per wave): <= 128 registers, so four
// waves per SIMD are resident and the store phase of one wave overlaps the exp phase of the others.
// WIDE (M > RC_MAX_M = 64): the dimensions pass through LDS in chunks of 64, the dot products accumulating in the MFMA accumulators across
// chunks; the fast instantiation stages the whole panels once, exactly as before.
template <bool CROSS, bool WIDE = false>
__global__ void __launch_bounds__(512, 4) k_gram(RcBP<double> outb, int64_t ld, RcBP<const double> Zrb, RcBP<const double> sqrb, int64_t nr_cross,
                                                 RcBP<const double> Zcb, RcBP<const double> sqcb, RcBN ncv, int M, RcBP<const double> FSb, int L,
                                                 int tb, int rb) {
  extern __shared__ double sm[];
  // the unit of this workgroup (blockIdx.z; a cross-Gram always has one). Valid rows: of the unit for the square Gram, the number of
  // new points for a cross-Gram.
  const int unit = CROSS ? 0 : blockIdx.z;
  double* __restrict__ out = outb.p[unit];
  const double* __restrict__ Zr = Zrb.p[unit];
  const double* __restrict__ sqr = sqrb.p[unit];
  const double* __restrict__ Zc = Zcb.p[unit];
  const double* __restrict__ sqc = sqcb.p[unit];
  const double* __restrict__ FS = FSb.p[unit];
  const int64_t nc_valid = ncv.v[unit], nr_valid = CROSS ? nr_cross : nc_valid;
  const int Mp = WIDE ? RC_MAX_M : ((M + 3) & ~3);     // dimensions (of a chunk) padded to a multiple of four (one fp64 MFMA consumes four), the padding zero
  double* zi = sm;                 // [Mp][ZST]
  double* zj = sm + Mp * ZST;      // [Mp][ZST]
  double* si = zj + Mp * ZST;      // [128]
  double* sj = si + 128;           // [128]
  int ti, tj;
  if (CROSS) {
    tj = blockIdx.x;
    ti = blockIdx.y;
  } else {
    tri_decode_g(blockIdx.x, ti, tj);
  }
  const int bi = CROSS ? rb : ti / tb, bj = tj / tb;
  const double var = FS[bi * L + bj], noise = CROSS ? 0.0 : FS[L * L + bi * L + bj];
  const int64_t ioff = CROSS ? 0 : (int64_t)bi * tb * 128, joff = (int64_t)bj * tb * 128;   // first row / column of the block
  const int t = threadIdx.x;
  if constexpr (!WIDE) {
    for (int e0 = t; e0 < 128 * M; e0 += 4 * 512) {    // four loads per panel in flight before the first LDS store (not one round trip each)
      double vi[4], vj[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = e0 + 512 * q;
        vi[q] = (e < 128 * M) ? Zr[(int64_t)ti * 128 * M + e] : 0.0;
        vj[q] = (e < 128 * M) ? Zc[(int64_t)tj * 128 * M + e] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = e0 + 512 * q;
        if (e < 128 * M) {
          const int rr = e / M, m = e - rr * M;
          zi[m * ZST + rr] = vi[q];
          zj[m * ZST + rr] = vj[q];
        }
      }
    }
  }
  if (t < 128) si[t] = sqr[(int64_t)ti * 128 + t];
  else if (t < 256) sj[t - 128] = sqc[(int64_t)tj * 128 + t - 128];
  if constexpr (!WIDE) {
    for (int e = t; e < (Mp - M) * 128; e += 512) {
      const int m = M + e / 128, rr = e & 127;
      zi[m * ZST + rr] = 0.0;
      zj[m * ZST + rr] = 0.0;
    }
    __syncthreads();
  }
  // z_i . z_j on the matrix cores (the -2 Z Z^T term of the squared-distance expansion IS a 128 x 128 x M product): 8 waves as 2 x 4, a wave
  // owns 64 x 32 = 4 x 2 MFMA tiles; A lane l = zi[k = l >> 4][row l & 15], B lane l = zj[k = l >> 4][col l & 15] -- both contiguous reads of
  // the m-major panels; the panels are zero-padded to a multiple of four dimensions. The M fp64 FMAs per element this replaces were 10 of the
  // ~34 VALU instructions per element at M = 10 (20 of 44 at M = 20), and the kernel is bound by VALU issue + stores.
  const int lane = t & 63, wave = t >> 6;
  const int wr = (wave >> 2) * 64, wc = (wave & 3) * 32;
  const int fr = lane & 15, fq = lane >> 4;
  v4d acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[a][c] = (v4d){0.0, 0.0, 0.0, 0.0};
  if constexpr (!WIDE) {
    for (int m0 = 0; m0 < Mp; m0 += 4) {
      double af[4], bf[2];
#pragma unroll
      for (int x = 0; x < 4; ++x) af[x] = zi[(m0 + fq) * ZST + wr + 16 * x + fr];
#pragma unroll
      for (int x = 0; x < 2; ++x) bf[x] = zj[(m0 + fq) * ZST + wc + 16 * x + fr];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
    }
  } else {
    for (int mc = 0; mc < M; mc += RC_MAX_M) {                   // dimension chunks [mc, mc + Mc)
      const int Mc = (M - mc < RC_MAX_M) ? M - mc : RC_MAX_M, Mcp = (Mc + 3) & ~3;
      __syncthreads();                                           // (everybody has read the previous chunk; si / sj are in place)
      for (int e = t; e < 128 * Mc; e += 512) {
        const int rr = e / Mc, m = e - rr * Mc;
        zi[m * ZST + rr] = Zr[((int64_t)ti * 128 + rr) * M + mc + m];
        zj[m * ZST + rr] = Zc[((int64_t)tj * 128 + rr) * M + mc + m];
      }
      for (int e = t; e < (Mcp - Mc) * 128; e += 512) {
        const int m = Mc + e / 128, rr = e & 127;
        zi[m * ZST + rr] = 0.0;
        zj[m * ZST + rr] = 0.0;
      }
      __syncthreads();
      for (int m0 = 0; m0 < Mcp; m0 += 4) {
        double af[4], bf[2];
#pragma unroll
        for (int x = 0; x < 4; ++x) af[x] = zi[(m0 + fq) * ZST + wr + 16 * x + fr];
#pragma unroll
        for (int x = 0; x < 2; ++x) bf[x] = zj[(m0 + fq) * ZST + wc + 16 * x + fr];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
      }
    }
  }
  // C/D layout: lane l, register r of tile (mi, ni) is element (row wr + 16 mi + 4 r + (l >> 4), column wc + 16 ni + (l & 15)): a store
  // instruction writes four rows x 128 contiguous bytes.
  const double sj0 = sj[wc + fr], sj1 = sj[wc + 16 + fr];
  // Interior tiles -- off the diagonal of their block pair and entirely inside the valid rows and columns: all but O(T) of the T^2/2
  // tiles -- need none of the per-element noise / padding tests.
  const int64_t i_lo = (int64_t)ti * 128 - ioff, j_lo = (int64_t)tj * 128 - joff;
  const bool interior = (CROSS || i_lo != j_lo || bi != bj) && i_lo + 128 <= nr_valid && j_lo + 128 <= nc_valid &&
                        (CROSS || i_lo >= j_lo + 128 || j_lo >= i_lo + 128);
  if (interior) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr + 16 * mi + 4 * r + fq;
        const double sia = si[row];
        double* dst_row = out + ((int64_t)ti * 128 + row) * ld + (int64_t)tj * 128 + wc + fr;
        __builtin_nontemporal_store(var * rc_exp(sia + sj0 + acc[mi][0][r]), dst_row);
        __builtin_nontemporal_store(var * rc_exp(sia + sj1 + acc[mi][1][r]), dst_row + 16);
      }
    return;
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wr + 16 * mi + 4 * r + fq;
      const int64_t i = (int64_t)ti * 128 + row, ii = i - ioff;            // ii, jj: in-block indices
      const double sia = si[row];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int64_t j = (int64_t)tj * 128 + wc + 16 * ni + fr, jj = j - joff;
        double v = var * rc_exp(sia + (ni ? sj1 : sj0) + acc[mi][ni][r]);
        if (CROSS) {
          if (ii >= nr_valid || jj >= nc_valid) v = 0.0;
        } else {
          if (ii == jj) v += noise;
          if (ii >= nr_valid || jj >= nc_valid) v = (i == j) ? 1.0 : 0.0;
        }
        __builtin_nontemporal_store(v, out + i * ld + j);   // written once, read next by another kernel: keep it out of the way in L2
      }
    }
}

static size_t gram_lds_bytes(int M) {
  const int Mc = (M < RC_MAX_M) ? M : RC_MAX_M;                  // the kernel stages at most RC_MAX_M dimensions at a time
  return (size_t)(2 * ((Mc + 3) & ~3) * ZST + 256) * sizeof(double);
}

int rc_launch_gram(rcgp_handle_s* h) {
  const int64_t T = h->Np / 128;
  const size_t lds = gram_lds_bytes(h->M);
  const bool wide = h->M > RC_MAX_M;
  RC_HIP(hipFuncSetAttribute(wide ? (const void*)k_gram<false, true> : (const void*)k_gram<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const double N = (double)h->N * (double)h->L;
  RC_BP(double, Ab, h->A)
  RC_BP(const double, Zb, h->Z)
  RC_BP(const double, sb, h->sq)
  RC_BP(const double, Fb, h->FS_d)
  // (profiling events ride on the dispatch itself: a pair of hipEventRecord markers brackets 15-60 us more than this 60-220 us kernel runs)
  RcProfScope ps(h, RC_K_GRAM, (double)h->nb * 8.0 * (N * (N + 1.0) / 2.0 + N * (double)h->M), true);   // algorithmic bytes
  const dim3 grid((unsigned)(T * (T + 1) / 2), 1, (unsigned)h->nb);
  if (wide) {
    RC_LAUNCH((k_gram<false, true>), grid, dim3(512), lds, Ab, h->Np, Zb, sb, (int64_t)0, Zb, sb, rc_bn(h), h->M, Fb, h->L, (int)(h->Nb / 128), 0);
  } else {
    RC_LAUNCH((k_gram<false, false>), grid, dim3(512), lds, Ab, h->Np, Zb, sb, (int64_t)0, Zb, sb, rc_bn(h), h->M, Fb, h->L, (int)(h->Nb / 128), 0);
  }
  RC_HIP(hipGetLastError());
  return 0;
}

int rc_launch_cross_gram(rcgp_handle_s* h, int64_t n, int64_t np, int out) {
  const size_t lds = gram_lds_bytes(h->M);
  const bool wide = h->M > RC_MAX_M;
  RC_HIP(hipFuncSetAttribute(wide ? (const void*)k_gram<true, true> : (const void*)k_gram<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  RcProfScope ps(h, RC_K_MISC, 0.0);
  RcBP<double> Kb = {{h->KsT}};
  RcBP<const double> Zsb = {{h->Zs}}, ssb = {{h->sqs}}, Zb = {{h->Z}}, sb = {{h->sq}}, Fb = {{h->FS_d}};
  RcBN nv;
  for (int u = 0; u < RC_MAX_BATCH; ++u) nv.v[u] = h->N;
  if (wide)
    hipLaunchKernelGGL((k_gram<true, true>), dim3((unsigned)(h->Np / 128), (unsigned)(np / 128)), dim3(512), lds, h->launch, Kb, h->Np, Zsb, ssb, n, Zb, sb,
                       nv, h->M, Fb, h->L, (int)(h->Nb / 128), out);
  else
    hipLaunchKernelGGL((k_gram<true, false>), dim3((unsigned)(h->Np / 128), (unsigned)(np / 128)), dim3(512), lds, h->launch, Kb, h->Np, Zsb, ssb, n, Zb, sb,
                       nv, h->M, Fb, h->L, (int)(h->Nb / 128), out);
  RC_HIP(hipGetLastError());
  return 0;
}
