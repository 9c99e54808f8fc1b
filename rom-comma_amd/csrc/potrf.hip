// Blocked right-looking Cholesky, in place on the lower triangle of A (K + noise I), fused with the forward substitution
// w = L^-1 y. Replaces tf.linalg.cholesky / triangular_solve inside GPflow's GPR.log_marginal_likelihood
// (reference call sites gpr/models.py:360, 427-439).
//
// Two-level blocking: outer panels of NB columns (K of the big MFMA updates; h->nb_outer, 512), inner blocks of 128:
//   k_diag2      : one workgroup factors the 128x128 diagonal block in LDS (16-blocked, MFMA), inverts it, emits w_j and log L_ii
//   k_prep_next  : (gemm.hip) the tile right below it solved, the next diagonal block completed -- the other critical kernel
//   k_trsm_panel : rows below <- rows below * inv(L_jj)^T  (a GEMM, gemm.hip), rhs update fused
//   k_gemm_nt_sub: K = 128 update of the block columns the chain keeps current; K = NB update of the window's column panels
//   k_syrk_lower : bulk trailing update with the whole outer panel (K = NB)
// Scheduling (main stream + chain + two column-work streams + bulk, events only): potrf_fine below; rc_potrf keeps the simpler coarse schedule behind RCGP_FINE=0 and for
// matrices of one or two panels.
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
// Two pivots (columns k, k+1) per barrier: the owners publish both raw columns, then every thread factors the 2x2 pivot
// block redundantly and applies the rank-2 update to its own elements -- half the synchronisations of a column-at-a-time sweep.
template <int KB>
__device__ __forceinline__ void chol_steps(double (&Lr)[4][8], double* colbuf, double* rsd, int* info, int64_t j0, const int tx,
                                           const int ty) {
  constexpr int A0 = KB >> 1;                                 // first register row-group that can lie on/below the diagonal
#pragma unroll 1
  for (int q = 0; q < 8; ++q) {
    const int kx = 2 * q, k = KB * 16 + kx;
    double* c0 = colbuf + (q & 1) * 256;
    double* c1 = c0 + 128;
    if (tx == kx) {
#pragma unroll
      for (int a = A0; a < 4; ++a) c0[ty + 32 * a] = Lr[a][KB];
    } else if (tx == kx + 1) {
#pragma unroll
      for (int a = A0; a < 4; ++a) c1[ty + 32 * a] = Lr[a][KB];
    }
    __syncthreads();
    double d0 = c0[k];
    if (!(d0 > 0.0)) {                      // not positive definite (or NaN): flag the leading minor, keep going finite
      if (threadIdx.x == 0) atomicCAS(info, 0, (int)(j0 + k + 1));
      d0 = 1.0;
    }
    const double rs0 = rc_rsqrt(d0);
    const double l10 = c0[k + 1] * rs0;                       // L[k+1][k]
    double d1 = __builtin_fma(-l10, l10, c1[k + 1]);
    if (!(d1 > 0.0)) {
      if (threadIdx.x == 0) atomicCAS(info, 0, (int)(j0 + k + 2));
      d1 = 1.0;
    }
    const double rs1 = rc_rsqrt(d1);
    if (threadIdx.x == 0) { rsd[k] = rs0; rsd[k + 1] = rs1; }
    double li0[4], li1[4], lj0[8], lj1[8];
#pragma unroll
    for (int a = A0; a < 4; ++a) {
      li0[a] = c0[ty + 32 * a] * rs0;
      li1[a] = __builtin_fma(-li0[a], l10, c1[ty + 32 * a]) * rs1;
    }
#pragma unroll
    for (int b = KB; b < 8; ++b) {
      lj0[b] = c0[tx + 16 * b] * rs0;
      lj1[b] = __builtin_fma(-lj0[b], l10, c1[tx + 16 * b]) * rs1;
    }
    if (tx == kx) {
#pragma unroll
      for (int a = A0; a < 4; ++a) Lr[a][KB] = li0[a];        // final L[i][k]   (row k itself: d0*rs0 = sqrt(d0))
    } else if (tx == kx + 1) {
#pragma unroll
      for (int a = A0; a < 4; ++a) Lr[a][KB] = li1[a];        // final L[i][k+1] (row k+1: d1*rs1 = sqrt(d1))
    }
#pragma unroll
    for (int b = KB; b < 8; ++b) {
      if (b == KB && tx <= kx + 1) continue;                  // only columns j > k+1
#pragma unroll
      for (int a = b >> 1; a < 4; ++a) Lr[a][b] = __builtin_fma(-li1[a], lj1[b], __builtin_fma(-li0[a], lj0[b], Lr[a][b]));
    }
  }
}

// Forward elimination on the identity, two rows (k, k+1) per barrier: rows k and k+1 of X = L^-1 become final, then
// rows i > k+1: X[i][j] -= L[i][k] X[k][j] + L[i][k+1] X[k+1][j].
template <int KB>
__device__ __forceinline__ void inv_steps(double (&Xr)[4][8], const double (&Lr)[4][8], double* colbuf, double* rowbuf, const double* rsd,
                                          const int tx, const int ty) {
  constexpr int AK = KB >> 1;                                 // register row-group holding rows k, k+1
#pragma unroll 1
  for (int q = 0; q < 8; ++q) {
    const int kx = 2 * q, k = KB * 16 + kx;
    const int kty = 16 * (KB & 1) + kx;                       // ty of the threads holding row k (row k+1: kty + 1)
    double* c0 = colbuf + (q & 1) * 256;
    double* c1 = c0 + 128;
    double* r0 = rowbuf + (q & 1) * 256;
    double* r1 = r0 + 128;
    if (ty == kty) {
#pragma unroll
      for (int b = 0; b <= KB; ++b) r0[tx + 16 * b] = Xr[AK][b];
    } else if (ty == kty + 1) {
#pragma unroll
      for (int b = 0; b <= KB; ++b) r1[tx + 16 * b] = Xr[AK][b];
    }
    if (tx == kx) {
#pragma unroll
      for (int a = AK; a < 4; ++a) c0[ty + 32 * a] = Lr[a][KB];
    } else if (tx == kx + 1) {
#pragma unroll
      for (int a = AK; a < 4; ++a) c1[ty + 32 * a] = Lr[a][KB];
    }
    __syncthreads();
    const double rk0 = rsd[k], rk1 = rsd[k + 1];
    const double l10 = c0[k + 1];                              // L[k+1][k]
    double x0[8], x1[8];
#pragma unroll
    for (int b = 0; b <= KB; ++b) {
      const int j = tx + 16 * b;
      x0[b] = (j < k) ? r0[j] * rk0 : (j == k ? rk0 : 0.0);
      x1[b] = (j <= k) ? __builtin_fma(-l10, x0[b], r1[j]) * rk1 : (j == k + 1 ? rk1 : 0.0);
    }
    if (ty == kty) {
#pragma unroll
      for (int b = 0; b <= KB; ++b) Xr[AK][b] = x0[b];
    } else if (ty == kty + 1) {
#pragma unroll
      for (int b = 0; b <= KB; ++b) Xr[AK][b] = x1[b];
    }
#pragma unroll
    for (int a = AK; a < 4; ++a) {
      if (a == AK && ty <= kty + 1) continue;                 // only rows i > k+1
      const double l0 = c0[ty + 32 * a], l1 = c1[ty + 32 * a];
#pragma unroll
      for (int b = 0; b <= KB; ++b) Xr[a][b] = __builtin_fma(-l1, x1[b], __builtin_fma(-l0, x0[b], Xr[a][b]));
    }
  }
}

__global__ void __launch_bounds__(512, 4) k_diag(double* __restrict__ A, int64_t ld, double* __restrict__ invL, double* __restrict__ rhs,
                                                 double* __restrict__ logdiag, int* __restrict__ info, int64_t j0) {
  __shared__ double colbuf[4 * 128], rowbuf[4 * 128], rsd[128], rv[128];
  __builtin_amdgcn_s_setprio(3);             // latency-critical: win issue arbitration against co-resident GEMM waves
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


// =====================================================================================================================
// k_diag2: the same job as k_diag (factor the 128x128 diagonal block, invert it, w_j, log L_ii) in ~1/3 of the time.
// 16-blocked right-looking Cholesky inside one workgroup (8 waves):
//   (a) the 16x16 pivot block is factored AND inverted by wave 0 alone, in registers, with v_readlane broadcasts (no LDS
//       round trips, no barriers inside the 16 sequential pivots);
//   (b) panel below:  L_rc = S_rc * inv(L_cc)^T        -- fp64 MFMA, operands from LDS, one 16x16 tile per wave at a time
//   (c) trailing:     S_{r,c2} -= L_rc * L_{c2,c}^T    -- fp64 MFMA
// then L^-1 by recursive doubling (block sizes 16, 32, 64) on MFMA: X21 = -C^-1 (B A^-1).
// Storage: S[128][LS] holds L in its lower triangle; the off-diagonal blocks of X = L^-1 live TRANSPOSED in the upper
// triangle of S, the diagonal 16-blocks of X in Xd. ~150 KB of LDS: runs on one of the CUs the bulk-update stream leaves free.
// =====================================================================================================================
#ifdef RC_DIAG_TIMING
__device__ long long g_diag_t[32];
__device__ long long g_diag_span[2 * 512];      // entry / exit stamp of the diagonal kernel of block j0 / 128
#define RC_T(i) do { if (threadIdx.x == 0) g_diag_t[i] = wall_clock64(); } while (0)
#define RC_SPAN(k) do { if (threadIdx.x == 0) g_diag_span[2 * ((j0 / 128) & 511) + (k)] = wall_clock64(); } while (0)
extern "C" __attribute__((visibility("default"))) int rcgp_debug_diag_times(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag_t), sizeof(long long) * 32);
}
extern "C" __attribute__((visibility("default"))) int rcgp_debug_diag_spans(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag_span), sizeof(long long) * 2 * 512);
}
#else
#define RC_T(i)
#define RC_SPAN(k)
#endif

#define LS 130
#define XS 18

__device__ __forceinline__ double rl_d(double v, int lane) {      // broadcast lane `lane` (compile-time constant) of v
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Wave 0 only (lane l mirrors row/column l & 15): Cholesky of the 16x16 block at (16c,16c) of S and its inverse.
// Lane r keeps row r in registers. Per pivot the current column is published through a 16-entry LDS line and read back as
// wave-uniform broadcasts (one LDS round trip per pivot, no barrier: a wave's LDS operations execute in order). The line is read
// in ONE batch of 16-byte loads into registers before any of it is used: left to itself the compiler issues read - wait - two FMAs
// eight times per pivot, i.e. eight exposed LDS latencies instead of one. The same for the inverse, whose column k of L comes
// from a transposed 16x16 copy (lt) as one contiguous batch per step.
template <bool INL>
__device__ __forceinline__ void pivot_block_16_body(double* S, double* Xd, double* rsd, double* pcol, double* lt, int* info, int64_t j0, int c,
                                                    int lane) {
#ifdef RC_DIAG2_NO_PIVOT
  if (lane < 16) { for (int i = 0; i < 16; ++i) Xd[(c * 16 + i) * XS + lane] = (i == lane) ? 1.0 : 0.0; rsd[16 * c + lane] = 1.0; }
  return;
#endif
  const int r = lane & 15;
  double* blk = S + (16 * c) * LS + 16 * c;
  double a[16], rsv[16];
  if (c == 0) RC_T(24);
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = (j <= r) ? blk[r * LS + j] : 0.0;
  if (c == 0) RC_T(25);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    double* line = pcol + (j & 1) * 16;
    line[r] = a[j];                                             // column j (unscaled): entry r from lane r
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double ln[16];
#pragma unroll
    for (int q = j >> 1; q < 8; ++q) {
      const double2 v = *reinterpret_cast<const double2*>(line + 2 * q);
      ln[2 * q] = v.x;
      ln[2 * q + 1] = v.y;
    }
    // The only chain from one pivot to the next is d -> 1/d -> tj -> a[j+1] -> line: the reciprocal gets its own short Newton
    // sequence (a dependent fp64 op costs ~40 cycles here); 1/sqrt(d), needed for the final scaling and the inverse only, and the
    // positivity check hang off it.
    const double draw = rl_d(a[j], j);                          // = line[j], without waiting for the LDS round trip
    const bool ok = draw > 0.0;                                 // not positive definite (or NaN): flag the leading minor, go on finite
    double rd = __builtin_amdgcn_rcp(draw);
    rd = __builtin_fma(__builtin_fma(-draw, rd, 1.0), rd, rd);
    rd = __builtin_fma(__builtin_fma(-draw, rd, 1.0), rd, rd);
    const double tj = a[j] * (ok ? rd : 1.0);                   // L[r][j] / sqrt(d) = A[r][j] / d
#pragma unroll
    for (int c2 = j + 1; c2 < 16; ++c2) a[c2] = __builtin_fma(-tj, ln[c2], a[c2]);
    if (!ok && lane == 0) atomicCAS(info, 0, (int)(j0 + 16 * c + j + 1));
    const double rs = rc_rsqrt(ok ? draw : 1.0);
    rsv[j] = rs;
    a[j] *= rs;                                                 // L[r][j]
  }
  if (c == 0) RC_T(26);
  // publish L_cc (rows from lanes 0..15) and its transpose, then invert it: lane j builds column j of X by forward substitution
  // with wave-uniform reads of L[i][k]
  if (lane < 16) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      blk[r * LS + j] = (j <= r) ? a[j] : 0.0;
      lt[j * 16 + r] = (j <= r) ? a[j] : 0.0;                   // lt[k][i] = L[i][k]
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  double x[16];
  if (c == 0) RC_T(27);
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = (i == r) ? 1.0 : 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {                                  // column-oriented: the dependent chain is 16 steps, not 120
    double lk[16];
#pragma unroll
    for (int q = (k + 1) >> 1; q < 8; ++q) {
      const double2 v = *reinterpret_cast<const double2*>(lt + k * 16 + 2 * q);
      lk[2 * q] = v.x;
      lk[2 * q + 1] = v.y;
    }
    x[k] *= rsv[k];
#pragma unroll
    for (int i = k + 1; i < 16; ++i) x[i] = __builtin_fma(-lk[i], x[k], x[i]);
  }
  if (c == 0) RC_T(28);
  if (lane < 16) {
#pragma unroll
    for (int i = 0; i < 16; ++i) Xd[(c * 16 + i) * XS + r] = (i >= r) ? x[i] : 0.0;      // Xd[c][i][j = r]
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (lane == j) rsd[16 * c + j] = rsv[j];
    // the strictly-upper part of the diagonal block belongs to nobody: restore zeros (blk was written with zeros above)
  }
  if (c == 0) RC_T(30);
}

// Out of line for the first form of k_diag2 (two call sites: inlined twice the kernel spills). NOTE what the call costs there: the callee
// saves and restores its 112 callee-saved VGPRs through scratch memory on every call (ISA: 112 scratch_store / scratch_load_dword around the
// body, ~2.5 us per pivot block by in-kernel stamps) -- the kernel-resource-usage remark of the KERNEL does not show the callee's frame.
// chol128_regs has ONE call site and inlines it.
__device__ __attribute__((noinline)) void pivot_block_16(double* S, double* Xd, double* rsd, double* pcol, double* lt, int* info, int64_t j0, int c,
                                               int lane) {
  pivot_block_16_body<false>(S, Xd, rsd, pcol, lt, info, j0, c, lane);
}

// element (i, j), i >= j, of X = L^-1 in its split storage
__device__ __forceinline__ double xval(const double* S, const double* Xd, int i, int j) {
  return ((i >> 4) == (j >> 4)) ? Xd[((i >> 4) * 16 + (i & 15)) * XS + (j & 15)] : S[j * LS + i];
}


// ---------------------------------------------------------------------------------------------------------------------
// pivot16: Cholesky of one 16x16 pivot block and its inverse by ONE wave (lane l mirrors row / column l & 15), written for a SHORT
// dependent chain and FEW registers, so that it can be inlined at its single call site in chol128_regs. (The first form,
// pivot_block_16, is a noinline function with two call sites: its call alone costs ~2.5 us per block -- 112 callee-saved VGPRs go
// through scratch memory on every call -- and per pivot it waits for an LDS round trip in the middle of an 8-operation fp64 chain.)
//   * Lane r keeps row r of the block in registers; column j (unscaled) is published through a 16-entry LDS line, as before -- but only
//     the entries j+2.. are read from it, one iteration after they were written, so the LDS latency is off the recurrence.
//   * The recurrence from pivot to pivot involves wave-uniform values only:  d_{j+1} = e - s^2 / d_j  with s = A[j+1][j] and
//     e = A[j+1][j+1] as they stand before pivot j's update, both fetched by v_readlane while 1/d_j is still being refined. The
//     chain per pivot is then fma -> rcp -> 2 Newton steps (6 dependent fp64 operations instead of 8 + LDS), everything else
//     (the row updates, 1/sqrt(d), the publishing stores) fills the issue slots beside it.
//   * Not positive definite (or NaN): the first failing leading minor is remembered in a register and flagged once, after the block;
//     the sweep goes on with d = 1 so that everything stays finite (as before).
// blk: the block inside S (row stride LS), holding the tile to factor, receives L_cc (zeros above the diagonal); Xdc: its slot in Xd;
// rsdc: its 16 entries of rsd (1 / L_ii).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pivot16(double* blk, double* Xdc, double* rsdc, double* pcol, double* lt, int* info, int64_t row0, int lane,
                                        bool stamp = false) {
  const int r = lane & 15;
  double a[16];
  if (stamp) RC_T(24);
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = (j <= r) ? blk[r * LS + j] : 0.0;
  if (stamp) RC_T(25);
  pcol[r] = a[0];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  double dv = rl_d(a[0], 0);
  int bad = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const double* line = pcol + (j & 1) * 16;
    double ln[16];
#pragma unroll
    for (int q = (j + 2) >> 1; q < 8; ++q) {                     // column j, entries >= j + 2 (written one iteration ago)
      const double2 v = *reinterpret_cast<const double2*>(line + 2 * q);
      ln[2 * q] = v.x;
      ln[2 * q + 1] = v.y;
    }
    const bool ok = dv > 0.0;
    bad = (!ok && bad == 0) ? j + 1 : bad;
    const double dd = ok ? dv : 1.0;
    double rd = __builtin_amdgcn_rcp(dd);
    rd = __builtin_fma(__builtin_fma(-dd, rd, 1.0), rd, rd);
    rd = __builtin_fma(__builtin_fma(-dd, rd, 1.0), rd, rd);
    const double tj = a[j] * rd;                                 // L[r][j] / sqrt(d) = A[r][j] / d
    if (j < 15) {
      const double s1 = rl_d(a[j], j + 1);                       // A[j+1][j]
      const double e = rl_d(a[j + 1], j + 1);                    // A[j+1][j+1] before this pivot's update
      dv = __builtin_fma(-(s1 * s1), rd, e);                     // the next pivot, wave-uniform
      a[j + 1] = __builtin_fma(-tj, s1, a[j + 1]);
      pcol[((j + 1) & 1) * 16 + r] = a[j + 1];                   // column j + 1 is complete: publish it
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int c2 = j + 2; c2 < 16; ++c2) a[c2] = __builtin_fma(-tj, ln[c2], a[c2]);
    const double rs = rc_rsqrt(dd);
    a[j] *= rs;                                                  // L[r][j]
    if (lane == j) rsdc[j] = rs;
  }
  if (stamp) RC_T(26);
  if (bad != 0 && lane == 0) atomicCAS(info, 0, (int)(row0 + bad));
  // publish L_cc (rows from lanes 0..15) and its transpose, then invert it: lane j builds column j of X by forward substitution
  if (lane < 16) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const double v = (j <= r) ? a[j] : 0.0;
      blk[r * LS + j] = v;
      lt[j * 16 + r] = v;                                        // lt[k][i] = L[i][k]
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  double x[16];
  if (stamp) RC_T(27);
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = (i == r) ? 1.0 : 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    double lk[16];
#pragma unroll
    for (int q = (k + 1) >> 1; q < 8; ++q) {
      const double2 v = *reinterpret_cast<const double2*>(lt + k * 16 + 2 * q);
      lk[2 * q] = v.x;
      lk[2 * q + 1] = v.y;
    }
    x[k] *= rsdc[k];
#pragma unroll
    for (int i = k + 1; i < 16; ++i) x[i] = __builtin_fma(-lk[i], x[k], x[i]);
  }
  if (stamp) RC_T(28);
  if (lane < 16) {
#pragma unroll
    for (int i = 0; i < 16; ++i) Xdc[i * XS + r] = (i >= r) ? x[i] : 0.0;                    // Xd[c][i][j = r]
  }
  if (stamp) RC_T(30);
}

// ---------------------------------------------------------------------------------------------------------------------
// chol128_regs: the 16-blocked Cholesky of the 128x128 block with the TRAILING MATRIX IN REGISTERS (k_diag2<MODE, true>).
// In k_diag2's first form every trailing tile update is LDS -> accumulator -> 4 MFMAs -> LDS (12 LDS reads + 4 writes per tile and
// step) and the pivot wave starts each step with one of them; here a tile of the trailing matrix stays in the accumulator registers of
// the wave that owns it from the first load to the panel step that makes it final, and LDS (the array S) only ever receives FINAL
// tiles of L -- which is also where the panel and trailing products read their operands from.
//   Tile layout, the same for every 16x16 tile held in registers: lane (fr = lane & 15, fq = lane >> 4), register q holds
//   T[fr][4q + fq]. For a tile of the symmetric trailing matrix, S_{rb,cb}, that IS the MFMA C/D layout of U = S_{rb,cb}^T
//   (row 4q + fq, column fr), and the same four registers are the B operand (k = 4s + fq, s = q) of  L_{rb,c}^T = X_cc * U  -- the panel
//   solve needs no transposition through LDS -- whose result comes out in the same layout again.
//   Trailing update in that layout:  U_{cb,rb} -= L_{cb,c} * L_{rb,c}^T  (A = L_{cb,c}, B = L_{rb,c}^T, both read from S; neg A by the MFMA).
//   Ownership (static; rc_tile_owner): wave k owns the diagonal tile (k,k) -- it factors pivot block k -- and the tile (k,k-1) to its
//   left; its three other tiles lie in block columns < k. So at step c the NEXT pivot wave (c + 1) has exactly one trailing update to do,
//   its own diagonal tile, and goes straight on to the pivot block while the other seven waves update everything else: nothing
//   but that one tile update (4 MFMAs) and the panel tile (c+1, c) (4 MFMAs) sits between two pivot blocks.
//   Two workgroup barriers per step: X_cc published -> panel column c; panel column c in S -> trailing updates.
// ---------------------------------------------------------------------------------------------------------------------
#define RC_REG_SLOTS 5
__device__ __constant__ signed char rc_tile_owner[8][RC_REG_SLOTS][2] = {      // [wave][slot] = {rb, cb}; rb < 0: no tile
    {{0, 0}, {-1, 0}, {-1, 0}, {-1, 0}, {-1, 0}},
    {{1, 1}, {1, 0}, {2, 0}, {3, 0}, {4, 0}},
    {{2, 2}, {2, 1}, {5, 0}, {6, 0}, {3, 1}},
    {{3, 3}, {3, 2}, {7, 0}, {4, 1}, {5, 1}},
    {{4, 4}, {4, 3}, {6, 1}, {7, 1}, {4, 2}},
    {{5, 5}, {5, 4}, {5, 2}, {6, 2}, {7, 2}},
    {{6, 6}, {6, 5}, {5, 3}, {6, 3}, {7, 3}},
    {{7, 7}, {7, 6}, {6, 4}, {7, 4}, {7, 5}}};

__device__ __forceinline__ void chol128_regs(double* S, double* Xd, double* rsd, double* pcol, double* lt, int* info, int64_t j0, double* rv = nullptr) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  int rbs[RC_REG_SLOTS], cbs[RC_REG_SLOTS];
  v4d acc[RC_REG_SLOTS];
#pragma unroll
  for (int k = 0; k < RC_REG_SLOTS; ++k) {
    rbs[k] = __builtin_amdgcn_readfirstlane((int)rc_tile_owner[wave][k][0]);
    cbs[k] = __builtin_amdgcn_readfirstlane((int)rc_tile_owner[wave][k][1]);
    acc[k] = (v4d){0.0, 0.0, 0.0, 0.0};
    if (rbs[k] >= 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[k][q] = S[(16 * rbs[k] + fr) * LS + 16 * cbs[k] + 4 * q + fq];
    }
  }
  // (no barrier: S is not written before every wave has passed the first one below -- wave 0 rewrites only its own tile (0,0))
#pragma unroll 1
  for (int c = 0; c < 8; ++c) {
    if (wave == c) {                                  // pivot block c: the owner puts its diagonal tile where pivot_block_16 works in place
#pragma unroll
      for (int q = 0; q < 4; ++q) S[(16 * c + fr) * LS + 16 * c + 4 * q + fq] = acc[0][q];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (c == 0) RC_T(29);
      // (an opaque zero per iteration: without it the compiler hoists the ~200 loop-invariant LDS addresses of the unrolled pivot
      // sweep out of the c loop and keeps each in a register of its own -- 140 SGPRs spilled to VGPR lanes, 30 VGPRs to scratch)
      int zero = 0;
      asm volatile("" : "+s"(zero));
      pivot16(S + (16 * c) * LS + 16 * c, Xd + c * 16 * XS, rsd + 16 * c, pcol + zero, lt + zero, info, j0 + 16 * c, lane + zero, c == 0);
      if (c == 0) RC_T(31);
    }
    __syncthreads();                                  // X_cc (Xd[c]) and L_cc are visible
    RC_T(2 + 2 * c);
    if (rv && wave == 0) {
      // forward substitution of the right-hand side, a side job of wave 0 (which owns a single tile): w_c = X_cc r_c now, and the rows
      // below -= L_{.,c} w_c once block column c is in S (after the next barrier). rv ends up holding w_j = L_jj^-1 rhs_j.
      double sw = 0.0;
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) sw = __builtin_fma(Xd[(c * 16 + lane) * XS + j], rv[16 * c + j], sw);
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < 16) rv[16 * c + lane] = sw;
    }
    if (c == 7) break;
    double xa[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xa[s] = Xd[(c * 16 + fr) * XS + 4 * s + fq];            // A operand: X_cc[fr][4s + fq]
#pragma unroll
    for (int k = 1; k < RC_REG_SLOTS; ++k) {          // panel: this wave's tiles of block column c become final
      if (rbs[k] >= 0 && cbs[k] == c) {
        v4d d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) d = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[s], acc[k][s], d, 0, 0, 0);
        acc[k] = d;
#pragma unroll
        for (int q = 0; q < 4; ++q) S[(16 * rbs[k] + fr) * LS + 16 * c + 4 * q + fq] = d[q];
      }
    }
    __syncthreads();                                  // block column c of L is in S
    RC_T(3 + 2 * c);
    if (rv && wave == 0) {
      for (int row = 16 * (c + 1) + lane; row < 128; row += 64) {
        double sw = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) sw = __builtin_fma(S[row * LS + 16 * c + j], rv[16 * c + j], sw);
        rv[row] -= sw;
      }
    }
#pragma unroll
    for (int k = 0; k < RC_REG_SLOTS; ++k) {          // trailing: slot 0 (the diagonal tile) first -- the next pivot wave has only that one
      if (rbs[k] >= 0 && cbs[k] > c) {
        double av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          av[s] = S[(16 * cbs[k] + fr) * LS + 16 * c + 4 * s + fq];                     // L_{cb,c}[fr][4s + fq], negated by the MFMA
          bv[s] = S[(16 * rbs[k] + fr) * LS + 16 * c + 4 * s + fq];                     // L_{rb,c}[fr][4s + fq]
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc[k], 0, 0, 1);
      }
    }
  }
  if (rv) __syncthreads();                            // w_7 (wave 0) before anybody reads rv
}

// MODE 0: the whole job (factor, invert, w_j). MODE 1: factor only -- L_jj, log L_ii and the eight 16x16 diagonal-block inverses
// (written into the diagonal 16-blocks of invL) -- which is all the NEXT chain step needs (k_prep1s solves the tile below by
// substitution): the 128x128 inverse and w_j are 15 us of this kernel and come off the critical path. MODE 2: the rest, as a kernel
// of its own on the column-work stream (MODE 3: the inversion alone, no right-hand side -- k_inv128_batched): reads L_jj and the diagonal-block inverses back, completes the right-hand side rows of
// this block (rhs_j -= L(j, j-1) w_{j-1}: the block row the chain no longer updates), inverts, emits invL and w_j.
template <int MODE, bool REG = false>
__device__ __forceinline__ void diag2_body(double* S, double* __restrict__ A, int64_t ld, double* __restrict__ invL, double* __restrict__ rhs,
                                           double* __restrict__ logdiag, int* __restrict__ info, int64_t j0) {
  // S: [128][LS], then Xd[8][16][XS], rsd[128], rv[128]   (dynamic shared memory of the calling kernel)
  double* Xd = S + 128 * LS;
  double* rsd = Xd + 8 * 16 * XS;
  double* rv = rsd + 128;
  double* pcol = rv + 128;                          // 2 x 16 pivot-column lines
  double* lt = pcol + 32;                           // 16 x 16: transposed copy of the current diagonal 16-block of L
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  double* At = A + j0 * ld + j0;
  RC_T(0);
  RC_SPAN(0);
  {
    // 16 bytes per lane; pairs entirely above the diagonal are not fetched. All sixteen loads of a lane are issued before the first is
    // waited for (as a loop this is sixteen memory round trips in a row: 5.5 us of the kernel on an idle chip, far more beside GEMMs).
    double2 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = t + 512 * q, i = e >> 6, j = (e & 63) * 2;
      v[q] = make_double2(0.0, 0.0);
      if (j <= i) v[q] = *reinterpret_cast<const double2*>(At + (int64_t)i * ld + j);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = t + 512 * q, i = e >> 6, j = (e & 63) * 2;
      S[i * LS + j] = v[q].x;
      S[i * LS + j + 1] = (j + 1 <= i) ? v[q].y : 0.0;
    }
  }
  if ((MODE == 0 || (MODE == 1 && REG)) && t < 128) rv[t] = rhs[j0 + t];
  if (MODE >= 2) {
    // the diagonal-block inverses of the factor-only kernel, and this block's right-hand side rows brought up to date with the tile
    // to the left (4 lanes per row, 16-byte loads all in flight)
    for (int e = t; e < 8 * 16 * 16; e += 512) {
      const int c = e >> 8, i = (e >> 4) & 15, j = e & 15;
      Xd[(c * 16 + i) * XS + j] = invL[(16 * c + i) * 128 + 16 * c + j];
    }
    const int i = t >> 2, h4 = t & 3;
    double sacc = 0.0;
    if (MODE == 2 && j0 > 0) {
      const double* Trow = A + (j0 + i) * ld + (j0 - 128);
      const double* wprev = rhs + (j0 - 128);
      double2 tv[16], wv2[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        tv[q] = *reinterpret_cast<const double2*>(Trow + 8 * q + 2 * h4);
        wv2[q] = *reinterpret_cast<const double2*>(wprev + 8 * q + 2 * h4);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) sacc = __builtin_fma(tv[q].y, wv2[q].y, __builtin_fma(tv[q].x, wv2[q].x, sacc));
      sacc += __shfl_xor(sacc, 1);
      sacc += __shfl_xor(sacc, 2);
    }
    if (MODE == 2 && h4 == 0) rv[i] = rhs[j0 + i] - sacc;
  }
  __syncthreads();
  RC_T(1);

  if (MODE < 2 && REG) {
    chol128_regs(S, Xd, rsd, pcol, lt, info, j0, MODE == 1 ? rv : nullptr);
  }
  if (MODE < 2 && !REG) {
  // ------------------------------------------------------------------ blocked Cholesky
  // Iteration c: (a) trailing update with block column c-1 of the lower tiles (rb, cb), c <= cb <= rb -- tile (c, c) goes to
  // wave 0, which then factors that pivot block while the other waves finish the remaining tiles; (b) panel below the pivot block.
  // (One call site for the pivot block: inlined twice it cost 180 B of scratch per lane.)
#pragma unroll 1
  for (int c = 0; c < 8; ++c) {
    const int cp = c - 1, nb = 8 - c;               // previous block column; remaining block rows/cols
    const int ntiles = (c > 0) ? nb * (nb + 1) / 2 : 0;
    auto trailing_tile = [&](int tile) {
      // tile 0 = (c, c); enumerate the lower triangle row by row
      int rr = 0, acc_t = tile;
      while (acc_t > rr) { acc_t -= rr + 1; ++rr; }
      const int rb = c + rr, cb = c + acc_t;
      v4d acc;
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = S[(16 * rb + fq + 4 * q) * LS + 16 * cb + fr];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k = 4 * s + fq;
        const double av = S[(16 * rb + fr) * LS + 16 * cp + k];          // L_{rb,cp}[i][k], negated by the MFMA (neg:[1,0,0])
        const double bv = S[(16 * cb + fr) * LS + 16 * cp + k];          // B[k][j] = L_{cb,cp}[j][k]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 1);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) S[(16 * rb + fq + 4 * q) * LS + 16 * cb + fr] = acc[q];
    };
    if (wave == 0) {
      if (ntiles > 0) trailing_tile(0);
      pivot_block_16(S, Xd, rsd, pcol, lt, info, j0, c, lane);
      for (int tile = 8; tile < ntiles; tile += 8) trailing_tile(tile);
    } else {
      for (int tile = wave; tile < ntiles; tile += 8) trailing_tile(tile);
    }
    __syncthreads();
    RC_T(2 + 2 * c);
    // (b) panel: L_rc = S_rc * Xcc^T, r = c+1..7, one tile per wave round-robin
    for (int rb = c + 1 + wave; rb < 8; rb += 8) {
      v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k = 4 * s + fq;
        const double av = S[(16 * rb + fr) * LS + 16 * c + k];          // A[i = fr][k]
        const double bv = Xd[(c * 16 + fr) * XS + k];                   // B[k][j = fr] = Xcc[j][k]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) S[(16 * rb + fq + 4 * q) * LS + 16 * c + fr] = acc[q];
    }
    __syncthreads();
    RC_T(3 + 2 * c);
  }
  }  // MODE < 2 && !REG
  if (MODE < 2) {
  // L back to global (lower + diagonal, zeros above), log-diagonal
  for (int e = t; e < 128 * 64; e += 512) {
    const int i = e >> 6, j = (e & 63) * 2;
    *reinterpret_cast<double2*>(At + (int64_t)i * ld + j) = make_double2((j <= i) ? S[i * LS + j] : 0.0, (j + 1 <= i) ? S[i * LS + j + 1] : 0.0);
  }
  if (t < 128) logdiag[j0 + t] = -log(rsd[t]);
  RC_T(19);
  if (MODE == 1) {                                   // the diagonal-block inverses, where the substitution kernel and MODE 2 find them
    for (int e = t; e < 8 * 16 * 8; e += 512) {
      const int c = e >> 7, i = (e >> 3) & 15, j = (e & 7) * 2;
      *reinterpret_cast<double2*>(invL + (16 * c + i) * 128 + 16 * c + j) =
          make_double2(Xd[(c * 16 + i) * XS + j], Xd[(c * 16 + i) * XS + j + 1]);
    }
    if (REG && t < 128) rhs[j0 + t] = rv[t];         // w_j from the fused forward substitution of chol128_regs
    return;
  }
  }  // MODE < 2

#ifndef RC_DIAG2_NO_INVERSE
  // ------------------------------------------------------------------ inverse by recursive doubling
  // level sb (block size in 16-blocks): pairs p; A part = block rows [2p*sb, 2p*sb+sb), C part = the next sb block rows.
  // A wave's second tile (level 64 only) takes the mirrored column, so that every wave gets the same total k-range; the eight
  // operands of a k-block are fetched from LDS as one batch before its four MFMAs (one exposed LDS latency per k-block, not four).
#pragma unroll 1
  for (int sb = 1; sb <= 4; sb *= 2) {
    const int npairs = 4 / sb, tiles_per_pair = sb * sb, ntiles = npairs * tiles_per_pair;
    auto tile_of = [&](int n, int& rb, int& cb) {
      const int tile = (n == 0) ? wave : 8 + (wave & 4) + (3 - (wave & 3));
      const int p = tile / tiles_per_pair, w = tile - p * tiles_per_pair;
      rb = 2 * p * sb + sb + w / sb;
      cb = 2 * p * sb + w % sb;
      return tile < ntiles;
    };
    const int ntw = (ntiles > 8) ? 2 : 1;
    // phase 1: T_{r,c} = sum_{k in A part, k >= c} L_{r,k} * Ainv_{k,c}  -> stored (transposed) in the X_{r,c} slot
    v4d tacc[2];
    for (int n = 0; n < ntw; ++n) {
      int rb, cb;
      if (!tile_of(n, rb, cb)) continue;
      const int kend = (cb / (2 * sb)) * 2 * sb + sb;             // end of the A part of this pair
      v4d acc = {0.0, 0.0, 0.0, 0.0};
      for (int kb = cb; kb < kend; ++kb) {
        double av[4], bv[4];
        const double* bsrc = (kb == cb) ? Xd + (kb * 16 + fq) * XS + fr : S + (16 * cb + fr) * LS + 16 * kb + fq;   // Ainv_{kb,cb}[k][j]
        const int bstep = (kb == cb) ? 4 * XS : 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          av[s] = S[(16 * rb + fr) * LS + 16 * kb + 4 * s + fq];                        // L_{r,kb}[i][k]
          bv[s] = bsrc[s * bstep];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc, 0, 0, 0);
      }
      tacc[n] = acc;
    }
    __syncthreads();                                  // nobody reads the target slots during phase 1, but keep phases apart
    for (int n = 0; n < ntw; ++n) {
      int rb, cb;
      if (!tile_of(n, rb, cb)) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) S[(16 * cb + fr) * LS + 16 * rb + fq + 4 * q] = tacc[n][q];   // T[i][j] at S[j][i]
    }
    __syncthreads();
    // phase 2: X_{r,c} = - sum_{k in C part, k <= r} Cinv_{r,k} * T_{k,c}
    for (int n = 0; n < ntw; ++n) {
      int rb, cb;
      if (!tile_of(n, rb, cb)) continue;
      const int kbeg = (cb / (2 * sb)) * 2 * sb + sb;             // start of the C part of this pair
      v4d acc = {0.0, 0.0, 0.0, 0.0};
      for (int kb = kbeg; kb <= rb; ++kb) {
        double av[4], bv[4];
        const double* asrc = (kb == rb) ? Xd + (rb * 16 + fr) * XS + fq : S + (16 * kb + fq) * LS + 16 * rb + fr;   // Cinv_{r,kb}[i][k]
        const int astep = (kb == rb) ? 4 : 4 * LS;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          av[s] = asrc[s * astep];                                                      // (negated by the MFMA)
          bv[s] = S[(16 * cb + fr) * LS + 16 * kb + 4 * s + fq];                        // T_{kb,cb}[k][j] at S[j][k]
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc, 0, 0, 1);
      }
      tacc[n] = acc;
    }
    __syncthreads();
    for (int n = 0; n < ntw; ++n) {
      int rb, cb;
      if (!tile_of(n, rb, cb)) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) S[(16 * cb + fr) * LS + 16 * rb + fq + 4 * q] = tacc[n][q];
    }
    __syncthreads();
    RC_T(20 + (sb == 1 ? 0 : sb == 2 ? 1 : 2));
  }

#endif
  // X out (row-major, zeros above the diagonal) and w_j = X * rhs_j
  for (int e = t; e < 128 * 64; e += 512) {
    const int i = e >> 6, j = (e & 63) * 2;
    *reinterpret_cast<double2*>(invL + i * 128 + j) = make_double2((j <= i) ? xval(S, Xd, i, j) : 0.0, (j + 1 <= i) ? xval(S, Xd, i, j + 1) : 0.0);
  }
  if (MODE != 3) {
    const int i = t >> 2, h4 = t & 3;                // 4 threads per row
    double s = 0.0;
    for (int j = h4; j <= i; j += 4) s = __builtin_fma(xval(S, Xd, i, j), rv[j], s);
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (h4 == 0) rhs[j0 + i] = s;
  }
  RC_T(23);
  RC_SPAN(1);
}

template <int MODE, bool REG = false>
__global__ void __launch_bounds__(512) k_diag2(double* __restrict__ A, int64_t ld, double* __restrict__ invL, double* __restrict__ rhs,
                                               double* __restrict__ logdiag, int* __restrict__ info, int64_t j0) {
  extern __shared__ double S[];
  diag2_body<MODE, REG>(S, A, ld, invL, rhs, logdiag, info, j0);
}

// Every 128x128 inverse of the factor's diagonal blocks in ONE launch (block b: L_bb from A, its eight 16x16 diagonal-block inverses from
// invdiag, the full inverse back into invdiag) -- after a factorisation whose chain never formed them (substitution-based solves), and
// only when L^-1 is wanted at all: they are level 0 of the recursive-doubling inverse (rc_trtri).
__global__ void __launch_bounds__(512) k_inv128_batched(double* __restrict__ A, int64_t ld, double* __restrict__ invdiag) {
  extern __shared__ double S[];
  diag2_body<3, false>(S, A, ld, invdiag + (size_t)blockIdx.x * 128 * 128, nullptr, nullptr, nullptr, (int64_t)blockIdx.x * 128);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_diag_loop: the diagonal kernel as ONE RESIDENT workgroup for the whole factorisation (RCGP_DLOOP=1). A kernel trace of the C2
// factorisation shows the chain's whole-CU kernels waiting MILLISECONDS for a CU while window pieces, bulk updates and column work
// keep both slots of every CU turning over out of phase; a workgroup that never leaves its CU cannot starve. It walks the diagonal
// blocks in order: waits until the host-side stream op behind P(jb-1) has raised `ready` to base + jb (hipStreamWriteValue64),
// acquires, factors + inverts block jb exactly as k_diag2<0> does, releases, raises `done` to base + jb + 1 -- the panel solve and
// the tile solve of that block wait for it with hipStreamWaitValue64. Every wait is bounded (wall clock): on a time-out the kernel
// flags the factorisation as failed (info = -9) and raises `done` past the last block, so no stream is left waiting.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) k_diag_loop(double* __restrict__ A, int64_t ld, double* __restrict__ invdiag, double* __restrict__ rhs,
                                                   double* __restrict__ logdiag, int* __restrict__ info, int nblocks,
                                                   unsigned long long* ready, unsigned long long* done, unsigned long long base) {
  extern __shared__ double S[];
  __shared__ int s_abort;
  if (threadIdx.x == 0) s_abort = 0;
  __syncthreads();
  for (int jb = 0; jb < nblocks; ++jb) {
    if (jb > 0) {
      if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < base + (unsigned long long)jb) {
          __builtin_amdgcn_s_sleep(8);
          if (wall_clock64() - t0 > 200000000ll) { s_abort = 1; break; }          // 2 s at 100 MHz
        }
      }
      __syncthreads();
      if (s_abort) {
        if (threadIdx.x == 0) {
          atomicExch(info, -9);
          __hip_atomic_store(done, base + (unsigned long long)nblocks + 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");              // what the update kernels wrote is read from memory, not from a stale line
    }
    diag2_body<0>(S, A, ld, invdiag + (size_t)jb * 128 * 128, rhs, logdiag, info, (int64_t)jb * 128);
    __threadfence();                                                   // L_jj, its inverse, w_j: out to memory before the word that announces them
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(done, base + (unsigned long long)jb + 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int rc_launch_diag_loop(rcgp_handle_s* h, unsigned long long base) {
  const size_t lds = (size_t)(128 * LS + 8 * 16 * XS + 256 + 32 + 256) * sizeof(double);
  if (!h->dloop_attr_set) {
    RC_HIP(hipFuncSetAttribute((const void*)k_diag_loop, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    h->dloop_attr_set = true;
  }
  RcProfScope ps(h, RC_K_DIAG, (double)(h->Np / 128) * 128.0 * 128.0 * 128.0 / 3.0, true);
  RC_LAUNCH(k_diag_loop, dim3(1), dim3(512), lds, h->A, h->Np, h->invdiag, h->w, h->logdiag, h->info, (int)(h->Np / 128),
            (unsigned long long*)h->sig_ready, (unsigned long long*)h->sig_done, base);
  RC_HIP(hipGetLastError());
  return 0;
}

int rc_launch_inv128_batched(rcgp_handle_s* h) {
  const size_t lds = (size_t)(128 * LS + 8 * 16 * XS + 256 + 32 + 256) * sizeof(double);
  static bool attr_set = false;                                // (device state; one device per process in practice, harmless to repeat)
  if (!attr_set) {
    RC_HIP(hipFuncSetAttribute((const void*)k_inv128_batched, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  RcProfScope ps(h, RC_K_DIAG, 0.0, true);
  RC_LAUNCH(k_inv128_batched, dim3((unsigned)(h->Np / 128)), dim3(512), lds, h->A, h->Np, h->invdiag);
  RC_HIP(hipGetLastError());
  h->invdiag_full = true;
  return 0;
}

int rc_launch_diag(rcgp_handle_s* h, int64_t j, int mode) {
  RcProfScope ps(h, RC_K_DIAG, mode == 2 ? 0.0 : 128.0 * 128.0 * 128.0 / 3.0, true);
  double* inv = h->invdiag + (j / 128) * 128 * 128;
  if (h->diag_variant == 1) {
    RC_LAUNCH(k_diag, dim3(1), dim3(512), 0, h->A, h->Np, inv, h->w, h->logdiag, h->info, j);
  } else {
    const size_t lds = (size_t)(128 * LS + 8 * 16 * XS + 256 + 32 + 256) * sizeof(double);
    if (!h->diag_attr_set) {                                   // per handle = per device (the attribute is device state)
      RC_HIP(hipFuncSetAttribute((const void*)k_diag2<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      RC_HIP(hipFuncSetAttribute((const void*)k_diag2<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      RC_HIP(hipFuncSetAttribute((const void*)k_diag2<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      RC_HIP(hipFuncSetAttribute((const void*)k_diag2<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      RC_HIP(hipFuncSetAttribute((const void*)k_diag2<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      h->diag_attr_set = true;
    }
    const bool reg = (h->diag_variant == 3 && h->prep_split != 3) || h->prep_split == 4;
    if (mode == 1) {
      if (reg) RC_LAUNCH((k_diag2<1, true>), dim3(1), dim3(512), lds, h->A, h->Np, inv, h->w, h->logdiag, h->info, j);
      else RC_LAUNCH(k_diag2<1>, dim3(1), dim3(512), lds, h->A, h->Np, inv, h->w, h->logdiag, h->info, j);
    } else if (mode == 2) {
      RC_LAUNCH(k_diag2<2>, dim3(1), dim3(512), lds, h->A, h->Np, inv, h->w, h->logdiag, h->info, j);
    } else {
      if (reg) RC_LAUNCH((k_diag2<0, true>), dim3(1), dim3(512), lds, h->A, h->Np, inv, h->w, h->logdiag, h->info, j);
      else RC_LAUNCH(k_diag2<0>, dim3(1), dim3(512), lds, h->A, h->Np, inv, h->w, h->logdiag, h->info, j);
    }
  }
  RC_HIP(hipGetLastError());
  return 0;
}

// Factor the outer panel of columns [J, Jend): for every 128-column block the diagonal kernel, the panel solve for all rows
// below, and the update of the panel's remaining columns. Launched on h->launch.
static int panel_factor(rcgp_handle_s* h, int64_t J, int64_t Jend) {
  const int64_t Np = h->Np;
  int rc;
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
  return 0;
}

// An event handed to the next launch (h->launch_stop) that no dispatch has taken (a launcher that returned early): record it.
static int flush_stop(rcgp_handle_s* h) {
  if (h->launch_stop) {
    RC_HIP(hipEventRecord(h->launch_stop, h->launch));
    h->launch_stop = nullptr;
  }
  return 0;
}

static int next_event(rcgp_handle_s* h, hipEvent_t* out) {
  if (h->la_cursor == h->la_events.size()) {
    hipEvent_t e;
    RC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    h->la_events.push_back(e);
  }
  *out = h->la_events[h->la_cursor++];
  return 0;
}

// Fine-grained blocked Cholesky. The critical path of the factorisation is the sequence of 128x128 diagonal blocks;
// everything else only has to be ready one step (or one panel) later. Per 128-column block j, on five streams:
//   C  (h->stream2, high priority): D(j) = diagonal kernel; P(j) = the tile (j+1, j) solved and the block (j+1, j+1) updated
//                                   (k_prep1 + k_prep2 on several CUs, or k_prep_next on one). D(j+1) follows P(j) in stream
//                                   order, so a chain step costs D + P, not D + T + G.
//   B  (h->stream5, high priority): T2(j) = panel solve of the rows from block j+2 on (after D(j));
//                                   near G(j) = K=128 update of the two block columns j+1, j+2 -- all that P(j+1) reads -- for the
//                                   rows from block j+2 on (after the solved tile of P(j) and the far part of G(j-1));
//   B2 (h->stream6, high priority): far G(j) = the same update of the block columns [j+3, cend) (after T2(j), in stream order
//                                   behind the earlier far parts; it is the far part that waits for the window piece below).
//   U1 (h->stream, main)          : when panel p = [pend-NB, pend) is complete, its K=NB update of the next `depth` column panels
//                                   from u0 on, one kernel each, nearest first (window pieces);
//   U2 (h->stream3)               : ... and of everything beyond them (the bulk of the flops), concurrently with the next chain.
// cend = pend + EXT: the G updates reach EXT columns past their own panel, so the first blocks of the NEXT panel are already
// up to date when the chain arrives there (u0 = pend + EXT): the chain itself never waits for a window piece, only the far part
// (and the near part once it reaches column u0) does. Events ride on the dispatches (RC_LAUNCH, h->launch_stop).
static int potrf_fine(rcgp_handle_s* h, bool overlap_inverse) {
  const int64_t Np = h->Np, NB = h->nb_outer, EXT = 128 * (int64_t)h->chain_ext;
  hipStream_t C = h->stream2, B = h->stream5, U1 = h->stream, U2 = h->stream3;
  int rc;
  hipEvent_t e0;
  if ((rc = next_event(h, &e0))) return rc;
  RC_HIP(hipEventRecord(e0, h->stream));
  RC_HIP(hipStreamWaitEvent(C, e0, 0));
  RC_HIP(hipStreamWaitEvent(B, e0, 0));
  hipEvent_t eG_prev = nullptr, eU1_prev = nullptr, eU2_prev = nullptr, eFar_prev = nullptr, eCU_prev1 = nullptr, eCU_prev2 = nullptr;
  hipStream_t B2 = h->stream6;
  RC_HIP(hipStreamWaitEvent(B2, e0, 0));
  bool near_waited = false, far_waited = false;                  // this panel's wait for the previous panel's window piece
  int64_t u0_prev = 0;                                           // first column of that piece
  // heavy: pieces + bulk of a finished panel are ONE persistent k_heavy_update on U2; "the window piece is done" is then a value in
  // the handle's signal word, waited for with hipStreamWaitValue64, not an event
  const bool heavy = h->heavy_mode && h->sig_flag && h->heavy_ctr;
  uint64_t hv_prev = 0;
  bool have_win = false;                                         // a previous panel's first column panel has to be waited for
  auto wait_window = [&](hipStream_t st) -> hipError_t {
    if (heavy) return hipStreamWaitValue64(st, h->sig_flag, hv_prev, hipStreamWaitValueGte, 0xffffffffffffffffull);
    return hipStreamWaitEvent(st, eU1_prev, 0);
  };
  const bool ext = h->ext_events && !h->profiling;              // (the profiling bracket records its own events around a launch)
  // late: the diagonal kernel only factors (k_diag2<1>); the 128x128 inverse and w_j follow on the column-work stream (k_diag2<2>, ahead
  // of the panel solve that needs them) and the chain's tile is solved by substitution (k_prep1s)
  const bool late = (h->prep_split == 3) && h->diag_variant != 1;
  // subst: no 128x128 inverse on the chain at all -- the diagonal kernel factors only (register-resident core, w_j fused), the chain's
  // tile and the column below are solved by blocked substitution (k_trsm_subst; the chain's instance also updates the next diagonal
  // block), and the inverses the L^-1 stage starts from are formed afterwards in one batched launch (rc_trtri_advance)
  const bool subst = (h->prep_split == 4) && h->diag_variant != 1;
  h->invdiag_full = !subst;
  // t2p: the panel solve T2(j) waits for P(j)'s solved tile instead of D(j), so that D(j) carries no completion event (a dispatch that
  // carries one delays its successor on the stream by ~5 us: kernel trace)
  const bool t2p = h->t2_after_p && !late;
  // dloop: the diagonal blocks are factored by ONE resident workgroup (k_diag_loop on stream4, launched here, before anything can crowd
  // it out); "block j is factored" and "block j may be factored" are values in two signal words, waited for / raised by stream
  // memory operations, and the chain's tile solve is the small-LDS k_prep1g
  const bool dloop = h->dloop && h->sig_ready && h->sig_done && h->stream4 && !late && h->diag_variant != 1 && h->chain_ext >= 2 && !h->profiling;
  uint64_t dl_base = 0;
  hipEvent_t eDL = nullptr;
  if (dloop) {
    h->dl_base += (1ull << 20);
    dl_base = h->dl_base;
    if ((rc = next_event(h, &eDL))) return rc;
    RC_HIP(hipStreamWaitEvent(h->stream4, e0, 0));
    h->launch = h->stream4;
    if (ext) h->launch_stop = eDL;
    if ((rc = rc_launch_diag_loop(h, dl_base)) || (rc = flush_stop(h))) return rc;
    if (!ext) RC_HIP(hipEventRecord(eDL, h->stream4));
  }
  for (int64_t j = 0; j < Np; j += 128) {
    const int64_t below = Np - (j + 128);
    hipEvent_t eD = nullptr, eP, eG;
    if ((below > 0 || late) && (rc = next_event(h, &eD))) return rc;
    h->launch = C;
    if (dloop) {
      // D(j) runs inside the resident kernel: C and B wait for its word instead of an event
      const uint64_t want = dl_base + (uint64_t)(j / 128) + 1;
      RC_HIP(hipStreamWaitValue64(C, h->sig_done, want, hipStreamWaitValueGte, 0xffffffffffffffffull));
      if (below > 0) RC_HIP(hipStreamWaitValue64(B, h->sig_done, want, hipStreamWaitValueGte, 0xffffffffffffffffull));
    } else {
    if (ext && !t2p) h->launch_stop = eD;
    if ((rc = rc_launch_diag(h, j, (late || subst) ? 1 : 0)) || (rc = flush_stop(h))) return rc;
    }
    if (late) {
      if (!ext) RC_HIP(hipEventRecord(eD, C));
      RC_HIP(hipStreamWaitEvent(B, eD, 0));
      h->launch = B;
      if ((rc = rc_launch_diag(h, j, 2))) return rc;              // inverse + w_j (+ this block's rhs rows from the tile to its left)
      h->launch = C;
    }
    if (below <= 0) break;
    const int64_t pend = (j / NB + 1) * NB;                      // end of the panel block j belongs to
    const int64_t cend = (pend + EXT < Np) ? pend + EXT : Np;    // G(j) covers the block columns [j + 128, cend)
    const bool first_of_panel = (j > 0 && j % NB == 0);
    double* P = h->A + (j + 128) * Np + j;                       // rows below the diagonal block, 128 columns
    const double* inv = h->invdiag + (j / 128) * 128 * 128;
    if ((rc = next_event(h, &eP)) || (rc = next_event(h, &eG))) return rc;
    if (!late && !t2p && !dloop) {
      if (!ext) RC_HIP(hipEventRecord(eD, C));
      RC_HIP(hipStreamWaitEvent(B, eD, 0));
    }
    if (eG_prev) RC_HIP(hipStreamWaitEvent(C, eG_prev, 0));
    if (first_of_panel && have_win && h->chain_ext < 2) RC_HIP(wait_window(C));   // P touches column j + 128 >= u0
    if (ext) h->launch_stop = eP;                                 // (with the split: taken by k_prep1 -- the column work needs the solved tile only)
    if (subst)
      rc = rc_launch_chain_tile(h, P, h->A + (j + 128) * Np + (j + 128), Np, h->A + j * Np + j, inv, h->w + j + 128, h->w + j);
    else if (late)
      rc = rc_launch_prep_subst(h, P, h->A + (j + 128) * Np + (j + 128), Np, h->A + j * Np + j, inv);
    else if (dloop)
      rc = (h->dloop == 2)
               ? rc_launch_prep_g(h, P, h->A + (j + 128) * Np + (j + 128), Np, inv, h->w + j + 128, h->w + j)
               : rc_launch_prep_q(h, P, h->A + (j + 128) * Np + (j + 128), Np, inv, h->w + j + 128, h->w + j, h->heavy_ctr + 2 * RC_MAX_PANELS,
                                  (unsigned long long*)h->sig_ready, dl_base + (uint64_t)(j / 128) + 1);
    else if (h->prep_small)
      rc = rc_launch_prep_q(h, P, h->A + (j + 128) * Np + (j + 128), Np, inv, h->w + j + 128, h->w + j, nullptr, nullptr, 0);
    else if (h->prep_split)
      rc = rc_launch_prep_split(h, P, h->A + (j + 128) * Np + (j + 128), Np, inv, h->w + j + 128, h->w + j);
    else
      rc = rc_launch_prep_next(h, P, h->A + (j + 128) * Np + (j + 128), Np, inv, h->w + j + 128, h->w + j);
    if (rc || (rc = flush_stop(h))) return rc;
    if (!ext) RC_HIP(hipEventRecord(eP, C));
    if (dloop && h->dloop == 2)                                   // (variant 2: a stream memory op raises the word; variant 1: k_prep2r's last workgroup)
      RC_HIP(hipStreamWriteValue64(C, h->sig_ready, dl_base + (uint64_t)(j / 128) + 1, 0));
    h->launch = B;
    if (t2p) RC_HIP(hipStreamWaitEvent(B, eP, 0));
    hipEvent_t ePanel = eG;                                       // everything of this step on B (and B2) done
    if (below > 128 && h->chain_split) {
      // G(j) in two kernels: the two block columns the NEXT chain step reads (near, on B; P(j+1) waits for it alone) and the
      // rest (far, on B2, in order behind the earlier far updates). Only the far part, and the near part once it reaches the
      // columns of the previous panel's window piece, wait for that piece: the chain does not stop at a panel boundary.
      hipEvent_t eT2, eFar;
      if ((rc = next_event(h, &eT2))) return rc;
      if (ext) h->launch_stop = eT2;
      if ((rc = subst ? rc_launch_trsm_subst(h, P + 128 * Np, Np, h->A + j * Np + j, inv, below - 128, h->w + j + 256, h->w + j)
                      : rc_launch_trsm_panel(h, P + 128 * Np, Np, inv, below - 128, h->w + j + 256, h->w + j)) ||
          (rc = flush_stop(h)))
        return rc;
      if (!ext) RC_HIP(hipEventRecord(eT2, B));
      const int64_t c0 = j + 128, nend = (c0 + 256 < cend) ? c0 + 256 : cend;
      // Catch-up mode (RCGP_CATCHUP = t, off by default) for the block columns taller than t blocks: instead of a K=128 far update at
      // every step -- one read-modify-write of the whole column per step, which is what makes the column work HBM-heavy -- a column c
      // receives ONE update with every finished column its window pieces do not deliver, [lo(c), c - 3 blocks), when it is four blocks
      // ahead of the chain (far stream; two steps of slack before anything waits for it), then the steps c-3 and c-2 together (K=256)
      // when it enters the near window as its second column, then step c-1 (K=128) as its first column: three passes instead of
      // ~EXT + NB/128. Panel q reaches column c through its window pieces iff c >= (q + 1) NB + EXT, so lo(c) = floor((c - EXT)/NB) NB;
      // EXT >= 3 blocks keeps the two steps of the K=256 update on the chain's side of that boundary.
      const int64_t ccut = (h->catchup_blocks > 0 && h->chain_ext >= 3 && Np > 128 * (int64_t)h->catchup_blocks)
                               ? Np - 128 * (int64_t)h->catchup_blocks : 0;
      const int64_t c2 = c0 + 128;                               // second near column
      const bool c2_catchup = (nend - c0 == 256 && c2 < ccut && j >= 128);
      RC_HIP(hipStreamWaitEvent(B, eP, 0));
      if (c2_catchup) {
        if (eCU_prev2) RC_HIP(hipStreamWaitEvent(B, eCU_prev2, 0));   // the catch-up of column c2 was issued two steps ago
      } else if (eFar_prev) {
        RC_HIP(hipStreamWaitEvent(B, eFar_prev, 0));
      }
      if (have_win && !near_waited && nend > u0_prev) { RC_HIP(wait_window(B)); near_waited = true; }
      if (c2_catchup) {
        if ((rc = rc_launch_gemm_nt_sub(h, h->A + (j + 256) * Np + c0, Np, P + 128 * Np, Np, P, Np, below - 128, 128, 128, j + 256, c0))) return rc;
        if (ext) h->launch_stop = eG;
        if ((rc = rc_launch_gemm_nt_sub(h, h->A + (j + 256) * Np + c2, Np, h->A + (j + 256) * Np + (j - 128), Np, h->A + c2 * Np + (j - 128), Np,
                                        below - 128, 128, 256, j + 256, c2)) ||
            (rc = flush_stop(h)))
          return rc;
      } else {
        if (ext) h->launch_stop = eG;
        if ((rc = rc_launch_gemm_nt_sub(h, h->A + (j + 256) * Np + c0, Np, P + 128 * Np, Np, P, Np, below - 128, nend - c0, 128, j + 256, c0)) ||
            (rc = flush_stop(h)))
          return rc;
      }
      if (!ext) RC_HIP(hipEventRecord(eG, B));
      const int64_t cc = c0 + 384;                               // the column four blocks ahead
      const int64_t lo = ((cc >= EXT) ? (cc - EXT) / NB : 0) * NB;
      const bool do_cu = (cc < cend && cc < ccut && j + 128 > lo);
      const int64_t f0 = (nend > ccut) ? nend : ccut;
      const bool do_far = (cend > f0);
      hipEvent_t eCU = nullptr;
      if (do_cu || do_far) {
        RC_HIP(hipStreamWaitEvent(B2, eT2, 0));
        if (have_win && !far_waited && ((do_far && cend > u0_prev) || (do_cu && cc + 128 > u0_prev))) {
          RC_HIP(wait_window(B2));
          far_waited = true;
        }
        h->launch = B2;
        if (do_cu) {
          if ((rc = next_event(h, &eCU))) return rc;
          if (ext) h->launch_stop = eCU;
          if ((rc = rc_launch_gemm_nt_sub(h, h->A + cc * Np + cc, Np, h->A + cc * Np + lo, Np, h->A + cc * Np + lo, Np, Np - cc, 128,
                                          j + 128 - lo, cc, cc)) ||
              (rc = flush_stop(h)))
            return rc;
          if (!ext) RC_HIP(hipEventRecord(eCU, B2));
        }
        if (do_far) {
          if ((rc = next_event(h, &eFar))) return rc;
          if (ext) h->launch_stop = eFar;
          if ((rc = rc_launch_gemm_nt_sub(h, h->A + (j + 256) * Np + f0, Np, P + 128 * Np, Np, P + (f0 - c0) * Np, Np, below - 128, cend - f0, 128,
                                          j + 256, f0)) ||
              (rc = flush_stop(h)))
            return rc;
          if (!ext) RC_HIP(hipEventRecord(eFar, B2));
          eFar_prev = eFar;
        }
      }
      eCU_prev2 = eCU_prev1;
      eCU_prev1 = eCU;
      if (j + 128 == pend) {                                      // the outer updates need both halves
        if ((rc = next_event(h, &ePanel))) return rc;
        if (eFar_prev) RC_HIP(hipStreamWaitEvent(B, eFar_prev, 0));
        RC_HIP(hipEventRecord(ePanel, B));
      }
    } else if (below > 128) {
      if ((rc = subst ? rc_launch_trsm_subst(h, P + 128 * Np, Np, h->A + j * Np + j, inv, below - 128, h->w + j + 256, h->w + j)
                      : rc_launch_trsm_panel(h, P + 128 * Np, Np, inv, below - 128, h->w + j + 256, h->w + j)))
        return rc;
      if (first_of_panel && have_win) RC_HIP(wait_window(B));
      RC_HIP(hipStreamWaitEvent(B, eP, 0));
      if (ext) h->launch_stop = eG;
      if ((rc = rc_launch_gemm_nt_sub(h, h->A + (j + 256) * Np + (j + 128), Np, P + 128 * Np, Np, P, Np, below - 128, cend - (j + 128), 128,
                                      j + 256, j + 128)) ||
          (rc = flush_stop(h)))
        return rc;
      if (!ext) RC_HIP(hipEventRecord(eG, B));
    } else {
      if (late) RC_HIP(hipStreamWaitEvent(B, eP, 0));            // the last block's inverse kernel reads the tile P(j) solves
      RC_HIP(hipEventRecord(eG, B));
    }
    eG_prev = eG;
    if (j + 128 == pend) {                                        // panel [pend - NB, pend) is final once B has finished this step
      const hipEvent_t eG = ePanel;                               // (shadows: the outer updates wait for the whole step)
      near_waited = far_waited = false;
      u0_prev = pend + EXT;
      // Outer (K = NB) updates with the finished panel, by target column panel: the next `depth` (shifted) panels one kernel
      // each, in column order on the main stream -- the first one is what the chain is waiting for -- and everything beyond
      // them in one bulk kernel on the bulk stream. A column panel leaves the bulk kernel's domain one panel before the
      // chain reaches it with depth 1, `depth` panels before with a deeper window: the chain may run that far ahead of the bulk.
      eU1_prev = nullptr;
      have_win = false;
      const int depth = h->chain_depth;
      const double* Lp0 = h->A + (pend - NB);                     // column offset of the finished panel
      hipEvent_t eR_new = nullptr;
      if (heavy) {
        const int64_t u0 = pend + EXT;
        const int64_t pidx = pend / NB - 1;
        if (u0 < Np && pidx < RC_MAX_PANELS) {
          if ((rc = next_event(h, &eR_new))) return rc;
          RC_HIP(hipStreamWaitEvent(U2, eG, 0));
          h->launch = U2;
          if (ext) h->launch_stop = eR_new;
          hv_prev = ++h->sig_value;
          if ((rc = rc_launch_heavy_update(h, h->A + u0 * Np + u0, Np, Lp0 + u0 * Np, Np, Np - u0, NB, NB, h->heavy_ctr + 2 * pidx, hv_prev)) ||
              (rc = flush_stop(h)))
            return rc;
          if (!ext) RC_HIP(hipEventRecord(eR_new, U2));
          have_win = true;
        }
      } else {
      for (int q = 0; q <= depth; ++q) {
        const int64_t u0 = pend + EXT + (int64_t)q * NB;
        if (u0 >= Np) break;
        const int64_t u1 = (u0 + NB < Np) ? u0 + NB : Np;
        if (q < depth) {                                          // window piece
          // pieces_on_bulk: the pieces go down the bulk stream, ahead of their panel's bulk kernel -- with RCGP_RESERVE_CUS that is ONE
          // CU-masked queue for every K = NB kernel (two active masked queues put the runtime in its slow regime)
          hipStream_t PS = h->pieces_on_bulk ? U2 : U1;
          if (q == 0) RC_HIP(hipStreamWaitEvent(PS, eG, 0));
          if (q == depth - 1 && eU2_prev && PS != U2) RC_HIP(hipStreamWaitEvent(PS, eU2_prev, 0));   // this panel was in the previous bulk kernel
          h->launch = PS;
          hipEvent_t eU1 = nullptr;
          if (q == 0) {
            if ((rc = next_event(h, &eU1))) return rc;
            if (ext) h->launch_stop = eU1;
          }
          if ((rc = rc_launch_gemm_nt_sub(h, h->A + u0 * Np + u0, Np, Lp0 + u0 * Np, Np, Lp0 + u0 * Np, Np, Np - u0, u1 - u0, NB, u0, u0)) ||
              (rc = flush_stop(h)))
            return rc;
          if (q == 0) {
            if (!ext) RC_HIP(hipEventRecord(eU1, PS));
            eU1_prev = eU1;
            have_win = true;
          }
        } else {                                                  // bulk: everything from u0 on
          if ((rc = next_event(h, &eR_new))) return rc;
          RC_HIP(hipStreamWaitEvent(U2, eG, 0));
          // bulk_after_piece: the bulk kernel of this panel starts only when the panel's FIRST window piece -- the update the chain is
          // waiting for -- is done, so that the piece has the chip (beside the previous bulk kernel's tail) instead of a third of it
          if (h->bulk_after_piece && eU1_prev) RC_HIP(hipStreamWaitEvent(U2, eU1_prev, 0));
          h->launch = U2;
          if (ext) h->launch_stop = eR_new;
          if ((rc = rc_launch_syrk_lower(h, h->A + u0 * Np + u0, Np, Lp0 + u0 * Np, Np, Np - u0, NB)) || (rc = flush_stop(h))) return rc;
          if (!ext) RC_HIP(hipEventRecord(eR_new, U2));
        }
      }
      }
      eU2_prev = eR_new;
      if (overlap_inverse && (pend / NB) % h->inv_every == 0) {   // rows < pend of L are final: feed the L^-1 kernels that only need those
        RC_HIP(hipStreamWaitEvent(h->stream4, eG, 0));
        h->launch = h->stream4;
        if ((rc = rc_trtri_advance(h, pend))) return rc;
      }
    }
  }
  h->launch = h->stream;
  hipEvent_t eC, eB, eB2;
  if ((rc = next_event(h, &eC)) || (rc = next_event(h, &eB)) || (rc = next_event(h, &eB2))) return rc;
  RC_HIP(hipEventRecord(eC, C));
  RC_HIP(hipEventRecord(eB, B));
  RC_HIP(hipEventRecord(eB2, B2));
  RC_HIP(hipStreamWaitEvent(h->stream, eC, 0));
  RC_HIP(hipStreamWaitEvent(h->stream, eB, 0));
  RC_HIP(hipStreamWaitEvent(h->stream, eB2, 0));
  if (eDL) RC_HIP(hipStreamWaitEvent(h->stream, eDL, 0));
  if (eU2_prev) RC_HIP(hipStreamWaitEvent(h->stream, eU2_prev, 0));
  if (overlap_inverse) {
    RC_HIP(hipEventRecord(h->ev_inv, h->stream4));
    RC_HIP(hipStreamWaitEvent(h->stream, h->ev_inv, 0));
  }
  return 0;
}

// Right-looking blocked Cholesky with one-panel look-ahead: as soon as the trailing update has finished the NEXT panel's
// columns, that panel is factored on the side streams while the bulk stream updates the rest of the trailing matrix.
int rc_potrf(rcgp_handle_s* h) {
  const int64_t Np = h->Np, NB = h->nb_outer;
  int rc;
  h->launch = h->stream;
  h->launch_stop = nullptr;                                        // (an earlier call may have failed half-way)
  h->prof_pending = -1;
  RC_HIP(hipMemcpyAsync(h->w, h->y, (size_t)Np * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  RC_HIP(hipMemsetAsync(h->info, 0, sizeof(int), h->stream));
  if (h->heavy_ctr) RC_HIP(hipMemsetAsync(h->heavy_ctr, 0, 2 * RC_MAX_PANELS * sizeof(int), h->stream));   // tile / completion counters of the heavy updates
  const int64_t npanels = (Np + NB - 1) / NB;
  const bool la = h->lookahead && Np >= 4 * 128;                   // (the multi-stream schedule needs no minimum number of panels)
  const bool inv = la && h->overlap_inverse;                      // feed L^-1 kernels into the idle CUs of the chain-bound tail
  h->tt_active = false;                                            // any earlier incremental schedule is void: L is being rebuilt
  h->la_cursor = 0;
  const bool fine = la && h->fine_chain;
  if (inv && (rc = rc_trtri_begin(h))) return rc;
  h->gram_fresh = false;                                           // consumed, whatever happens below
  if (fine) {
    if ((rc = potrf_fine(h, inv))) return rc;
    h->factored = true;
    h->inverted = false;
    return 0;
  }
  h->invdiag_full = true;                                          // (the coarse schedule's diagonal kernel forms every inverse)
  if ((rc = panel_factor(h, 0, NB < Np ? NB : Np))) return rc;
  for (int64_t J = 0; J + NB < Np; J += NB) {
    const int64_t Jend = J + NB;
    const int64_t Jend2 = (Jend + NB < Np) ? Jend + NB : Np;
    const double* P = h->A + Jend * Np + J;                      // panel J below its own rows: (Np - Jend) x NB
    // (main) the next panel's columns first
    if ((rc = rc_launch_gemm_nt_sub(h, h->A + Jend * Np + Jend, Np, P, Np, P, Np, Np - Jend, Jend2 - Jend, NB, Jend, Jend))) return rc;
    if (la) {
      hipEvent_t ev_next, ev_panel, ev_rest;
      if ((rc = next_event(h, &ev_next)) || (rc = next_event(h, &ev_rest))) return rc;
      RC_HIP(hipEventRecord(ev_next, h->stream));
      // (side streams) factor panel J+1
      if ((rc = next_event(h, &ev_panel))) return rc;
      RC_HIP(hipStreamWaitEvent(h->stream2, ev_next, 0));
      h->launch = h->stream2;
      rc = panel_factor(h, Jend, Jend2);
      h->launch = h->stream;
      if (rc) return rc;
      RC_HIP(hipEventRecord(ev_panel, h->stream2));
      if (inv) {                                                   // rows < Jend2 of L are final once panel J+1 is done
        RC_HIP(hipStreamWaitEvent(h->stream4, ev_panel, 0));
        h->launch = h->stream4;
        rc = rc_trtri_advance(h, Jend2);
        h->launch = h->stream;
        if (rc) return rc;
      }
      // (bulk stream) the rest of the trailing matrix, concurrently with the panel
      RC_HIP(hipStreamWaitEvent(h->stream3, ev_next, 0));
      if (Np - Jend2 > 0) {
        h->launch = h->stream3;
        rc = rc_launch_syrk_lower(h, h->A + Jend2 * Np + Jend2, Np, h->A + Jend2 * Np + J, Np, Np - Jend2, NB);
        h->launch = h->stream;
        if (rc) return rc;
      }
      RC_HIP(hipEventRecord(ev_rest, h->stream3));
      RC_HIP(hipStreamWaitEvent(h->stream, ev_panel, 0));
      RC_HIP(hipStreamWaitEvent(h->stream, ev_rest, 0));
    } else {
      if (Np - Jend2 > 0) {
        if ((rc = rc_launch_syrk_lower(h, h->A + Jend2 * Np + Jend2, Np, h->A + Jend2 * Np + J, Np, Np - Jend2, NB))) return rc;
      }
      if ((rc = panel_factor(h, Jend, Jend2))) return rc;
    }
  }
  if (inv) {
    RC_HIP(hipEventRecord(h->ev_inv, h->stream4));
    RC_HIP(hipStreamWaitEvent(h->stream, h->ev_inv, 0));
  }
  h->factored = true;
  h->inverted = false;
  return 0;
}
