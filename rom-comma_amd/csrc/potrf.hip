// Blocked right-looking Cholesky, in place on the lower triangle of A (K + noise I), fused with the forward substitution
// w = L^-1 y. Replaces tf.linalg.cholesky / triangular_solve inside GPflow's GPR.log_marginal_likelihood
// (reference call sites gpr/models.py:360, 427-439).
//
// Two-level blocking: outer panels of NB columns (K of the big MFMA updates; h->nb_outer, 1024), inner blocks of 128. Nothing on the
// factorisation's critical path forms a 128x128 inverse:
//   k_diag_factor : one workgroup factors the 128x128 diagonal block with the trailing matrix in registers (chol128_regs, pivot16),
//                   emits L_jj, log L_ii, the eight 16x16 diagonal-block inverses and w_j = L_jj^-1 rhs_j
//   k_trsm_subst  : (gemm.hip) rows below <- rows below * L_jj^-T by blocked substitution against L_jj, rhs update fused; the chain's
//                   own tile as eight one-strip workgroups, the rest of the column in 64-row workgroups
//   k_prep2       : (gemm.hip) the next diagonal block's last update, from the chain's tile
//   k_gemm_nt_sub : K = 128 update of the block columns the chain keeps current; K = NB update of the window's column panels
//   k_syrk_lower  : bulk trailing update with the whole outer panel (K = NB)
//   k_inv128_batched : afterwards and only when L^-1 is wanted: every 128x128 diagonal-block inverse in one launch (level 0 of rc_trtri)
// Scheduling (main stream + chain + two column-work streams + bulk, events only): potrf_fine below; rc_potrf keeps the simpler
// coarse schedule behind RCGP_FINE=0 / RCGP_LOOKAHEAD=0 and for matrices of fewer than four blocks.
// (Round 2's chain -- explicit inverse inside the diagonal kernel, GEMM-form solves -- and its schedule experiments: branch exp/r2-schedule-variants.)
#include "common.h"

// Newton-refined reciprocal square root (v_rsq_f64 seed, 2^-24 relative: tools/fp64_latency.hip): relative error ~1 ulp.
__device__ __forceinline__ double rc_rsqrt(double d) {
  double y = __builtin_amdgcn_rsq(d);
  y = y * __builtin_fma(-0.5 * d * y, y, 1.5);
  y = y * __builtin_fma(-0.5 * d * y, y, 1.5);
  return y;
}

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

// LDS of the two diagonal-block kernels (dynamic): S[128][LS] -- L in its lower triangle (and, in the inverse kernel, the off-diagonal
// blocks of X = L^-1 TRANSPOSED in the upper triangle) -- then Xd[8][16][XS] (the 16x16 diagonal-block inverses), rsd[128] (1 / L_ii),
// rv[128] (right-hand side), pcol[32] (two pivot-column lines), lt[256] (transposed copy of the current diagonal 16-block).
#define LS 129          // ODD: a row per lane (the pivot block's loads / stores) and a row per fragment lane (the MFMA operand reads) both
                        // fall on 16 different bank pairs (130: lanes r and r + 8 collide)
#define XS 18
#define RC_DIAG_LDS ((size_t)(128 * LS + 8 * 16 * XS + 256 + 32 + 256) * sizeof(double))

__device__ __forceinline__ double rl_d(double v, int lane) {      // broadcast lane `lane` (compile-time constant) of v
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------------------------------
// pivot16: Cholesky of one 16x16 pivot block and its inverse by ONE wave (lane l mirrors row / column l & 15), written for a SHORT
// dependent chain and FEW registers, so that it can be inlined at its single call site in chol128_regs. (Round 2's form was a
// noinline function with two call sites: its call alone cost ~2.5 us per block -- the callee saved and restored its 112 callee-saved
// VGPRs through scratch memory on every call, which the kernel-resource-usage remark of the calling KERNEL does not show -- and per
// pivot it waited for an LDS round trip in the middle of an 8-operation fp64 chain. In-kernel stamps: 7.0 -> 4.0 us per pivot block.)
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
  double dv = rl_d(a[0], 0), dsave = 1.0;
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
    dsave = (r == j) ? dd : dsave;                               // lane j keeps pivot j
  }
  // 1 / sqrt(d_j) for all sixteen pivots at once (lane j its own: one Newton-refined rsqrt per lane instead of one per pivot and wave),
  // handed round through rsd, then the columns scaled: L[r][j] = A[r][j] / sqrt(d_j)
  if (lane < 16) rsdc[r] = rc_rsqrt(dsave);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const double2 v = *reinterpret_cast<const double2*>(rsdc + 2 * q);
    a[2 * q] *= v.x;
    a[2 * q + 1] *= v.y;
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
// chol128_regs: the 16-blocked Cholesky of the 128x128 block with the TRAILING MATRIX IN REGISTERS (k_diag_factor).
// In round 2's kernel every trailing tile update was LDS -> accumulator -> 4 MFMAs -> LDS (12 LDS reads + 4 writes per tile and
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
    if (wave == c) {                                  // pivot block c: the owner puts its diagonal tile where pivot16 works in place
#pragma unroll
      for (int q = 0; q < 4; ++q) S[(16 * c + fr) * LS + 16 * c + 4 * q + fq] = acc[0][q];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // (an opaque zero per iteration: without it the compiler hoists the ~200 loop-invariant LDS addresses of the unrolled pivot
      // sweep out of the c loop and keeps each in a register of its own -- 140 SGPRs spilled to VGPR lanes, 30 VGPRs to scratch)
      int zero = 0;
      asm volatile("" : "+s"(zero));
      pivot16(S + (16 * c) * LS + 16 * c, Xd + c * 16 * XS, rsd + 16 * c, pcol + zero, lt + zero, info, j0 + 16 * c, lane + zero, c == 0);
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

// The diagonal kernel of the chain: factor the block at (j0, j0) in place, emit log L_ii, the eight 16x16 diagonal-block inverses (into the
// diagonal blocks of invL, where k_trsm_subst and k_inv128_batched find them) and w_j = L_jj^-1 rhs_j (in place in rhs).
__global__ void __launch_bounds__(512) k_diag_factor(RcBP<double> Ab, int64_t ld, RcBP<double> invLb, RcBP<double> rhsb, RcBP<double> logdiagb,
                                                     RcBP<int> infob, int64_t j0) {
  extern __shared__ double S[];
  double* __restrict__ A = Ab.p[blockIdx.z];                      // one workgroup per unit of the batch
  double* __restrict__ invL = invLb.p[blockIdx.z];
  double* __restrict__ rhs = rhsb.p[blockIdx.z];
  double* __restrict__ logdiag = logdiagb.p[blockIdx.z];
  int* __restrict__ info = infob.p[blockIdx.z];
  double* Xd = S + 128 * LS;
  double* rsd = Xd + 8 * 16 * XS;
  double* rv = rsd + 128;
  double* pcol = rv + 128;
  double* lt = pcol + 32;
  const int t = threadIdx.x;
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
  if (t < 128) rv[t] = rhs[j0 + t];
  __syncthreads();
  RC_T(1);
  chol128_regs(S, Xd, rsd, pcol, lt, info, j0, rv);
  // L back to global (lower + diagonal, zeros above), log-diagonal, the diagonal-block inverses, w_j
  for (int e = t; e < 128 * 64; e += 512) {
    const int i = e >> 6, j = (e & 63) * 2;
    *reinterpret_cast<double2*>(At + (int64_t)i * ld + j) = make_double2((j <= i) ? S[i * LS + j] : 0.0, (j + 1 <= i) ? S[i * LS + j + 1] : 0.0);
  }
  if (t < 128) logdiag[j0 + t] = -log(rsd[t]);              // log L_ii = log sqrt(d) = -log(1/sqrt(d))
  for (int e = t; e < 8 * 16 * 8; e += 512) {
    const int c = e >> 7, i = (e >> 3) & 15, j = (e & 7) * 2;
    *reinterpret_cast<double2*>(invL + (16 * c + i) * 128 + 16 * c + j) = make_double2(Xd[(c * 16 + i) * XS + j], Xd[(c * 16 + i) * XS + j + 1]);
  }
  if (t < 128) rhs[j0 + t] = rv[t];
  RC_T(19);
  RC_SPAN(1);
}

// element (i, j), i >= j, of X = L^-1 in its split storage
__device__ __forceinline__ double xval(const double* S, const double* Xd, int i, int j) {
  return ((i >> 4) == (j >> 4)) ? Xd[((i >> 4) * 16 + (i & 15)) * XS + (j & 15)] : S[j * LS + i];
}

// Every 128x128 inverse of the factor's diagonal blocks in ONE launch, straight into the diagonal blocks of W = L^-1 (block b: L_bb from
// A, its eight 16x16 diagonal-block inverses from invdiag). They are level 0 of the recursive-doubling inverse (rc_trtri); the
// factorisation itself never needs them. Recursive doubling inside the block too (block sizes 16, 32, 64) on fp64 MFMA from LDS:
// X21 = -C^-1 (B A^-1); the off-diagonal blocks of X live TRANSPOSED in the upper triangle of S.
__global__ void __launch_bounds__(512) k_inv128_batched(RcBP<const double> Ab, int64_t ld, RcBP<const double> invdiagb, RcBP<double> Wb,
                                                        int64_t ldw) {
  extern __shared__ double S[];
  const double* __restrict__ A = Ab.p[blockIdx.z];
  const double* __restrict__ invdiag = invdiagb.p[blockIdx.z];
  double* __restrict__ W = Wb.p[blockIdx.z];
  double* Xd = S + 128 * LS;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int64_t j0 = (int64_t)blockIdx.x * 128;
  const double* At = A + j0 * ld + j0;
  const double* invL = invdiag + (size_t)blockIdx.x * 128 * 128;
  {
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
  for (int e = t; e < 8 * 16 * 16; e += 512) {
    const int c = e >> 8, i = (e >> 4) & 15, j = e & 15;
    Xd[(c * 16 + i) * XS + j] = invL[(16 * c + i) * 128 + 16 * c + j];
  }
  __syncthreads();
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
  }

  double* Wt = W + j0 * ldw + j0;
  for (int e = t; e < 128 * 64; e += 512) {
    const int i = e >> 6, j = (e & 63) * 2;
    *reinterpret_cast<double2*>(Wt + (int64_t)i * ldw + j) = make_double2((j <= i) ? xval(S, Xd, i, j) : 0.0, (j + 1 <= i) ? xval(S, Xd, i, j + 1) : 0.0);
  }
}

static int set_diag_attributes(rcgp_handle_s* h) {
  if (!h->diag_attr_set) {                                       // per handle = per device (the attribute is device state)
    RC_HIP(hipFuncSetAttribute((const void*)k_diag_factor, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RC_DIAG_LDS));
    RC_HIP(hipFuncSetAttribute((const void*)k_inv128_batched, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RC_DIAG_LDS));
    h->diag_attr_set = true;
  }
  return 0;
}

int rc_launch_inv128_batched(rcgp_handle_s* h) {
  int rc;
  if ((rc = set_diag_attributes(h))) return rc;
  RC_BP(const double, Ab, h->A)
  RC_BP(const double, ib, h->invdiag)
  RC_BP(double, Wb, h->Linv)
  RcProfScope ps(h, RC_K_DIAG, 0.0, true);
  RC_LAUNCH(k_inv128_batched, dim3((unsigned)(h->Np / 128), 1, (unsigned)h->nb), dim3(512), RC_DIAG_LDS, Ab, h->Np, ib, Wb, h->Np);
  RC_HIP(hipGetLastError());
  return 0;
}

int rc_launch_diag(rcgp_handle_s* h, int64_t j) {
  int rc;
  if ((rc = set_diag_attributes(h))) return rc;
  RC_BP(double, Ab, h->A)
  RC_BP(double, ib, h->invdiag + (j / 128) * 128 * 128)
  RC_BP(double, wb, h->w)
  RC_BP(double, lb, h->logdiag)
  RC_BP(int, fb, h->info)
  RcProfScope ps(h, RC_K_DIAG, (double)h->nb * 128.0 * 128.0 * 128.0 / 3.0, true);
  RC_LAUNCH(k_diag_factor, dim3(1, 1, (unsigned)h->nb), dim3(512), RC_DIAG_LDS, Ab, h->Np, ib, wb, lb, fb, j);
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
    if ((rc = rc_launch_trsm_subst(h, P, Np, h->A + j * Np + j, h->invdiag + (j / 128) * 128 * 128, below, h->w + j + 128, h->w + j))) return rc;
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
//   C  (h->stream2, high priority): D(j) = k_diag_factor; P(j) = the tile (j+1, j) solved (k_trsm_subst<1>, eight strips on eight CUs) and
//                                   the block (j+1, j+1) updated (k_prep2). D(j+1) follows P(j) in stream order, so a chain step
//                                   costs D + P, not D + T + G.
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
// (and the near part once it reaches column u0) does. Dependency events ride on the dispatches (RC_LAUNCH, h->launch_stop): one queue
// packet less between two dependent kernels than a hipEventRecord marker.
static int potrf_fine(rcgp_handle_s* h) {
  // A batched call (h->nb units per launch) runs narrower outer panels and a shorter tail than one unit alone: with several units the K = NB
  // updates fill the chip from smaller trailing matrices on, and the all-fine-grained tail -- which trades throughput for latency -- pays
  // only for the last 16 block columns (N = 8192, 4 units: 16.6 -> 15.4 ms; N = 4096, 16 units: 9.2 -> 8.8; profiles/r04_batch_knob_sweeps.txt).
  // The schedule does not enter a unit's arithmetic (every tile adds its k-slabs in ascending order whatever delivers them), so the
  // results stay bit-identical to the single-unit call: tests/test_gpu_batch.py.
  const bool batched = h->nb > 1;
  const int tail_blocks = batched ? h->batch_tail_blocks : h->tail_blocks;
  const int64_t Np = h->Np, NB = batched ? h->batch_nb_outer : h->nb_outer, EXT = 128 * (int64_t)h->chain_ext;
  hipStream_t C = h->stream2, B = h->stream5, B2 = h->stream6, U1 = h->stream, U2 = h->stream3;
  int rc;
  hipEvent_t e0;
  if ((rc = next_event(h, &e0))) return rc;
  RC_HIP(hipEventRecord(e0, h->stream));
  RC_HIP(hipStreamWaitEvent(C, e0, 0));
  RC_HIP(hipStreamWaitEvent(B, e0, 0));
  RC_HIP(hipStreamWaitEvent(B2, e0, 0));
  hipEvent_t eG_prev = nullptr, eU1_prev = nullptr, eU2_prev = nullptr, eFar_prev = nullptr;
  hipEvent_t eFar_last = nullptr, eFarB = nullptr;              // the last far launch of all; the last far B no far A has followed yet
  int64_t farB_lo = 0;                                           // ... and its first column
  bool near_waited = false, far_waited = false;                  // this panel's wait for the previous panel's window piece
  int64_t u0_prev = 0;                                           // first column of that piece
  // The TAIL: the last `tail_blocks` block columns (rounded to whole panels) are ONE panel -- no K = NB updates any more, every update
  // fine-grained. Down there a panel's bulk update is a few hundred tiles, and it still stops the chain for its whole duration (the
  // chain's whole-CU kernel finds no empty CU beside it), while the K = 128 / 256 column work of a 40-block column hides behind the
  // chain's 73 us steps. The tail's updates reach to the last column, so they wait for EVERY outstanding K = NB update first.
  int64_t tail0 = Np;
  if (tail_blocks > 0) {
    const int64_t t = Np - 128 * (int64_t)tail_blocks;
    tail0 = (t <= 0) ? 0 : ((t + NB - 1) / NB) * NB;
    if (tail0 >= Np) tail0 = Np;
  }
  hipEvent_t eU1_all = nullptr, eU2_last = nullptr;              // every window piece / every bulk update launched so far
  bool near_all_waited = false, far_all_waited = false;
  const bool ext = !h->profiling;                                // (a profiling bracket records its own events around a launch)
  for (int64_t j = 0; j < Np; j += 128) {
    const int64_t below = Np - (j + 128);
    hipEvent_t eD = nullptr, eP, eG;
    if (below > 0 && (rc = next_event(h, &eD))) return rc;
    // LEAN steps (the column below is at most lean_blocks blocks tall: the latency-bound part of a factorisation): the diagonal kernel carries
    // no completion signal -- the tile solve follows it in stream order without the microseconds a signal costs its successor -- and the
    // column work waits for the tile solve's signal alone. It then starts ~10 us later, which a short column can afford and a tall one
    // cannot (there the panel solve is on the critical path: all steps lean costs 1 % at C2). N = 4096 2.39 -> 2.31 ms, N = 8192 6.70 -> 6.58.
    // Dropping the panel solve's signal as well (far update behind the near one) was slower: the far update then runs beside the next tile solve.
    const bool lean = ext && below > 0 && below <= 128 * (int64_t)h->lean_blocks;
    h->launch = C;
    if (ext && !lean) h->launch_stop = eD;
    if ((rc = rc_launch_diag(h, j)) || (rc = flush_stop(h))) return rc;
    if (below <= 0) break;
    const bool in_tail = (j >= tail0);
    const int64_t p0 = in_tail ? tail0 : (j / NB) * NB;          // the panel block j belongs to: [p0, pend)
    const int64_t pend = in_tail ? Np : p0 + NB;
    const int64_t cend = (pend + EXT < Np) ? pend + EXT : Np;    // G(j) covers the block columns [j + 128, cend)
    const bool first_of_panel = (j > 0 && j == p0);
    double* P = h->A + (j + 128) * Np + j;                       // rows below the diagonal block, 128 columns
    const double* Ljj = h->A + j * Np + j;
    const double* inv = h->invdiag + (j / 128) * 128 * 128;
    if ((rc = next_event(h, &eP)) || (rc = next_event(h, &eG))) return rc;
    if (!ext) RC_HIP(hipEventRecord(eD, C));
    if (!lean) RC_HIP(hipStreamWaitEvent(B, eD, 0));
    if (eG_prev) RC_HIP(hipStreamWaitEvent(C, eG_prev, 0));
    if (first_of_panel && eU1_prev && h->chain_ext < 2) RC_HIP(hipStreamWaitEvent(C, eU1_prev, 0));   // P touches column j + 128 >= u0
    if (ext) h->launch_stop = eP;                                 // (taken by the tile solve: the column work needs the solved tile only)
    if ((rc = rc_launch_chain_tile(h, P, h->A + (j + 128) * Np + (j + 128), Np, Ljj, inv, h->w + j + 128, h->w + j)) || (rc = flush_stop(h)))
      return rc;
    if (!ext) RC_HIP(hipEventRecord(eP, C));
    h->launch = B;
    hipEvent_t ePanel = eG;                                       // everything of this step on B (and B2) done
    if (below > 128) {
      // G(j) in two kernels: the two block columns the NEXT chain step reads (near, on B; P(j+1) waits for it alone) and the
      // rest (far, on B2, in order behind the earlier far updates). Only the far part, and the near part once it reaches the
      // columns of the previous panel's window piece, wait for that piece: the chain does not stop at a panel boundary.
      hipEvent_t eT2, eFar;
      if ((rc = next_event(h, &eT2))) return rc;
      if (lean) RC_HIP(hipStreamWaitEvent(B, eP, 0));
      if (ext) h->launch_stop = eT2;
      if ((rc = rc_launch_trsm_subst(h, P + 128 * Np, Np, Ljj, inv, below - 128, h->w + j + 256, h->w + j)) || (rc = flush_stop(h))) return rc;
      if (!ext) RC_HIP(hipEventRecord(eT2, B));
      // Far updates in GROUPS of G steps (G = far_group, 2): within a panel, step r of a group (r = 0 .. G-1, counted from the panel's first
      // column) sends no far update unless it is the group's last -- its near kernel takes G + 1 - r block columns instead of two, the ones that
      // enter the near window before the group is complete -- and the last step applies the whole group's 128 G columns to everything from the
      // third block column on as ONE K = 128 G product: 1/G as many passes over the far C tiles (a K = 128 update moves a 128 KB tile in and out
      // for 4.2 MFLOP). A panel that ends inside a group sends the incomplete group with its last step.
      const int64_t c0 = j + 128;
      const int G = h->far_group, r = (int)(((j - p0) / 128) % G);
      const bool group_end = (r == G - 1), panel_end = (j + 128 == pend);
      const int64_t nwidth = 128 * (int64_t)(G + 1 - r);
      const int64_t nend = (c0 + nwidth < cend) ? c0 + nwidth : cend;
      if (!lean) RC_HIP(hipStreamWaitEvent(B, eP, 0));
      if (eFar_prev) RC_HIP(hipStreamWaitEvent(B, eFar_prev, 0));
      if (eFarB && nend > farB_lo) RC_HIP(hipStreamWaitEvent(B, eFarB, 0));      // (never in the regular pattern: a near window inside the last far B)
      if (eU1_prev && !near_waited && nend > u0_prev) { RC_HIP(hipStreamWaitEvent(B, eU1_prev, 0)); near_waited = true; }
      if (in_tail && !near_all_waited && nend > u0_prev + NB) {    // past the first window piece's columns: everything else that is outstanding
        if (eU1_all) RC_HIP(hipStreamWaitEvent(B, eU1_all, 0));
        if (eU2_last) RC_HIP(hipStreamWaitEvent(B, eU2_last, 0));
        near_all_waited = true;
      }
      if (ext) h->launch_stop = eG;
      if ((rc = rc_launch_gemm_nt_sub(h, h->A + (j + 256) * Np + c0, Np, P + 128 * Np, Np, P, Np, below - 128, nend - c0, 128, j + 256, c0)) ||
          (rc = flush_stop(h)))
        return rc;
      if (!ext) RC_HIP(hipEventRecord(eG, B));
      if (cend > nend && (group_end || panel_end)) {
        RC_HIP(hipStreamWaitEvent(B2, eT2, 0));
        if (eU1_prev && !far_waited && cend > u0_prev) { RC_HIP(hipStreamWaitEvent(B2, eU1_prev, 0)); far_waited = true; }
        if (in_tail && !far_all_waited) {
          if (eU1_all) RC_HIP(hipStreamWaitEvent(B2, eU1_all, 0));
          if (eU2_last) RC_HIP(hipStreamWaitEvent(B2, eU2_last, 0));
          far_all_waited = true;
        }
        h->launch = B2;
        // Two launches: far A = the G block columns that the NEXT group's near windows write -- the only part a near update has to wait for
        // (two kernels must not read-modify-write one tile at a time) -- then far B = everything beyond, which nothing touches before the
        // next group's far A (behind it in stream order). As one kernel the whole far update (14 GFLOP, 280 us at the first steps of
        // N = 8192) sat between this step's panel solve and the next step's near update.
        const int64_t kk = 128 * (int64_t)(r + 1), kcol = j - 128 * (int64_t)r;            // the L columns [kcol, kcol + kk) of the rows below
        const int64_t aend = (below > 128 * (int64_t)h->far_split_min && nend + 128 * (int64_t)G < cend) ? nend + 128 * (int64_t)G : cend;
        if ((rc = next_event(h, &eFar))) return rc;
        if (ext) h->launch_stop = eFar;
        if ((rc = rc_launch_gemm_nt_sub(h, h->A + (j + 256) * Np + nend, Np, h->A + (j + 256) * Np + kcol, Np, h->A + nend * Np + kcol, Np, below - 128,
                                        aend - nend, kk, j + 256, nend)) ||
            (rc = flush_stop(h)))
          return rc;
        if (!ext) RC_HIP(hipEventRecord(eFar, B2));
        eFar_prev = eFar_last = eFar;
        eFarB = nullptr;                                          // (an earlier far B is behind this far A in stream order)
        if (cend > aend) {
          if ((rc = next_event(h, &eFarB))) return rc;
          if (ext) h->launch_stop = eFarB;
          if ((rc = rc_launch_gemm_nt_sub(h, h->A + (j + 256) * Np + aend, Np, h->A + (j + 256) * Np + kcol, Np, h->A + aend * Np + kcol, Np, below - 128,
                                          cend - aend, kk, j + 256, aend)) ||
              (rc = flush_stop(h)))
            return rc;
          if (!ext) RC_HIP(hipEventRecord(eFarB, B2));
          eFar_last = eFarB;
          farB_lo = aend;
        }
      }
      if (j + 128 == pend) {                                      // the outer updates need both halves
        if ((rc = next_event(h, &ePanel))) return rc;
        if (eFar_last) RC_HIP(hipStreamWaitEvent(B, eFar_last, 0));
        RC_HIP(hipEventRecord(ePanel, B));
      }
    } else {
      RC_HIP(hipEventRecord(eG, B));
    }
    eG_prev = eG;
    if (j + 128 == pend) {                                        // panel [pend - NB, pend) is final once B has finished this step
      near_waited = far_waited = false;
      u0_prev = pend + EXT;
      // Outer (K = NB) updates with the finished panel, by target column panel: the next `depth` (shifted) panels one kernel
      // each, in column order on the main stream -- the first one is what the chain is waiting for -- and everything beyond
      // them in one bulk kernel on the bulk stream. A column panel leaves the bulk kernel's domain one panel before the
      // chain reaches it with depth 1, `depth` panels before with a deeper window: the chain may run that far ahead of the bulk.
      eU1_prev = nullptr;
      const int depth = h->chain_depth;
      const double* Lp0 = h->A + p0;                              // column offset of the finished panel
      hipEvent_t eR_new = nullptr;
      for (int q = 0; q <= depth; ++q) {
        const int64_t u0 = pend + EXT + (int64_t)q * NB;
        if (u0 >= Np) break;
        const int64_t u1 = (u0 + NB < Np) ? u0 + NB : Np;
        if (q < depth) {                                          // window piece
          if (q == 0) RC_HIP(hipStreamWaitEvent(U1, ePanel, 0));
          if (q == depth - 1 && eU2_prev) RC_HIP(hipStreamWaitEvent(U1, eU2_prev, 0));   // this panel was in the previous bulk kernel
          h->launch = U1;
          hipEvent_t eU1 = nullptr;
          if (q == 0) {
            if ((rc = next_event(h, &eU1))) return rc;
            if (ext) h->launch_stop = eU1;
          }
          if ((rc = rc_launch_gemm_nt_sub(h, h->A + u0 * Np + u0, Np, Lp0 + u0 * Np, Np, Lp0 + u0 * Np, Np, Np - u0, u1 - u0, NB, u0, u0)) ||
              (rc = flush_stop(h)))
            return rc;
          if (q == 0) {
            if (!ext) RC_HIP(hipEventRecord(eU1, U1));
            eU1_prev = eU1;
          }
        } else {                                                  // bulk: everything from u0 on
          if ((rc = next_event(h, &eR_new))) return rc;
          RC_HIP(hipStreamWaitEvent(U2, ePanel, 0));
          h->launch = U2;
          if (ext) h->launch_stop = eR_new;
          if ((rc = rc_launch_syrk_lower(h, h->A + u0 * Np + u0, Np, Lp0 + u0 * Np, Np, Np - u0, NB)) || (rc = flush_stop(h))) return rc;
          if (!ext) RC_HIP(hipEventRecord(eR_new, U2));
        }
      }
      eU2_prev = eR_new;
      if (eR_new) eU2_last = eR_new;
      if (pend == tail0 && pend + EXT < Np) {                     // the tail starts here: one event behind every window piece
        if ((rc = next_event(h, &eU1_all))) return rc;
        RC_HIP(hipEventRecord(eU1_all, U1));
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
  if (eU2_last) RC_HIP(hipStreamWaitEvent(h->stream, eU2_last, 0));
  return 0;
}

// w = y (the right-hand side the factorisation carries along) and status word = 0, every unit of a batch in one launch
__global__ void k_potrf_init(RcBP<double> wb, RcBP<const double> yb, RcBP<int> infob, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) wb.p[blockIdx.z][i] = yb.p[blockIdx.z][i];
  if (i == 0) *infob.p[blockIdx.z] = 0;
}

// Right-looking blocked Cholesky. Default: the fine-grained multi-stream schedule above. RCGP_FINE=0: one-panel look-ahead (as soon as the
// trailing update has finished the NEXT panel's columns, that panel is factored on the chain stream while the bulk stream updates the
// rest of the trailing matrix); RCGP_LOOKAHEAD=0 or a matrix of fewer than four blocks: strictly sequential on the main stream.
int rc_potrf(rcgp_handle_s* h) {
  const int64_t Np = h->Np, NB = h->nb_outer;
  int rc;
  h->launch = h->stream;
  h->launch_stop = nullptr;                                        // (an earlier call may have failed half-way)
  h->prof_pending = -1;
  if (h->nb > 1) {                                                // a batched call: w = y and status = 0 for every unit in ONE launch (2 nb copies otherwise)
    RC_BP(double, wb, h->w)
    RC_BP(const double, yb, (const double*)h->y)
    RC_BP(int, fb, h->info)
    hipLaunchKernelGGL(k_potrf_init, dim3((unsigned)((Np + 255) / 256), 1, (unsigned)h->nb), dim3(256), 0, h->stream, wb, yb, fb, Np);
    RC_HIP(hipGetLastError());
    for (int u = 0; u < h->nb; ++u) h->bh[u]->gram_fresh = false;  // consumed, whatever happens below
  } else {
    RC_HIP(hipMemcpyAsync(h->w, h->y, (size_t)Np * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    RC_HIP(hipMemsetAsync(h->info, 0, sizeof(int), h->stream));
    h->gram_fresh = false;                                         // consumed, whatever happens below
  }
  g_rc_stat[0] += h->nb;
  const bool la = h->lookahead && Np >= 4 * 128;                   // (the multi-stream schedule needs no minimum number of panels)
  h->la_cursor = 0;
  auto mark_factored = [&]() {
    for (int u = 0; u < h->nb; ++u) {
      rcgp_handle_s* hu = (h->nb > 1) ? h->bh[u] : h;
      hu->factored = true;
      hu->inverted = false;
    }
  };
  if (la && h->fine_chain) {
    if ((rc = potrf_fine(h))) return rc;
    mark_factored();
    return 0;
  }
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
      // (chain stream) factor panel J+1
      if ((rc = next_event(h, &ev_panel))) return rc;
      RC_HIP(hipStreamWaitEvent(h->stream2, ev_next, 0));
      h->launch = h->stream2;
      rc = panel_factor(h, Jend, Jend2);
      h->launch = h->stream;
      if (rc) return rc;
      RC_HIP(hipEventRecord(ev_panel, h->stream2));
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
  mark_factored();
  return 0;
}
