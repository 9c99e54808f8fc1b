// fp64 MFMA GEMM family for gfx950 (v_mfma_f64_16x16x4_f64).
//
// One workgroup = 512 threads = 8 waves (2x4), one 128x128 output tile; each wave owns 64x32 = 4x2 MFMA tiles
// (8 accumulators x 4 doubles; geometry: struct Geo below).  Operands are staged global -> registers -> LDS in k-slabs of 16
// with a two-stage LDS ring (one barrier per slab); fragment reads are bank-conflict free (row strides LDK / LDR below).
//
// Two operand storage kinds are supported, selected at compile time:
//   KC = true  : element (r, k) of the operand lives at src[r*ld + k]   (k contiguous: "A[i][k]" or "B^T[j][k]")
//   KC = false : element (r, k) of the operand lives at src[k*ld + r]   (r contiguous: "A^T[k][i]" or "B[k][j]")
// where r is the output-row index for the A operand and the output-column index for the B operand.
//
// MFMA lane maps (cdna_hip_programming.md section 3): A lane l holds A[i=l&15][k=l>>4]; B lane l holds B[k=l>>4][j=l&15];
// C/D lane l register r holds C[row=(l>>4)+4r][col=l&15].
#include "common.h"
#include "rc_math.h"
#include <algorithm>

#ifndef LDK
#define LDK 17          // row stride (doubles) of a KC=true LDS slab: [128][17]. ODD: hipcc fuses the fragment reads into
                        // ds_read2_b64, which is banked over 32 dwords in groups of 16 consecutive lanes (16 rows of one k):
                        // stride 18 was two-way conflicted there (SQ_LDS_BANK_CONFLICT 5 cycles per LDS instruction), 17 is not
#endif
#define LDR 144         // row stride (doubles) of a KC=false LDS slab: [16][144]
#define SLAB 2304       // doubles per operand slab (both layouts)
#define GEMM_LDS (4 * SLAB)

// WN = number of wave columns of the workgroup: 2 -> 256 threads, waves 2x2, 64x64 per wave (4x4 MFMA tiles);
//                                                4 -> 512 threads, waves 2x4, 64x32 per wave (4x2 MFMA tiles).
// The 512-thread shape keeps a wave under 128 registers, so two workgroups fit a CU = 4 waves per SIMD: the fp64 MFMA
// pipe sustains 36 TFLOP/s with one wave per SIMD, 46 with two and 48-49 with four (tools/mfma_peak.hip).
#ifndef RC_WN
#define RC_WN 4
#endif
template <int WN> struct Geo {
  static constexpr int NT = 128 * WN;          // threads
  static constexpr int NI = 8 / WN;            // MFMA tile columns per wave (wave tile = 64 x 16*NI)
  static constexpr int NP = 1024 / NT;         // double2 loads per thread per operand slab
};

template <int WN> struct RegsN { double2 v[Geo<WN>::NP]; };

template <bool KC, int WN>
__device__ __forceinline__ void slab_load(const double* __restrict__ src, int64_t ld, int64_t r0, int64_t k0, RegsN<WN>& rg) {
  const int t = threadIdx.x;
  constexpr int NT = Geo<WN>::NT, NP = Geo<WN>::NP;
  if (KC) {
    const int kk = (t & 7) * 2, r = t >> 3;
#pragma unroll
    for (int p = 0; p < NP; ++p)
      rg.v[p] = *reinterpret_cast<const double2*>(src + (r0 + r + (NT / 8) * p) * ld + k0 + kk);
  } else {
    const int r = (t & 63) * 2, k = t >> 6;
#pragma unroll
    for (int p = 0; p < NP; ++p)
      rg.v[p] = *reinterpret_cast<const double2*>(src + (k0 + k + (NT / 64) * p) * ld + r0 + r);
  }
}

template <bool KC, int WN>
__device__ __forceinline__ void slab_store(double* lds, const RegsN<WN>& rg) {
  const int t = threadIdx.x;
  constexpr int NT = Geo<WN>::NT, NP = Geo<WN>::NP;
  if (KC) {
    const int kk = (t & 7) * 2, r = t >> 3;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      double* d = lds + (r + (NT / 8) * p) * LDK + kk;           // rows are only 8-byte aligned with an odd stride
      d[0] = rg.v[p].x;
      d[1] = rg.v[p].y;
    }
  } else {
    const int r = (t & 63) * 2, k = t >> 6;
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<double2*>(lds + (k + (NT / 64) * p) * LDR + r) = rg.v[p];
  }
}

template <bool KC>
__device__ __forceinline__ double frag_read(const double* lds, int row, int k) {
  return KC ? lds[row * LDK + k] : lds[k * LDR + row];
}

// acc (+/-)= A(i, k0:k1) * B(j, k0:k1)^T for the 128x128 tile (i from a0, j from b0). k0,k1 multiples of 16, k1 > k0.
// NEG = true subtracts the product, so that an update kernel can start from acc = C: the A operand is negated BY THE MFMA ITSELF (the
// last builtin argument, BLGP, is the NEG field of the fp64 MFMAs on gfx940+: 1 = neg:[1,0,0]). A v_xor per A fragment instead -- 16 VALU
// instructions per slab and wave -- cost the update kernels 5-7 % (tools/gemm_rate.hip: 71.3 -> 67.5 TFLOP/s at K = 1024 on idle operands).
template <bool AKC, bool BKC, int WN, bool NEG = false>
__device__ __forceinline__ void gemm_mainloop(const double* __restrict__ A, int64_t lda, int64_t a0, const double* __restrict__ B,
                                              int64_t ldb, int64_t b0, int64_t k0, int64_t k1, v4d (&acc)[4][Geo<WN>::NI], double* lds) {
  constexpr int NI = Geo<WN>::NI;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = (wave / WN) * 64, wc = (wave % WN) * (16 * NI);
  const int fr = lane & 15, fq = lane >> 4;
  const int nk = (int)((k1 - k0) >> 4);
  RegsN<WN> ra, rb;
  slab_load<AKC, WN>(A, lda, a0, k0, ra);
  slab_load<BKC, WN>(B, ldb, b0, k0, rb);
  slab_store<AKC, WN>(lds, ra);
  slab_store<BKC, WN>(lds + SLAB, rb);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const double* la = lds + (kt & 1) * 2 * SLAB;
    const double* lb = la + SLAB;
    // Branch-free prefetch: the last iteration re-reads its own slab (an L2 hit) and parks it in the idle LDS stage, so the
    // staging registers never become conditionally live (hipcc otherwise keeps them in scratch memory).
    const int64_t kn = k0 + (int64_t)((kt + 1 < nk) ? kt + 1 : kt) * 16;
    slab_load<AKC, WN>(A, lda, a0, kn, ra);
    slab_load<BKC, WN>(B, ldb, b0, kn, rb);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      double af[4], bf[NI];
#pragma unroll
      for (int x = 0; x < 4; ++x) af[x] = frag_read<AKC>(la, wr + 16 * x + fr, 4 * s + fq);
#pragma unroll
      for (int x = 0; x < NI; ++x) bf[x] = frag_read<BKC>(lb, wc + 16 * x + fr, 4 * s + fq);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi], bf[ni], acc[mi][ni], 0, 0, NEG ? 1 : 0);
    }
    double* na = lds + ((kt + 1) & 1) * 2 * SLAB;
    slab_store<AKC, WN>(na, ra);
    slab_store<BKC, WN>(na + SLAB, rb);
    __syncthreads();
  }
}

template <int NI>
__device__ __forceinline__ void acc_zero(v4d (&acc)[4][NI]) {
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (v4d){0.0, 0.0, 0.0, 0.0};
}

// Visit every accumulator element with its (row, col) inside the 128x128 tile.
#define RC_FOR_ACC(mi, ni, r, row, col)                                  \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                       \
  _Pragma("unroll") for (int ni = 0; ni < NI_; ++ni)                     \
  _Pragma("unroll") for (int r = 0; r < 4; ++r)                          \
    for (int row = wr_ + 16 * mi + 4 * r + fq_, col = wc_ + 16 * ni + fr_, once_ = 1; once_; once_ = 0)

#define RC_LANE_VARS(WN)                                                          \
  constexpr int NI_ = Geo<WN>::NI;                                                \
  const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;                   \
  const int wr_ = (wave_ / WN) * 64, wc_ = (wave_ % WN) * (16 * NI_);             \
  const int fr_ = lane_ & 15, fq_ = lane_ >> 4;

#define RC_BOUNDS(WN) __launch_bounds__(128 * WN, WN)

__device__ __forceinline__ void tri_decode(int64_t id, int& ti, int& tj) {
  int t = (int)((sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
  while ((int64_t)(t + 1) * (t + 2) / 2 <= id) ++t;
  while ((int64_t)t * (t + 1) / 2 > id) --t;
  ti = t;
  tj = (int)(id - (int64_t)t * (t + 1) / 2);
}

// acc = C tile (the MFMA C/D layout), for kernels that update C in place
template <int WN>
__device__ __forceinline__ void acc_load(v4d (&acc)[4][Geo<WN>::NI], const double* __restrict__ Ct, int64_t ldc) {
  RC_LANE_VARS(WN)
  RC_FOR_ACC(mi, ni, r, row, col) { acc[mi][ni][r] = Ct[(int64_t)row * ldc + col]; }
}

template <int WN>
__device__ __forceinline__ void acc_store(const v4d (&acc)[4][Geo<WN>::NI], double* __restrict__ Ct, int64_t ldc) {
  RC_LANE_VARS(WN)
  RC_FOR_ACC(mi, ni, r, row, col) { Ct[(int64_t)row * ldc + col] = acc[mi][ni][r]; }
}

// The same through LDS, 64 tile rows at a time ([64][LDW] doubles = the whole operand ring, free outside the main loop): the MFMA
// C/D layout gives a lane four DIFFERENT rows of one column, i.e. 64 eight-byte accesses per lane for a tile; staged, every
// lane moves 16 sixteen-byte pieces of whole rows instead. LDW = 144: the two rows a half-wave touches per ds access fall on
// disjoint bank halves.
#define LDW 144
template <int WN>
__device__ __forceinline__ void acc_load_staged(v4d (&acc)[4][Geo<WN>::NI], const double* Ct, int64_t ldc, double* lds) {
  RC_LANE_VARS(WN)
  const int row0 = threadIdx.x >> 6, c2 = (threadIdx.x & 63) * 2;
  constexpr int RSTEP = 2 * WN;                              // rows covered by one pass of the workgroup
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
    for (int p = 0; p < 64 / RSTEP; ++p) {
      const int row = row0 + RSTEP * p;
      *reinterpret_cast<double2*>(lds + row * LDW + c2) = *reinterpret_cast<const double2*>(Ct + (int64_t)(64 * hh + row) * ldc + c2);
    }
    __syncthreads();
    if (wr_ == 64 * hh) {
      RC_FOR_ACC(mi, ni, r, row, col) { acc[mi][ni][r] = lds[(row - wr_) * LDW + col]; }
    }
    __syncthreads();
  }
}

template <int WN>
__device__ __forceinline__ void acc_store_staged(const v4d (&acc)[4][Geo<WN>::NI], double* Ct, int64_t ldc, double* lds) {
  RC_LANE_VARS(WN)
  const int row0 = threadIdx.x >> 6, c2 = (threadIdx.x & 63) * 2;
  constexpr int RSTEP = 2 * WN;
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    if (wr_ == 64 * hh) {
      RC_FOR_ACC(mi, ni, r, row, col) { lds[(row - wr_) * LDW + col] = acc[mi][ni][r]; }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 64 / RSTEP; ++p) {
      const int row = row0 + RSTEP * p;
      *reinterpret_cast<double2*>(Ct + (int64_t)(64 * hh + row) * ldc + c2) = *reinterpret_cast<const double2*>(lds + row * LDW + c2);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// C[i][j] -= sum_k P[i][k] P[j][k] on the lower tiles of an n x n matrix (trailing update of the blocked Cholesky).
// The accumulators start from the C tile (its loads overlap the first operand slabs) and the product is subtracted in
// the MFMA chain, so the epilogue is stores only.
// ---------------------------------------------------------------------------------------------------------------------
template <int WN, int STAGED>
__global__ void RC_BOUNDS(WN) k_syrk_lower(RcBP<double> Cb, int64_t ldc, RcBP<const double> Pb, int64_t ldp, int kk) {
  __shared__ double lds[GEMM_LDS];
  double* __restrict__ C = Cb.p[blockIdx.z];
  const double* __restrict__ P = Pb.p[blockIdx.z];
  int ti, tj;
  tri_decode(blockIdx.x, ti, tj);
  v4d acc[4][Geo<WN>::NI];
  double* Ct = C + (int64_t)ti * 128 * ldc + (int64_t)tj * 128;
  if (STAGED & 1) acc_load_staged<WN>(acc, Ct, ldc, lds); else acc_load<WN>(acc, Ct, ldc);
  gemm_mainloop<true, true, WN, true>(P, ldp, (int64_t)ti * 128, P, ldp, (int64_t)tj * 128, 0, kk, acc, lds);
  if (STAGED & 2) acc_store_staged<WN>(acc, Ct, ldc, lds); else acc_store<WN>(acc, Ct, ldc);
}

int rc_launch_syrk_lower(rcgp_handle_s* h, double* C, int64_t ldc, const double* P, int64_t ldp, int64_t n, int64_t kk) {
  const int64_t T = n / 128;
  if (T <= 0) return 0;
  RC_BP(double, Cb, C)
  RC_BP(const double, Pb, P)
  RcProfScope ps(h, RC_K_GEMM, (double)h->nb * (double)n * (double)(n + 128) * (double)kk, true);   // 2*kk flops per lower-tile element
  const dim3 grid((unsigned)(T * (T + 1) / 2), 1, (unsigned)h->nb), block(128 * RC_WN);
  RC_LAUNCH((k_syrk_lower<RC_WN, 3>), grid, block, 0, Cb, ldc, Pb, ldp, (int)kk);
  RC_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// C (m x n) -= A (m x kk) * B (n x kk)^T, skipping tiles strictly above the global diagonal.
// ---------------------------------------------------------------------------------------------------------------------
template <int WN, int STAGED>
__global__ void RC_BOUNDS(WN) k_gemm_nt_sub(RcBP<double> Cb, int64_t ldc, RcBP<const double> Ab, int64_t lda, RcBP<const double> Bb, int64_t ldb,
                                            int kk, int64_t row0, int64_t col0) {
  __shared__ double lds[GEMM_LDS];
  const int tj = blockIdx.x, ti = blockIdx.y;
  if (col0 + (int64_t)tj * 128 > row0 + (int64_t)ti * 128) return;
  double* __restrict__ C = Cb.p[blockIdx.z];
  const double* __restrict__ A = Ab.p[blockIdx.z];
  const double* __restrict__ B = Bb.p[blockIdx.z];
  v4d acc[4][Geo<WN>::NI];
  double* Ct = C + (int64_t)ti * 128 * ldc + (int64_t)tj * 128;
  if (STAGED & 1) acc_load_staged<WN>(acc, Ct, ldc, lds); else acc_load<WN>(acc, Ct, ldc);
  gemm_mainloop<true, true, WN, true>(A, lda, (int64_t)ti * 128, B, ldb, (int64_t)tj * 128, 0, kk, acc, lds);
  if (STAGED & 2) acc_store_staged<WN>(acc, Ct, ldc, lds); else acc_store<WN>(acc, Ct, ldc);
}

// The same update on 64 x 128 HALF tiles for short k (kk = 128 or 256: the near / far updates of the panel chain). A 128^2 tile with K = 128 is
// 13.7 us of MFMA time per CU behind a C tile and eight operand slabs that have to arrive first; with two workgroups per CU that took 23-27 us
// per tile. Half tiles: 32 accumulator registers, the C half tile and the first four operand slabs requested up front (register ring as in
// gemm_mainloop_m64, both operands k-contiguous), two workgroups per CU at <= 128 registers.
__global__ void __launch_bounds__(512, 4) k_gemm_nt_sub_h64(RcBP<double> Cb, int64_t ldc, RcBP<const double> Ab, int64_t lda, RcBP<const double> Bb,
                                                             int64_t ldb, int kk, int64_t row0, int64_t col0) {
  __shared__ double lds[2 * 3 * 64 * LDK];                      // two stages of A [64][LDK] + B [128][LDK]
  const int tj = blockIdx.x, th = blockIdx.y;
  if (col0 + (int64_t)tj * 128 > row0 + (int64_t)th * 64) return;    // (row0, col0 multiples of 128: a half tile is below the diagonal with its tile)
  double* __restrict__ C = Cb.p[blockIdx.z];
  const double* __restrict__ A = Ab.p[blockIdx.z];
  const double* __restrict__ B = Bb.p[blockIdx.z];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = (wave >> 2) * 32, wc = (wave & 3) * 32;
  const int fr = lane & 15, fq = lane >> 4;
  const int nk = kk >> 4;
  const int skk = (t & 7) * 2, sr = t >> 3;                     // slab element of this thread: row sr (A; B rows sr and sr + 64), k pair skk
  const double* ap = A + ((int64_t)th * 64 + sr) * lda + skk;
  const double* bp = B + ((int64_t)tj * 128 + sr) * ldb + skk;
  constexpr int STAGE = 3 * 64 * LDK;
  double* las = lds + sr * LDK + skk;
  double* lbs = lds + 64 * LDK + sr * LDK + skk;
#define RC_H64_LOAD(KK, RA, RB0, RB1)                                  \
  RA = *reinterpret_cast<const double2*>(ap + (KK));                   \
  RB0 = *reinterpret_cast<const double2*>(bp + (KK));                  \
  RB1 = *reinterpret_cast<const double2*>(bp + 64 * ldb + (KK));
#define RC_H64_STORE(ST, RA, RB0, RB1)                                 \
  las[(ST) * STAGE] = RA.x;  las[(ST) * STAGE + 1] = RA.y;             \
  lbs[(ST) * STAGE] = RB0.x; lbs[(ST) * STAGE + 1] = RB0.y;            \
  lbs[(ST) * STAGE + 64 * LDK] = RB1.x; lbs[(ST) * STAGE + 64 * LDK + 1] = RB1.y;
  double2 ra0, ra1, ra2, ra3, rb00, rb01, rb10, rb11, rb20, rb21, rb30, rb31;
  RC_H64_LOAD(0, ra0, rb00, rb01)
  RC_H64_LOAD(16, ra1, rb10, rb11)
  RC_H64_LOAD(32, ra2, rb20, rb21)
  RC_H64_LOAD(48, ra3, rb30, rb31)
  v4d acc[2][2];
  double* Ct = C + ((int64_t)th * 64 + wr + fq) * ldc + (int64_t)tj * 128 + wc + fr;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = Ct[(int64_t)(16 * mi + 4 * r) * ldc + 16 * ni];
  RC_H64_STORE(0, ra0, rb00, rb01)
  __syncthreads();
#define RC_H64_STEP(U, RA, RB0, RB1, NA, NB0, NB1)                                                                              \
  {                                                                                                                             \
    const int kt = kt0 + U;                                                                                                     \
    const double* la = lds + (U & 1) * STAGE;                                                                                   \
    const double* lb = la + 64 * LDK;                                                                                           \
    const int kn = ((kt + 4 < nk) ? kt + 4 : nk - 1) * 16;                   /* branch-free: the tail re-reads the last slab */ \
    RC_H64_LOAD(kn, RA, RB0, RB1)                                                                                               \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                             \
      double af[2], bf[2];                                                                                                      \
      _Pragma("unroll") for (int x = 0; x < 2; ++x) af[x] = la[(wr + 16 * x + fr) * LDK + 4 * s + fq];                          \
      _Pragma("unroll") for (int x = 0; x < 2; ++x) bf[x] = lb[(wc + 16 * x + fr) * LDK + 4 * s + fq];                          \
      _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                                          \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                                          \
        acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi], bf[ni], acc[mi][ni], 0, 0, 1);                               \
    }                                                                                                                           \
    RC_H64_STORE((U + 1) & 1, NA, NB0, NB1)                                                                                     \
    __syncthreads();                                                                                                            \
  }
  for (int kt0 = 0; kt0 < nk; kt0 += 4) {
    RC_H64_STEP(0, ra0, rb00, rb01, ra1, rb10, rb11)
    RC_H64_STEP(1, ra1, rb10, rb11, ra2, rb20, rb21)
    RC_H64_STEP(2, ra2, rb20, rb21, ra3, rb30, rb31)
    RC_H64_STEP(3, ra3, rb30, rb31, ra0, rb00, rb01)
  }
#undef RC_H64_LOAD
#undef RC_H64_STORE
#undef RC_H64_STEP
  int64_t ldo = ldc;                                            // opaque copy: the sixteen store addresses are formed here, not kept from the loads
  asm volatile("" : "+s"(ldo));
  double* Co = C + ((int64_t)th * 64 + wr + fq) * ldo + (int64_t)tj * 128 + wc + fr;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) Co[(int64_t)(16 * mi + 4 * r) * ldo + 16 * ni] = acc[mi][ni][r];
}

int rc_launch_gemm_nt_sub(rcgp_handle_s* h, double* C, int64_t ldc, const double* A, int64_t lda, const double* B, int64_t ldb,
                          int64_t m, int64_t n, int64_t kk, int64_t row0, int64_t col0) {
  if (m <= 0 || n <= 0) return 0;
  RC_BP(double, Cb, C)
  RC_BP(const double, Ab, A)
  RC_BP(const double, Bb, B)
  RcProfScope ps(h, RC_K_GEMM, (double)h->nb * 2.0 * (double)m * (double)n * (double)kk, true);
  if (kk <= 512 && kk % 64 == 0) {     // the panel chain's near / far updates (K = 128 ... 512): half tiles, everything requested up front
    RC_LAUNCH(k_gemm_nt_sub_h64, dim3((unsigned)(n / 128), (unsigned)(m / 64), (unsigned)h->nb), dim3(512), 0, Cb, ldc, Ab, lda, Bb, ldb, (int)kk, row0,
              col0);
  } else {
    RC_LAUNCH((k_gemm_nt_sub<RC_WN, 3>), dim3((unsigned)(n / 128), (unsigned)(m / 128), (unsigned)h->nb), dim3(128 * RC_WN), 0, Cb, ldc, Ab, lda, Bb, ldb,
              (int)kk, row0, col0);
  }
  RC_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_prep2: the chain's last update of the NEXT diagonal block, D -= T T^T from the solved tile T (128 x 128) right below the block just
// factored: ten workgroups, one lower 32x32 block of D each (four waves: its 16x16 tiles), operands in LDS with an odd row stride (LP:
// the 16 lanes of a fragment read are 16 rows of one k), every global load in flight before the first wait.
// ---------------------------------------------------------------------------------------------------------------------
#define LP 129
__global__ void __launch_bounds__(256) k_prep2(RcBP<const double> Ltb, RcBP<double> Db, int64_t ld) {
  extern __shared__ double sm2[];                  // La[32][LP], Lb[32][LP]
  const double* __restrict__ Lt = Ltb.p[blockIdx.z];
  double* D = Db.p[blockIdx.z];
  double* La = sm2;
  double* Lb = La + 32 * LP;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fr = lane & 15, fq = lane >> 4;
  int bi = 0, rem = blockIdx.x;                     // lower 32x32 blocks of the 4x4 block grid, row by row
  while (rem > bi) { rem -= bi + 1; ++bi; }
  const int bj = rem;
  double2 va[8], vb[8];                             // all 16 loads in flight together
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int e = t + 256 * q, i = e >> 6, j2 = (e & 63) * 2;
    va[q] = *reinterpret_cast<const double2*>(Lt + (int64_t)(32 * bi + i) * ld + j2);
    vb[q] = *reinterpret_cast<const double2*>(Lt + (int64_t)(32 * bj + i) * ld + j2);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int e = t + 256 * q, i = e >> 6, j2 = (e & 63) * 2;
    La[i * LP + j2] = va[q].x;
    La[i * LP + j2 + 1] = va[q].y;
    Lb[i * LP + j2] = vb[q].x;
    Lb[i * LP + j2 + 1] = vb[q].y;
  }
  const int ti = wave >> 1, tj = wave & 1;
  double* Dt = D + (int64_t)(32 * bi + 16 * ti) * ld + 32 * bj + 16 * tj;
  const bool needed = !(bi == bj && tj > ti);       // the strictly upper 16x16 tile of a diagonal block is never read
  v4d acc = {0.0, 0.0, 0.0, 0.0};
  if (needed) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = Dt[(int64_t)(fq + 4 * q) * ld + fr];
  }
  __syncthreads();
  if (needed) {
#pragma unroll 2
    for (int kt = 0; kt < 8; ++kt) {
      double av[4], bv[4];
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        av[s2] = La[(16 * ti + fr) * LP + 16 * kt + 4 * s2 + fq];
        bv[s2] = Lb[(16 * tj + fr) * LP + 16 * kt + 4 * s2 + fq];
      }
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s2], bv[s2], acc, 0, 0, 1);     // neg:[1,0,0]
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) Dt[(int64_t)(fq + 4 * q) * ld + fr] = acc[q];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_trsm_subst: a 128-row tile of the panel solve  X L_jj^T = T  by blocked forward substitution against L_jj ITSELF -- no 128x128
// inverse anywhere on the factorisation's critical path (the diagonal kernel then only factors: 11 us less per chain step). What
// it needs of L_jj are its 28 strictly-lower 16x16 blocks and the inverses of its 8 diagonal blocks (which the factor-only diagonal
// kernel leaves in the diagonal blocks of invL); they are packed block by block into 77 KB of LDS (PB doubles per block, odd row
// stride), so two workgroups share a CU.
// One wave per 16-row strip, the whole recurrence in registers, with Y = X^T the C/D layout of a 16x16
// MFMA result (lane (fr, fq), register r: row fq + 4r, column fr) is the B-operand layout of the next products (k = 4s + fq, s = r):
//     Y_c = inv(L_cc) (T^T_c - sum_{k<c} L_ck Y_k),  c = 0..7:   4c + 4 MFMAs, 144 per strip, the same count as the product with the
// explicit triangular inverse; sequential along c, but an fp64 MFMA occupies the pipe for its whole 64-cycle latency anyway.
// The strip is read from / written to global memory in that layout directly (lane fr = its row, 32 columns fq + 4r + 16c).
// Fused: rhs rows -= X w_j (forward substitution of the right-hand side, as in the GEMM form of the panel solve).
// NS = strips (computing waves) per workgroup; all eight waves of the workgroup bring L_jj into LDS. NS = 4 for the column below (64 rows
// per workgroup, one computing wave per SIMD), NS = 1 for the chain's own tile (eight workgroups on eight CUs: 12 us against 40 for one
// workgroup that solves all eight strips and then updates the next diagonal block from them -- measured, that form is gone).
// ---------------------------------------------------------------------------------------------------------------------
#define PBS 17                     // row stride of a packed 16x16 block
#define PB (16 * PBS)              // doubles per packed block
__device__ __forceinline__ int rc_packed_block(int rb, int cb) { return rb * (rb + 1) / 2 + cb; }

template <int NS>
__global__ void __launch_bounds__(512, 4) k_trsm_subst(RcBP<double> Pb, int64_t ldp, RcBP<const double> Ljjb, int64_t ldl, RcBP<const double> invLb,
                                                        RcBP<double> rhsb, RcBP<const double> wjb) {
  extern __shared__ double smts[];                 // Lp[36][PB], wv[128]
  double* __restrict__ P = Pb.p[blockIdx.z];
  const double* __restrict__ Ljj = Ljjb.p[blockIdx.z];
  const double* __restrict__ invL = invLb.p[blockIdx.z];
  double* __restrict__ rhs = rhsb.p[blockIdx.z];
  const double* __restrict__ wj = wjb.p[blockIdx.z];
  double* Lp = smts;
  double* wv = Lp + 36 * PB;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fr = lane & 15, fq = lane >> 4;
  double* T = P + (int64_t)blockIdx.x * (16 * NS) * ldp;
  double* rhs_t = rhs + (int64_t)blockIdx.x * (16 * NS);
  // every global load of the kernel in flight before the first wait: L_jj (its 36 lower blocks enumerated directly: 36 x 16 rows x 8
  // pairs = 9 loads per thread; the diagonal ones from invL) and, for the computing waves, their strip
  double2 vl[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int e = t + 512 * q, b = e >> 7, i16 = (e >> 3) & 15, j2 = (e & 7) * 2;
    int rb = 0;
    while ((rb + 1) * (rb + 2) / 2 <= b) ++rb;
    const int cb = b - rb * (rb + 1) / 2, i = 16 * rb + i16, j = 16 * cb + j2;
    vl[q] = (rb == cb) ? *reinterpret_cast<const double2*>(invL + i * 128 + j) : *reinterpret_cast<const double2*>(Ljj + (int64_t)i * ldl + j);
  }
  v4d Y[8];
  double* Tw = T + (int64_t)(16 * (wave < NS ? wave : 0) + fr) * ldp + fq;
  if (wave < NS) {
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) Y[c][r] = Tw[16 * c + 4 * r];
  }
  const double wreg = (t < 128) ? wj[t] : 0.0;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int e = t + 512 * q, b = e >> 7, i16 = (e >> 3) & 15, j2 = (e & 7) * 2;
    double* d = Lp + b * PB + i16 * PBS + j2;      // (block b of the enumeration IS packed block b)
    d[0] = vl[q].x;                                // (the inverse blocks are lower triangular with stored zeros above their diagonal)
    d[1] = vl[q].y;
  }
  if (t < 128) wv[t] = wreg;
  __syncthreads();
  if (wave >= NS) return;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    v4d acc = Y[c];
#pragma unroll
    for (int k = 0; k < c; ++k) {
      const double* blk = Lp + rc_packed_block(c, k) * PB + fr * PBS + fq;
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(blk[4 * s2], Y[k][s2], acc, 0, 0, 1);   // - L_ck Y_k
    }
    const double* inv = Lp + rc_packed_block(c, c) * PB + fr * PBS + fq;
    v4d y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) y = __builtin_amdgcn_mfma_f64_16x16x4f64(inv[4 * s2], acc[s2], y, 0, 0, 0);
    Y[c] = y;
    // this block column of X is final: out it goes while the recurrence continues
#pragma unroll
    for (int r = 0; r < 4; ++r) Tw[16 * c + 4 * r] = y[r];
  }
  // rhs rows -= X w_j: lane (fr, fq) holds X[fr][16c + fq + 4r]; the four fq lanes of a row are summed in a fixed order
  double part = 0.0;
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) part = __builtin_fma(Y[c][r], wv[16 * c + 4 * r + fq], part);
  part += __shfl_xor(part, 16);
  part += __shfl_xor(part, 32);
  if (fq == 0) rhs_t[16 * wave + fr] -= part;
}

#define RC_SUBST_LDS ((size_t)(36 * PB + 128) * sizeof(double))

// Panel solve by substitution for the m rows below (m a multiple of 128): P <- P L_jj^-T, rhs -= P_new w_j. Four strips (64 rows) per
// workgroup, one computing wave per SIMD: the recurrence of a strip is 144 dependent fp64 MFMAs (64 cycles each: 3.8 us), two strips on
// one SIMD take twice that.
int rc_launch_trsm_subst(rcgp_handle_s* h, double* P, int64_t ldp, const double* Ljj, const double* invL, int64_t m, double* rhs, const double* wj) {
  if (m <= 0) return 0;
  if (!h->subst_attr_set) {
    RC_HIP(hipFuncSetAttribute((const void*)k_trsm_subst<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RC_SUBST_LDS));
    RC_HIP(hipFuncSetAttribute((const void*)k_trsm_subst<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RC_SUBST_LDS));
    h->subst_attr_set = true;
  }
  RC_BP(double, Pb, P)
  RC_BP(const double, Lb, Ljj)
  RC_BP(const double, ib, invL)
  RC_BP(double, rb, rhs)
  RC_BP(const double, wb, wj)
  RcProfScope ps(h, RC_K_GEMM, (double)h->nb * (double)m * 128.0 * 128.0, true);
  RC_LAUNCH(k_trsm_subst<4>, dim3((unsigned)(m / 64), 1, (unsigned)h->nb), dim3(512), RC_SUBST_LDS, Pb, ldp, Lb, ldp, ib, rb, wb);
  RC_HIP(hipGetLastError());
  return 0;
}

// The chain's tile: T (128 x 128, right below L_jj) solved -- eight workgroups, one strip each, on eight CUs -- and its rhs rows updated
// (this dispatch carries a pending h->launch_stop: the column work needs only the solved tile), then D (the next diagonal block)
// -= T_new T_new^T (k_prep2).
int rc_launch_chain_tile(rcgp_handle_s* h, double* T, double* D, int64_t ld, const double* Ljj, const double* invL, double* rhs, const double* wj) {
  const size_t lds2 = (size_t)(64 * LP) * sizeof(double);
  if (!h->subst_attr_set) {
    RC_HIP(hipFuncSetAttribute((const void*)k_trsm_subst<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RC_SUBST_LDS));
    RC_HIP(hipFuncSetAttribute((const void*)k_trsm_subst<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RC_SUBST_LDS));
    h->subst_attr_set = true;
  }
  if (!h->prep_attr_set) {
    RC_HIP(hipFuncSetAttribute((const void*)k_prep2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    h->prep_attr_set = true;
  }
  RC_BP(double, Tb, T)
  RC_BP(const double, Tcb, (const double*)T)
  RC_BP(double, Db, D)
  RC_BP(const double, Lb, Ljj)
  RC_BP(const double, ib, invL)
  RC_BP(double, rb, rhs)
  RC_BP(const double, wb, wj)
  {
    RcProfScope ps(h, RC_K_GEMM, (double)h->nb * 128.0 * 128.0 * 128.0, true);
    RC_LAUNCH(k_trsm_subst<1>, dim3(8, 1, (unsigned)h->nb), dim3(512), RC_SUBST_LDS, Tb, ld, Lb, ld, ib, rb, wb);
    RC_HIP(hipGetLastError());
  }
  {
    RcProfScope ps(h, RC_K_GEMM, (double)h->nb * 128.0 * 129.0 * 128.0, true);
    RC_LAUNCH(k_prep2, dim3(10, 1, (unsigned)h->nb), dim3(256), lds2, Tcb, Db, ld);
    RC_HIP(hipGetLastError());
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Triangular inverse by recursive doubling. With L = [[A,0],[B,C]]: L^-1 = [[A^-1,0],[-C^-1 B A^-1, C^-1]].
// Level s: for each pair p (colA = 2ps, rowC = colA + s):  T = B * A^-1 -> S ; X21 = -C^-1 * T -> W.
// ---------------------------------------------------------------------------------------------------------------------
template <int WN>
__global__ void RC_BOUNDS(WN) k_trtri_T(RcBP<const double> Lmb, RcBP<const double> Wb, RcBP<double> Sb, int64_t ld, int64_t Np, int64_t s, int pair0,
                                        int ti0, int npairs) {
  __shared__ double lds[GEMM_LDS];
  const int ti = ti0 + blockIdx.x, tj = blockIdx.y;           // tj slow: the longest k-ranges (small tj) are dispatched first
  const int unit = blockIdx.z / npairs;                       // blockIdx.z = unit * npairs + pair
  const int64_t colA = 2 * s * (int64_t)(pair0 + blockIdx.z - unit * npairs), rowC = colA + s;
  if (rowC + (int64_t)ti * 128 >= Np) return;
  const double* __restrict__ Lm = Lmb.p[unit];
  const double* __restrict__ W = Wb.p[unit];
  double* __restrict__ S = Sb.p[unit];
  v4d acc[4][Geo<WN>::NI];
  acc_zero(acc);
  // A operand: B block of L, element (i,k) at Lm[(rowC+i)*ld + colA + k]; B operand: A^-1 element (k,j) at W[(colA+k)*ld + colA + j]
  gemm_mainloop<true, false, WN>(Lm + rowC * ld + colA, ld, (int64_t)ti * 128, W + colA * ld + colA, ld, (int64_t)tj * 128,
                                 (int64_t)tj * 128, s, acc, lds);
  acc_store<WN>(acc, S + (rowC + (int64_t)ti * 128) * ld + colA + (int64_t)tj * 128, ld);
}

template <int WN>
__global__ void RC_BOUNDS(WN) k_trtri_X(RcBP<double> Wb, RcBP<const double> Sb, int64_t ld, int64_t Np, int64_t s, int pair0, int npairs) {
  __shared__ double lds[GEMM_LDS];
  const int tj = blockIdx.x, ti = (int)gridDim.y - 1 - (int)blockIdx.y;   // ti slow and reversed: longest k-ranges first
  const int unit = blockIdx.z / npairs;
  const int64_t colA = 2 * s * (int64_t)(pair0 + blockIdx.z - unit * npairs), rowC = colA + s;
  if (rowC + (int64_t)ti * 128 >= Np) return;
  double* __restrict__ W = Wb.p[unit];
  const double* __restrict__ S = Sb.p[unit];
  v4d acc[4][Geo<WN>::NI];
  acc_zero(acc);
  // A operand: C^-1 element (i,k) at W[(rowC+i)*ld + rowC + k], zero for k > i; B operand: T element (k,j) at S[(rowC+k)*ld + colA + j]
  gemm_mainloop<true, false, WN, true>(W + rowC * ld + rowC, ld, (int64_t)ti * 128, S + rowC * ld + colA, ld, (int64_t)tj * 128, 0,
                                       (int64_t)(ti + 1) * 128, acc, lds);
  acc_store<WN>(acc, W + (rowC + (int64_t)ti * 128) * ld + colA + (int64_t)tj * 128, ld);
}

// The same two products on 64 x 128 HALF tiles (512 threads, waves 2x4, 32x32 per wave): for the levels whose launches have too few
// tiles to balance -- a launch of <= 2 generations of 128^2 tiles with k-ranges from 128 to s lasts as long as its LONGEST tile
// (C1, s = 2048: 512 tiles, 0.53 ms for 0.28 ms of work); twice as many workgroups of half the length pack behind each other
// (0.33 ms; L^-1 at N = 8192 3.90 -> 3.39 ms, at N = 4096 1.13 -> 0.72, C2 22.9 -> 22.5). Pairing a long and a short half tile in one
// workgroup (equal k totals) on top of that: no further gain, 12-24 spilled registers -- not kept.
template <bool NEG>
__device__ __forceinline__ void gemm_mainloop_m64(const double* __restrict__ A, int64_t lda, int64_t a0, const double* __restrict__ B, int64_t ldb,
                                                  int64_t b0, int64_t k0, int64_t k1, v4d (&acc)[2][2], double* lds) {
  // (k1 - k0) must be a multiple of 64. A half tile has 16 MFMAs per wave and slab -- 0.2 us of pipe time -- so a one-slab prefetch leaves it
  // waiting for memory (measured 1.1-2.2 us per slab); its 32 accumulator registers leave room for a FOUR-slab register ring instead:
  // slab kt + 4 is requested when slab kt has moved to LDS.
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = (wave >> 2) * 32, wc = (wave & 3) * 32;
  const int fr = lane & 15, fq = lane >> 4;
  const int nk = (int)((k1 - k0) >> 4);
  const int akk = (t & 7) * 2, ar = t >> 3;                    // A (k contiguous): 64 rows x 16 k = one double2 per thread
  const double* ap = A + (a0 + ar) * lda + akk;
  const int bj = (t & 63) * 2, bk = t >> 6;                    // B (column contiguous): 16 k x 128 columns = two double2 per thread (k, k + 8)
  const double* bp = B + (int64_t)bk * ldb + b0 + bj;
  double* lbs = lds + SLAB + bk * LDR + bj;
  double* las = lds + ar * LDK + akk;
#define RC_M64_LOAD(KK, RA, RB0, RB1)                                   \
  RA = *reinterpret_cast<const double2*>(ap + (KK));                    \
  RB0 = *reinterpret_cast<const double2*>(bp + (KK) * ldb);             \
  RB1 = *reinterpret_cast<const double2*>(bp + ((KK) + 8) * ldb);
#define RC_M64_STORE(STAGE, RA, RB0, RB1)                               \
  las[(STAGE) * 2 * SLAB] = RA.x;                                       \
  las[(STAGE) * 2 * SLAB + 1] = RA.y;                                   \
  *reinterpret_cast<double2*>(lbs + (STAGE) * 2 * SLAB) = RB0;          \
  *reinterpret_cast<double2*>(lbs + (STAGE) * 2 * SLAB + 8 * LDR) = RB1;
  double2 ra0, ra1, ra2, ra3, rb00, rb01, rb10, rb11, rb20, rb21, rb30, rb31;
  RC_M64_LOAD(k0, ra0, rb00, rb01)
  RC_M64_LOAD(k0 + 16, ra1, rb10, rb11)
  RC_M64_LOAD(k0 + 32, ra2, rb20, rb21)
  RC_M64_LOAD(k0 + 48, ra3, rb30, rb31)
  RC_M64_STORE(0, ra0, rb00, rb01)
  __syncthreads();
  // one slab: reload the register set whose slab is in LDS already (<- slab kt + 4), multiply slab kt from LDS stage U & 1, move the next slab
  // (the N set) into the other stage
#define RC_M64_STEP(U, RA, RB0, RB1, NA, NB0, NB1)                                                                              \
  {                                                                                                                             \
    const int kt = kt0 + U;                                                                                                     \
    const double* la = lds + (U & 1) * 2 * SLAB;                                                                                \
    const double* lb = la + SLAB;                                                                                               \
    const int64_t kn = k0 + (int64_t)((kt + 4 < nk) ? kt + 4 : nk - 1) * 16; /* branch-free: the tail re-reads the last slab */ \
    RC_M64_LOAD(kn, RA, RB0, RB1)                                                                                               \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                             \
      double af[2], bf[2];                                                                                                      \
      _Pragma("unroll") for (int x = 0; x < 2; ++x) af[x] = la[(wr + 16 * x + fr) * LDK + 4 * s + fq];                          \
      _Pragma("unroll") for (int x = 0; x < 2; ++x) bf[x] = lb[(4 * s + fq) * LDR + wc + 16 * x + fr];                          \
      _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                                          \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                                          \
        acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi], bf[ni], acc[mi][ni], 0, 0, NEG ? 1 : 0);                     \
    }                                                                                                                           \
    RC_M64_STORE((U + 1) & 1, NA, NB0, NB1)                                                                                     \
    __syncthreads();                                                                                                            \
  }
  for (int kt0 = 0; kt0 < nk; kt0 += 4) {
    RC_M64_STEP(0, ra0, rb00, rb01, ra1, rb10, rb11)
    RC_M64_STEP(1, ra1, rb10, rb11, ra2, rb20, rb21)
    RC_M64_STEP(2, ra2, rb20, rb21, ra3, rb30, rb31)
    RC_M64_STEP(3, ra3, rb30, rb31, ra0, rb00, rb01)
  }
#undef RC_M64_LOAD
#undef RC_M64_STORE
#undef RC_M64_STEP
}

__device__ __forceinline__ void acc_store_m64(const v4d (&acc)[2][2], double* __restrict__ Ct, int64_t ldc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = (wave >> 2) * 32, wc = (wave & 3) * 32, fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ct[(int64_t)(wr + 16 * mi + 4 * r + fq) * ldc + wc + 16 * ni + fr] = acc[mi][ni][r];
}

__global__ void __launch_bounds__(512, 4) k_trtri_T_half(RcBP<const double> Lmb, RcBP<const double> Wb, RcBP<double> Sb, int64_t ld, int64_t Np,
                                                          int64_t s, int pair0, int npairs) {
  __shared__ double lds[GEMM_LDS];
  const int th = blockIdx.x, tj = blockIdx.y;                 // th: 64-row half tile of the C part; tj slow: longest k-ranges first
  const int unit = blockIdx.z / npairs;
  const int64_t colA = 2 * s * (int64_t)(pair0 + blockIdx.z - unit * npairs), rowC = colA + s;
  if (rowC + (int64_t)th * 64 >= Np) return;
  const double* __restrict__ Lm = Lmb.p[unit];
  const double* __restrict__ W = Wb.p[unit];
  double* __restrict__ S = Sb.p[unit];
  v4d acc[2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i >> 1][i & 1] = (v4d){0.0, 0.0, 0.0, 0.0};
  gemm_mainloop_m64<false>(Lm + rowC * ld + colA, ld, (int64_t)th * 64, W + colA * ld + colA, ld, (int64_t)tj * 128, (int64_t)tj * 128, s, acc, lds);
  acc_store_m64(acc, S + (rowC + (int64_t)th * 64) * ld + colA + (int64_t)tj * 128, ld);
}

__global__ void __launch_bounds__(512, 4) k_trtri_X_half(RcBP<double> Wb, RcBP<const double> Sb, int64_t ld, int64_t Np, int64_t s, int pair0,
                                                          int npairs) {
  __shared__ double lds[GEMM_LDS];
  const int tj = blockIdx.x, th = (int)gridDim.y - 1 - (int)blockIdx.y;     // th slow and reversed: longest k-ranges first
  const int unit = blockIdx.z / npairs;
  const int64_t colA = 2 * s * (int64_t)(pair0 + blockIdx.z - unit * npairs), rowC = colA + s;
  if (rowC + (int64_t)th * 64 >= Np) return;
  double* __restrict__ W = Wb.p[unit];
  const double* __restrict__ S = Sb.p[unit];
  v4d acc[2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i >> 1][i & 1] = (v4d){0.0, 0.0, 0.0, 0.0};
  // C^-1 is lower triangular down to the element (its diagonal blocks are stored with their zeros): rows [64 th, 64 th + 64) need k < 64 (th + 1)
  gemm_mainloop_m64<true>(W + rowC * ld + rowC, ld, (int64_t)th * 64, S + rowC * ld + colA, ld, (int64_t)tj * 128, 0, (int64_t)(th + 1) * 64, acc, lds);
  acc_store_m64(acc, W + (rowC + (int64_t)th * 64) * ld + colA + (int64_t)tj * 128, ld);
}

// T phase of `npairs` pairs starting at pair0 on level s, C-part row tiles [ti0, ti0 + nti); X phase of whole pairs.
int rc_launch_trtri_T(rcgp_handle_s* h, int64_t s, int pair0, int npairs, int ti0, int nti) {
  if (npairs <= 0 || nti <= 0) return 0;
  const int64_t st = s / 128;
  RC_BP(const double, Ab, h->A)
  RC_BP(const double, Wb, h->Linv)
  RC_BP(double, Sb, h->S)
  RcProfScope ps(h, RC_K_GEMM, (double)h->nb * (double)npairs * (double)nti * 128.0 * (double)s * (double)s);
  // (half tiles by the tile count of ONE unit. Both forms add up a tile's k-slabs in the same order, so which one runs does not change a
  // bit of the result; measured with 2 and 4 units per launch the half tiles stay the faster form at these sizes: N = 8192, 4 units,
  // L^-1 12.3 ms with them, 12.8 without)
  if (ti0 == 0 && (int64_t)npairs * nti * st <= h->trtri_half_tiles) {
    hipLaunchKernelGGL(k_trtri_T_half, dim3((unsigned)(2 * nti), (unsigned)st, (unsigned)(npairs * h->nb)), dim3(512), 0, h->launch, Ab, Wb, Sb, h->Np,
                       h->Np, s, pair0, npairs);
    RC_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(k_trtri_T<RC_WN>, dim3((unsigned)nti, (unsigned)st, (unsigned)(npairs * h->nb)), dim3(128 * RC_WN), 0, h->launch, Ab, Wb, Sb,
                     h->Np, h->Np, s, pair0, ti0, npairs);
  RC_HIP(hipGetLastError());
  return 0;
}

int rc_launch_trtri_X(rcgp_handle_s* h, int64_t s, int pair0, int npairs) {
  if (npairs <= 0) return 0;
  const int64_t st = s / 128;
  RC_BP(double, Wb, h->Linv)
  RC_BP(const double, Sb, h->S)
  RcProfScope ps(h, RC_K_GEMM, (double)h->nb * (double)npairs * (double)s * (double)s * (double)s);
  if ((int64_t)npairs * st * st <= h->trtri_half_tiles) {
    hipLaunchKernelGGL(k_trtri_X_half, dim3((unsigned)st, (unsigned)(2 * st), (unsigned)(npairs * h->nb)), dim3(512), 0, h->launch, Wb, Sb, h->Np, h->Np,
                       s, pair0, npairs);
    RC_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(k_trtri_X<RC_WN>, dim3((unsigned)st, (unsigned)st, (unsigned)(npairs * h->nb)), dim3(128 * RC_WN), 0, h->launch, Wb, Sb, h->Np,
                     h->Np, s, pair0, npairs);
  RC_HIP(hipGetLastError());
  return 0;
}

// Sum of v over the 64 lanes of a wave, the same value in every lane. Four butterfly steps inside each row of 16 lanes as DPP moves (VALU
// speed: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror -- any pairing will do for a sum), then the four row sums by
// v_readlane. The shuffle form (__shfl_xor = two ds_bpermute_b32 per step and a wait for each of six dependent steps) cost ~600 cycles per sum,
// and the gradient epilogue needs 4 M + 2 of them per wave and tile: ~45 us per tile at M = 10 (round 3), of which this form leaves about half.
// A fixed order, so results stay bit-reproducible from run to run.
template <int CTRL>
__device__ __forceinline__ double rc_dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rc_readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double rc_wave_sum(double v) {
  v += rc_dpp_f64<0xB1>(v);
  v += rc_dpp_f64<0x4E>(v);
  v += rc_dpp_f64<0x141>(v);
  v += rc_dpp_f64<0x140>(v);
  return (rc_readlane_f64(v, 0) + rc_readlane_f64(v, 16)) + (rc_readlane_f64(v, 32) + rc_readlane_f64(v, 48));
}

// ---------------------------------------------------------------------------------------------------------------------
// K^-1 = L^-T L^-1 tile by tile, fused with the reduction for the LML gradient:
//   Wij = alpha_i alpha_j - Kinv_ij ; G_m = sum Wij Kij (z_im - z_jm)^2 ; G_var = sum Wij Kij ; G_noise = tr W
// K^-1 is never written: each lower tile is reduced in the epilogue to M+2 partial sums (row blockIdx of h->partial).
// ---------------------------------------------------------------------------------------------------------------------
// Tiles go out in row order of the lower triangle, heaviest k-ranges first by construction of tri_decode's enumeration (an 8 x 8
// super-block order per XCD was measured: HBM reads -7 %, time +13 % -- DESIGN.md Appendix A.3).
// WIDE (M > RC_MAX_M = 64, up to RC_MAX_M_WIDE): the epilogue stages the Z panels of the tile in chunks of LZ - 1 = 32 dimensions -- per 16-row
// group once for the dot products and once for the per-dimension sums -- instead of once per tile; the per-wave sums take RC_MAX_M_WIDE + 2 LDS
// slots (one workgroup per CU then: a path for the rare wide design, not a fast one). The fast instantiations compile as before.
template <int LZ, int WN, bool WIDE = false>
__global__ void RC_BOUNDS(WN) k_grad(RcBP<const double> Linvb, int64_t ld, int64_t Np, RcBN Nv, int M, RcBP<const double> Zb, RcBP<const double> sqb,
                                     RcBP<const double> alphab, RcBP<const double> FSb, RcBP<double> partialb) {
  const double* __restrict__ Linv = Linvb.p[blockIdx.z];
  const double* __restrict__ Z = Zb.p[blockIdx.z];
  const double* __restrict__ sq = sqb.p[blockIdx.z];
  const double* __restrict__ alpha = alphab.p[blockIdx.z];
  double* __restrict__ partial = partialb.p[blockIdx.z];
  const int64_t N = Nv.v[blockIdx.z];
  const double var = FSb.p[blockIdx.z][0];                    // the unit's kernel variance (F[0][0] of its hyper-parameter block)
  constexpr int ZL = 2 * 128 * LZ;
  constexpr int NW = 2 * WN;                                  // waves per workgroup
  constexpr int ZA = ZL + 4 * 128;                            // + alpha and sq of the tile's 128 rows and 128 columns
  constexpr int RS = (WIDE ? RC_MAX_M_WIDE : RC_MAX_M) + 2;   // per-wave slots of the gradient sums
  constexpr int MCH = LZ - 1;                                 // WIDE: dimensions staged at a time
  __shared__ double lds[(GEMM_LDS > ZA ? GEMM_LDS : ZA) + NW * RS];
  int ti, tj;
  tri_decode(blockIdx.x, ti, tj);
  const int64_t kstart = (int64_t)ti * 128;
  const int64_t tile_id = (int64_t)ti * (ti + 1) / 2 + tj;     // row of `partial`
  v4d acc[4][Geo<WN>::NI];
  acc_zero(acc);
  gemm_mainloop<false, false, WN>(Linv, ld, (int64_t)ti * 128, Linv, ld, (int64_t)tj * 128, kstart, Np, acc, lds);
  RC_LANE_VARS(WN)
  double* zi = lds;
  double* zj = lds + 128 * LZ;
  double* ars = lds + ZL;                                     // alpha_i[128], sq_i[128], alpha_j[128], sq_j[128]
  double* red = lds + (GEMM_LDS > ZA ? GEMM_LDS : ZA);
  // dimensions [c0, c0 + mc) of the tile's Z rows and columns into LDS (WIDE: called by every wave at the same points of uniform loops)
  auto stage_z = [&](int c0, int mc) {
    for (int e = threadIdx.x; e < 128 * mc; e += 128 * WN) {
      const int rr = e / mc, m = e - rr * mc;
      zi[rr * LZ + m] = Z[((int64_t)ti * 128 + rr) * M + c0 + m];
      zj[rr * LZ + m] = Z[((int64_t)tj * 128 + rr) * M + c0 + m];
    }
  };
  if constexpr (!WIDE) stage_z(0, M);
  if (threadIdx.x < 128) {                                    // (through LDS: no 64-bit address arithmetic per element in the epilogue)
    ars[threadIdx.x] = alpha[(int64_t)ti * 128 + threadIdx.x];
    ars[128 + threadIdx.x] = sq[(int64_t)ti * 128 + threadIdx.x];
    ars[256 + threadIdx.x] = alpha[(int64_t)tj * 128 + threadIdx.x];
    ars[384 + threadIdx.x] = sq[(int64_t)tj * 128 + threadIdx.x];
  }
  __syncthreads();
  double gvar = 0.0, gnoise = 0.0;
  double aj[NI_], sj[NI_];
#pragma unroll
  for (int ni = 0; ni < NI_; ++ni) {
    aj[ni] = ars[256 + wc_ + 16 * ni + fr_];
    sj[ni] = ars[384 + wc_ + 16 * ni + fr_];
  }
  const int row0 = ti * 128, col0 = tj * 128;                 // (N < 2^31)
  auto wave_sum = [](double v) { return rc_wave_sum(v); };
  if constexpr (!WIDE) {
    if (lane_ < M) red[wave_ * RS + lane_] = 0.0;              // this wave's M gradient sums, accumulated group by group (M <= 64 lanes)
  } else {
    for (int m = lane_; m < M; m += 64) red[wave_ * RS + m] = 0.0;
  }
  // One 16-row group (8 elements per lane) at a time, START TO END: its dot products in 8 registers, its eight W.K values, and at once
  // their M gradient sums (reduced over the wave and added into the wave's LDS slots by lane 0) -- so the group's accumulators are dead
  // when the next group starts. Written as 32 independent element evaluations followed by one pass per m over all 32, the compiler
  // jams the 32 dynamic-length dot loops together and keeps everything alive: 260 registers per lane in scratch (704 B: 4.4 GB of HBM
  // writes per launch at C2, profiles/r02_pmc_c2.json). The epilogue's instruction count is irrelevant next to a tile's main loop.
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    double dot[4][NI_];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ni = 0; ni < NI_; ++ni) dot[r][ni] = 0.0;
    // (the trip count passes through an opaque statement together with the running sum of the previous group: otherwise the compiler
    // runs the four dot loops first, parks 32 dot products in scratch memory, and evaluates the 32 exps afterwards)
    int mcount = M;
    asm volatile("" : "+s"(mcount), "+v"(gvar));
    auto dot_over = [&](int mc) {                              // dot products over the mc dimensions in LDS
#pragma unroll 1
      for (int m = 0; m < mc; ++m) {
        double zr[4], zc[NI_];
#pragma unroll
        for (int r = 0; r < 4; ++r) zr[r] = zi[(wr_ + 16 * mi + 4 * r + fq_) * LZ + m];
#pragma unroll
        for (int ni = 0; ni < NI_; ++ni) zc[ni] = zj[(wc_ + 16 * ni + fr_) * LZ + m];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ni = 0; ni < NI_; ++ni) dot[r][ni] = fma(zr[r], zc[ni], dot[r][ni]);
      }
    };
    if constexpr (!WIDE) {
      dot_over(mcount);                                        // all M dimensions, staged once per tile
    } else {
#pragma unroll 1
      for (int c0 = 0; c0 < mcount; c0 += MCH) {
        const int mc = (mcount - c0 < MCH) ? mcount - c0 : MCH;
        __syncthreads();                                       // (everybody has read the chunk staged before)
        stage_z(c0, mc);
        __syncthreads();
        dot_over(mc);
      }
    }
    double wk[4][NI_];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wr_ + 16 * mi + 4 * r + fq_;
      const int i = row0 + row;
      const double ai = ars[row], si = ars[128 + row];
#pragma unroll
      for (int ni = 0; ni < NI_; ++ni) {
        const int col = wc_ + 16 * ni + fr_;
        const int j = col0 + col;
        const double kij = var * rc_exp(si + sj[ni] + dot[r][ni]);
        const double wij = ai * aj[ni] - acc[mi][ni][r];
        const bool valid = (i < (int)N) && (j <= i);
        const double wgt = valid ? (j == i ? 1.0 : 2.0) : 0.0;
        wk[r][ni] = wgt * wij * kij;
        gvar += wk[r][ni];
        if (valid && j == i) gnoise += wij;
      }
    }
    auto sums_over = [&](int c0, int mc) {                     // the per-dimension gradient sums of the mc dimensions in LDS (dimension c0 + m)
#pragma unroll 1
      for (int m = 0; m < mc; ++m) {
        double g = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double zim = zi[(wr_ + 16 * mi + 4 * r + fq_) * LZ + m];
#pragma unroll
          for (int ni = 0; ni < NI_; ++ni) {
            const double d = zim - zj[(wc_ + 16 * ni + fr_) * LZ + m];
            g = fma(wk[r][ni], d * d, g);
          }
        }
        g = wave_sum(g);
        if (lane_ == 0) red[wave_ * RS + c0 + m] += g;
      }
    };
    if constexpr (!WIDE) {
      sums_over(0, mcount);
    } else {
#pragma unroll 1
      for (int c0 = 0; c0 < mcount; c0 += MCH) {
        const int mc = (mcount - c0 < MCH) ? mcount - c0 : MCH;
        __syncthreads();
        stage_z(c0, mc);
        __syncthreads();
        sums_over(c0, mc);
      }
    }
  }
  gvar = wave_sum(gvar);
  gnoise = wave_sum(gnoise);
  if (lane_ == 0) {
    red[wave_ * RS + M] = gvar;
    red[wave_ * RS + M + 1] = gnoise;
  }
  __syncthreads();
  if (threadIdx.x < M + 2) {
    const int m = threadIdx.x;
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w * RS + m];      // fixed order: bit-reproducible
    partial[tile_id * (M + 2) + m] = s;
  }
}

int rc_launch_grad(rcgp_handle_s* h, int* nrows) {
  const int64_t T = h->Np / 128;
  const int64_t nb = T * (T + 1) / 2;
  int rc = rc_ensure_partial(h, (size_t)nb * (h->M + 2));      // (every unit of a batched call)
  if (rc) return rc;
  const double np = (double)h->Np;
  g_rc_stat[2] += h->nb;
  RC_BP(const double, Lb, h->Linv)
  RC_BP(const double, Zb, h->Z)
  RC_BP(const double, sb, h->sq)
  RC_BP(const double, ab, h->alpha)
  RC_BP(const double, Fb, h->FS_d)
  RC_BP(double, pb, h->partial)
  RcProfScope ps(h, RC_K_GRAD, (double)h->nb * np * np * np / 3.0);
  const dim3 grid((unsigned)nb, 1, (unsigned)h->nb);
  if (h->M <= 32)
    hipLaunchKernelGGL((k_grad<33, RC_WN>), grid, dim3(128 * RC_WN), 0, h->launch, Lb, h->Np, h->Np, rc_bn(h), h->M, Zb, sb, ab, Fb, pb);
  else if (h->M <= RC_MAX_M)
    hipLaunchKernelGGL((k_grad<65, RC_WN>), grid, dim3(128 * RC_WN), 0, h->launch, Lb, h->Np, h->Np, rc_bn(h), h->M, Zb, sb, ab, Fb, pb);
  else
    hipLaunchKernelGGL((k_grad<33, RC_WN, true>), grid, dim3(128 * RC_WN), 0, h->launch, Lb, h->Np, h->Np, rc_bn(h), h->M, Zb, sb, ab, Fb, pb);
  RC_HIP(hipGetLastError());
  *nrows = (int)nb;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same fused K^-1 + gradient reduction for a covariant GP of L outputs (gpf/kernels.py, gpf/likelihoods.py; the
// reference differentiates by TF autodiff, gpr/models.py:359-361). System row a = (output block, sample); tiles never
// straddle blocks (every block is padded to a multiple of 128 rows). With W = alpha alpha^T - K^-1, E the unit-variance
// kernel, K = F_lj E, u_a = x_n / ell_l and d = u_a - u_b, each lower tile of block pair (bi, bj) yields 2M + 2 sums over its
// valid elements, weighted 2 off the system diagonal (the mirrored element lives in the upper triangle):
//   [m]       bi == bj: sum W K d_m^2        bi != bj: sum W K d_m u_am     (goes to ell[bi][m])
//   [M + m]   bi == bj: 0                    bi != bj: sum W K d_m u_bm     (goes to ell[bj][m], negated)
//   [2M]      sum W E                        (d LML / d F)
//   [2M + 1]  sum W over the elements with equal in-block index   (d LML / d Sigma)
// rc_grad_finish_mo adds the tiles of every block pair and applies the factors.
// ---------------------------------------------------------------------------------------------------------------------
// WIDE (M > RC_MAX_M = 64): Z chunks of LZ - 1 = 32 dimensions per 16-row group, as in k_grad<., ., WIDE>.
template <int LZ, int WN, bool WIDE = false>
__global__ void RC_BOUNDS(WN) k_grad_mo(const double* __restrict__ Linv, int64_t ld, int64_t Np, int64_t N, int M,
                                        const double* __restrict__ Z, const double* __restrict__ sq, const double* __restrict__ alpha,
                                        const double* __restrict__ FS, int L, int tb, double* __restrict__ partial) {
  constexpr int ZL = 2 * 128 * LZ;
  constexpr int NW = 2 * WN;
  constexpr int RW = 2 * (WIDE ? RC_MAX_M_WIDE : RC_MAX_M) + 2;
  constexpr int MCH = LZ - 1;                                 // WIDE: dimensions staged at a time
  constexpr int ZA = ZL + 4 * 128;                            // + alpha and sq of the tile's 128 rows and 128 columns
  __shared__ double lds[(GEMM_LDS > ZA ? GEMM_LDS : ZA) + NW * RW];
  int ti, tj;
  tri_decode(blockIdx.x, ti, tj);
  v4d acc[4][Geo<WN>::NI];
  acc_zero(acc);
  gemm_mainloop<false, false, WN>(Linv, ld, (int64_t)ti * 128, Linv, ld, (int64_t)tj * 128, (int64_t)ti * 128, Np, acc, lds);
  RC_LANE_VARS(WN)
  double* zi = lds;
  double* zj = lds + 128 * LZ;
  double* ars = lds + ZL;                                     // alpha_i[128], sq_i[128], alpha_j[128], sq_j[128]
  double* red = lds + (GEMM_LDS > ZA ? GEMM_LDS : ZA);
  auto stage_z = [&](int c0, int mc) {                        // dimensions [c0, c0 + mc) of the tile's Z rows and columns
    for (int e = threadIdx.x; e < 128 * mc; e += 128 * WN) {
      const int rr = e / mc, m = e - rr * mc;
      zi[rr * LZ + m] = Z[((int64_t)ti * 128 + rr) * M + c0 + m];
      zj[rr * LZ + m] = Z[((int64_t)tj * 128 + rr) * M + c0 + m];
    }
  };
  if constexpr (!WIDE) stage_z(0, M);
  if (threadIdx.x < 128) {                                    // (through LDS: no 64-bit address arithmetic per element in the epilogue)
    ars[threadIdx.x] = alpha[(int64_t)ti * 128 + threadIdx.x];
    ars[128 + threadIdx.x] = sq[(int64_t)ti * 128 + threadIdx.x];
    ars[256 + threadIdx.x] = alpha[(int64_t)tj * 128 + threadIdx.x];
    ars[384 + threadIdx.x] = sq[(int64_t)tj * 128 + threadIdx.x];
  }
  __syncthreads();
  const int bi = ti / tb, bj = tj / tb;
  const bool same = (bi == bj);
  const double var = FS[bi * L + bj];
  const int64_t ioff = (int64_t)bi * tb * 128, joff = (int64_t)bj * tb * 128;
  double ge = 0.0, gdiag = 0.0;
  double aj[NI_], sj[NI_];
#pragma unroll
  for (int ni = 0; ni < NI_; ++ni) {
    aj[ni] = ars[256 + wc_ + 16 * ni + fr_];
    sj[ni] = ars[384 + wc_ + 16 * ni + fr_];
  }
  auto wave_sum = [](double v) { return rc_wave_sum(v); };
  for (int m = lane_; m < 2 * M; m += 64) red[wave_ * RW + m] = 0.0;   // this wave's 2M gradient sums, accumulated group by group
  // one 16-row group at a time, start to end, as in k_grad (its accumulators are dead when the next group starts: no scratch)
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    double dot[4][NI_];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ni = 0; ni < NI_; ++ni) dot[r][ni] = 0.0;
    int mcount = M;
    asm volatile("" : "+s"(mcount), "+v"(ge));
    auto dot_over = [&](int mc) {
#pragma unroll 1
      for (int m = 0; m < mc; ++m) {
        double zr[4], zc[NI_];
#pragma unroll
        for (int r = 0; r < 4; ++r) zr[r] = zi[(wr_ + 16 * mi + 4 * r + fq_) * LZ + m];
#pragma unroll
        for (int ni = 0; ni < NI_; ++ni) zc[ni] = zj[(wc_ + 16 * ni + fr_) * LZ + m];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ni = 0; ni < NI_; ++ni) dot[r][ni] = fma(zr[r], zc[ni], dot[r][ni]);
      }
    };
    if constexpr (!WIDE) {
      dot_over(mcount);
    } else {
#pragma unroll 1
      for (int c0 = 0; c0 < mcount; c0 += MCH) {
        const int mc = (mcount - c0 < MCH) ? mcount - c0 : MCH;
        __syncthreads();                                       // (everybody has read the chunk staged before)
        stage_z(c0, mc);
        __syncthreads();
        dot_over(mc);
      }
    }
    double wk[4][NI_];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wr_ + 16 * mi + 4 * r + fq_;
      const int i = ti * 128 + row;
      const double ai = ars[row], si = ars[128 + row];
#pragma unroll
      for (int ni = 0; ni < NI_; ++ni) {
        const int col = wc_ + 16 * ni + fr_;
        const int j = tj * 128 + col;
        const double eij = rc_exp(si + sj[ni] + dot[r][ni]);
        const double wij = ai * aj[ni] - acc[mi][ni][r];
        const int ii = i - (int)ioff, jj = j - (int)joff;
        const bool valid = (ii < (int)N) && (jj < (int)N) && (j <= i);
        const double wgt = valid ? (j == i ? 1.0 : 2.0) : 0.0;
        const double we = wgt * wij * eij;
        wk[r][ni] = var * we;
        ge += we;
        if (ii == jj) gdiag += wgt * wij;
      }
    }
    auto sums_over = [&](int c0, int mc) {                     // the sums of the mc dimensions in LDS (dimension c0 + m)
#pragma unroll 1
      for (int m = 0; m < mc; ++m) {
        double ga = 0.0, gb = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double zim = zi[(wr_ + 16 * mi + 4 * r + fq_) * LZ + m];
#pragma unroll
          for (int ni = 0; ni < NI_; ++ni) {
            const double zjm = zj[(wc_ + 16 * ni + fr_) * LZ + m];
            const double d = zim - zjm;
            const double wd = wk[r][ni] * d;
            ga = fma(wd, same ? d : zim, ga);
            gb = fma(wd, same ? 0.0 : zjm, gb);
          }
        }
        ga = wave_sum(ga);
        gb = wave_sum(gb);
        if (lane_ == 0) {
          red[wave_ * RW + c0 + m] += ga;
          red[wave_ * RW + M + c0 + m] += gb;
        }
      }
    };
    if constexpr (!WIDE) {
      sums_over(0, mcount);
    } else {
#pragma unroll 1
      for (int c0 = 0; c0 < mcount; c0 += MCH) {
        const int mc = (mcount - c0 < MCH) ? mcount - c0 : MCH;
        __syncthreads();
        stage_z(c0, mc);
        __syncthreads();
        sums_over(c0, mc);
      }
    }
  }
  ge = wave_sum(ge);
  gdiag = wave_sum(gdiag);
  if (lane_ == 0) {
    red[wave_ * RW + 2 * M] = ge;
    red[wave_ * RW + 2 * M + 1] = gdiag;
  }
  __syncthreads();
  for (int m = threadIdx.x; m < 2 * M + 2; m += 128 * WN) {     // (one pass up to M = 64)
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w * RW + m];
    partial[(int64_t)blockIdx.x * (2 * M + 2) + m] = s;
  }
}

int rc_launch_grad_mo(rcgp_handle_s* h, int* nrows) {
  const int64_t T = h->Np / 128;
  const int64_t nb = T * (T + 1) / 2;
  int rc = rc_ensure_partial(h, (size_t)nb * (2 * h->M + 2) + (size_t)(h->L * (h->L + 1) / 2) * (2 * h->M + 2));
  if (rc) return rc;
  const double np = (double)h->Np;
  const int tb = (int)(h->Nb / 128);
  RcProfScope ps(h, RC_K_GRAD, np * np * np / 3.0);
  if (h->M <= 32)
    hipLaunchKernelGGL((k_grad_mo<33, RC_WN>), dim3((unsigned)nb), dim3(128 * RC_WN), 0, h->launch, h->Linv, h->Np, h->Np, h->N, h->M, h->Z,
                       h->sq, h->alpha, h->FS_d, h->L, tb, h->partial);
  else if (h->M <= RC_MAX_M)
    hipLaunchKernelGGL((k_grad_mo<65, RC_WN>), dim3((unsigned)nb), dim3(128 * RC_WN), 0, h->launch, h->Linv, h->Np, h->Np, h->N, h->M, h->Z,
                       h->sq, h->alpha, h->FS_d, h->L, tb, h->partial);
  else
    hipLaunchKernelGGL((k_grad_mo<33, RC_WN, true>), dim3((unsigned)nb), dim3(128 * RC_WN), 0, h->launch, h->Linv, h->Np, h->Np, h->N, h->M,
                       h->Z, h->sq, h->alpha, h->FS_d, h->L, tb, h->partial);
  RC_HIP(hipGetLastError());
  *nrows = (int)nb;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Predictive variance term: colsum((L^-1 K*)^2). A = Linv (Np x Np, lower) times KsT^T (KsT is np x Np, one test point
// per row). The product is never written: each tile is squared and column-summed into partial[ti][j].
// ---------------------------------------------------------------------------------------------------------------------
template <int WN>
__global__ void RC_BOUNDS(WN) k_predict_var(const double* __restrict__ Linv, int64_t ld, const double* __restrict__ KsT, int64_t ldk,
                                            int64_t np, double* __restrict__ partial) {
  __shared__ double lds[GEMM_LDS];
  __shared__ double colsum[2][128];
  const int tj = blockIdx.x;
  const int ti = gridDim.y - 1 - blockIdx.y;          // heavy row tiles first
  v4d acc[4][Geo<WN>::NI];
  acc_zero(acc);
  gemm_mainloop<true, true, WN>(Linv, ld, (int64_t)ti * 128, KsT, ldk, (int64_t)tj * 128, 0, (int64_t)(ti + 1) * 128, acc, lds);
  RC_LANE_VARS(WN)
#pragma unroll
  for (int ni = 0; ni < NI_; ++ni) {
    double s = 0.0;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) s = fma(acc[mi][ni][r], acc[mi][ni][r], s);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (fq_ == 0) colsum[wave_ / WN][wc_ + 16 * ni + fr_] = s;
  }
  __syncthreads();
  if (threadIdx.x < 128)
    partial[(int64_t)ti * np + (int64_t)tj * 128 + threadIdx.x] = colsum[0][threadIdx.x] + colsum[1][threadIdx.x];
}

__global__ void k_colreduce(const double* __restrict__ partial, int64_t rows, int64_t n, double* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double s = 0.0;
  for (int64_t r = 0; r < rows; ++r) s += partial[r * n + j];
  out[j] = s;
}

int rc_launch_predict_var(rcgp_handle_s* h, int64_t np) {
  const int64_t T = h->Np / 128;
  int rc = rc_ensure_partial(h, (size_t)T * np);
  if (rc) return rc;
  {
    RcProfScope ps(h, RC_K_GEMM, (double)h->Np * (double)h->Np * (double)np);
    hipLaunchKernelGGL(k_predict_var<RC_WN>, dim3((unsigned)(np / 128), (unsigned)T), dim3(128 * RC_WN), 0, h->launch, h->Linv, h->Np, h->KsT,
                       h->Np, np, h->partial);
    RC_HIP(hipGetLastError());
  }
  {
    RcProfScope ps(h, RC_K_MISC, 0.0);
    hipLaunchKernelGGL(k_colreduce, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, h->launch, h->partial, T, np, h->pvar);
    RC_HIP(hipGetLastError());
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// predict_gradient (reference gpr/models.py:386-415): V = L^-1 D^T stored (D = d k(X, x)/dx, one row per (point, dim)),
// then C = V^T V. Two thin kernels on the same mainloop with plain store epilogues.
// ---------------------------------------------------------------------------------------------------------------------
template <int WN>
__global__ void RC_BOUNDS(WN) k_linv_times_rows(const double* __restrict__ Linv, int64_t ld, const double* __restrict__ D, int64_t ldd,
                                                double* __restrict__ V, int64_t ldv) {
  __shared__ double lds[GEMM_LDS];
  const int tj = blockIdx.x;
  const int ti = gridDim.y - 1 - blockIdx.y;
  v4d acc[4][Geo<WN>::NI];
  acc_zero(acc);
  gemm_mainloop<true, true, WN>(Linv, ld, (int64_t)ti * 128, D, ldd, (int64_t)tj * 128, 0, (int64_t)(ti + 1) * 128, acc, lds);
  acc_store<WN>(acc, V + (int64_t)ti * 128 * ldv + (int64_t)tj * 128, ldv);
}

template <int WN>
__global__ void RC_BOUNDS(WN) k_vtv(const double* __restrict__ V, int64_t ldv, int64_t k0, int64_t k1, double* __restrict__ C, int64_t ldc) {
  __shared__ double lds[GEMM_LDS];
  const int tj = blockIdx.x, ti = blockIdx.y;
  v4d acc[4][Geo<WN>::NI];
  acc_zero(acc);
  gemm_mainloop<false, false, WN>(V, ldv, (int64_t)ti * 128, V, ldv, (int64_t)tj * 128, k0, k1, acc, lds);
  acc_store<WN>(acc, C + (int64_t)ti * 128 * ldc + (int64_t)tj * 128, ldc);
}

// V[:, col0 : col0 + rows_chunk) = Linv * D^T for the chunk of derivative rows currently in KsT (rows_chunk a multiple of 128); V is
// Np x ldv row-major. predict_gradient feeds the rows through KsT 4096 at a time, so the number of gradient components is bounded by
// memory only.
int rc_launch_linv_rows(rcgp_handle_s* h, int64_t rows_chunk, double* V, int64_t ldv, int64_t col0) {
  const int64_t T = h->Np / 128, R = rows_chunk / 128;
  RcProfScope ps(h, RC_K_GEMM, (double)h->Np * (double)h->Np * (double)rows_chunk);
  hipLaunchKernelGGL(k_linv_times_rows<RC_WN>, dim3((unsigned)R, (unsigned)T), dim3(128 * RC_WN), 0, h->launch, h->Linv, h->Np, h->KsT, h->Np,
                     V + col0, ldv);
  RC_HIP(hipGetLastError());
  return 0;
}

// C = V^T V over all rows of V; with per_block one product per output block of rows (a covariant GP's predict_gradient keeps
// the training output index, gpr/models.py:398: 'LNlOM, LNlom -> OLolMm'), stored one after the other in C.
int rc_launch_vtv(rcgp_handle_s* h, int64_t rows_padded, const double* V, double* C, bool per_block) {
  const int64_t R = rows_padded / 128;
  const int nprod = per_block ? h->L : 1;
  const int64_t span = per_block ? h->Nb : h->Np;
  for (int b = 0; b < nprod; ++b) {
    RcProfScope ps(h, RC_K_GEMM, 2.0 * (double)span * (double)rows_padded * (double)rows_padded);
    hipLaunchKernelGGL(k_vtv<RC_WN>, dim3((unsigned)R, (unsigned)R), dim3(128 * RC_WN), 0, h->launch, V, rows_padded, b * span, (b + 1) * span,
                       C + (size_t)b * rows_padded * rows_padded, rows_padded);
    RC_HIP(hipGetLastError());
  }
  return 0;
}
