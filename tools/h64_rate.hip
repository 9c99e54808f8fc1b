// Rate of the half-tile update kernel (k_gemm_nt_sub_h64: the near / far updates of the panel chain) on a far-update-shaped launch, by K, beside
// experimental variants: a two-slab register ring at <= 84 VGPRs (three workgroups per CU) and the 128 x 128 kernel at the same K.
//   hipcc -O3 --offload-arch=gfx950 tools/h64_rate.hip -o tools/dev/h64_rate
#include "../rom-comma_amd/csrc/gemm.hip"
#include <stdio.h>
int rc_ensure_partial(rcgp_handle_s*, size_t) { return 0; }

// Variant: two-slab ring, capped at 6 waves per SIMD (84 VGPRs): three workgroups per CU (3 x 52 KB of LDS).
__global__ void __launch_bounds__(512, 6) k_h64_r2(double* __restrict__ C, int64_t ldc, const double* __restrict__ A, int64_t lda,
                                                    const double* __restrict__ B, int64_t ldb, int kk, int64_t row0, int64_t col0) {
  __shared__ double lds[2 * 3 * 64 * LDK];
  const int tj = blockIdx.x, th = blockIdx.y;
  if (col0 + (int64_t)tj * 128 > row0 + (int64_t)th * 64) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = (wave >> 2) * 32, wc = (wave & 3) * 32;
  const int fr = lane & 15, fq = lane >> 4;
  const int nk = kk >> 4;
  const int skk = (t & 7) * 2, sr = t >> 3;
  const double* ap = A + ((int64_t)th * 64 + sr) * lda + skk;
  const double* bp = B + ((int64_t)tj * 128 + sr) * ldb + skk;
  constexpr int STAGE = 3 * 64 * LDK;
  double* las = lds + sr * LDK + skk;
  double* lbs = lds + 64 * LDK + sr * LDK + skk;
  double2 ra0, rb00, rb01, ra1, rb10, rb11;
#define LD(KK, RA, RB0, RB1) RA = *reinterpret_cast<const double2*>(ap + (KK)); RB0 = *reinterpret_cast<const double2*>(bp + (KK)); RB1 = *reinterpret_cast<const double2*>(bp + 64 * ldb + (KK));
#define ST(S, RA, RB0, RB1) las[(S) * STAGE] = RA.x; las[(S) * STAGE + 1] = RA.y; lbs[(S) * STAGE] = RB0.x; lbs[(S) * STAGE + 1] = RB0.y; lbs[(S) * STAGE + 64 * LDK] = RB1.x; lbs[(S) * STAGE + 64 * LDK + 1] = RB1.y;
  LD(0, ra0, rb00, rb01)
  LD(16, ra1, rb10, rb11)
  v4d acc[2][2];
  double* Ct = C + ((int64_t)th * 64 + wr + fq) * ldc + (int64_t)tj * 128 + wc + fr;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = Ct[(int64_t)(16 * mi + 4 * r) * ldc + 16 * ni];
  ST(0, ra0, rb00, rb01)
  __syncthreads();
#define STEP(U, RA, RB0, RB1, NA, NB0, NB1)                                                       \
  {                                                                                               \
    const int kt = kt0 + U;                                                                       \
    const double* la = lds + (U & 1) * STAGE;                                                     \
    const double* lb = la + 64 * LDK;                                                             \
    const int kn = ((kt + 2 < nk) ? kt + 2 : nk - 1) * 16;                                        \
    LD(kn, RA, RB0, RB1)                                                                          \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                               \
      double af[2], bf[2];                                                                        \
      _Pragma("unroll") for (int x = 0; x < 2; ++x) af[x] = la[(wr + 16 * x + fr) * LDK + 4 * s + fq]; \
      _Pragma("unroll") for (int x = 0; x < 2; ++x) bf[x] = lb[(wc + 16 * x + fr) * LDK + 4 * s + fq]; \
      _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                            \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                            \
        acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi], bf[ni], acc[mi][ni], 0, 0, 1); \
    }                                                                                             \
    ST((U + 1) & 1, NA, NB0, NB1)                                                                 \
    __syncthreads();                                                                              \
  }
  for (int kt0 = 0; kt0 < nk; kt0 += 2) {
    STEP(0, ra0, rb00, rb01, ra1, rb10, rb11)
    STEP(1, ra1, rb10, rb11, ra0, rb00, rb01)
  }
  int64_t ldo = ldc;
  asm volatile("" : "+s"(ldo));
  double* Co = C + ((int64_t)th * 64 + wr + fq) * ldo + (int64_t)tj * 128 + wc + fr;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) Co[(int64_t)(16 * mi + 4 * r) * ldo + 16 * ni] = acc[mi][ni][r];
}

__global__ void k_fill(double* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed * 40503u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = ((double)(x & 0xffffff) / 16777216.0 - 0.5) * 1e-3;
  }
}

template <typename F>
static float best_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0, 0);
    launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const int64_t ld = 8192;
  double* Lm; double* Cm;
  hipMalloc(&Lm, ld * ld * sizeof(double)); hipMalloc(&Cm, ld * ld * sizeof(double));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, Lm, (size_t)ld * ld, 1u);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, Cm, (size_t)ld * ld, 2u);
  hipDeviceSynchronize();
  // the far update of step j (blocks): C[rows >= j+2, cols j+4 ..] -= L[rows, j-1..j+1) L[cols, j-1..j+1)^T, lower tiles only
  for (int blocks : {56, 40, 24}) {
    const int64_t j = 64 - blocks - 2;                       // so that `blocks` block rows remain below
    const int64_t row0 = (j + 2) * 128, col0 = (j + 4) * 128, m = ld - row0, n = ld - col0;
    double tiles = 0;
    for (int64_t r = 0; r < m / 128; ++r) for (int64_t c = 0; c < n / 128; ++c) if (col0 + c * 128 <= row0 + r * 128) tiles += 1;
    for (int kk : {128, 256, 512}) {
      if (kk / 128 > j + 1) continue;
      const double* A = Lm + row0 * ld + (j + 1) * 128 - kk;
      const double* B = Lm + col0 * ld + (j + 1) * 128 - kk;
      double* C = Cm + row0 * ld + col0;
      const double flops = tiles * 2.0 * 128 * 128 * kk;
      float t0 = best_ms([&] { hipLaunchKernelGGL(k_gemm_nt_sub_h64, dim3((unsigned)(n / 128), (unsigned)(m / 64)), dim3(512), 0, 0, C, ld, A, ld, B, ld, kk, row0, col0); });
      float t1 = best_ms([&] { hipLaunchKernelGGL(k_h64_r2, dim3((unsigned)(n / 128), (unsigned)(m / 64)), dim3(512), 0, 0, C, ld, A, ld, B, ld, kk, row0, col0); });
      float t2 = best_ms([&] { hipLaunchKernelGGL((k_gemm_nt_sub<4, 3>), dim3((unsigned)(n / 128), (unsigned)(m / 128)), dim3(512), 0, 0, C, ld, A, ld, B, ld, kk, row0, col0); });
      printf("%2d blocks below, K=%3d (%5.0f tiles, %6.2f GFLOP): h64 %7.1f us %5.1f TF/s | h64 two-slab ring, 3 per CU %7.1f us %5.1f TF/s | 128x128 %7.1f us %5.1f TF/s\n", blocks, kk,
             tiles, flops / 1e9, t0 * 1e3, flops / t0 / 1e9, t1 * 1e3, flops / t1 / 1e9, t2 * 1e3, flops / t2 / 1e9);
    }
  }
  return 0;
}
