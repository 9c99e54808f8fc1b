// Where does a generation of half tiles of the K <= 512 update kernel (k_gemm_nt_sub_h64, rom-comma_amd/csrc/gemm.hip) spend its time? The
// same kernel body with its C-tile load and / or store switched off, and a variant that walks TP consecutive half tiles per workgroup
// with the NEXT tile's C values and first operand slabs requested while the current tile is multiplied -- on the far-update-shaped
// launches of a factorisation at N = 8192 (56 / 40 / 24 block rows below), K = 128 / 256 / 512.
//   hipcc -O3 --offload-arch=gfx950 tools/h64_parts.hip -o tools/dev/h64_parts && tools/dev/h64_parts
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
#define LDK 17

// ---- the product kernel's body: one 64 x 128 half tile per workgroup (th, tj from the grid), four-slab register ring
template <bool LOADC, bool STOREC>
__global__ void __launch_bounds__(512, 4) k_h64(double* __restrict__ C, int64_t ldc, const double* __restrict__ A, int64_t lda, const double* __restrict__ B,
                                                int64_t ldb, int kk, int64_t row0, int64_t col0) {
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
#define H_LOAD(KK, RA, RB0, RB1) RA = *reinterpret_cast<const double2*>(ap + (KK)); RB0 = *reinterpret_cast<const double2*>(bp + (KK)); RB1 = *reinterpret_cast<const double2*>(bp + 64 * ldb + (KK));
#define H_STORE(ST, RA, RB0, RB1) las[(ST) * STAGE] = RA.x; las[(ST) * STAGE + 1] = RA.y; lbs[(ST) * STAGE] = RB0.x; lbs[(ST) * STAGE + 1] = RB0.y; lbs[(ST) * STAGE + 64 * LDK] = RB1.x; lbs[(ST) * STAGE + 64 * LDK + 1] = RB1.y;
  double2 ra0, ra1, ra2, ra3, rb00, rb01, rb10, rb11, rb20, rb21, rb30, rb31;
  H_LOAD(0, ra0, rb00, rb01)
  H_LOAD(16, ra1, rb10, rb11)
  H_LOAD(32, ra2, rb20, rb21)
  H_LOAD(48, ra3, rb30, rb31)
  v4d acc[2][2];
  double* Ct = C + ((int64_t)th * 64 + wr + fq) * ldc + (int64_t)tj * 128 + wc + fr;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = LOADC ? Ct[(int64_t)(16 * mi + 4 * r) * ldc + 16 * ni] : 0.0;
  H_STORE(0, ra0, rb00, rb01)
  __syncthreads();
#define H_STEP(U, RA, RB0, RB1, NA, NB0, NB1)                                                        \
  {                                                                                                  \
    const int kt = kt0 + U;                                                                          \
    const double* la = lds + (U & 1) * STAGE;                                                        \
    const double* lb = la + 64 * LDK;                                                                \
    const int kn = ((kt + 4 < nk) ? kt + 4 : nk - 1) * 16;                                           \
    H_LOAD(kn, RA, RB0, RB1)                                                                         \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                  \
      double af[2], bf[2];                                                                           \
      _Pragma("unroll") for (int x = 0; x < 2; ++x) af[x] = la[(wr + 16 * x + fr) * LDK + 4 * s + fq]; \
      _Pragma("unroll") for (int x = 0; x < 2; ++x) bf[x] = lb[(wc + 16 * x + fr) * LDK + 4 * s + fq]; \
      _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                               \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                               \
        acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi], bf[ni], acc[mi][ni], 0, 0, 1);    \
    }                                                                                                \
    H_STORE((U + 1) & 1, NA, NB0, NB1)                                                               \
    __syncthreads();                                                                                 \
  }
  for (int kt0 = 0; kt0 < nk; kt0 += 4) {
    H_STEP(0, ra0, rb00, rb01, ra1, rb10, rb11)
    H_STEP(1, ra1, rb10, rb11, ra2, rb20, rb21)
    H_STEP(2, ra2, rb20, rb21, ra3, rb30, rb31)
    H_STEP(3, ra3, rb30, rb31, ra0, rb00, rb01)
  }
  int64_t ldo = ldc;
  asm volatile("" : "+s"(ldo));
  double* Co = C + ((int64_t)th * 64 + wr + fq) * ldo + (int64_t)tj * 128 + wc + fr;
  if (STOREC) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) Co[(int64_t)(16 * mi + 4 * r) * ldo + 16 * ni] = acc[mi][ni][r];
  } else if (acc[0][0][0] == 1.2345e300) {
    Co[0] = acc[1][1][3] + acc[0][1][2] + acc[1][0][1];                 // (keeps the products alive)
  }
}

// ---- TP half tiles per workgroup, walking down a block column (th = TP * blockIdx.y + p): two-slab register ring running ACROSS the tile
// boundaries (the slabs of tile p + 1 are requested during the last slabs of tile p), the C values of tile p + 1 requested while tile p is
// multiplied, the C stores of tile p draining behind the first slabs of tile p + 1.
template <int TP>
__global__ void __launch_bounds__(512, 4) k_h64_pipe(double* __restrict__ C, int64_t ldc, const double* __restrict__ A, int64_t lda,
                                                      const double* __restrict__ B, int64_t ldb, int kk, int64_t row0, int64_t col0, int nth) {
  __shared__ double lds[2 * 3 * 64 * LDK];
  const int tj = blockIdx.x;
  int th0 = blockIdx.y * TP;
  // tiles strictly above the diagonal are skipped: the first valid half tile of this block column
  const int64_t first = (col0 + (int64_t)tj * 128 - row0 + 63) / 64;
  if (first > th0) th0 = (int)first;
  int th1 = blockIdx.y * TP + TP;
  if (th1 > nth) th1 = nth;
  if (th0 >= th1) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = (wave >> 2) * 32, wc = (wave & 3) * 32;
  const int fr = lane & 15, fq = lane >> 4;
  const int nk = kk >> 4;
  const int skk = (t & 7) * 2, sr = t >> 3;
  const double* bp = B + ((int64_t)tj * 128 + sr) * ldb + skk;
  constexpr int STAGE = 3 * 64 * LDK;
  double* las = lds + sr * LDK + skk;
  double* lbs = lds + 64 * LDK + sr * LDK + skk;
  const int total = (th1 - th0) * nk;                         // slabs of the whole walk
  auto a_ptr = [&](int g) { const int p = g / nk, k = g - p * nk; return A + ((int64_t)(th0 + p) * 64 + sr) * lda + skk + 16 * k; };
  auto b_off = [&](int g) { return 16 * (g % nk); };
  double2 ra0, rb00, rb01, ra1, rb10, rb11;
#define P_LOAD(G, RA, RB0, RB1) { const int g_ = ((G) < total) ? (G) : total - 1; RA = *reinterpret_cast<const double2*>(a_ptr(g_)); RB0 = *reinterpret_cast<const double2*>(bp + b_off(g_)); RB1 = *reinterpret_cast<const double2*>(bp + 64 * ldb + b_off(g_)); }
  P_LOAD(0, ra0, rb00, rb01)
  P_LOAD(1, ra1, rb10, rb11)
  v4d acc[2][2], nxt[2][2];
  auto c_ptr = [&](int th) { return C + ((int64_t)th * 64 + wr + fq) * ldc + (int64_t)tj * 128 + wc + fr; };
  {
    const double* Ct = c_ptr(th0);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mi][ni][r] = Ct[(int64_t)(16 * mi + 4 * r) * ldc + 16 * ni];
  }
  H_STORE(0, ra0, rb00, rb01)
  __syncthreads();
#define P_STEP(U, RA, RB0, RB1, NA, NB0, NB1)                                                        \
  {                                                                                                  \
    const double* la = lds + (U & 1) * STAGE;                                                        \
    const double* lb = la + 64 * LDK;                                                                \
    P_LOAD(g + U + 2, RA, RB0, RB1)                                                                  \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                  \
      double af[2], bf[2];                                                                           \
      _Pragma("unroll") for (int x = 0; x < 2; ++x) af[x] = la[(wr + 16 * x + fr) * LDK + 4 * s + fq]; \
      _Pragma("unroll") for (int x = 0; x < 2; ++x) bf[x] = lb[(wc + 16 * x + fr) * LDK + 4 * s + fq]; \
      _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                               \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                               \
        acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi], bf[ni], acc[mi][ni], 0, 0, 1);    \
    }                                                                                                \
    H_STORE((U + 1) & 1, NA, NB0, NB1)                                                               \
    __syncthreads();                                                                                 \
  }
  int g = 0;
  for (int th = th0; th < th1; ++th) {
    const bool more = th + 1 < th1;
    if (more) {                                               // the next tile's C values: in flight for the whole of this tile's multiplication
      const double* Cn = c_ptr(th + 1);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) nxt[mi][ni][r] = Cn[(int64_t)(16 * mi + 4 * r) * ldc + 16 * ni];
    }
    for (int k = 0; k < nk; k += 2, g += 2) {
      P_STEP(0, ra0, rb00, rb01, ra1, rb10, rb11)
      P_STEP(1, ra1, rb10, rb11, ra0, rb00, rb01)
    }
    double* Co = c_ptr(th);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) Co[(int64_t)(16 * mi + 4 * r) * ldc + 16 * ni] = acc[mi][ni][r];
    if (more) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = nxt[mi][ni];
    }
  }
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

__global__ void k_diff(const double* a, const double* b, size_t n, double* out) {
  double m = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmax(m, fabs(a[i] - b[i]));
  if (m > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(out), __double_as_longlong(m));
}

int main() {
  const int64_t ld = 8192;
  double *Lm, *Cm, *C2, *dmax;
  hipMalloc(&Lm, ld * ld * sizeof(double)); hipMalloc(&Cm, ld * ld * sizeof(double)); hipMalloc(&C2, ld * ld * sizeof(double)); hipMalloc(&dmax, 8);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, Lm, (size_t)ld * ld, 1u);
  for (int units : {1, 2}) {                                   // (two launches back to back stand for a batch of two units)
    for (int blocks : {56, 40, 24}) {
      const int64_t j = 64 - blocks - 2;
      const int64_t row0 = (j + 2) * 128, col0 = (j + 4) * 128, m = ld - row0, n = ld - col0;
      double tiles = 0;
      for (int64_t r = 0; r < m / 128; ++r) for (int64_t c = 0; c < n / 128; ++c) if (col0 + c * 128 <= row0 + r * 128) tiles += 1;
      for (int kk : {128, 256, 512}) {
        if (kk / 128 > j + 1) continue;
        const double* A = Lm + row0 * ld + (j + 1) * 128 - kk;
        const double* B = Lm + col0 * ld + (j + 1) * 128 - kk;
        double* C = Cm + row0 * ld + col0;
        double* Cb = C2 + row0 * ld + col0;
        const double flops = units * tiles * 2.0 * 128 * 128 * kk;
        const dim3 grid((unsigned)(n / 128), (unsigned)(m / 64));
        const int nth = (int)(m / 64);
#define RUN(KERNEL, GRID, ...) best_ms([&] { for (int u = 0; u < units; ++u) hipLaunchKernelGGL(KERNEL, GRID, dim3(512), 0, 0, (u ? Cb : C), ld, A, ld, B, ld, kk, row0, col0, ##__VA_ARGS__); })
        const float t11 = RUN((k_h64<true, true>), grid);
        const float t01 = RUN((k_h64<false, true>), grid);
        const float t10 = RUN((k_h64<true, false>), grid);
        const float t00 = RUN((k_h64<false, false>), grid);
        const float p2 = RUN((k_h64_pipe<2>), dim3(grid.x, (grid.y + 1) / 2), nth);
        const float p4 = RUN((k_h64_pipe<4>), dim3(grid.x, (grid.y + 3) / 4), nth);
        const float p8 = RUN((k_h64_pipe<8>), dim3(grid.x, (grid.y + 7) / 8), nth);
        // the pipelined walk computes what the product kernel computes: same k order per tile -> the same bits
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, Cm, (size_t)ld * ld, 2u);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, C2, (size_t)ld * ld, 2u);
        hipLaunchKernelGGL((k_h64<true, true>), grid, dim3(512), 0, 0, C, ld, A, ld, B, ld, kk, row0, col0);
        hipLaunchKernelGGL((k_h64_pipe<4>), dim3(grid.x, (grid.y + 3) / 4), dim3(512), 0, 0, Cb, ld, A, ld, B, ld, kk, row0, col0, nth);
        hipMemset(dmax, 0, 8);
        hipLaunchKernelGGL(k_diff, dim3(1024), dim3(256), 0, 0, Cm, C2, (size_t)ld * ld, dmax);
        double hd = 0; hipMemcpy(&hd, dmax, 8, hipMemcpyDeviceToHost);
        printf("units %d, %2d blocks below, K=%3d (%5.0f tiles): product %6.1f us %5.1f TF/s | no C load %6.1f | no C store %6.1f | neither %6.1f (%5.1f TF/s) | "
               "walk of 2 %6.1f us %5.1f TF/s, of 4 %6.1f us %5.1f, of 8 %6.1f us %5.1f | max |diff| %.1e\n",
               units, blocks, kk, tiles, t11 * 1e3, flops / t11 / 1e9, t01 * 1e3, t10 * 1e3, t00 * 1e3, flops / t00 / 1e9, p2 * 1e3, flops / p2 / 1e9, p4 * 1e3,
               flops / p4 / 1e9, p8 * 1e3, flops / p8 / 1e9, hd);
      }
    }
  }
  return 0;
}
