// Micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate (and the in-kernel clock it runs at) and the fp64 VALU FMA rate on
// every CU -- peak calibration for the roofline fractions quoted in DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(double* out, int iters, unsigned long long* clk) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ void __launch_bounds__(256) k_fma(double* out, int iters) {
  double x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3 + i;
  const double a = 1.0000001, b = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float time_ms(F f) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  f();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int NACC>
static void run_mfma(double* out, unsigned long long* clk, int wg_per_cu, int threads) {
  const int iters = 20000, blocks = 256 * wg_per_cu;
  float ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, clk); });
  unsigned long long h[2];
  (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / (double)h[1] * 0.1;      // s_memrealtime ticks at 100 MHz
  const double waves_per_simd = wg_per_cu * threads / 256.0;
  const double flops = (double)blocks * (threads / 64) * iters * NACC * 2048.0;
  printf("mfma_f64_16x16x4 acc=%d waves/SIMD=%.0f: %6.2f TFLOP/s, in-kernel clock %.2f GHz, %.1f cycles/mfma/SIMD at that clock\n", NACC,
         waves_per_simd, flops / ms / 1e9, ghz, (double)h[0] / ((double)iters * NACC * waves_per_simd));
}

int main() {
  double* out; (void)hipMalloc(&out, 8192 * 512 * sizeof(double));
  unsigned long long* clk; (void)hipMalloc(&clk, 8192 * 2 * sizeof(unsigned long long));
  run_mfma<1>(out, clk, 1, 256);
  run_mfma<4>(out, clk, 1, 256);
  run_mfma<8>(out, clk, 1, 256);
  run_mfma<16>(out, clk, 1, 256);
  run_mfma<4>(out, clk, 2, 256);
  run_mfma<16>(out, clk, 2, 256);
  run_mfma<4>(out, clk, 4, 256);
  run_mfma<4>(out, clk, 8, 256);
  const int iters = 20000;
  for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
    const int blocks = 256 * wg_per_cu;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters); });
    double flops = (double)blocks * 256 * iters * 16 * 2.0;
    printf("v_fma_f64, %d waves/SIMD: %.2f TFLOP/s\n", wg_per_cu, flops / ms / 1e9);
  }
  return 0;
}
