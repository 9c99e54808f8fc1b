// Micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate and fp64 VALU FMA rate on every CU (peak calibration for roofline).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(double* out, int iters) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
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
  hipEventCreate(&e0); hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double* out; hipMalloc(&out, 4096 * 256 * sizeof(double));
  const int iters = 20000;
  for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
    const int blocks = 256 * wg_per_cu;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(256), 0, 0, out, iters); });
    double flops = (double)blocks * 4 /*waves*/ * iters * 4 /*acc*/ * 2048.0;
    printf("mfma_f64 16x16x4, 4 acc, %d waves/SIMD: %.2f TFLOP/s (%.1f cycles/mfma/SIMD @2.4GHz)\n", wg_per_cu, flops / ms / 1e9,
           ms * 1e-3 * 2.4e9 / ((double)wg_per_cu * iters * 4));
    ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(256), 0, 0, out, iters); });
    flops = (double)blocks * 4 * iters * 1 * 2048.0;
    printf("mfma_f64 16x16x4, 1 acc (dependent), %d waves/SIMD: %.2f TFLOP/s\n", wg_per_cu, flops / ms / 1e9);
  }
  for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {
    const int blocks = 256 * wg_per_cu;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters); });
    double flops = (double)blocks * 256 * iters * 16 * 2.0;
    printf("v_fma_f64, %d waves/SIMD: %.2f TFLOP/s\n", wg_per_cu, flops / ms / 1e9);
  }
  return 0;
}
