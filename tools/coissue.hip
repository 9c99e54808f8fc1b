// Micro-benchmark: do fp64 MFMA and fp64 VALU FMA share an execution unit on gfx950? Both have the same spec peak (78.6 TFLOP/s).
// Half of the waves of every SIMD run a v_mfma_f64_16x16x4 loop, the other half a v_fma_f64 loop of about the same length; timed
// alone and together. together ~ max(alone) -> separate units (a GEMM could use both); together ~ sum -> one unit.
//   hipcc -O3 --offload-arch=gfx950 tools/coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));

// 1024 threads = 16 waves = 4 per SIMD; wave w sits on SIMD w % 4, role = (w / 4) & 1: every SIMD gets two MFMA and two VALU waves
// per workgroup. mode bit 0: MFMA waves work, bit 1: VALU waves work.
__global__ void __launch_bounds__(1024) k_mix(double* out, int iters_mfma, int iters_fma, int mode) {
  const int w = threadIdx.x >> 6;
  const int role = (w >> 2) & 1;
  double s = 0;
  if (role == 0) {
    if (!(mode & 1)) return;
    v4d acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters_mfma; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    if (!(mode & 2)) return;
    double x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3 + i;
    const double a = 1.0000001, b = 1e-9;
    for (int it = 0; it < iters_fma; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
    for (int i = 0; i < 16; ++i) s += x[i];
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float time_ms(int blocks, double* out, int im, int iv, int mode) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(1024), 0, 0, out, im, iv, mode);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(1024), 0, 0, out, im, iv, mode);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double* out; (void)hipMalloc(&out, (size_t)1024 * 1024 * sizeof(double));
  for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
    const int blocks = 256 * wg_per_cu, im = 20000, iv = 4 * im;
    const double f_m = (double)blocks * 8 * im * 4 * 2048.0, f_v = (double)blocks * 8 * 64 * (double)iv * 16 * 2.0;
    const float t1 = time_ms(blocks, out, im, iv, 1), t2 = time_ms(blocks, out, im, iv, 2), t3 = time_ms(blocks, out, im, iv, 3);
    printf("%d workgroup(s) of 16 waves per CU (%d MFMA + %d VALU waves per SIMD):\n", wg_per_cu, 2 * wg_per_cu, 2 * wg_per_cu);
    printf("  MFMA alone  %8.3f ms = %6.2f TFLOP/s\n", t1, f_m / t1 / 1e9);
    printf("  VALU alone  %8.3f ms = %6.2f TFLOP/s\n", t2, f_v / t2 / 1e9);
    printf("  together    %8.3f ms = %6.2f TFLOP/s (MFMA %.2f + VALU %.2f); sum of alone %.3f ms, max %.3f ms\n", t3, (f_m + f_v) / t3 / 1e9,
           f_m / t3 / 1e9, f_v / t3 / 1e9, t1 + t2, t1 > t2 ? t1 : t2);
  }
  return 0;
}
