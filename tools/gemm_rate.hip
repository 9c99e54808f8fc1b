// Main-loop rate of the fp64 MFMA GEMM core (rom-comma_amd/csrc/gemm.hip) by operand orientation, sign handling and K, on full
// rounds of identical tiles (n = 8192: 4096 tiles = 8 rounds of 512 resident workgroups): what separates the update kernels'
// 54-58 TFLOP/s from k_grad's 65?   hipcc -O3 --offload-arch=gfx950 tools/gemm_rate.hip -o tools/gemm_rate
#include "../rom-comma_amd/csrc/gemm.hip"
#include <stdio.h>
int rc_ensure_partial(rcgp_handle_s*, size_t) { return 0; }

template <bool AKC, bool BKC, bool NEG, int EPI>
__global__ void RC_BOUNDS(4) k_rate(const double* __restrict__ A, int64_t lda, const double* __restrict__ B, int64_t ldb, int kk,
                                    double* __restrict__ C, int64_t ldc) {
  __shared__ double lds[GEMM_LDS];
  const int tj = blockIdx.x, ti = blockIdx.y;
  v4d acc[4][Geo<4>::NI];
  double* Ct = C + (int64_t)ti * 128 * ldc + (int64_t)tj * 128;
  if (EPI == 2) acc_load_staged<4>(acc, Ct, ldc, lds); else acc_zero(acc);
  gemm_mainloop<AKC, BKC, 4, NEG>(A, lda, (int64_t)ti * 128, B, ldb, (int64_t)tj * 128, 0, kk, acc, lds);
  if (EPI == 2) acc_store_staged<4>(acc, Ct, ldc, lds); else acc_store<4>(acc, Ct, ldc);
}

__global__ void k_fill(double* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed * 40503u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = ((double)(x & 0xffffff) / 16777216.0 - 0.5) * 1e-3;
  }
}

template <bool AKC, bool BKC, bool NEG, int EPI>
static void run(const char* name, const double* A, const double* B, double* C, int n, int kk, int64_t ld) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const dim3 grid(n / 128, n / 128), block(512);
  hipLaunchKernelGGL((k_rate<AKC, BKC, NEG, EPI>), grid, block, 0, 0, A, ld, B, ld, kk, C, ld);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_rate<AKC, BKC, NEG, EPI>), grid, block, 0, 0, A, ld, B, ld, kk, C, ld);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%-34s K=%5d  %8.3f ms  %6.1f TFLOP/s\n", name, kk, best, 2.0 * n * (double)n * kk / (best * 1e-3) / 1e12);
}

int main() {
  const int n = 8192; const int64_t ld = 16384;
  double *A, *B, *C;
  hipMalloc(&A, ld * ld * sizeof(double)); hipMalloc(&B, ld * ld * sizeof(double)); hipMalloc(&C, ld * ld * sizeof(double));
  hipMemset(A, 0, ld * ld * sizeof(double)); hipMemset(B, 0, ld * ld * sizeof(double)); hipMemset(C, 0, ld * ld * sizeof(double));
  for (int pass = 0; pass < 2; ++pass) {
  if (pass == 1) {            // random operands: the chip lowers its clock under real fp64 MFMA load (zero-filled operands flatter every kernel)
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, A, (size_t)ld * ld, 1u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, B, (size_t)ld * ld, 2u);
    hipDeviceSynchronize();
    printf("--- random operands\n");
  } else printf("--- zero-filled operands\n");
  for (int kk : {1024, 8192}) {
    run<true, true, false, 0>("NT  (k-contiguous both)", A, B, C, n, kk, ld);
    run<true, true, true, 0>("NT  negated A", A, B, C, n, kk, ld);
    run<true, true, true, 2>("NT  negated A, C in/out staged", A, B, C, n, kk, ld);
    run<false, false, false, 0>("TN  (row-contiguous both: k_grad)", A, B, C, n, kk, ld);
    run<true, false, false, 0>("NN  (k_trtri)", A, B, C, n, kk, ld);
    run<true, true, false, 0>("NT  same operand (A == B)", A, A, C, n, kk, ld);
  }
  }
  // the library's own bulk update (lower tiles of an n x n trailing matrix, K = NB) and a window piece, alone on the chip, random operands
  rcgp_handle_s hh;
  hh.launch = nullptr;
  rcgp_handle_s* h = &hh;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nn : {12800, 8704, 4608}) {
    for (int kk : {512, 1024, 2048}) {
      rc_launch_syrk_lower(h, C, ld, A, ld, nn, kk);
      hipDeviceSynchronize();
      float best = 1e30f;
      for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0, 0);
        rc_launch_syrk_lower(h, C, ld, A, ld, nn, kk);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double tiles = (nn / 128) * (nn / 128 + 1) / 2.0;
      printf("k_syrk_lower n=%5d K=%4d: %7.3f ms  %5.1f TFLOP/s  (%5.0f tiles = %.2f rounds of 512)\n", nn, kk, best,
             (double)nn * (nn + 128.0) * kk / (best * 1e-3) / 1e12, tiles, tiles / 512.0);
    }
  }
  return 0;
}
