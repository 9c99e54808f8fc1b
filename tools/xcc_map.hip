// Which XCD does block b of a 1-D grid land on? (s_getreg_b32 HW_REG_XCC_ID.) Grid shaped like the GEMM kernels: 512 threads, 37 KB LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void __launch_bounds__(512) k_where(int* xcc, long long* t0, int spin) {
  __shared__ double pad[4608];
  pad[threadIdx.x] = threadIdx.x;
  if (threadIdx.x == 0) {
    xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;
    t0[blockIdx.x] = wall_clock64();
  }
  long long s = wall_clock64();
  while (wall_clock64() - s < spin) {}
  if (pad[(threadIdx.x + 1) & 511] < 0) xcc[0] = -1;
}
int main() {
  const int nb = 2048;
  int* d; long long* t;
  hipMalloc(&d, nb * sizeof(int)); hipMalloc(&t, nb * sizeof(long long));
  hipLaunchKernelGGL(k_where, dim3(nb), dim3(512), 0, 0, d, t, 2000);
  hipDeviceSynchronize();
  std::vector<int> h(nb); std::vector<long long> ht(nb);
  hipMemcpy(h.data(), d, nb * sizeof(int), hipMemcpyDeviceToHost);
  hipMemcpy(ht.data(), t, nb * sizeof(long long), hipMemcpyDeviceToHost);
  int agree = 0, first_gen = 0;
  for (int b = 0; b < nb; ++b) if (h[b] == h[b % 8]) ++agree;
  printf("blocks whose XCC equals that of block b %% 8: %d of %d\n", agree, nb);
  printf("first 24 blocks: "); for (int b = 0; b < 24; ++b) printf("%d ", h[b]); printf("\n");
  printf("blocks 512..535: "); for (int b = 512; b < 536; ++b) printf("%d ", h[b]); printf("\n");
  long long tmin = ht[0]; for (int b = 0; b < nb; ++b) if (ht[b] < tmin) tmin = ht[b];
  printf("start times (us) of blocks 0,8,16,...,504 (one XCD, first generation): ");
  for (int b = 0; b < 512; b += 8) printf("%.1f ", (ht[b] - tmin) * 0.01); printf("\n");
  return 0;
}
