// Physical placement of the workgroups of a 512-block launch shaped like the GEMM kernels (512 threads, 37 KB LDS, 128 VGPRs):
// XCC_ID and the HW_ID fields (cu_id, sh_id, se_id). Which (xcc, se, sh, cu) tuples exist, and how many blocks land on each CU?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
#include <tuple>
__global__ void __launch_bounds__(512, 4) k_where(unsigned* out, int spin) {
  __shared__ double pad[4608];
  pad[threadIdx.x] = threadIdx.x;
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);          // XCC_ID[3:0]
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_ID, 32 bits
  }
  long long s = wall_clock64();
  while (wall_clock64() - s < spin) {}
  if (pad[(threadIdx.x + 1) & 511] < 0) out[0] = 0;
}
int main() {
  const int nb = 512;
  unsigned* d;
  hipMalloc(&d, 2 * nb * sizeof(unsigned));
  hipLaunchKernelGGL(k_where, dim3(nb), dim3(512), 0, 0, d, 5000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(2 * nb);
  hipMemcpy(h.data(), d, 2 * nb * sizeof(unsigned), hipMemcpyDeviceToHost);
  std::map<std::tuple<int, int, int, int>, int> count;
  std::map<int, int> cu_ids, se_ids, sh_ids;
  for (int b = 0; b < nb; ++b) {
    const unsigned xcc = h[2 * b] & 15, hw = h[2 * b + 1];
    const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    ++count[std::make_tuple((int)xcc, se, sh, cu)];
    ++cu_ids[cu]; ++se_ids[se]; ++sh_ids[sh];
  }
  printf("distinct (xcc, se, sh, cu): %zu\n", count.size());
  printf("cu_id histogram: "); for (auto& kv : cu_ids) printf("%d:%d ", kv.first, kv.second); printf("\n");
  printf("se_id histogram: "); for (auto& kv : se_ids) printf("%d:%d ", kv.first, kv.second); printf("\n");
  printf("sh_id histogram: "); for (auto& kv : sh_ids) printf("%d:%d ", kv.first, kv.second); printf("\n");
  std::map<int, int> per;
  for (auto& kv : count) ++per[kv.second];
  printf("blocks per CU histogram: "); for (auto& kv : per) printf("%d blocks: %d CUs  ", kv.first, kv.second); printf("\n");
  printf("xcc 0: "); for (auto& kv : count) if (std::get<0>(kv.first) == 0) printf("(se%d sh%d cu%d)x%d ", std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first), kv.second); printf("\n");
  return 0;
}
