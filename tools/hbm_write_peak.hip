// Write-bandwidth reference for the Gram kernel: plain fill kernels over 1.07 GB (the lower-triangle footprint of the C2 Gram
// matrix) with the store shapes the Gram kernel could use.   hipcc --offload-arch=gfx950 -O3 tools/hbm_write_peak.hip -o tools/hbm_write_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void fill8(double* p, size_t n, double v) {            // 8 B per lane, grid-stride
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill16(double2* p, size_t n, double v) {          // 16 B per lane
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_double2(v, v);
}
__global__ void fill8nt(double* p, size_t n, double v) {          // 8 B per lane, nontemporal
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, p + i);
}
__global__ void fill16nt(double* p, size_t n, double v) {         // 2 x 8 B per lane, nontemporal, 16 B contiguous per lane
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n / 2; i += (size_t)gridDim.x * blockDim.x) {
    __builtin_nontemporal_store(v, p + 2 * i);
    __builtin_nontemporal_store(v, p + 2 * i + 1);
  }
}

template <typename F> static void run(const char* name, F launch, size_t bytes) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 20; ++r) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-28s %7.3f ms  %7.0f GB/s\n", name, ms / 20, bytes / (ms / 20 * 1e-3) / 1e9);
}

int main() {
  const size_t n = (size_t)16384 * 16385 / 2 + 16384 * 10;        // doubles: lower triangle + X
  const size_t bytes = n * 8;
  double* p; hipMalloc(&p, bytes + 64);
  for (int grid : {2048, 8192, 32768}) {
    printf("grid %d x 512 threads\n", grid);
    run("fill 8 B/lane", [&] { fill8<<<grid, 512>>>(p, n, 1.0); }, bytes);
    run("fill 16 B/lane", [&] { fill16<<<grid, 512>>>((double2*)p, n / 2, 1.0); }, bytes);
    run("fill 8 B/lane nontemporal", [&] { fill8nt<<<grid, 512>>>(p, n, 1.0); }, bytes);
    run("fill 2x8 B/lane nontemporal", [&] { fill16nt<<<grid, 512>>>(p, n, 1.0); }, bytes);
  }
  run("hipMemsetAsync", [&] { hipMemsetAsync(p, 0, bytes, 0); }, bytes);
  hipFree(p);
  return 0;
}
