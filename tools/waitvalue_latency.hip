// Cross-stream hand-over latency: kernel A on stream 1, kernel B on stream 2 that must not start before A has finished. Three ways:
//   (1) hipEventRecord / hipStreamWaitEvent           (marker packets)
//   (2) hipExtLaunchKernelGGL stop event + StreamWaitEvent   (what the panel chain uses)
//   (3) A's last instruction stores a flag, stream 2 waits with hipStreamWaitValue64
// Both kernels stamp wall_clock64 (100 MHz constant clock): latency = B's first stamp - A's last stamp. A chain of `steps` ping-pongs.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

__global__ void k_work(long long* stamps, int idx, uint64_t* flag, uint64_t value, int spin) {
  if (threadIdx.x == 0) stamps[2 * idx] = wall_clock64();
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}                      // ~spin * 10 ns of "work"
  __syncthreads();
  if (threadIdx.x == 0) {
    stamps[2 * idx + 1] = wall_clock64();
    if (flag) {
      __threadfence_system();
      __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t s1, s2;
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi));
  const int steps = 40;
  long long* stamps;
  CK(hipMalloc(&stamps, sizeof(long long) * 4 * steps));
  uint64_t *flag = nullptr, *flag2 = nullptr;                 // (signal memory comes in 8-byte allocations)
  if (can) { CK(hipExtMallocWithFlags((void**)&flag, sizeof(uint64_t), hipMallocSignalMemory)); CK(hipExtMallocWithFlags((void**)&flag2, sizeof(uint64_t), hipMallocSignalMemory)); }
  std::vector<hipEvent_t> ev(2 * steps);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (int mode = 1; mode <= (can ? 3 : 2); ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (flag) { CK(hipMemset(flag, 0, sizeof(uint64_t))); CK(hipMemset(flag2, 0, sizeof(uint64_t))); }
      CK(hipMemset(stamps, 0, sizeof(long long) * 4 * steps));
      CK(hipDeviceSynchronize());
      for (int i = 0; i < steps; ++i) {
        // A(i) on s1 waits for B(i-1); B(i) on s2 waits for A(i)
        if (mode == 1) {
          if (i > 0) CK(hipStreamWaitEvent(s1, ev[2 * (i - 1) + 1], 0));
          hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, s1, stamps, 2 * i, (uint64_t*)nullptr, (uint64_t)0, 500);
          CK(hipEventRecord(ev[2 * i], s1));
          CK(hipStreamWaitEvent(s2, ev[2 * i], 0));
          hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, s2, stamps, 2 * i + 1, (uint64_t*)nullptr, (uint64_t)0, 500);
          CK(hipEventRecord(ev[2 * i + 1], s2));
        } else if (mode == 2) {
          if (i > 0) CK(hipStreamWaitEvent(s1, ev[2 * (i - 1) + 1], 0));
          hipExtLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, s1, nullptr, ev[2 * i], 0, stamps, 2 * i, (uint64_t*)nullptr, (uint64_t)0, 500);
          CK(hipStreamWaitEvent(s2, ev[2 * i], 0));
          hipExtLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, s2, nullptr, ev[2 * i + 1], 0, stamps, 2 * i + 1, (uint64_t*)nullptr, (uint64_t)0, 500);
        } else {
          if (i > 0) CK(hipStreamWaitValue64(s1, flag2, (uint64_t)i, hipStreamWaitValueGte, 0xffffffffffffffffull));
          hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, s1, stamps, 2 * i, flag, (uint64_t)(i + 1), 500);
          CK(hipStreamWaitValue64(s2, flag, (uint64_t)(i + 1), hipStreamWaitValueGte, 0xffffffffffffffffull));
          hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, s2, stamps, 2 * i + 1, flag2, (uint64_t)(i + 1), 500);
        }
      }
      CK(hipDeviceSynchronize());
      std::vector<long long> h(4 * steps);
      CK(hipMemcpy(h.data(), stamps, sizeof(long long) * 4 * steps, hipMemcpyDeviceToHost));
      double sum = 0; int n = 0; double mn = 1e9, mx = 0;
      for (int k = 1; k < 2 * steps; ++k) {                  // kernel k starts after kernel k-1 ends
        const double us = (h[2 * k] - h[2 * (k - 1) + 1]) * 0.01;
        if (k > 4) { sum += us; ++n; mn = us < mn ? us : mn; mx = us > mx ? us : mx; }
      }
      printf("mode %d rep %d: hand-over latency mean %.2f us  min %.2f  max %.2f  (kernel body %.2f us)\n", mode, rep, sum / n, mn, mx,
             (h[1] - h[0]) * 0.01);
    }
  }
  return 0;
}
