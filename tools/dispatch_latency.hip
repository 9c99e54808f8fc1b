// How long does a ONE-workgroup kernel on a high-priority stream wait for a place on the chip while long-running GEMM-shaped workgroups
// (512 threads, <= 128 VGPRs, 64 KB of LDS: two per CU, like the K = NB update kernels of the Cholesky) keep every slot busy?
//   hog   : `gens` generations of 512 workgroups, each spinning `tile_us` microseconds, on one or two normal-priority streams;
//   probe : one workgroup of `threads` threads with `lds` bytes of dynamic LDS, launched (and waited for) again and again on a
//           high-priority stream while the hog runs; its latency = host time from launch to completion, minus the same on an idle chip.
// Build: hipcc -O3 --offload-arch=gfx950 tools/dispatch_latency.hip -o tools/dispatch_latency      Run: tools/dispatch_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <algorithm>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

template <int PAD>
__global__ void __launch_bounds__(512, 4) k_hog(int ticks, double* sink, int jitter) {      // (512, 4): at most 128 VGPRs, as the update kernels
  if (jitter == 1) ticks += (ticks * (int)(blockIdx.x % 8)) / 16;  // tiles of unequal length: up to +44 %
  if (jitter == 2 && blockIdx.x < 512) {                          // equal tiles, first generation staggered over one tile time
    const long long until = wall_clock64() + (long long)((blockIdx.x * 37u) & 63u) * ticks / 64;
    while (wall_clock64() < until) __builtin_amdgcn_s_sleep(32);
  }
  __shared__ double pad[PAD];                                // 8192 doubles = 64 KB: two workgroups per CU; 10496 = 82 KB: ONE per CU
  asm volatile("v_mov_b32 v127, 0" ::: "v127");             // 128 VGPRs: two workgroups fill the register file as well
  pad[threadIdx.x] = (double)threadIdx.x;
  const long long t0 = wall_clock64();                       // 100 MHz
  while (wall_clock64() - t0 < ticks) {}
  __syncthreads();
  if (sink && pad[(threadIdx.x * 7) & 8191] < -1.0) sink[0] = pad[threadIdx.x];
}

template <int REGS>
__global__ void __launch_bounds__(512) k_probe(long long* stamp, int ticks) {
  extern __shared__ double dyn[];
  if (REGS == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if (REGS == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
  if (REGS == 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
  if (threadIdx.x == 0) stamp[0] = wall_clock64();
  dyn[threadIdx.x] = 1.0;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  __syncthreads();
  if (threadIdx.x == 0) stamp[1] = wall_clock64();
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  const int tile_us = argc > 1 ? atoi(argv[1]) : 260, gens = argc > 2 ? atoi(argv[2]) : 12, jitter = argc > 3 ? atoi(argv[3]) : 0;
  const int one_per_cu = argc > 4 ? atoi(argv[4]) : 0;       // 1: the hog's workgroups take 82 KB of LDS, so a CU holds ONE and keeps a slot free
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t hogA, probeS, s3, s4, hogB;                      // creation ranks 1..5 after the null stream: hogA and hogB share a pipe (1 and 5)
  CK(hipStreamCreateWithFlags(&hogA, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&probeS, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&s3, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&s4, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithFlags(&hogB, hipStreamNonBlocking));
  long long* stamp;
  CK(hipMalloc(&stamp, 64));
  CK(hipFuncSetAttribute((const void*)k_probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_probe<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_probe<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_probe<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  struct Shape { int threads, lds, regs; const char* what; };
  const Shape shapes[] = {{64, 1024, 0, "1 wave, 1 KB, few regs"},          {512, 10240, 0, "8 waves, 10 KB, few regs"},
                          {512, 10240, 64, "8 waves, 10 KB, 64 regs"},      {512, 10240, 128, "8 waves, 10 KB, 128 regs (k_diag)"},
                          {256, 10240, 128, "4 waves, 10 KB, 128 regs"},    {256, 10240, 256, "4 waves, 10 KB, 256 regs"},
                          {512, 50688, 128, "8 waves, 49.5 KB, 128 regs"},  {256, 50688, 128, "4 waves, 49.5 KB, 128 regs"},
                          {512, 141312, 128, "8 waves, 138 KB (k_prep1)"},  {512, 153600, 128, "8 waves, 150 KB (k_diag2)"},
                          {256, 30720, 256, "4 waves, 30 KB, 256 regs (slim diag)"}, {512, 78848, 128, "8 waves, 77 KB, 128 regs (k_trsm_subst)"},
                          {512, 75776, 128, "8 waves, 74 KB, 128 regs (update)"}};
  auto launch_probe = [&](const Shape& sh) {
    if (sh.regs == 64) hipLaunchKernelGGL(k_probe<64>, dim3(1), dim3(sh.threads), sh.lds, probeS, stamp, 500);
    else if (sh.regs == 128) hipLaunchKernelGGL(k_probe<128>, dim3(1), dim3(sh.threads), sh.lds, probeS, stamp, 500);
    else if (sh.regs == 256) hipLaunchKernelGGL(k_probe<256>, dim3(1), dim3(sh.threads), sh.lds, probeS, stamp, 500);
    else hipLaunchKernelGGL(k_probe<0>, dim3(1), dim3(sh.threads), sh.lds, probeS, stamp, 500);
  };
  for (int nhog = 1; nhog <= 2; ++nhog)
    for (const Shape& sh : shapes) {
      // idle-chip latency of the probe
      std::vector<double> idle, busy;
      for (int i = 0; i < 20; ++i) {
        const double t0 = now_us();
        launch_probe(sh);
        CK(hipStreamSynchronize(probeS));
        idle.push_back(now_us() - t0);
      }
      std::sort(idle.begin(), idle.end());
      CK(hipDeviceSynchronize());
      const double h0 = now_us();
      const int hog_gens = one_per_cu ? gens / 2 : gens;       // the same wall time: half as many workgroups run at once
      if (one_per_cu) {
        hipLaunchKernelGGL(k_hog<10496>, dim3(512 * hog_gens), dim3(512), 0, hogA, tile_us * 100, (double*)nullptr, jitter);
        if (nhog == 2) hipLaunchKernelGGL(k_hog<10496>, dim3(512 * hog_gens), dim3(512), 0, hogB, tile_us * 100, (double*)nullptr, jitter);
      } else {
        hipLaunchKernelGGL(k_hog<8192>, dim3(512 * gens), dim3(512), 0, hogA, tile_us * 100, (double*)nullptr, jitter);
        if (nhog == 2) hipLaunchKernelGGL(k_hog<8192>, dim3(512 * gens), dim3(512), 0, hogB, tile_us * 100, (double*)nullptr, jitter);
      }
      const double total = (double)tile_us * gens * nhog;
      while (now_us() - h0 < 0.8 * total) {
        const double t0 = now_us();
        launch_probe(sh);
        CK(hipStreamSynchronize(probeS));
        busy.push_back(now_us() - t0);
      }
      CK(hipDeviceSynchronize());
      const double hog_time = now_us() - h0;
      std::vector<double> sorted = busy;
      std::sort(sorted.begin(), sorted.end());
      printf("%d hog stream(s), probe %-34s idle %6.1f us | beside the hog: n %3zu  first %8.1f  median %8.1f  max %8.1f us  (hog %.0f us, ideal %.0f)\n", nhog,
             sh.what, idle[idle.size() / 2], busy.size(), busy.empty() ? 0.0 : busy[0], sorted.empty() ? 0.0 : sorted[sorted.size() / 2],
             sorted.empty() ? 0.0 : sorted.back(), hog_time, total);
    }
  return 0;
}
