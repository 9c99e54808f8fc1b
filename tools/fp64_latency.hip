// Dependent-issue latencies of the instructions on the Cholesky pivot chain, one wave alone on a CU (gfx950), and the accuracy of
// v_rcp_f64 / v_rsq_f64 seeds.   hipcc --offload-arch=gfx950 -O3 tools/fp64_latency.hip -o tools/fp64_latency && tools/fp64_latency
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
#define REP 256
// A timestamp that cannot be taken before the value x has been produced (v_readfirstlane waits for the VALU result) nor moved by the compiler.
__device__ __forceinline__ long long tick(double& x) {
  long long t;
  int lo = __double2loint(x), dummy;
  asm volatile("v_readfirstlane_b32 %1, %2\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(dummy), "+v"(lo)::"memory");
  x = __hiloint2double(__double2hiint(x), lo);
  return t;
}
__global__ void k_lat(double* out, long long* cyc, double x0) {
  double x = x0 + threadIdx.x * 1e-9, y = 1.0000001;
  long long t0, t1;
  const long long w0 = wall_clock64();
  const long long m0 = tick(x);
  // dependent v_fma_f64
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) x = __builtin_fma(x, y, 1e-12);
  t1 = tick(x);
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  // dependent v_rcp_f64
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) x = __builtin_amdgcn_rcp(x);
  t1 = tick(x);
  if (threadIdx.x == 0) cyc[1] = t1 - t0;
  // dependent readlane -> valu (v_readlane x2 then v_fma with the scalar)
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), 3), hi = __builtin_amdgcn_readlane(__double2hiint(x), 3);
    x = __builtin_fma(__hiloint2double(hi, lo), y, x);
  }
  t1 = tick(x);
  if (threadIdx.x == 0) cyc[2] = t1 - t0;
  // dependent mfma f64 16x16x4 (through the accumulator)
  v4d acc = {x, x, x, x};
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, acc, 0, 0, 0);
  { double a = acc[0]; t1 = tick(a); acc[0] = a; }
  if (threadIdx.x == 0) cyc[3] = t1 - t0;
  // dependent mfma through the B operand (the substitution recurrence: result -> B of the next)
  double b = x;
  t0 = tick(b);
#pragma unroll
  for (int i = 0; i < REP; ++i) { v4d z = {0, 0, 0, 0}; z = __builtin_amdgcn_mfma_f64_16x16x4f64(y, b, z, 0, 0, 0); b = z[0]; }
  t1 = tick(b);
  if (threadIdx.x == 0) cyc[4] = t1 - t0;
  // independent mfma (4 accumulators round-robin): issue interval of one wave
  v4d a0 = acc, a1 = acc, a2 = acc, a3 = acc;
  t0 = tick(b);
#pragma unroll
  for (int i = 0; i < REP / 4; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
  }
  { double a = a0[0] + a1[0] + a2[0] + a3[0]; t1 = tick(a); b += a; }
  if (threadIdx.x == 0) cyc[5] = t1 - t0;
  // LDS write -> read round trip (dependent)
  __shared__ double sh[64];
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) { sh[threadIdx.x] = x; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); x = sh[(threadIdx.x + 1) & 63] + 1e-9; }
  t1 = tick(x);
  if (threadIdx.x == 0) cyc[6] = t1 - t0;
  // independent v_fma_f64 (8 chains): issue interval
  double c0 = x, c1 = x + 1, c2 = x + 2, c3 = x + 3, c4 = x + 4, c5 = x + 5, c6 = x + 6, c7 = x + 7;
  t0 = tick(c7);
#pragma unroll
  for (int i = 0; i < REP / 8; ++i) {
    c0 = __builtin_fma(c0, y, 1e-12); c1 = __builtin_fma(c1, y, 1e-12); c2 = __builtin_fma(c2, y, 1e-12); c3 = __builtin_fma(c3, y, 1e-12);
    c4 = __builtin_fma(c4, y, 1e-12); c5 = __builtin_fma(c5, y, 1e-12); c6 = __builtin_fma(c6, y, 1e-12); c7 = __builtin_fma(c7, y, 1e-12);
  }
  { double a = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7; t1 = tick(a); c0 = a; }
  if (threadIdx.x == 0) cyc[7] = t1 - t0;
  // the same 8 independent chains with only lanes 0..15 active: does the VALU skip the three fully-masked quarter-waves?
  {
    long long ta = tick(c0);
    if (threadIdx.x < 16) {
#pragma unroll
      for (int i = 0; i < REP / 8; ++i) {
        c0 = __builtin_fma(c0, y, 1e-12); c1 = __builtin_fma(c1, y, 1e-12); c2 = __builtin_fma(c2, y, 1e-12); c3 = __builtin_fma(c3, y, 1e-12);
        c4 = __builtin_fma(c4, y, 1e-12); c5 = __builtin_fma(c5, y, 1e-12); c6 = __builtin_fma(c6, y, 1e-12); c7 = __builtin_fma(c7, y, 1e-12);
      }
    }
    double a = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    long long tb = tick(a);
    c0 = a;
    if (threadIdx.x == 0) cyc[10] = tb - ta;
  }
  { const long long m1 = tick(x); const long long w1 = wall_clock64(); if (threadIdx.x == 0) { cyc[8] = m1 - m0; cyc[9] = w1 - w0; } }
  out[threadIdx.x] = x + acc[0] + a0[1] + a1[0] + a2[0] + a3[0] + b + c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
__global__ void k_acc(const double* in, double* r, double* q, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { r[i] = __builtin_amdgcn_rcp(in[i]); q[i] = __builtin_amdgcn_rsq(in[i]); }
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 12 * 8);
  for (int rep = 0; rep < 2; ++rep) k_lat<<<1, 64>>>(out, cyc, 1.25);
  long long h[12]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("s_memtime ticks %lld over %lld wall-clock ticks (100 MHz): s_memtime runs at %.1f MHz\n", h[8], h[9], 100.0 * h[8] / h[9]);
  const char* names[8] = {"dependent v_fma_f64", "dependent v_rcp_f64", "readlane x2 -> v_fma_f64", "dependent mfma_f64_16x16x4 (accumulator)",
                          "dependent mfma (result -> B operand)", "independent mfma, 4 accumulators", "LDS write -> barrier -> read", "independent v_fma_f64, 8 chains"};
  for (int i = 0; i < 8; ++i) printf("%-44s %7.1f cycles each (s_memtime / readcyclecounter units)\n", names[i], (double)h[i] / REP);
  printf("%-44s %7.1f cycles each\n", "independent v_fma_f64, 8 chains, 16 lanes active", (double)h[10] / REP);
  const int n = 1 << 16;
  std::vector<double> x(n), r(n), q(n);
  for (int i = 0; i < n; ++i) x[i] = std::ldexp(1.0 + (double)rand() / RAND_MAX, (rand() % 40) - 20);
  double *dx, *dr, *dq; hipMalloc(&dx, n * 8); hipMalloc(&dr, n * 8); hipMalloc(&dq, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k_acc<<<n / 256, 256>>>(dx, dr, dq, n);
  hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost); hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost);
  double er = 0, eq = 0;
  for (int i = 0; i < n; ++i) { er = std::fmax(er, std::fabs(r[i] * x[i] - 1.0)); eq = std::fmax(eq, std::fabs(q[i] * q[i] * x[i] - 1.0) / 2); }
  printf("v_rcp_f64 max relative error %.3e (2^%.1f); v_rsq_f64 %.3e (2^%.1f)\n", er, std::log2(er), eq, std::log2(eq));
  return 0;
}
