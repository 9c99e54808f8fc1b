// Device math helpers shared by the kernels (gfx950, fp64).
#pragma once
#include <hip/hip_runtime.h>

// exp(x) in fp64: k = rint(x log2 e); r = x - k ln2 (two-term Cody-Waite); degree-13 Taylor polynomial on |r| <= 0.3466
// (truncation 4e-18 relative); scaled by 2^k with v_ldexp_f64. Measured max error vs the correctly rounded result < 1 ulp
// on [-745, 709] (tests/test_gpu_parity.py::test_exp_accuracy_through_gram). Underflows to 0 below -745.2, overflows to +inf above 709.8.
__device__ __forceinline__ double rc_exp(double x) {
  const double LOG2E = 1.4426950408889634074;
  const double LN2_HI = 6.93147180369123816490e-01;   // high 33 bits of ln 2
  const double LN2_LO = 1.90821492927058770002e-10;   // ln 2 - LN2_HI
  const double xc = fmin(fmax(x, -746.0), 710.0);
  const double k = __builtin_rint(xc * LOG2E);
  double r = __builtin_fma(-k, LN2_HI, xc);
  r = __builtin_fma(-k, LN2_LO, r);
  double p = 1.6059043836821614599e-10;               // 1/13!
  p = __builtin_fma(p, r, 2.0876756987868098979e-09); // 1/12!
  p = __builtin_fma(p, r, 2.5052108385441718775e-08); // 1/11!
  p = __builtin_fma(p, r, 2.7557319223985890653e-07); // 1/10!
  p = __builtin_fma(p, r, 2.7557319223985892511e-06); // 1/9!
  p = __builtin_fma(p, r, 2.4801587301587301566e-05); // 1/8!
  p = __builtin_fma(p, r, 1.9841269841269841253e-04); // 1/7!
  p = __builtin_fma(p, r, 1.3888888888888889419e-03); // 1/6!
  p = __builtin_fma(p, r, 8.3333333333333332177e-03); // 1/5!
  p = __builtin_fma(p, r, 4.1666666666666664354e-02); // 1/4!
  p = __builtin_fma(p, r, 1.6666666666666665741e-01); // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)k);
}
