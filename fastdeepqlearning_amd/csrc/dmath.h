// exp / log / tanh in double for the policy heads, accurate to ~1e-13 relative - far below half an fp32 ulp (6e-8), which is
// what their callers keep: every use rounds the result to float once ((float)dm_exp((double)x)), so that the log-prob of a
// tanh-Gaussian action (ill-conditioned near |a| -> 1, gaussian_mlp.py:15-39) sees correctly rounded fp32 functions, as with the
// device library's double routines - which are correct to the last double bit and ~150 instructions each: four of them per
// (row, action) made k_policy_fwd a 60 us launch at config 4.  Here: argument reduction + one short polynomial (~25 FMAs).
// Host-compilable (tests/test_dmath.py runs the same code against libm on the CPU).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define FDQL_HD __host__ __device__ __forceinline__
#else
#define FDQL_HD inline
#endif

namespace fdql {

// exp(r) - 1 for |r| <= 0.35: Taylor to r^12 (remainder < 3e-15 relative to r)
FDQL_HD double dm_expm1_small(double r) {
  double p = 1.0 / 479001600.0;
  p = __builtin_fma(p, r, 1.0 / 39916800.0);
  p = __builtin_fma(p, r, 1.0 / 3628800.0);
  p = __builtin_fma(p, r, 1.0 / 362880.0);
  p = __builtin_fma(p, r, 1.0 / 40320.0);
  p = __builtin_fma(p, r, 1.0 / 5040.0);
  p = __builtin_fma(p, r, 1.0 / 720.0);
  p = __builtin_fma(p, r, 1.0 / 120.0);
  p = __builtin_fma(p, r, 1.0 / 24.0);
  p = __builtin_fma(p, r, 1.0 / 6.0);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  return p * r;
}

// x = n ln2 + r, |r| <= ln2 / 2 (two-constant reduction: exact product for |n| < 2^20)
FDQL_HD double dm_reduce_ln2(double x, int &n) {
  const double inv_ln2 = 1.4426950408889634074, ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double t = __builtin_rint(x * inv_ln2);
  n = (int)t;
  return __builtin_fma(t, -ln2_lo, __builtin_fma(t, -ln2_hi, x));
}

// Finite x: the argument is clamped to [-745.2, 709.8] (exp underflows to 0 / overflows to inf beyond it, and the reduction's
// (int) conversion needs |n| small); exp(-inf) = 0, exp(inf) = inf through the clamp and ldexp; NaN in -> NaN out, as libm.
FDQL_HD double dm_exp(double x) {
  if (x != x) return x;
  const double xc = x < -745.2 ? -745.2 : (x > 709.8 ? 709.8 : x);
  int n;
  const double r = dm_reduce_ln2(xc, n);
  // two-step scaling: 1 + p in [0.7, 1.5], n in [-1075, 1024] - one ldexp by n < -1022 or n = 1024 would lose the result
  const int n1 = n / 2;
  return __builtin_ldexp(__builtin_ldexp(1.0 + dm_expm1_small(r), n1), n - n1);
}

// x > 0, normal
FDQL_HD double dm_log(double x) {
  int e;
  double m = __builtin_frexp(x, &e);          // [0.5, 1)
  if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }   // [sqrt(1/2), sqrt(2))
  const double s = (m - 1.0) / (m + 1.0);     // |s| <= 0.1716; log m = 2 atanh s
  const double z = s * s;
  double p = 1.0 / 19.0;
  p = __builtin_fma(p, z, 1.0 / 17.0);
  p = __builtin_fma(p, z, 1.0 / 15.0);
  p = __builtin_fma(p, z, 1.0 / 13.0);
  p = __builtin_fma(p, z, 1.0 / 11.0);
  p = __builtin_fma(p, z, 1.0 / 9.0);
  p = __builtin_fma(p, z, 1.0 / 7.0);
  p = __builtin_fma(p, z, 1.0 / 5.0);
  p = __builtin_fma(p, z, 1.0 / 3.0);
  const double lm = __builtin_fma(p * z, 2.0 * s, 2.0 * s);
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  return __builtin_fma((double)e, ln2_hi, __builtin_fma((double)e, ln2_lo, lm));
}

// tanh x = em / (em + 2), em = expm1(2 |x|)
FDQL_HD double dm_tanh(double x) {
  if (x != x) return x;   // NaN in -> NaN out (a diverged policy mean must not come back as a saturated action)
  const double ax = __builtin_fabs(x);
  const double u = 2.0 * (ax < 20.0 ? ax : 20.0);   // tanh(20) rounds to 1 in double
  int n;
  const double r = dm_reduce_ln2(u, n);
  const double p = dm_expm1_small(r);
  const double em = n == 0 ? p : __builtin_ldexp(1.0 + p, n) - 1.0;
  const double t = em / (em + 2.0);
  return __builtin_copysign(t, x);
}

}  // namespace fdql
