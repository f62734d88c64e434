// faddeeva.hpp -- Re w(x + iy) on the device, for the Lyman-series regime 0 < y <= ~1e-3.
//
// Replaces the third-party call the reference makes at voigt.c:288 (libcerf voigt(x, sigma, gamma)
// = Re w((x + i gamma)/(sqrt2 sigma)) / (sqrt(2 pi) sigma)).  Three tiers by |x|:
//
//   |x| >= 30      rew_wing   : (y rho/sqrt(pi)) [T(rho) - 2 y^2 rho^2], rho = 1/(x^2+y^2), T the
//                               6-term asymptotic (2m+1)!!/2^m series.  One reciprocal + 9 FMAs.
//                               Relative error <= 6e-16 (tests/test_oracle_voigt.py grid, mpmath).
//   8 <= |x| < 30  rew_series : (i/sqrt(pi) z) Sum (2m-1)!!/(2 z^2)^m, 16 terms, complex Horner.
//   |x| < 8        rew_core   : trapezoid rule on the Voigt integral with the pole correction
//                               folded analytically into the n = 0 node (no cancellation);
//                               h = 0.4 -> quadrature error exp(-pi^2/h^2) = 1.6e-27.
//
// These tiers serve gpdla_voigt (k_voigt_raw) directly and, evaluated in long double on the host,
// are what near_tables.hpp fits its per-line polynomials to.  The sweep kernels evaluate the wing
// formula (economised, sweep_kernels.hpp) branch-free for every (pixel, line) and, under a
// wave-uniform vote when some lane is within 30 Doppler widths of a line centre (about 11 pixels
// per line; samples are processed in z order so those lanes coincide), those polynomials.
#pragma once
#include <hip/hip_runtime.h>

#define GPDLA_HD __host__ __device__

namespace gpdla {

// 1/a to ~1 ulp: hardware seed (2^-24) + two Newton steps on the device, plain division on the host.
GPDLA_HD __forceinline__ double fd_rcp(double a) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(a);
  double e = fma(-a, r, 1.0);
  r = fma(r, e, r);
  e = fma(-a, r, 1.0);
  return fma(r, e, r);
#else
  return 1.0 / a;
#endif
}

constexpr double kInvSqrtPi = 0.56418958354775628694807945156;
constexpr double kPi = 3.14159265358979323846;

// (2m+1)!!/2^m, m = 0..6
constexpr double kT0 = 1.0, kT1 = 1.5, kT2 = 3.75, kT3 = 13.125, kT4 = 59.0625,
                 kT5 = 324.84375, kT6 = 2111.484375;

GPDLA_HD __forceinline__ double rew_wing(double x, double y) {
  const double y2 = y * y;
  const double rho = 1.0 / fma(x, x, y2);
  double t = fma(kT6, rho, kT5);
  t = fma(t, rho, kT4);
  t = fma(t, rho, kT3);
  t = fma(t, rho, kT2);
  t = fma(t, rho, kT1);
  t = fma(t, rho, kT0);
  t = fma(-2.0 * y2 * rho, rho, t);
  return (y * rho) * kInvSqrtPi * t;
}

// 8 <= |x|: asymptotic series in u = 1/z^2, complex Horner, coefficients (2m-1)!!/2^m.
GPDLA_HD __forceinline__ double rew_series(double x, double y) {
  const double rho = 1.0 / fma(x, x, y * y);
  const double ur = (x * x - y * y) * rho * rho;
  const double ui = -2.0 * x * y * rho * rho;
  // c_m, m = 16 .. 0
  constexpr double c[17] = {1.0, 0.5, 0.75, 1.875, 6.5625, 29.53125, 162.421875, 1055.7421875,
                            7918.06640625, 67303.564453125, 639383.8623046875, 6713530.554199219,
                            77205601.37329102, 965070017.1661377, 13028445231.74286,
                            188912455860.2715, 2928143065834.208};
  double sr = c[16], si = 0.0;
#pragma unroll
  for (int m = 15; m >= 0; --m) {
    const double nr = fma(sr, ur, fma(-si, ui, c[m]));
    const double ni = fma(sr, ui, si * ur);
    sr = nr;
    si = ni;
  }
  return rho * kInvSqrtPi * fma(sr, y, -si * x);
}

// |x| < 8 (valid to |x| ~ 9): see header comment.  exp(-(t0 + m h)^2) = exp(-t0^2) a^m E[m] with
// a = exp(-2 t0 h), E[m] = exp(-m^2 h^2) tabulated, |t0| <= h/2.
GPDLA_HD inline double rew_core(double x, double y) {
  constexpr double h = 0.4;
  constexpr int NT = 17;
  constexpr double E[NT + 1] = {
      1.0, 0.8521437889662113, 0.5272924240430485, 0.23692775868212165,
      0.07730474044329971, 0.01831563888873418, 0.0031511115984444358, 0.0003936690406550776,
      3.5712849641635144e-05, 2.352575200009771e-06, 1.1253517471925912e-07, 3.90893843426485e-09,
      9.859505575991446e-11, 1.8058314375132107e-12, 2.4017347816209437e-14, 2.319522830243569e-16,
      1.6266646214532314e-18, 8.2836770076828e-21};
  const double y2 = y * y;
  const double n0f = -rint(x * (1.0 / h));
  const double t0 = fma(n0f, h, x);  // in [-h/2, h/2]
  const int n0 = (int)n0f;
  const double e0 = exp(-t0 * t0);
  const double a = exp(-2.0 * h * t0);
  const double ai = 1.0 / a;
  double s = 0.0;
  if (n0 != 0) {
    const double nh = n0f * h;
    s = e0 / fma(nh, nh, y2);
  }
  double ap = 1.0, am = 1.0;
#pragma unroll 1
  for (int m = 1; m <= NT; ++m) {
    ap *= a;
    am *= ai;
    const double em = e0 * E[m];
    const int np = n0 + m, nm = n0 - m;
    if (np != 0) {
      const double nh = (double)np * h;
      s = fma(em * ap, fd_rcp(fma(nh, nh, y2)), s);
    }
    if (nm != 0) {
      const double nh = (double)nm * h;
      s = fma(em * am, fd_rcp(fma(nh, nh, y2)), s);
    }
  }
  s *= y * h / kPi;
  const double q = 2.0 * kPi * y / h;
  double b;  // 1/q - 1/expm1(q)
  if (q < 0.5) {
    const double q2 = q * q;
    b = 0.5 - q * (1.0 / 12.0 - q2 * (1.0 / 720.0 - q2 * (1.0 / 30240.0 -
                  q2 * (1.0 / 1209600.0 - q2 / 47900160.0))));
  } else {
    b = 1.0 / q - 1.0 / expm1(q);
  }
  const double sxy = sin(x * y);
  const double core =
      exp(-x * x) * (2.0 * b + 2.0 * (2.0 * sxy * sxy - expm1(y2) * cos(2.0 * x * y)) / expm1(q));
  return s + core;
}

GPDLA_HD inline double rew_full(double x, double y) {
  x = fabs(x);
  if (x >= 30.0) return rew_wing(x, y);
  if (x >= 8.0) return rew_series(x, y);
  return rew_core(x, y);
}

}  // namespace gpdla
