// near_tables.hpp -- host side of the sweep kernel's accurate tier: piecewise polynomials for
// f_j(x) = Re w(x + i y_j), 0 <= x < 32, one set per Lyman line j (y_j = gamma_j / (sqrt2 sigma) is a
// constant of the line, voigt.c:146-220, so within 30 Doppler widths of a line centre the Voigt
// function is a function of x alone).
//
// Why tables: inside |x| < 30 the asymptotic wing formula is not accurate enough and Re w needs the
// series / trapezoid tiers of faddeeva.hpp (130 / 1200 instructions, divergent).  In the sweep
// kernel that meant a call with all accumulators live in ~5 % of the K-steps, 8.5 % of the run
// time (tools/ablate.sh) plus the chunk barrier waiting for whichever wave drew it.  A degree-11
// polynomial per interval costs ~35 instructions, inline.
//
// Layout: kNearIntervals = 64 intervals of width 1/8 on [0, 8) (Gaussian core), then 48 of width
// 1/2 on [8, 32) (damping wing), kNearCoef = 12 monomial coefficients each in the local variable
// t in [-1/2, 1/2] (interval midpoint = 0).  Interpolation error at these widths: <= 1.4e-15
// relative (checked against 40-digit mpmath for y from 1e-6 to 4.7e-4; tests/test_near_tables.py
// repeats the check through gpdla_debug_near_poly).
//
// The node values are computed here in long double (x87, 64-bit mantissa), so the tables carry
// interpolation error only: the same trapezoid rule with the pole correction folded in as
// faddeeva.hpp's rew_core for x < 8 and the asymptotic series (24 terms) beyond.
#pragma once
#include <cmath>
#include <vector>

namespace gpdla {

constexpr int kNearCore = 64;       // intervals of width 1/8 on [0, 8)
constexpr int kNearOuter = 48;      // intervals of width 1/2 on [8, 32)
constexpr int kNearIntervals = kNearCore + kNearOuter;
constexpr int kNearCoef = 12;
constexpr int kNearLineDoubles = kNearIntervals * kNearCoef;

namespace near_detail {

typedef long double ld;
constexpr ld kPiL = 3.14159265358979323846264338327950288L;

// x >= 8: Re w(z) = Re[(i / (sqrt(pi) z)) Sum_m (2m-1)!! / (2 z^2)^m]; smallest term at x = 8 is
// m ~ 64, 24 terms leave < 1e-21 relative.
inline ld rew_series_ld(ld x, ld y) {
  const ld rho = 1.0L / (x * x + y * y);
  const ld ur = (x * x - y * y) * rho * rho;  // u = 1/z^2
  const ld ui = -2.0L * x * y * rho * rho;
  constexpr int M = 24;
  ld c[M + 1];
  c[0] = 1.0L;
  for (int m = 1; m <= M; ++m) c[m] = c[m - 1] * (ld)(2 * m - 1) / 2.0L;
  ld sr = c[M], si = 0.0L;
  for (int m = M - 1; m >= 0; --m) {
    const ld nr = sr * ur - si * ui + c[m];
    const ld ni = sr * ui + si * ur;
    sr = nr;
    si = ni;
  }
  return rho * (sr * y - si * x) / std::sqrt(kPiL);
}

// x < 8 (valid to ~9): trapezoid rule, step h = 0.4, on Re w = (y/pi) Int exp(-t^2) / ((x-t)^2 + y^2) dt
// with nodes t = x + (n0 + m) h ... written, as in faddeeva.hpp, with the node grid anchored at the
// origin and the pole correction folded into the node nearest to x.
inline ld rew_core_ld(ld x, ld y) {
  const ld h = 0.4L;
  constexpr int NT = 19;
  const ld y2 = y * y;
  const ld n0f = -std::nearbyint(x / h);
  const ld t0 = n0f * h + x;  // in [-h/2, h/2]
  const int n0 = (int)n0f;
  const ld e0 = std::exp(-t0 * t0);
  const ld a = std::exp(-2.0L * h * t0);
  const ld ai = 1.0L / a;
  ld s = 0.0L;
  if (n0 != 0) {
    const ld nh = n0f * h;
    s = e0 / (nh * nh + y2);
  }
  ld ap = 1.0L, am = 1.0L;
  for (int m = 1; m <= NT; ++m) {
    ap *= a;
    am *= ai;
    const ld em = e0 * std::exp(-(ld)(m * m) * h * h);
    const int np = n0 + m, nm = n0 - m;
    if (np != 0) {
      const ld nh = (ld)np * h;
      s += em * ap / (nh * nh + y2);
    }
    if (nm != 0) {
      const ld nh = (ld)nm * h;
      s += em * am / (nh * nh + y2);
    }
  }
  s *= y * h / kPiL;
  const ld q = 2.0L * kPiL * y / h;
  ld b;  // 1/q - 1/expm1(q); the difference cancels for the small q of the Lyman lines
  if (q < 0.05L) {
    const ld q2 = q * q;
    b = 0.5L - q * (1.0L / 12.0L - q2 * (1.0L / 720.0L - q2 * (1.0L / 30240.0L -
               q2 * (1.0L / 1209600.0L - q2 / 47900160.0L))));
  } else {
    b = 1.0L / q - 1.0L / std::expm1(q);
  }
  const ld sxy = std::sin(x * y);
  const ld core = std::exp(-x * x) *
                  (2.0L * b + 2.0L * (2.0L * sxy * sxy - std::expm1(y2) * std::cos(2.0L * x * y)) / std::expm1(q));
  return s + core;
}

inline ld rew_ld(ld x, ld y) { return x >= 8.0L ? rew_series_ld(x, y) : rew_core_ld(x, y); }

// Interpolating polynomial of degree kNearCoef-1 through the Chebyshev nodes of [lo, lo + width],
// as monomial coefficients in t = (x - lo)/width - 1/2.  Gaussian elimination with partial
// pivoting in long double on the 12 x 12 Vandermonde system.
inline void fit_interval(ld lo, ld width, ld y, double *coef) {
  constexpr int n = kNearCoef;
  ld V[n][n + 1];
  for (int i = 0; i < n; ++i) {
    const ld t = -0.5L * std::cos((ld)(2 * i + 1) * kPiL / (ld)(2 * n));
    ld p = 1.0L;
    for (int j = 0; j < n; ++j) {
      V[i][j] = p;
      p *= t;
    }
    V[i][n] = rew_ld(lo + (t + 0.5L) * width, y);
  }
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r)
      if (std::fabs(V[r][c]) > std::fabs(V[piv][c])) piv = r;
    if (piv != c)
      for (int j = 0; j <= n; ++j) std::swap(V[c][j], V[piv][j]);
    for (int r = c + 1; r < n; ++r) {
      const ld f = V[r][c] / V[c][c];
      for (int j = c; j <= n; ++j) V[r][j] -= f * V[c][j];
    }
  }
  ld sol[n];
  for (int r = n - 1; r >= 0; --r) {
    ld acc = V[r][n];
    for (int j = r + 1; j < n; ++j) acc -= V[r][j] * sol[j];
    sol[r] = acc / V[r][r];
  }
  for (int r = 0; r < n; ++r) coef[r] = (double)sol[r];
}

}  // namespace near_detail

// Interval index and local variable of |x| < 32.
inline void near_locate(double ax, int *idx, double *t) {
  if (ax < 8.0) {
    const double u = ax * 8.0;
    const int i = (int)u;
    *idx = i;
    *t = (u - (double)i) - 0.5;
  } else {
    const double u = (ax - 8.0) * 2.0;
    const int i = (int)u;
    *idx = kNearCore + i;
    *t = (u - (double)i) - 0.5;
  }
}

// Host evaluation of one line's table (the device does the same Horner, sweep_kernels.hpp).
inline double near_poly_host(const double *line_tab, double ax) {
  int idx;
  double t;
  near_locate(ax, &idx, &t);
  const double *c = line_tab + (size_t)idx * kNearCoef;
  double p = c[kNearCoef - 1];
  for (int k = kNearCoef - 2; k >= 0; --k) p = std::fma(p, t, c[k]);
  return p;
}

// Tables for nlines lines with damping parameters y[j]: out[j][interval][coef].
inline void build_near_tables(const double *y, int nlines, std::vector<double> &out) {
  out.assign((size_t)nlines * kNearLineDoubles, 0.0);
  for (int j = 0; j < nlines; ++j) {
    double *tab = out.data() + (size_t)j * kNearLineDoubles;
    for (int i = 0; i < kNearIntervals; ++i) {
      const long double lo = i < kNearCore ? (long double)i / 8.0L : 8.0L + (long double)(i - kNearCore) / 2.0L;
      const long double width = i < kNearCore ? 0.125L : 0.5L;
      near_detail::fit_interval(lo, width, (long double)y[j], tab + (size_t)i * kNearCoef);
    }
  }
}

}  // namespace gpdla
