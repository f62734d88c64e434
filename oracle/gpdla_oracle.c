/* gpdla_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see gpdla_oracle.h for the rules).
 *
 * Plain C, fp64, scalar loops in the reference's as-written operation order.  Build with
 * -ffp-contract=off so that a*b+c stays two roundings, as in the reference's x86 MEX/MATLAB build.
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#include "gpdla_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/gpdla_lyman_series.h"

/* ---- line data (voigt.c:31-220), unpacked from the generated X-macro ---- */
#define COL_WL(i, wl, f, G, lead, gam) wl,
#define COL_F(i, wl, f, G, lead, gam) f,
#define COL_LEAD(i, wl, f, G, lead, gam) lead,
#define COL_GAM(i, wl, f, G, lead, gam) gam,
static const double transition_wavelengths[GPDLA_MAX_LINES] = {GPDLA_LYMAN_SERIES(COL_WL)};
static const double oscillator_strengths[GPDLA_MAX_LINES] = {GPDLA_LYMAN_SERIES(COL_F)};
static const double leading_constants[GPDLA_MAX_LINES] = {GPDLA_LYMAN_SERIES(COL_LEAD)};
static const double gammas[GPDLA_MAX_LINES] = {GPDLA_LYMAN_SERIES(COL_GAM)};
static const double instrument_profile[2 * GPDLA_CONV_WIDTH + 1] = GPDLA_INSTRUMENT_PROFILE;
static const double c_cgs = GPDLA_SPEED_OF_LIGHT_CGS; /* voigt.c:22  */
static const double sigma_cgs = GPDLA_GAUSS_SIGMA_CGS; /* voigt.c:146 */

/* ------------------------------------------------------------------------------------------
 * Faddeeva function, real part, for 0 < y << 1.
 *
 * libcerf is absent (gpdla_oracle.h), so its published definition is evaluated with an
 * independent method:
 *   |x| <  8 : Re w = (y/pi) Int exp(-t^2)/((x-t)^2+y^2) dt, trapezoid rule in s = t - x with
 *              step h.  The integrand's poles s = +-iy add the exact correction
 *              -2 exp(y^2-x^2) cos(2xy)/(exp(2 pi y/h)-1); that and the n = 0 node (both
 *              ~ h/(pi y) exp(-x^2)) are combined analytically so nothing cancels.
 *              Remaining quadrature error ~ exp(-pi^2/h^2) = 1.6e-27 for h = 0.4.
 *   |x| >= 8 : Laplace continued fraction  w(z) = (i/sqrt(pi)) / (z - (1/2)/(z - 1/(z - ...))),
 *              16 levels (12 already give 1e-15 at |x| = 8; exp(-x^2) < 4e-23 of Re w there).
 * Checked against 40-digit mpmath and against scipy.special.wofz in tests/test_oracle_voigt.py.
 * ------------------------------------------------------------------------------------------ */
double gpdla_oracle_faddeeva_re(double x, double y) {
  const double pi = 3.14159265358979323846;
  x = fabs(x);
  if (y <= 0.0) return exp(-x * x);
  if (x >= 8.0) {
    double wr = x, wi = y;
    for (int lev = 16; lev >= 1; --lev) {
      double kk = 0.5 * lev;
      double den = kk / (wr * wr + wi * wi);
      wr = x - wr * den;
      wi = y + wi * den;
    }
    return wi / (sqrt(pi) * (wr * wr + wi * wi));
  }
  const double h = 0.4;
  double q = 2.0 * pi * y / h;
  long n0 = -lround(x / h);
  double s = 0.0;
  for (long n = n0 - 17; n <= n0 + 17; ++n) {
    if (n == 0) continue;
    double t = x + n * h;
    double nh = n * h;
    s += exp(-t * t) / (nh * nh + y * y);
  }
  s *= y * h / pi;
  double b; /* 1/q - 1/expm1(q) */
  if (q < 0.5) {
    double q2 = q * q;
    b = 0.5 - q * (1.0 / 12.0 - q2 * (1.0 / 720.0 - q2 * (1.0 / 30240.0 -
              q2 * (1.0 / 1209600.0 - q2 / 47900160.0))));
  } else {
    b = 1.0 / q - 1.0 / expm1(q);
  }
  double sxy = sin(x * y);
  double core = exp(-x * x) *
                (2.0 * b + 2.0 * (2.0 * sxy * sxy - expm1(y * y) * cos(2.0 * x * y)) / expm1(q));
  return s + core;
}

/* libcerf voigt(x, sigma, gamma), call site voigt.c:288. */
double gpdla_oracle_voigt_line(double x, double sigma, double gamma) {
  const double pi = 3.14159265358979323846;
  double zr = x / sqrt(2.0) / sigma;
  double zi = gamma / sqrt(2.0) / sigma;
  return gpdla_oracle_faddeeva_re(zr, zi) / sqrt(2.0 * pi) / sigma;
}

/* voigt.c:277-292 */
int gpdla_oracle_voigt_raw(const double *lambdas, int64_t num_points, double z, double N,
                           int num_lines, double *raw_profile) {
  double multipliers[GPDLA_MAX_LINES];
  if (num_lines < 1 || num_lines > GPDLA_MAX_LINES) return -2;
  for (int i = 0; i < num_lines; i++) /* voigt.c:278-279 */
    multipliers[i] = c_cgs / (transition_wavelengths[i] * (1 + z)) / 1e8;
  for (int64_t i = 0; i < num_points; i++) { /* voigt.c:282-292 */
    double total = 0;
    for (int j = 0; j < num_lines; j++) {
      double velocity = lambdas[i] * multipliers[j] - c_cgs;
      total += -leading_constants[j] * gpdla_oracle_voigt_line(velocity, sigma_cgs, gammas[j]);
    }
    raw_profile[i] = exp(N * total);
  }
  return 0;
}

/* voigt.c:253-304 */
static int voigt_ws(const double *lambdas, int64_t num_points, double z, double N, int num_lines,
                    double *profile, double *raw_profile);

int gpdla_oracle_voigt(const double *lambdas, int64_t num_points, double z, double N,
                       int num_lines, double *profile) {
  if (num_points <= 2 * GPDLA_CONV_WIDTH) return -1;
  double *raw_profile = (double *)malloc((size_t)num_points * sizeof(double)); /* voigt.c:275 */
  int rc = voigt_ws(lambdas, num_points, z, N, num_lines, profile, raw_profile);
  free(raw_profile);                                                           /* voigt.c:302 */
  return rc;
}

static int voigt_ws(const double *lambdas, int64_t num_points, double z, double N, int num_lines,
                    double *profile, double *raw_profile) {
  const int width = GPDLA_CONV_WIDTH;
  int rc = gpdla_oracle_voigt_raw(lambdas, num_points, z, N, num_lines, raw_profile);
  if (rc) return rc;
  int64_t num_out = num_points - 2 * width; /* voigt.c:271, :294 */
  for (int64_t i = 0; i < num_out; i++) {   /* voigt.c:297-299; output starts zero-filled (:271) */
    double acc = 0.0;
    for (int64_t j = i, k = 0; j <= i + 2 * width; j++, k++) acc += raw_profile[j] * instrument_profile[k];
    profile[i] = acc;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * log_mvnpdf_low_rank.m:5-34, as written (including the k x n matrix C of :26).
 * MATLAB's chol returns upper-triangular R with R'R = B (:24).
 * ------------------------------------------------------------------------------------------ */
/* doubles of scratch the evaluation needs (y, D_inv_y: n each; D_inv_M, C: n k each; B, R: k k
 * each; Cy: k) */
size_t gpdla_oracle_lowrank_scratch_doubles(int64_t n, int k) {
  return 2 * (size_t)n + 2 * (size_t)n * k + 2 * (size_t)k * k + (size_t)k;
}

/* The evaluation proper, on caller-provided scratch (so that a sweep allocates once per thread,
 * not once per sample: at n = 1500 each of these arrays is past glibc's mmap threshold, and 7
 * mmap/munmap pairs per sample serialise every thread on the kernel's address-space lock). */
static int lowrank_ws(const double *y_in, const double *mu, const double *M, const double *d,
                      int64_t n, int k, double *log_p, double *ws) {
  const double log_2pi = 1.83787706640934534; /* :7 */
  double *y = ws;
  double *D_inv_y = y + n;
  double *D_inv_M = D_inv_y + n;
  double *C = D_inv_M + (size_t)n * k; /* k x n, column-major */
  double *B = C + (size_t)n * k;
  double *R = B + (size_t)k * k;
  double *Cy = R + (size_t)k * k;
  memset(B, 0, sizeof(double) * ((size_t)2 * k * k + k)); /* B, R, Cy start at zero */
  int rc = 0;

  for (int64_t i = 0; i < n; i++) y[i] = y_in[i] - mu[i]; /* :11 */
  for (int64_t i = 0; i < n; i++) {                       /* :13-15 */
    double d_inv = 1.0 / d[i];
    D_inv_y[i] = d_inv * y[i];
    for (int a = 0; a < k; a++) D_inv_M[i + a * n] = d_inv * M[i + a * n];
  }
  for (int a = 0; a < k; a++) /* :22  B = M' * D_inv_M */
    for (int b = 0; b < k; b++) {
      double acc = 0.0;
      for (int64_t i = 0; i < n; i++) acc += M[i + a * n] * D_inv_M[i + b * n];
      B[a + b * k] = acc;
    }
  for (int a = 0; a < k; a++) B[a + a * k] += 1.0; /* :23 */
  /* :24  R = chol(B), upper: R(j,j) = sqrt(B(j,j) - sum_{m<j} R(m,j)^2),
   *      R(j,i) = (B(j,i) - sum_{m<j} R(m,j) R(m,i)) / R(j,j) for i > j */
  for (int j = 0; j < k && !rc; j++) {
    double s = B[j + j * k];
    for (int m = 0; m < j; m++) s -= R[m + j * k] * R[m + j * k];
    if (!(s > 0.0)) {
      rc = -1;
      break;
    }
    double rjj = sqrt(s);
    R[j + j * k] = rjj;
    for (int i = j + 1; i < k; i++) {
      double t = B[j + i * k];
      for (int m = 0; m < j; m++) t -= R[m + j * k] * R[m + i * k];
      R[j + i * k] = t / rjj;
    }
  }
  if (rc) {
    *log_p = NAN;
    return rc;
  }
  /* :26  C = R \ (R' \ D_inv_M')  -- one column (pixel) at a time */
  for (int64_t i = 0; i < n; i++) {
    double *c = C + (size_t)i * k;
    for (int a = 0; a < k; a++) { /* forward: R' t = D_inv_M(i,:)' */
      double t = D_inv_M[i + a * n];
      for (int m = 0; m < a; m++) t -= R[m + a * k] * c[m];
      c[a] = t / R[a + a * k];
    }
    for (int a = k - 1; a >= 0; a--) { /* backward: R c = t */
      double t = c[a];
      for (int m = a + 1; m < k; m++) t -= R[a + m * k] * c[m];
      c[a] = t / R[a + a * k];
    }
  }
  /* :28  K_inv_y = D_inv_y - D_inv_M * (C * y) */
  for (int64_t i = 0; i < n; i++)
    for (int a = 0; a < k; a++) Cy[a] += C[(size_t)i * k + a] * y[i];
  double quad = 0.0;
  for (int64_t i = 0; i < n; i++) {
    double t = 0.0;
    for (int a = 0; a < k; a++) t += D_inv_M[i + a * n] * Cy[a];
    double K_inv_y = D_inv_y[i] - t;
    quad += y[i] * K_inv_y; /* y' * K_inv_y of :32 */
  }
  double log_det_K = 0.0; /* :30 */
  for (int64_t i = 0; i < n; i++) log_det_K += log(d[i]);
  double sld = 0.0;
  for (int a = 0; a < k; a++) sld += log(R[a + a * k]);
  log_det_K += 2 * sld;
  *log_p = -0.5 * (quad + log_det_K + (double)n * log_2pi); /* :32 */
  return rc;
}

int gpdla_oracle_log_mvnpdf_low_rank(const double *y_in, const double *mu, const double *M,
                                     const double *d, int64_t n, int k, double *log_p) {
  double *ws = (double *)malloc(sizeof(double) * gpdla_oracle_lowrank_scratch_doubles(n, k));
  int rc = lowrank_ws(y_in, mu, M, d, n, k, log_p, ws);
  free(ws);
  return rc;
}

/* ------------------------------------------------------------------------------------------
 * Helpers for the drivers.
 * ------------------------------------------------------------------------------------------ */

/* griddedInterpolant(grid, values, 'linear') evaluated at x (process_qsos.m:66-71,138-141).
 * MATLAB does not document the rounding of its linear kernel; restated as
 * v0 + (v1 - v0) * (x - g0)/(g1 - g0) on the bracketing interval (linear extrapolation outside,
 * MATLAB's default for 'linear'; never reached because :104-105 confines x to the grid). */
static void bracket(const double *grid, int G, double x, int *i0, double *t) {
  int lo = 0, hi = G - 1;
  if (x <= grid[0]) {
    lo = 0;
  } else if (x >= grid[G - 1]) {
    lo = G - 2;
  } else {
    while (hi - lo > 1) {
      int mid = (lo + hi) / 2;
      if (grid[mid] <= x) lo = mid; else hi = mid;
    }
  }
  *i0 = lo;
  *t = (x - grid[lo]) / (grid[lo + 1] - grid[lo]);
}

static double lerp(const double *v, int i0, double t) { return v[i0] + (v[i0 + 1] - v[i0]) * t; }

/* logspace(a, b, 3) of process_qsos.m:169-175: 10.^linspace(a,b,3); linspace sets the end points
 * exactly and the middle to a + 1*(b-a)/2. */
static void logspace3(double a, double b, double *out) {
  out[0] = pow(10.0, a);
  out[1] = pow(10.0, a + 1.0 * (b - a) / 2.0);
  out[2] = pow(10.0, b);
}

typedef struct {
  int64_t n;            /* kept pixels */
  int64_t n_u;          /* unmasked-range pixels */
  double *wavelengths;  /* [n] kept */
  double *rest;         /* [n] */
  double *flux;         /* [n] */
  double *noise;        /* [n] */
  double *unmasked_wl;  /* [n_u] */
  uint8_t *keep;        /* [n_u]  ~pixel_mask(unmasked_ind), process_qsos.m:181 */
} selection;

/* process_qsos.m:102-119 */
static int select_pixels(const gpdla_oracle_params *prm, int64_t num_pixels, const double *wl,
                         const double *flux, const double *nv, const uint8_t *mask, double z_qso,
                         selection *s) {
  memset(s, 0, sizeof(*s));
  s->wavelengths = (double *)malloc(sizeof(double) * (size_t)(num_pixels + 1));
  s->rest = (double *)malloc(sizeof(double) * (size_t)(num_pixels + 1));
  s->flux = (double *)malloc(sizeof(double) * (size_t)(num_pixels + 1));
  s->noise = (double *)malloc(sizeof(double) * (size_t)(num_pixels + 1));
  s->unmasked_wl = (double *)malloc(sizeof(double) * (size_t)(num_pixels + 1));
  s->keep = (uint8_t *)malloc((size_t)(num_pixels + 1));
  for (int64_t i = 0; i < num_pixels; i++) {
    double rest = wl[i] / (1 + z_qso);                                      /* :102 */
    int unmasked = (rest >= prm->min_lambda) && (rest <= prm->max_lambda);  /* :104-105 */
    if (!unmasked) continue;
    s->unmasked_wl[s->n_u] = wl[i];                                         /* :108 */
    s->keep[s->n_u] = (uint8_t)(!mask[i]);                                  /* :181 */
    s->n_u++;
    if (mask[i]) continue;                                                  /* :110 */
    s->wavelengths[s->n] = wl[i];                                           /* :112-115 */
    s->rest[s->n] = rest;
    s->flux[s->n] = flux[i];
    s->noise[s->n] = nv[i];
    s->n++;
  }
  return s->n > 0 ? 0 : -1;
}

static void free_selection(selection *s) {
  free(s->wavelengths);
  free(s->rest);
  free(s->flux);
  free(s->noise);
  free(s->unmasked_wl);
  free(s->keep);
}

static double vmin(const double *v, int64_t n) {
  double m = v[0];
  for (int64_t i = 1; i < n; i++) if (v[i] < m) m = v[i];
  return m;
}
static double vmax(const double *v, int64_t n) {
  double m = v[0];
  for (int64_t i = 1; i < n; i++) if (v[i] > m) m = v[i];
  return m;
}

/* process_qsos.m:168-176 */
static double *padded_grid(const gpdla_oracle_params *prm, const selection *s) {
  int w = prm->width;
  double *p = (double *)malloc(sizeof(double) * (size_t)(s->n_u + 2 * w));
  double lo = log10(vmin(s->unmasked_wl, s->n_u));
  double hi = log10(vmax(s->unmasked_wl, s->n_u));
  /* width is 3 in every parameter file (set_parameters.m:59; voigt.c:229 hard-codes it) */
  logspace3(lo - w * prm->pixel_spacing, lo - prm->pixel_spacing, p);
  memcpy(p + w, s->unmasked_wl, sizeof(double) * (size_t)s->n_u);
  logspace3(hi + prm->pixel_spacing, hi + w * prm->pixel_spacing, p + w + s->n_u);
  return p;
}

/* set_parameters.m:65-73 */
static void z_dla_range(const gpdla_oracle_params *prm, const selection *s, double z_qso,
                        double *zmin, double *zmax) {
  double wmin = vmin(s->wavelengths, s->n), wmax = vmax(s->wavelengths, s->n);
  *zmax = (wmax / prm->lya_wavelength - 1) - prm->max_z_cut;
  double a = wmin / prm->lya_wavelength - 1;
  double b = prm->lyman_limit * (1 + z_qso) / prm->lya_wavelength - 1 + prm->min_z_cut;
  *zmin = a > b ? a : b;
}

static size_t sample_scratch_doubles(int64_t n, int k) {
  return (size_t)n * (2 + k) + gpdla_oracle_lowrank_scratch_doubles(n, k);
}

/* the body of the sweep, process_qsos.m:187-198: one sample's log-likelihood given the (already
 * multiplied, for the multi-DLA driver) absorption on the unmasked grid */
static double sample_loglik(const selection *s, int k, const double *absorption_u,
                            const double *this_mu, const double *this_M, const double *this_omega2,
                            double *scratch /* sample_scratch_doubles(n, k) */) {
  int64_t n = s->n;
  double *dla_mu = scratch, *dla_d = scratch + n, *dla_M = scratch + 2 * n;
  double *ws = dla_M + (size_t)n * k;
  int64_t j = 0;
  for (int64_t i = 0; i < s->n_u; i++) { /* absorption(ind), :190 */
    if (!s->keep[i]) continue;
    double a = absorption_u[i];
    dla_mu[j] = this_mu[j] * a;                          /* :192 */
    for (int c = 0; c < k; c++) dla_M[j + c * n] = this_M[j + c * n] * a; /* :193 */
    dla_d[j] = this_omega2[j] * (a * a) + s->noise[j];   /* :194, :198 */
    j++;
  }
  double lp;
  lowrank_ws(s->flux, dla_mu, dla_M, dla_d, n, k, &lp, ws); /* :196-198 */
  return lp;
}

/* process_qsos.m:203-210 (max / exp / mean / log), NaN-aware like nanmax/nanmean when
 * nan_aware != 0 (multi :400-409).  Returns NaN if every entry is NaN. */
static double log_mean_exp(const double *ll, int64_t S, int nan_aware, double *max_out) {
  double mx = -INFINITY;
  int any = 0;
  for (int64_t i = 0; i < S; i++) {
    if (isnan(ll[i])) {
      if (!nan_aware) { mx = NAN; any = 1; break; }
      continue;
    }
    if (!any || ll[i] > mx) mx = ll[i];
    any = 1;
  }
  if (!any) mx = NAN;
  if (max_out) *max_out = mx;
  double sum = 0.0;
  int64_t cnt = 0;
  for (int64_t i = 0; i < S; i++) {
    double p = exp(ll[i] - mx);
    if (nan_aware && isnan(p)) continue;
    sum += p;
    cnt++;
  }
  return mx + log(sum / (double)cnt);
}

/* ------------------------------------------------------------------------------------------
 * process_qsos.m:96-213 for one quasar.
 * ------------------------------------------------------------------------------------------ */
int gpdla_oracle_process_spectrum(const gpdla_oracle_params *prm, const gpdla_oracle_model *mdl,
                                  int64_t S, const double *offset_samples,
                                  const double *nhi_samples, int64_t num_pixels,
                                  const double *wavelengths, const double *flux,
                                  const double *noise_variance, const uint8_t *pixel_mask,
                                  double z_qso, int num_threads, double *min_z_dla,
                                  double *max_z_dla, double *log_likelihood_no_dla,
                                  double *sample_ll, double *log_likelihood_dla,
                                  gpdla_oracle_dump *dump) {
  selection s;
  const int k = mdl->k, G = mdl->num_rest;
  if (select_pixels(prm, num_pixels, wavelengths, flux, noise_variance, pixel_mask, z_qso, &s)) {
    free_selection(&s);
    return -1;
  }
  const int64_t n = s.n;
  double c_0 = exp(mdl->log_c_0), tau_0 = exp(mdl->log_tau_0), beta = exp(mdl->log_beta); /* :84-86 */
  double *this_mu = (double *)malloc(sizeof(double) * (size_t)n);
  double *this_M = (double *)malloc(sizeof(double) * (size_t)n * k);
  double *this_omega2 = (double *)malloc(sizeof(double) * (size_t)n);
  double *d0 = (double *)malloc(sizeof(double) * (size_t)n);
  for (int64_t i = 0; i < n; i++) {
    int i0;
    double t;
    bracket(mdl->rest_wavelengths, G, s.rest[i], &i0, &t);
    this_mu[i] = lerp(mdl->mu, i0, t);                                       /* :138 */
    for (int c = 0; c < k; c++) this_M[i + c * n] = lerp(mdl->M + (size_t)c * G, i0, t); /* :139 */
    double this_log_omega = lerp(mdl->log_omega, i0, t);                     /* :141 */
    double omega2 = exp(2 * this_log_omega);                                 /* :142 */
    double lya_z = (s.wavelengths[i] - prm->lya_wavelength) / prm->lya_wavelength; /* :117-119 */
    double scaling = 1 - exp(-tau_0 * pow(1 + lya_z, beta)) + c_0;           /* :144 */
    this_omega2[i] = omega2 * (scaling * scaling);                           /* :146 */
    d0[i] = this_omega2[i] + s.noise[i];                                     /* :151 */
  }
  gpdla_oracle_log_mvnpdf_low_rank(s.flux, this_mu, this_M, d0, n, k, log_likelihood_no_dla); /* :149 */
  z_dla_range(prm, &s, z_qso, min_z_dla, max_z_dla);                          /* :159-160 */
  double *padded = padded_grid(prm, &s);                                     /* :168-176 */
  int64_t n_pad = s.n_u + 2 * prm->width;
  if (dump) {
    if (dump->n_kept) *dump->n_kept = n;
    if (dump->n_unmasked) *dump->n_unmasked = s.n_u;
    if (dump->this_mu) memcpy(dump->this_mu, this_mu, sizeof(double) * (size_t)n);
    if (dump->this_M) memcpy(dump->this_M, this_M, sizeof(double) * (size_t)n * k);
    if (dump->this_omega2) memcpy(dump->this_omega2, this_omega2, sizeof(double) * (size_t)n);
    if (dump->padded_wavelengths) memcpy(dump->padded_wavelengths, padded, sizeof(double) * (size_t)n_pad);
  }
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
#else
  (void)num_threads;
#endif
#pragma omp parallel
  {
    /* per-thread scratch, allocated once outside the sample loop (a MATLAB parfor worker likewise
     * keeps its heap); the operations per sample are unchanged */
    double *absorption = (double *)malloc(sizeof(double) * (size_t)s.n_u);
    double *raw = (double *)malloc(sizeof(double) * (size_t)n_pad);
    double *scratch = (double *)malloc(sizeof(double) * sample_scratch_doubles(n, k));
#pragma omp for schedule(static)
    for (int64_t i = 0; i < S; i++) { /* parfor, :185 */
      double z_dla = *min_z_dla + (*max_z_dla - *min_z_dla) * offset_samples[i]; /* :162-164 */
      if (dump && dump->sample_z_dlas) dump->sample_z_dlas[i] = z_dla;
      voigt_ws(padded, n_pad, z_dla, nhi_samples[i], prm->num_lines, absorption, raw); /* :187 */
      sample_ll[i] = sample_loglik(&s, k, absorption, this_mu, this_M, this_omega2, scratch);
    }
    free(absorption);
    free(raw);
    free(scratch);
  }
  *log_likelihood_dla = log_mean_exp(sample_ll, S, 0, NULL); /* :203-210 */
  free(padded);
  free(this_mu);
  free(this_M);
  free(this_omega2);
  free(d0);
  free_selection(&s);
  return 0;
}

/* Mean-flux suppression of one observed pixel, multi :267-285: exp(-Sum_l tau_l (1 + z_l)^beta) over the
 * first num_forest_lines Lyman lines, tau_l = prev_tau_0 f_l / f_Lya lambda_l / lambda_Lya (:269-273),
 * z_l = (lambda - lambda_l) / lambda_l (:184-186), lines other than Lyman-alpha counted only where
 * z_l <= z_qso (:279-282; nansum skips the NaN-flagged entries).  The reference has its own Python
 * restatement of exactly this, QSOLoader.total_scale_factor (CDDF_analysis/qso_loader.py:1777-1822):
 * tests/golden/mean_flux.npz holds its output and pins this function. */
double gpdla_oracle_mean_flux_suppression(double wavelength, double z_qso, double lya_wavelength,
                                          double prev_tau_0, double prev_beta, int num_forest_lines) {
  const double lya_f = oscillator_strengths[0]; /* lya_oscillator_strength, set_parameters_multi.m */
  double total = 0.0;
  for (int l = 0; l < num_forest_lines; l++) {
    double wl_l = transition_wavelengths[l] * 1e8;
    double z_l = (wavelength - wl_l) / wl_l;                              /* multi :184-186 */
    double this_tau_0 = prev_tau_0 * oscillator_strengths[l] / lya_f * wl_l / lya_wavelength;
    double od = this_tau_0 * pow(1 + z_l, prev_beta);                     /* multi :275-276 */
    if (l > 0 && z_l > z_qso) continue;                                   /* multi :279-282 */
    total += od;
  }
  return exp(-total);                                                     /* multi :285 */
}

/* ------------------------------------------------------------------------------------------
 * multi_dlas/process_qsos_multiple_dlas_meanflux.m:141-477 for one quasar ("multi :N" below).
 * ------------------------------------------------------------------------------------------ */
static int cmp_double(const void *a, const void *b) {
  double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

int gpdla_oracle_process_spectrum_multi(
    const gpdla_oracle_params *prm, const gpdla_oracle_model *mdl, const gpdla_oracle_multi *mul,
    int64_t S, const double *offset_samples, const double *nhi_samples, int64_t num_pixels,
    const double *wavelengths, const double *flux, const double *noise_variance,
    const uint8_t *pixel_mask, double z_qso, int num_threads, double *min_z_dla,
    double *max_z_dla, double *log_likelihood_no_dla, double *sample_ll /* [S x max_dlas] */,
    double *log_likelihoods_dla, double *sample_ll_lls, double *log_likelihood_lls,
    double *MAP_z_dlas, double *MAP_log_nhis, double *MAP_inds) {
  selection s;
  const int k = mdl->k, G = mdl->num_rest, max_dlas = mul->max_dlas, L = mul->num_forest_lines;
  const double lya_f = oscillator_strengths[0]; /* lya_oscillator_strength, set_parameters_multi.m */
  for (int64_t i = 0; i < S * max_dlas; i++) sample_ll[i] = NAN;       /* multi :146 */
  for (int m = 0; m < max_dlas; m++) log_likelihoods_dla[m] = NAN;     /* multi :117 */
  for (int64_t i = 0; i < S; i++) sample_ll_lls[i] = NAN;              /* multi :125 */
  for (int m = 0; m < max_dlas * max_dlas; m++) MAP_z_dlas[m] = MAP_log_nhis[m] = MAP_inds[m] = NAN;
  *log_likelihood_lls = NAN;
  *log_likelihood_no_dla = NAN;
  *min_z_dla = *max_z_dla = NAN;
  if (select_pixels(prm, num_pixels, wavelengths, flux, noise_variance, pixel_mask, z_qso, &s)) {
    free_selection(&s);
    return -1; /* multi :227-238 */
  }
  const int64_t n = s.n;
  double c_0 = exp(mdl->log_c_0), tau_0 = exp(mdl->log_tau_0), beta = exp(mdl->log_beta); /* :134-136 */
  double *this_mu = (double *)malloc(sizeof(double) * (size_t)n);
  double *this_M = (double *)malloc(sizeof(double) * (size_t)n * k);
  double *this_omega2 = (double *)malloc(sizeof(double) * (size_t)n);
  double *d0 = (double *)malloc(sizeof(double) * (size_t)n);
  for (int64_t i = 0; i < n; i++) {
    int i0;
    double t;
    bracket(mdl->rest_wavelengths, G, s.rest[i], &i0, &t);
    double mu_i = lerp(mdl->mu, i0, t);                                   /* multi :228 */
    double omega2 = exp(2 * lerp(mdl->log_omega, i0, t));                 /* multi :240-241 */
    double lya_z = (s.wavelengths[i] - prm->lya_wavelength) / prm->lya_wavelength; /* multi :175-177 */
    /* noise-model scaling with the Lyman series, multi :245-263 */
    double lya_optical_depth = tau_0 * pow(1 + lya_z, beta);
    for (int l = 1; l < L; l++) {
      double wl_1 = transition_wavelengths[0] * 1e8, wl_l = transition_wavelengths[l] * 1e8;
      double lyman_1pz = wl_1 * (1 + lya_z) / wl_l;                       /* multi :248-249 */
      double indicator = (lyman_1pz <= (1 + z_qso)) ? 1.0 : 0.0;          /* multi :252 */
      lyman_1pz = lyman_1pz * indicator;                                  /* multi :253 */
      double tau = tau_0 * wl_l * oscillator_strengths[l] / (wl_1 * oscillator_strengths[0]); /* :255 */
      lya_optical_depth = lya_optical_depth + tau * pow(lyman_1pz, beta); /* multi :258 */
    }
    double scaling = 1 - exp(-lya_optical_depth) + c_0;                   /* multi :261 */
    omega2 = omega2 * (scaling * scaling);                                /* multi :263 */
    /* mean-flux suppression, multi :267-285 */
    double lya_absorption = gpdla_oracle_mean_flux_suppression(s.wavelengths[i], z_qso, prm->lya_wavelength,
                                                               mul->prev_tau_0, mul->prev_beta, L);
    this_mu[i] = mu_i * lya_absorption;                                   /* multi :287 */
    for (int c = 0; c < k; c++)
      this_M[i + c * n] = lerp(mdl->M + (size_t)c * G, i0, t) * lya_absorption; /* multi :229,:288 */
    this_omega2[i] = omega2 * (lya_absorption * lya_absorption);          /* multi :293 */
    d0[i] = this_omega2[i] + s.noise[i];
  }
  gpdla_oracle_log_mvnpdf_low_rank(s.flux, this_mu, this_M, d0, n, k, log_likelihood_no_dla); /* :296 */
  z_dla_range(prm, &s, z_qso, min_z_dla, max_z_dla);                       /* multi :306-307 */
  double *padded = padded_grid(prm, &s);                                  /* multi :322-330 */
  int64_t n_pad = s.n_u + 2 * prm->width;
  double *sample_z = (double *)malloc(sizeof(double) * (size_t)S);
  for (int64_t i = 0; i < S; i++)
    sample_z[i] = *min_z_dla + (*max_z_dla - *min_z_dla) * offset_samples[i]; /* multi :309-311 */
  const double log_S = log((double)S);
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
#else
  (void)num_threads;
#endif
  for (int num_dlas = 1; num_dlas <= max_dlas; num_dlas++) { /* multi :337 */
    double *col = sample_ll + (size_t)(num_dlas - 1) * S;
#pragma omp parallel
    {
      double *absorption = (double *)malloc(sizeof(double) * (size_t)s.n_u);
      double *other = (double *)malloc(sizeof(double) * (size_t)s.n_u);
      double *raw = (double *)malloc(sizeof(double) * (size_t)n_pad);
      double *scratch = (double *)malloc(sizeof(double) * sample_scratch_doubles(n, k));
#pragma omp for schedule(static)
      for (int64_t i = 0; i < S; i++) { /* parfor, multi :340 */
        voigt_ws(padded, n_pad, sample_z[i], nhi_samples[i], prm->num_lines, absorption, raw);
        for (int j = 1; j <= num_dlas - 1; j++) { /* multi :346-351 */
          int64_t kk = (int64_t)mul->base_sample_inds[(j - 1) + (size_t)i * (max_dlas - 1)] - 1;
          voigt_ws(padded, n_pad, sample_z[kk], nhi_samples[kk], prm->num_lines, other, raw);
          for (int64_t p = 0; p < s.n_u; p++) absorption[p] = absorption[p] * other[p];
        }
        col[i] = sample_loglik(&s, k, absorption, this_mu, this_M, this_omega2, scratch) - log_S; /* :359-361 */
        if (num_dlas == 1) { /* multi :365-380 */
          voigt_ws(padded, n_pad, sample_z[i], mul->lls_nhi_samples[i], prm->num_lines, absorption, raw);
          sample_ll_lls[i] = sample_loglik(&s, k, absorption, this_mu, this_M, this_omega2, scratch) - log_S;
        }
      }
      free(absorption);
      free(other);
      free(raw);
      free(scratch);
    }
    if (num_dlas > 1) { /* multi :386-392: any(diff(sort(all_z_dlas)) < min_z_separation) */
      double zs[16];
      for (int64_t i = 0; i < S; i++) {
        zs[0] = sample_z[i];
        for (int j = 1; j <= num_dlas - 1; j++)
          zs[j] = sample_z[(int64_t)mul->base_sample_inds[(j - 1) + (size_t)i * (max_dlas - 1)] - 1];
        qsort(zs, (size_t)num_dlas, sizeof(double), cmp_double);
        for (int j = 1; j < num_dlas; j++)
          if (zs[j] - zs[j - 1] < mul->min_z_separation) col[i] = NAN;
      }
    }
    double mx;
    double lme = log_mean_exp(col, S, 1, &mx);                            /* multi :400-408 */
    log_likelihoods_dla[num_dlas - 1] = lme - log_S * (num_dlas - 1);     /* multi :407-409 */
    if (num_dlas == 1)                                                    /* multi :416-426 */
      *log_likelihood_lls = log_mean_exp(sample_ll_lls, S, 1, NULL) - log_S * (num_dlas - 1);
    /* MAP bookkeeping, multi :439-445: first index of the nanmax */
    int64_t maxidx = -1;
    for (int64_t i = 0; i < S; i++)
      if (!isnan(col[i]) && (maxidx < 0 || col[i] > col[maxidx])) maxidx = i;
    if (maxidx < 0) maxidx = 0; /* MATLAB nanmax of all-NaN returns index 1 */
    for (int j = 0; j < num_dlas; j++) {
      int64_t idx = (j == 0) ? maxidx
                             : (int64_t)mul->base_sample_inds[(j - 1) + (size_t)maxidx * (max_dlas - 1)] - 1;
      size_t at = (size_t)(num_dlas - 1) + (size_t)j * max_dlas; /* (model, slot) column-major */
      MAP_inds[at] = (double)(idx + 1);
      MAP_z_dlas[at] = sample_z[idx];
      MAP_log_nhis[at] = mul->log_nhi_samples[idx];
    }
    if (num_dlas == max_dlas) break;                                      /* multi :452-454 */
    if (isnan(log_likelihoods_dla[num_dlas - 1])) break;                  /* multi :460-464 */
    /* multi :467-472 draws base_sample_inds(num_dlas,:) with MATLAB's RNG; here it is an input. */
  }
  free(sample_z);
  free(padded);
  free(this_mu);
  free(this_M);
  free(this_omega2);
  free(d0);
  free_selection(&s);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Training objective: spectrum_loss.m:14-76 and objective.m:12-75,
 * as written.  M is n x k column-major; dM likewise.
 * ------------------------------------------------------------------------------------------ */
/* The two spectrum losses differ in the optical depth only: spectrum_loss.m:22 (Lyman alpha) or
 * multi_dlas/spectrum_loss_lyseries.m:22-39 (Lyman alpha plus num_forest_lines - 1 higher lines, each
 * switched off where its redshift would exceed the quasar's).  Lines :23-75 of the former are lines
 * :40-92 of the latter, word for word (line cites below: spectrum_loss.m). */
static int spectrum_loss_body(const double *y, const double *lya_1pz, const double *noise_variance,
                              const double *M, const double *omega2, int64_t n, int k, double c_0,
                              double tau_0, double beta, int num_forest_lines,
                              const double *all_transition_wavelengths, const double *all_oscillator_strengths,
                              double zqso_1pz, double *nlog_p, double *dM, double *dlog_omega,
                              double *dlog_c_0, double *dlog_tau_0, double *dlog_beta) {
  const double log_2pi = 1.83787706640934534; /* :17 */
  size_t nn = (size_t)n;
  double *lya_optical_depth = (double *)malloc(sizeof(double) * nn);
  double *lya_absorption = (double *)malloc(sizeof(double) * nn);
  double *scaling_factor = (double *)malloc(sizeof(double) * nn);
  double *absorption_noise = (double *)malloc(sizeof(double) * nn);
  double *d = (double *)malloc(sizeof(double) * nn);
  double *d_inv = (double *)malloc(sizeof(double) * nn);
  double *D_inv_y = (double *)malloc(sizeof(double) * nn);
  double *D_inv_M = (double *)malloc(sizeof(double) * nn * k);
  double *B = (double *)calloc((size_t)k * k, sizeof(double));
  double *R = (double *)calloc((size_t)k * k, sizeof(double));
  double *C = (double *)malloc(sizeof(double) * nn * k); /* k x n, column-major */
  double *Cy = (double *)calloc((size_t)k, sizeof(double));
  double *CM = (double *)calloc((size_t)k * k, sizeof(double)); /* C * M, k x k */
  double *K_inv_y = (double *)malloc(sizeof(double) * nn);
  double *K_inv_M = (double *)malloc(sizeof(double) * nn * k);
  double *diag_K_inv = (double *)malloc(sizeof(double) * nn);
  double *yM = (double *)calloc((size_t)k, sizeof(double));
  int rc = 0;
  for (int64_t i = 0; i < n; i++) {
    lya_optical_depth[i] = tau_0 * pow(lya_1pz[i], beta);                 /* :22 */
    for (int l = 1; l < num_forest_lines; l++) {                          /* spectrum_loss_lyseries.m:27-38 */
      double lyman_1pz = all_transition_wavelengths[0] * lya_1pz[i] / all_transition_wavelengths[l];
      const double indicator = lyman_1pz <= zqso_1pz ? 1.0 : 0.0;
      lyman_1pz = lyman_1pz * indicator;
      const double tau = tau_0 * all_transition_wavelengths[l] * all_oscillator_strengths[l] /
                         (all_transition_wavelengths[0] * all_oscillator_strengths[0]);
      lya_optical_depth[i] = lya_optical_depth[i] + tau * pow(lyman_1pz, beta);
    }
    lya_absorption[i] = exp(-lya_optical_depth[i]);                       /* :23 */
    scaling_factor[i] = 1 - lya_absorption[i] + c_0;                      /* :26 */
    absorption_noise[i] = omega2[i] * (scaling_factor[i] * scaling_factor[i]); /* :27 */
    d[i] = noise_variance[i] + absorption_noise[i];                       /* :29 */
    d_inv[i] = 1.0 / d[i];                                                /* :31 */
    D_inv_y[i] = d_inv[i] * y[i];                                         /* :32 */
    for (int a = 0; a < k; a++) D_inv_M[i + a * nn] = d_inv[i] * M[i + a * nn]; /* :33 */
  }
  for (int a = 0; a < k; a++) /* :40 */
    for (int b = 0; b < k; b++) {
      double acc = 0.0;
      for (int64_t i = 0; i < n; i++) acc += M[i + a * nn] * D_inv_M[i + b * nn];
      B[a + b * k] = acc;
    }
  for (int a = 0; a < k; a++) B[a + a * k] += 1.0; /* :41 */
  for (int j = 0; j < k && !rc; j++) {             /* :42 chol, upper */
    double s = B[j + j * k];
    for (int m = 0; m < j; m++) s -= R[m + j * k] * R[m + j * k];
    if (!(s > 0.0)) { rc = -1; break; }
    double rjj = sqrt(s);
    R[j + j * k] = rjj;
    for (int i = j + 1; i < k; i++) {
      double t = B[j + i * k];
      for (int m = 0; m < j; m++) t -= R[m + j * k] * R[m + i * k];
      R[j + i * k] = t / rjj;
    }
  }
  if (rc) { *nlog_p = NAN; goto done; }
  for (int64_t i = 0; i < n; i++) { /* :44  C = L \ (L' \ D_inv_M') */
    double *c = C + (size_t)i * k;
    for (int a = 0; a < k; a++) {
      double t = D_inv_M[i + a * nn];
      for (int m = 0; m < a; m++) t -= R[m + a * k] * c[m];
      c[a] = t / R[a + a * k];
    }
    for (int a = k - 1; a >= 0; a--) {
      double t = c[a];
      for (int m = a + 1; m < k; m++) t -= R[a + m * k] * c[m];
      c[a] = t / R[a + a * k];
    }
  }
  for (int64_t i = 0; i < n; i++)
    for (int a = 0; a < k; a++) Cy[a] += C[(size_t)i * k + a] * y[i];
  double quad = 0.0, log_det_K = 0.0;
  for (int64_t i = 0; i < n; i++) { /* :46 */
    double t = 0.0;
    for (int a = 0; a < k; a++) t += D_inv_M[i + a * nn] * Cy[a];
    K_inv_y[i] = D_inv_y[i] - t;
    quad += y[i] * K_inv_y[i];
    log_det_K += log(d[i]);
  }
  {
    double sld = 0.0;
    for (int a = 0; a < k; a++) sld += log(R[a + a * k]);
    log_det_K += 2 * sld; /* :48 */
  }
  *nlog_p = 0.5 * (quad + log_det_K + (double)n * log_2pi); /* :52 */
  /* :55  K_inv_M = D_inv_M - D_inv_M * (C * M) */
  for (int a = 0; a < k; a++)
    for (int b = 0; b < k; b++) {
      double acc = 0.0;
      for (int64_t i = 0; i < n; i++) acc += C[(size_t)i * k + a] * M[i + b * nn];
      CM[a + b * k] = acc;
    }
  for (int64_t i = 0; i < n; i++)
    for (int b = 0; b < k; b++) {
      double t = 0.0;
      for (int a = 0; a < k; a++) t += D_inv_M[i + a * nn] * CM[a + b * k];
      K_inv_M[i + b * nn] = D_inv_M[i + b * nn] - t;
    }
  for (int64_t i = 0; i < n; i++)
    for (int b = 0; b < k; b++) yM[b] += K_inv_y[i] * M[i + b * nn];
  for (int64_t i = 0; i < n; i++) /* :56 */
    for (int b = 0; b < k; b++) dM[i + b * nn] = -(K_inv_y[i] * yM[b] - K_inv_M[i + b * nn]);
  *dlog_c_0 = *dlog_tau_0 = *dlog_beta = 0.0;
  double c1 = 0, c2 = 0, t1 = 0, t2 = 0, b1 = 0, b2 = 0;
  for (int64_t i = 0; i < n; i++) {
    double s = 0.0;
    for (int a = 0; a < k; a++) s += C[(size_t)i * k + a] * D_inv_M[i + a * nn];
    diag_K_inv[i] = d_inv[i] - s;                                                     /* :59 */
    dlog_omega[i] = -(absorption_noise[i] * (K_inv_y[i] * K_inv_y[i] - diag_K_inv[i])); /* :62 */
    double da = c_0 * omega2[i] * scaling_factor[i];                                  /* :65 */
    c1 += (K_inv_y[i] * da) * K_inv_y[i];
    c2 += diag_K_inv[i] * da;
    da = omega2[i] * scaling_factor[i] * lya_optical_depth[i] * lya_absorption[i];    /* :69 */
    t1 += (K_inv_y[i] * da) * K_inv_y[i];
    t2 += diag_K_inv[i] * da;
    da = da * log(lya_1pz[i]) * beta;                                                 /* :73 */
    b1 += (K_inv_y[i] * da) * K_inv_y[i];
    b2 += diag_K_inv[i] * da;
  }
  *dlog_c_0 = -c1 + c2;   /* :66 */
  *dlog_tau_0 = -t1 + t2; /* :70 */
  *dlog_beta = -b1 + b2;  /* :74 */
done:
  free(lya_optical_depth); free(lya_absorption); free(scaling_factor); free(absorption_noise);
  free(d); free(d_inv); free(D_inv_y); free(D_inv_M); free(B); free(R); free(C); free(Cy); free(CM);
  free(K_inv_y); free(K_inv_M); free(diag_K_inv); free(yM);
  return rc;
}

int gpdla_oracle_spectrum_loss(const double *y, const double *lya_1pz, const double *noise_variance,
                               const double *M, const double *omega2, int64_t n, int k, double c_0,
                               double tau_0, double beta, double *nlog_p, double *dM,
                               double *dlog_omega, double *dlog_c_0, double *dlog_tau_0,
                               double *dlog_beta) {
  return spectrum_loss_body(y, lya_1pz, noise_variance, M, omega2, n, k, c_0, tau_0, beta, 1, NULL, NULL, 0.0,
                            nlog_p, dM, dlog_omega, dlog_c_0, dlog_tau_0, dlog_beta);
}

/* multi_dlas/spectrum_loss_lyseries.m:14-93 */
int gpdla_oracle_spectrum_loss_lyseries(const double *y, const double *lya_1pz, const double *noise_variance,
                                        const double *M, const double *omega2, int64_t n, int k, double c_0,
                                        double tau_0, double beta, int num_forest_lines,
                                        const double *all_transition_wavelengths,
                                        const double *all_oscillator_strengths, double zqso_1pz,
                                        double *nlog_p, double *dM, double *dlog_omega, double *dlog_c_0,
                                        double *dlog_tau_0, double *dlog_beta) {
  return spectrum_loss_body(y, lya_1pz, noise_variance, M, omega2, n, k, c_0, tau_0, beta, num_forest_lines,
                            all_transition_wavelengths, all_oscillator_strengths, zqso_1pz, nlog_p, dM,
                            dlog_omega, dlog_c_0, dlog_tau_0, dlog_beta);
}

/* objective.m:12-75.  The three data matrices are [num_quasars x num_pixels] column-major with NaN
 * marking missing pixels (:42); x = [vec M; log omega; log c0; log tau0; log beta] (:5). */
static int objective_body(const double *x, int64_t num_quasars, int64_t num_pixels, int k,
                          const double *centered_rest_fluxes, const double *lya_1pzs,
                          const double *rest_noise_variances, int num_forest_lines,
                          const double *all_transition_wavelengths, const double *all_oscillator_strengths,
                          int num_threads, double *f, double *g) {
  const size_t G = (size_t)num_pixels;
  const double *M = x, *log_omega = x + G * k;
  const double log_c_0 = x[G * (k + 1)], log_tau_0 = x[G * (k + 1) + 1], log_beta = x[G * (k + 1) + 2];
  const double c_0 = exp(log_c_0), tau_0 = exp(log_tau_0), beta = exp(log_beta); /* :30-32 */
  double *omega2 = (double *)malloc(sizeof(double) * G);
  for (size_t p = 0; p < G; p++) omega2[p] = exp(2 * log_omega[p]); /* :29 */
  double fsum = 0.0, dc = 0.0, dt = 0.0, db = 0.0;
  memset(g, 0, sizeof(double) * (G * (k + 1) + 3));
  int fail = 0;
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
#else
  (void)num_threads;
#endif
#pragma omp parallel
  {
    double *y = (double *)malloc(sizeof(double) * G), *l1 = (double *)malloc(sizeof(double) * G);
    double *nv = (double *)malloc(sizeof(double) * G), *om = (double *)malloc(sizeof(double) * G);
    double *Mi = (double *)malloc(sizeof(double) * G * k), *dM = (double *)malloc(sizeof(double) * G * k);
    double *dlo = (double *)malloc(sizeof(double) * G);
    int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * G);
    double *gl = (double *)calloc(G * (k + 1) + 3, sizeof(double));
    double fl = 0.0;
#pragma omp for schedule(dynamic, 4)
    for (int64_t i = 0; i < num_quasars; i++) { /* :41 */
      int64_t n = 0;
      for (size_t p = 0; p < G; p++) {
        double v = centered_rest_fluxes[i + p * num_quasars];
        if (isnan(v)) continue; /* :42 */
        idx[n] = (int64_t)p;
        y[n] = v;
        l1[n] = lya_1pzs[i + p * num_quasars];
        nv[n] = rest_noise_variances[i + p * num_quasars];
        om[n] = omega2[p];
        n++;
      }
      if (n == 0) continue;
      for (int a = 0; a < k; a++)
        for (int64_t j = 0; j < n; j++) Mi[j + a * n] = M[idx[j] + a * G];
      double tf, tc, tt, tb;
      /* objective_lyseries.m:46: zqso + 1 is the quasar's last lya_1pz */
      const double zqso_1pz = lya_1pzs[i + (G - 1) * num_quasars];
      if (spectrum_loss_body(y, l1, nv, Mi, om, n, k, c_0, tau_0, beta, num_forest_lines,
                             all_transition_wavelengths, all_oscillator_strengths, zqso_1pz, &tf, dM, dlo,
                             &tc, &tt, &tb)) {
#pragma omp atomic write
        fail = 1;
        continue;
      }
      fl += tf; /* :50-55 */
      for (int a = 0; a < k; a++)
        for (int64_t j = 0; j < n; j++) gl[idx[j] + a * G] += dM[j + a * n];
      for (int64_t j = 0; j < n; j++) gl[G * k + idx[j]] += dlo[j];
      gl[G * (k + 1)] += tc;
      gl[G * (k + 1) + 1] += tt;
      gl[G * (k + 1) + 2] += tb;
    }
#pragma omp critical
    {
      fsum += fl;
      for (size_t e = 0; e < G * (k + 1); e++) g[e] += gl[e];
      dc += gl[G * (k + 1)];
      dt += gl[G * (k + 1) + 1];
      db += gl[G * (k + 1) + 2];
    }
    free(y); free(l1); free(nv); free(om); free(Mi); free(dM); free(dlo); free(idx); free(gl);
  }
  /* priors of Kim et al. (2007), :59-71 (added to the gradient only, as the reference does) */
  const double tau_0_mu = 0.0023, tau_0_sigma = 0.0007, beta_mu = 3.65, beta_sigma = 0.21;
  dt += tau_0 * (tau_0 - tau_0_mu) / (tau_0_sigma * tau_0_sigma);
  db += beta * (beta - beta_mu) / (beta_sigma * beta_sigma);
  g[G * (k + 1)] = dc;
  g[G * (k + 1) + 1] = dt;
  g[G * (k + 1) + 2] = db;
  *f = fsum;
  free(omega2);
  return fail ? -1 : 0;
}

int gpdla_oracle_objective(const double *x, int64_t num_quasars, int64_t num_pixels, int k,
                           const double *centered_rest_fluxes, const double *lya_1pzs,
                           const double *rest_noise_variances, int num_threads, double *f,
                           double *g) {
  return objective_body(x, num_quasars, num_pixels, k, centered_rest_fluxes, lya_1pzs, rest_noise_variances, 1,
                        NULL, NULL, num_threads, f, g);
}

/* multi_dlas/objective_lyseries.m:12-78 (objective.m with spectrum_loss_lyseries in place of spectrum_loss) */
int gpdla_oracle_objective_lyseries(const double *x, int64_t num_quasars, int64_t num_pixels, int k,
                                    const double *centered_rest_fluxes, const double *lya_1pzs,
                                    const double *rest_noise_variances, int num_forest_lines,
                                    const double *all_transition_wavelengths,
                                    const double *all_oscillator_strengths, int num_threads, double *f,
                                    double *g) {
  if (num_forest_lines < 1 || !all_transition_wavelengths || !all_oscillator_strengths) return -2;
  return objective_body(x, num_quasars, num_pixels, k, centered_rest_fluxes, lya_1pzs, rest_noise_variances,
                        num_forest_lines, all_transition_wavelengths, all_oscillator_strengths, num_threads, f, g);
}
