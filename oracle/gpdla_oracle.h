/* gpdla_oracle.h -- CPU ORACLE for the gp_dla_detection hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (gp_dla_detection_amd/, the C-ABI in
 * include/gpdla.h) may include, link or call this.  Allowed users: tests/, __graft_entry__.smoke()
 * and the cpu_baseline leg of bench.py -- as the checker / the reported CPU baseline, never as the
 * thing shipped.
 *
 * What it is: a plain-C, fp64, as-written restatement of the reference's
 *   voigt.c:253-304                 (MEX gateway: Lyman-series Voigt profile + 7-tap convolution)
 *   log_mvnpdf_low_rank.m:5-34      (Woodbury low-rank Gaussian log-pdf, incl. the k x n matrix C)
 *   process_qsos.m:96-213           (per-spectrum driver: selection, interpolation, sweep, evidence)
 *   multi_dlas/process_qsos_multiple_dlas_meanflux.m:141-477 (multi-DLA / LLS / mean-flux driver)
 *   spectrum_loss.m:14-76, objective.m:12-75 (training objective and gradient, "next" row N3)
 *   multi_dlas/spectrum_loss_lyseries.m:14-93, objective_lyseries.m:12-78 (the same for the mean-flux model)
 *
 * Third-party arithmetic: voigt.c:288 calls libcerf's  double voigt(double x, double sigma,
 * double gamma)  (libcerf is NOT in /root/reference and its version is not pinned anywhere,
 * README.md:210-218).  libcerf documents voigt(x,sigma,gamma) = Re w((x + i gamma)/(sqrt2 sigma))
 * / (sqrt(2 pi) sigma); that published definition is restated here with this file's own Faddeeva
 * routine.
 *
 * PARITY PIN.  Voigt half: PINNED against golden vectors produced by importing the reference's own
 * CDDF_analysis/voigt.py (scipy wofz) in the build container (tests/golden/make_golden.py).
 * MATLAB half (log_mvnpdf_low_rank.m, process_qsos.m, the multi-DLA driver): PARITY UNPINNED -- no
 * MATLAB/Octave exists in this pipeline and the reference ships no tests or fixtures, so no
 * reference-produced number exists for it; it is a line-by-line restatement cross-checked against an
 * independent dense evaluation and an independent NumPy restatement (tests/test_oracle_driver.py).
 */
#ifndef GPDLA_ORACLE_H
#define GPDLA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Re w(x + i y), y > 0 small (Lyman-series damping parameters are <= 4.8e-4). */
double gpdla_oracle_faddeeva_re(double x, double y);

/* libcerf voigt(x, sigma, gamma) as documented (call site voigt.c:288). */
double gpdla_oracle_voigt_line(double x, double sigma, double gamma);

/* voigt.c:253-304.  profile has num_points - 6 entries.  Returns 0. */
int gpdla_oracle_voigt(const double *lambdas, int64_t num_points, double z, double N,
                       int num_lines, double *profile);

/* Same without the instrument convolution (what CDDF_analysis/voigt.py:230-275 returns). */
int gpdla_oracle_voigt_raw(const double *lambdas, int64_t num_points, double z, double N,
                           int num_lines, double *raw_profile);

/* log_mvnpdf_low_rank.m:5-34.  M is column-major n x k.  Returns 0, or -1 if B is not PD
 * (MATLAB's chol would throw, :24); *log_p is NaN then. */
int gpdla_oracle_log_mvnpdf_low_rank(const double *y, const double *mu, const double *M,
                                     const double *d, int64_t n, int k, double *log_p);
/* doubles of scratch one evaluation uses (the drivers allocate it once per thread) */
size_t gpdla_oracle_lowrank_scratch_doubles(int64_t n, int k);

/* Constants of set_parameters.m that the driver reads (process_qsos.m:104-105,118,159-176,188). */
typedef struct {
  double min_lambda;      /* set_parameters.m:33 */
  double max_lambda;      /* set_parameters.m:34 */
  double lya_wavelength;  /* set_parameters.m:5  */
  double lyman_limit;     /* set_parameters.m:7  */
  double pixel_spacing;   /* set_parameters.m:60 */
  double max_z_cut;       /* set_parameters.m:65 */
  double min_z_cut;       /* set_parameters.m:69 */
  int32_t width;          /* set_parameters.m:59 */
  int32_t num_lines;      /* set_parameters.m:63 */
} gpdla_oracle_params;

/* Learned GP model, learn_qso_model.m:113-123 (loaded at process_qsos.m:30-35). */
typedef struct {
  int32_t num_rest;               /* G */
  int32_t k;
  const double *rest_wavelengths; /* [G] */
  const double *mu;               /* [G] */
  const double *M;                /* [G x k] column-major */
  const double *log_omega;        /* [G] */
  double log_c_0, log_tau_0, log_beta;
} gpdla_oracle_model;

/* Optional dumps of the intermediates of process_qsos.m:138-176 (any pointer may be NULL). */
typedef struct {
  int64_t *n_kept;            /* number of pixels after :110-115 */
  int64_t *n_unmasked;        /* numel(this_unmasked_wavelengths), :108 */
  double *this_mu;            /* [n_kept] :138 */
  double *this_M;             /* [n_kept x k] column-major :139 */
  double *this_omega2;        /* [n_kept] after :146 */
  double *padded_wavelengths; /* [n_unmasked + 2*width] :168-176 */
  double *sample_z_dlas;      /* [S] :162-164 */
} gpdla_oracle_dump;

/* process_qsos.m:96-213 for one quasar (the prior of :122-131 is host logic, not here).
 * Outputs: min_z_dla, max_z_dla (:159-160), log_likelihood_no_dla (:149-151),
 * sample_log_likelihoods_dla[S] (:185-199), log_likelihood_dla (:203-210).
 * num_threads mirrors the parfor worker count (:185).  Returns 0, or -1 when no pixel survives. */
int gpdla_oracle_process_spectrum(const gpdla_oracle_params *prm, const gpdla_oracle_model *mdl,
                                  int64_t num_samples, const double *offset_samples,
                                  const double *nhi_samples, int64_t num_pixels,
                                  const double *wavelengths, const double *flux,
                                  const double *noise_variance, const uint8_t *pixel_mask,
                                  double z_qso, int num_threads, double *min_z_dla,
                                  double *max_z_dla, double *log_likelihood_no_dla,
                                  double *sample_log_likelihoods_dla, double *log_likelihood_dla,
                                  gpdla_oracle_dump *dump);

/* Extra inputs of the multi-DLA driver. */
typedef struct {
  int32_t max_dlas;            /* multi :32 */
  int32_t num_forest_lines;    /* set_parameters_multi.m (31) */
  double min_z_separation;     /* multi :33 */
  double prev_tau_0;           /* multi :36 */
  double prev_beta;            /* multi :37 */
  const double *lls_nhi_samples;   /* [S] set_lls_parameters.m:59-63 */
  /* base_sample_inds[(max_dlas-1) x S], 1-BASED like MATLAB (multi :313, :471-472); supplied as an
   * input because MATLAB's rng('default')+randsample stream (:143, :471) is not reproducible. */
  const uint32_t *base_sample_inds;
  const double *log_nhi_samples;   /* [S] for MAP bookkeeping :389,:445 */
} gpdla_oracle_multi;

/* Mean-flux suppression factor of one observed pixel (multi :267-285); pinned by the reference's own
 * QSOLoader.total_scale_factor through tests/golden/mean_flux.npz. */
double gpdla_oracle_mean_flux_suppression(double wavelength, double z_qso, double lya_wavelength,
                                          double prev_tau_0, double prev_beta, int num_forest_lines);

/* multi_dlas/process_qsos_multiple_dlas_meanflux.m:141-477 for one quasar.
 * sample_log_likelihoods_dla is [S x max_dlas] column-major (this_sample_log_likelihoods_dla, :146),
 * log_likelihoods_dla [max_dlas], sample_log_likelihoods_lls [S], MAP_* [max_dlas x max_dlas]
 * column-major (model index first) with NaN for unused slots, MAP_inds 1-based as doubles (:131).
 * Models after an early NaN exit (:460-464) stay NaN.  Returns 0, or -1 for an empty spectrum
 * (all_exceptions, :230-234). */
int gpdla_oracle_process_spectrum_multi(
    const gpdla_oracle_params *prm, const gpdla_oracle_model *mdl, const gpdla_oracle_multi *mul,
    int64_t num_samples, const double *offset_samples, const double *nhi_samples,
    int64_t num_pixels, const double *wavelengths, const double *flux,
    const double *noise_variance, const uint8_t *pixel_mask, double z_qso, int num_threads,
    double *min_z_dla, double *max_z_dla, double *log_likelihood_no_dla,
    double *sample_log_likelihoods_dla, double *log_likelihoods_dla,
    double *sample_log_likelihoods_lls, double *log_likelihood_lls, double *MAP_z_dlas,
    double *MAP_log_nhis, double *MAP_inds);

/* spectrum_loss.m:14-76 (row N3 of SURVEY.md section 8f).  M, dM: n x k column-major. */
int gpdla_oracle_spectrum_loss(const double *y, const double *lya_1pz, const double *noise_variance,
                               const double *M, const double *omega2, int64_t n, int k, double c_0,
                               double tau_0, double beta, double *nlog_p, double *dM,
                               double *dlog_omega, double *dlog_c_0, double *dlog_tau_0,
                               double *dlog_beta);

/* objective.m:12-75.  Data matrices [num_quasars x num_pixels] column-major, NaN = missing;
 * x and g have num_pixels*(k+1) + 3 entries ([vec M; log omega; log c0; log tau0; log beta]). */
int gpdla_oracle_objective(const double *x, int64_t num_quasars, int64_t num_pixels, int k,
                           const double *centered_rest_fluxes, const double *lya_1pzs,
                           const double *rest_noise_variances, int num_threads, double *f,
                           double *g);

/* multi_dlas/spectrum_loss_lyseries.m:14-93 and multi_dlas/objective_lyseries.m:12-78: the training
 * objective of the mean-flux (multi-DLA) model -- the optical depth sums num_forest_lines Lyman
 * lines, each switched off beyond the quasar's own redshift (zqso_1pz = the quasar's last lya_1pz). */
int gpdla_oracle_spectrum_loss_lyseries(const double *y, const double *lya_1pz, const double *noise_variance,
                                        const double *M, const double *omega2, int64_t n, int k, double c_0,
                                        double tau_0, double beta, int num_forest_lines,
                                        const double *all_transition_wavelengths,
                                        const double *all_oscillator_strengths, double zqso_1pz,
                                        double *nlog_p, double *dM, double *dlog_omega, double *dlog_c_0,
                                        double *dlog_tau_0, double *dlog_beta);
int gpdla_oracle_objective_lyseries(const double *x, int64_t num_quasars, int64_t num_pixels, int k,
                                    const double *centered_rest_fluxes, const double *lya_1pzs,
                                    const double *rest_noise_variances, int num_forest_lines,
                                    const double *all_transition_wavelengths,
                                    const double *all_oscillator_strengths, int num_threads, double *f,
                                    double *g);

#ifdef __cplusplus
}
#endif
#endif
