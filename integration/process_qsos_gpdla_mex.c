/* process_qsos_gpdla_mex.c -- reference-side MEX gateway for the hot loop of process_qsos.m.
 *
 * Replaces process_qsos.m:88-233 (the serial loop over quasars with its parfor over the DLA samples,
 * and the posterior softmax) by ONE call into libgpdla.so.  The script keeps its loading code
 * (:4-61) and its save (:236-250); between them it calls
 *
 *     res = process_qsos_gpdla(model, samples, spectra, prior [, params]);
 *
 *   model    struct: rest_wavelengths, mu, M, log_omega, log_c_0, log_tau_0, log_beta   (the variables
 *            process_qsos.m:30-35 loads from learned_qso_model_*.mat)
 *   samples  struct: offset_samples, log_nhi_samples, nhi_samples                       (:38-40)
 *   spectra  struct: wavelengths, flux, noise_variance, pixel_mask -- the cell arrays all_wavelengths,
 *            all_flux, all_noise_variance, all_pixel_mask after the test_ind subset of :56-61 -- and
 *            z_qsos, a vector with one entry per cell
 *   prior    struct: z_qsos, dla_ind -- the training release's quasars after the Lyman-limit filter
 *            of :15-25 (the struct the script calls `prior`)
 *   params   optional struct; any of: prior_z_qso_increase, num_lines, min_lambda, max_lambda,
 *            lya_wavelength, lyman_limit, pixel_spacing, max_z_cut, min_z_cut, device_id
 *            (set_parameters.m:5-73; defaults are the reference's)
 *   res      struct with the variables the script saves (:239-244), shaped as the script shapes
 *            them (:74-82): min_z_dlas, max_z_dlas, log_priors_no_dla, log_priors_dla,
 *            log_likelihoods_no_dla, log_likelihoods_dla, log_posteriors_no_dla, log_posteriors_dla,
 *            p_no_dlas, p_dlas [nq x 1]; sample_log_likelihoods_dla [nq x S]; model_posteriors [nq x 2].
 *            A quasar with no pixel in range keeps NaN, as the NaN pre-fill of :74-82 leaves it.
 *
 * Build (MATLAB):  mex process_qsos_gpdla_mex.c -output process_qsos_gpdla -I<repo>/include ...
 *                      -L<repo>/gp_dla_detection_amd/csrc -lgpdla -lamdhip64
 * Only documented mex.h / matrix.h calls are used; tests/test_integration.py compiles this file
 * (syntax and types) against declarations of exactly those calls, because no MATLAB exists in the
 * build image.  The gateway owns no state: every buffer it allocates is freed before it returns.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "mex.h"

#include "gpdla.h"

#define AT_LEAST_ONE(n) ((n) ? (n) : 1) /* mxMalloc(0) may return NULL */

static const mxArray *need_field(const mxArray *s, const char *arg, const char *name) {
  const mxArray *f;
  if (!mxIsStruct(s)) mexErrMsgIdAndTxt("gpdla:arg", "%s must be a struct", arg);
  f = mxGetField(s, 0, name);
  if (f == NULL) mexErrMsgIdAndTxt("gpdla:arg", "%s.%s is missing", arg, name);
  return f;
}

static const double *need_doubles(const mxArray *s, const char *arg, const char *name, size_t *count) {
  const mxArray *f = need_field(s, arg, name);
  if (!mxIsDouble(f) || mxIsComplex(f)) mexErrMsgIdAndTxt("gpdla:arg", "%s.%s must be a real double array", arg, name);
  if (count) *count = mxGetNumberOfElements(f);
  return mxGetPr(f);
}

static double need_scalar(const mxArray *s, const char *arg, const char *name) {
  size_t n;
  const double *p = need_doubles(s, arg, name, &n);
  if (n != 1) mexErrMsgIdAndTxt("gpdla:arg", "%s.%s must be a scalar", arg, name);
  return p[0];
}

static double optional_scalar(const mxArray *params, const char *name, double fallback) {
  const mxArray *f;
  if (params == NULL || !mxIsStruct(params)) return fallback;
  f = mxGetField(params, 0, name);
  if (f == NULL || mxGetNumberOfElements(f) != 1) return fallback;
  return mxGetScalar(f);
}

static mxArray *column(size_t rows, size_t cols) {
  mxArray *a = mxCreateDoubleMatrix(rows, cols, mxREAL);
  double *p = mxGetPr(a);
  size_t i;
  for (i = 0; i < rows * cols; ++i) p[i] = mxGetNaN(); /* the NaN pre-fill of process_qsos.m:74-82 */
  return a;
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  static const char *fields[] = {"min_z_dlas", "max_z_dlas", "log_priors_no_dla", "log_priors_dla",
                                 "log_likelihoods_no_dla", "sample_log_likelihoods_dla", "log_likelihoods_dla",
                                 "log_posteriors_no_dla", "log_posteriors_dla", "model_posteriors", "p_no_dlas",
                                 "p_dlas"}; /* process_qsos.m:239-244 */
  enum { F_MIN_Z, F_MAX_Z, F_LP_NO, F_LP_DLA, F_LL_NO, F_SLL, F_LL_DLA, F_LPOST_NO, F_LPOST_DLA, F_POST, F_P_NO,
         F_P_DLA, NUM_FIELDS };
  const mxArray *m_model, *m_samples, *m_spectra, *m_prior, *m_params;
  const mxArray *c_wl, *c_flux, *c_nv, *c_mask;
  gpdla_model model;
  gpdla_samples samples;
  gpdla_spectra_cells spectra;
  gpdla_config cfg;
  gpdla_results res;
  mxArray *out[NUM_FIELDS];
  size_t nq, S, G, n_prior, n_prior_flags, count, q, i, total;
  int64_t *npix;
  const double **p_wl, **p_flux, **p_nv;
  const uint8_t **p_mask;
  double *lp_no, *lp_dla, *sll, *post;
  uint8_t *mask;
  const double *prior_z;
  const mxArray *prior_flags;
  double z_increase;
  int device_id, rc, f;

  if (nrhs < 4 || nrhs > 5)
    mexErrMsgIdAndTxt("gpdla:arg", "usage: res = process_qsos_gpdla(model, samples, spectra, prior [, params])");
  if (nlhs > 1) mexErrMsgIdAndTxt("gpdla:arg", "one output");
  m_model = prhs[0];
  m_samples = prhs[1];
  m_spectra = prhs[2];
  m_prior = prhs[3];
  m_params = nrhs > 4 ? prhs[4] : NULL;

  /* ---- the learned model, process_qsos.m:30-35 ---- */
  memset(&model, 0, sizeof model);
  model.rest_wavelengths = need_doubles(m_model, "model", "rest_wavelengths", &G);
  model.mu = need_doubles(m_model, "model", "mu", &count);
  if (count != G) mexErrMsgIdAndTxt("gpdla:arg", "model.mu must have numel(rest_wavelengths) entries");
  model.log_omega = need_doubles(m_model, "model", "log_omega", &count);
  if (count != G) mexErrMsgIdAndTxt("gpdla:arg", "model.log_omega must have numel(rest_wavelengths) entries");
  model.M = need_doubles(m_model, "model", "M", &count); /* [G x k], column-major as the C-ABI takes it */
  if (mxGetM(need_field(m_model, "model", "M")) != G)
    mexErrMsgIdAndTxt("gpdla:arg", "model.M must have numel(rest_wavelengths) rows");
  model.num_rest_pixels = (int32_t)G;
  model.k = (int32_t)mxGetN(need_field(m_model, "model", "M"));
  model.log_c_0 = need_scalar(m_model, "model", "log_c_0");
  model.log_tau_0 = need_scalar(m_model, "model", "log_tau_0");
  model.log_beta = need_scalar(m_model, "model", "log_beta");

  /* ---- the DLA parameter samples, :38-40 ---- */
  memset(&samples, 0, sizeof samples);
  samples.offset_samples = need_doubles(m_samples, "samples", "offset_samples", &S);
  samples.nhi_samples = need_doubles(m_samples, "samples", "nhi_samples", &count);
  if (count != S) mexErrMsgIdAndTxt("gpdla:arg", "samples.nhi_samples must match offset_samples");
  samples.log_nhi_samples = need_doubles(m_samples, "samples", "log_nhi_samples", &count);
  if (count != S) mexErrMsgIdAndTxt("gpdla:arg", "samples.log_nhi_samples must match offset_samples");
  samples.lls_nhi_samples = NULL;
  samples.num_dla_samples = (int64_t)S;

  /* ---- the ragged cell arrays of preloaded_qsos.mat (:43-61), handed over cell by cell ---- */
  c_wl = need_field(m_spectra, "spectra", "wavelengths");
  c_flux = need_field(m_spectra, "spectra", "flux");
  c_nv = need_field(m_spectra, "spectra", "noise_variance");
  c_mask = need_field(m_spectra, "spectra", "pixel_mask");
  if (!mxIsCell(c_wl) || !mxIsCell(c_flux) || !mxIsCell(c_nv) || !mxIsCell(c_mask))
    mexErrMsgIdAndTxt("gpdla:arg", "spectra.wavelengths / flux / noise_variance / pixel_mask must be cell arrays");
  nq = mxGetNumberOfElements(c_wl);
  if (mxGetNumberOfElements(c_flux) != nq || mxGetNumberOfElements(c_nv) != nq || mxGetNumberOfElements(c_mask) != nq)
    mexErrMsgIdAndTxt("gpdla:arg", "the four cell arrays must have one cell per quasar");
  spectra.z_qsos = need_doubles(m_spectra, "spectra", "z_qsos", &count);
  if (count != nq) mexErrMsgIdAndTxt("gpdla:arg", "spectra.z_qsos must have one entry per cell");
  /* No flattening: the library takes one array per quasar (gpdla_spectra_cells) and copies block by block
   * into its batch slots beside the sweeps.  Logical masks are one byte per pixel and go as they are;
   * a mask held as doubles is converted here (into one byte buffer, freed below). */
  npix = (int64_t *)mxMalloc(AT_LEAST_ONE(nq) * sizeof(int64_t));
  p_wl = (const double **)mxMalloc(AT_LEAST_ONE(nq) * sizeof(double *));
  p_flux = (const double **)mxMalloc(AT_LEAST_ONE(nq) * sizeof(double *));
  p_nv = (const double **)mxMalloc(AT_LEAST_ONE(nq) * sizeof(double *));
  p_mask = (const uint8_t **)mxMalloc(AT_LEAST_ONE(nq) * sizeof(uint8_t *));
  total = 0;
  for (q = 0; q < nq; ++q) {
    const mxArray *w = mxGetCell(c_wl, q), *fl = mxGetCell(c_flux, q), *v = mxGetCell(c_nv, q), *mk = mxGetCell(c_mask, q);
    size_t n = w ? mxGetNumberOfElements(w) : 0;
    if (n && (!fl || !v || !mk || mxGetNumberOfElements(fl) != n || mxGetNumberOfElements(v) != n ||
              mxGetNumberOfElements(mk) != n))
      mexErrMsgIdAndTxt("gpdla:arg", "quasar %d: the four cells differ in length", (int)(q + 1));
    if (n && (!mxIsDouble(w) || !mxIsDouble(fl) || !mxIsDouble(v) || !(mxIsLogical(mk) || mxIsDouble(mk))))
      mexErrMsgIdAndTxt("gpdla:arg", "quasar %d: double wavelengths / flux / noise_variance and a logical mask", (int)(q + 1));
    npix[q] = (int64_t)n;
    p_wl[q] = n ? mxGetPr(w) : NULL;
    p_flux[q] = n ? mxGetPr(fl) : NULL;
    p_nv[q] = n ? mxGetPr(v) : NULL;
    p_mask[q] = n && mxIsLogical(mk) ? (const uint8_t *)mxGetLogicals(mk) : NULL;
    if (n && !mxIsLogical(mk)) total += n;
  }
  mask = (uint8_t *)mxMalloc(AT_LEAST_ONE(total));
  total = 0;
  for (q = 0; q < nq; ++q) {
    const mxArray *mk = mxGetCell(c_mask, q);
    size_t n = (size_t)npix[q];
    if (n && !mxIsLogical(mk)) {
      const double *b = mxGetPr(mk);
      for (i = 0; i < n; ++i) mask[total + i] = b[i] != 0.0;
      p_mask[q] = mask + total;
      total += n;
    }
  }

  /* ---- the model prior, :122-131: counts of prior quasars with z < z_qso + prior_z_qso_increase ---- */
  prior_z = need_doubles(m_prior, "prior", "z_qsos", &n_prior);
  prior_flags = need_field(m_prior, "prior", "dla_ind");
  n_prior_flags = mxGetNumberOfElements(prior_flags);
  if (n_prior_flags != n_prior || !(mxIsLogical(prior_flags) || mxIsDouble(prior_flags)))
    mexErrMsgIdAndTxt("gpdla:arg", "prior.dla_ind must be a logical vector the size of prior.z_qsos");
  z_increase = optional_scalar(m_params, "prior_z_qso_increase", 30000.0 / 299792.458); /* kms_to_z(30000), set_parameters.m:56 */
  for (f = 0; f < NUM_FIELDS; ++f)
    out[f] = column(nq, f == F_SLL ? S : f == F_POST ? 2 : 1);
  lp_no = mxGetPr(out[F_LP_NO]);
  lp_dla = mxGetPr(out[F_LP_DLA]);
  for (q = 0; q < nq; ++q) {
    double num_quasars = 0.0, num_dlas = 0.0;
    const double limit = spectra.z_qsos[q] + z_increase; /* :122 */
    for (i = 0; i < n_prior; ++i) {
      if (prior_z[i] < limit) {
        const int is_dla = mxIsLogical(prior_flags) ? (mxGetLogicals(prior_flags)[i] != 0) : (mxGetPr(prior_flags)[i] != 0.0);
        num_quasars += 1.0;             /* :124 */
        num_dlas += is_dla ? 1.0 : 0.0; /* :125 */
      }
    }
    lp_dla[q] = log(num_dlas) - log(num_quasars);               /* :128-129 */
    lp_no[q] = log(num_quasars - num_dlas) - log(num_quasars);  /* :130-131 */
  }

  spectra.num_quasars = (int64_t)nq;
  spectra.num_pixels = npix;
  spectra.wavelengths = p_wl;
  spectra.flux = p_flux;
  spectra.noise_variance = p_nv;
  spectra.pixel_mask = p_mask;
  spectra.log_priors_no_dla = lp_no;
  spectra.log_priors_dla = lp_dla;
  spectra.log_priors_lls = NULL;

  /* ---- set_parameters.m values the loop reads ---- */
  gpdla_default_config(&cfg);
  cfg.num_lines = (int32_t)optional_scalar(m_params, "num_lines", cfg.num_lines);          /* :188 */
  cfg.min_lambda = optional_scalar(m_params, "min_lambda", cfg.min_lambda);                /* :104 */
  cfg.max_lambda = optional_scalar(m_params, "max_lambda", cfg.max_lambda);                /* :105 */
  cfg.lya_wavelength = optional_scalar(m_params, "lya_wavelength", cfg.lya_wavelength);    /* :118 */
  cfg.lyman_limit = optional_scalar(m_params, "lyman_limit", cfg.lyman_limit);
  cfg.pixel_spacing = optional_scalar(m_params, "pixel_spacing", cfg.pixel_spacing);       /* :169-175 */
  cfg.max_z_cut = optional_scalar(m_params, "max_z_cut", cfg.max_z_cut);                   /* :159-160 */
  cfg.min_z_cut = optional_scalar(m_params, "min_z_cut", cfg.min_z_cut);
  device_id = (int)optional_scalar(m_params, "device_id", 0.0);

  /* ---- the loop, :88-233.  The library's tables are [nq][S] / [nq][2] with the quasar slowest
   * (row-major); MATLAB's are column-major: those two are transposed on the way out. ---- */
  sll = (double *)mxMalloc((nq > 0 && S > 0 ? nq * S : 1) * sizeof(double));
  post = (double *)mxMalloc((nq ? 2 * nq : 1) * sizeof(double));
  memset(&res, 0, sizeof res);
  res.min_z_dlas = mxGetPr(out[F_MIN_Z]);
  res.max_z_dlas = mxGetPr(out[F_MAX_Z]);
  res.log_likelihoods_no_dla = mxGetPr(out[F_LL_NO]);
  res.sample_log_likelihoods_dla = sll;
  res.log_likelihoods_dla = mxGetPr(out[F_LL_DLA]);
  res.log_posteriors_no_dla = mxGetPr(out[F_LPOST_NO]);
  res.log_posteriors_dla = mxGetPr(out[F_LPOST_DLA]);
  res.model_posteriors = post;
  res.p_no_dlas = mxGetPr(out[F_P_NO]);
  res.p_dlas = mxGetPr(out[F_P_DLA]);
  rc = nq ? gpdla_process_cells(&model, &samples, &spectra, &cfg, &res, device_id) : GPDLA_OK;
  if (rc == GPDLA_OK) {
    double *sll_m = mxGetPr(out[F_SLL]), *post_m = mxGetPr(out[F_POST]);
    for (q = 0; q < nq; ++q) {
      for (i = 0; i < S; ++i) sll_m[q + i * nq] = sll[q * S + i]; /* sample_log_likelihoods_dla(quasar_ind, i), :196 */
      post_m[q] = post[2 * q];                                    /* model_posteriors(:, 1), :230 */
      post_m[q + nq] = post[2 * q + 1];
    }
  }
  mxFree(sll);
  mxFree(post);
  mxFree(mask);
  mxFree((void *)p_mask);
  mxFree((void *)p_nv);
  mxFree((void *)p_flux);
  mxFree((void *)p_wl);
  mxFree(npix);
  if (rc != GPDLA_OK) {
    for (f = 0; f < NUM_FIELDS; ++f) mxDestroyArray(out[f]);
    mexErrMsgIdAndTxt("gpdla:process_qsos", "%s", gpdla_last_error());
  }
  plhs[0] = mxCreateStructMatrix(1, 1, NUM_FIELDS, fields);
  for (f = 0; f < NUM_FIELDS; ++f) mxSetFieldByNumber(plhs[0], 0, f, out[f]);
}
