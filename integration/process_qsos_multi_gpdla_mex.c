/* process_qsos_multi_gpdla_mex.c -- reference-side MEX gateway for the hot loop of
 * multi_dlas/process_qsos_multiple_dlas_meanflux.m.
 *
 * Replaces multi :141-495 (the serial loop over quasars -- mean-flux suppression, up to max_dlas
 * stacked Voigt profiles per sample, the sub-DLA model, MAP extraction, the weighted resampling --
 * and the (2 + max_dlas)-way softmax) by ONE call into libgpdla.so.  The script keeps its loading
 * code (:1-99) and its save (:498-523); between them it calls
 *
 *     res = process_qsos_multi_gpdla(model, samples, spectra, prior, Z_lls, Z_dla [, params [, base_sample_inds]]);
 *
 *   model    struct: rest_wavelengths, mu, M, log_omega, log_c_0, log_tau_0, log_beta
 *            (learned_qso_model_lyseries_variance_kim_*.mat, loaded as in process_qsos.m:30-35)
 *   samples  struct: offset_samples, log_nhi_samples, nhi_samples (dla_samples.mat) and
 *            lls_nhi_samples (set_lls_parameters.m:59-63)
 *   spectra  struct: wavelengths, flux, noise_variance, pixel_mask (the all_* cell arrays after the
 *            test_ind subset) and z_qsos, one entry per cell
 *   prior    struct: z_qsos, dla_ind -- the training release's quasars after the Lyman-limit filter
 *            of multi :56-63
 *   Z_lls, Z_dla   the partition functions of set_lls_parameters.m:69-71 (multi :208-216)
 *   params   optional struct; any of: prior_z_qso_increase, num_lines, max_dlas, min_z_separation,
 *            prev_tau_0, prev_beta, num_forest_lines, min_lambda, max_lambda, lya_wavelength,
 *            lyman_limit, pixel_spacing, max_z_cut, min_z_cut, rng_seed, device_id
 *            (set_parameters_multi.m, multi :31-37; defaults are the reference's)
 *   base_sample_inds   optional uint32 [nq x S x (max_dlas - 1)], 1-based: the resampling indices of a
 *            previous run (the variable the script saves, :476) to replay; omitted or [], they are
 *            drawn on the GPU (MATLAB's rng('default') stream of :143 cannot be reproduced outside
 *            MATLAB -- gpdla.h, gpdla_process_batch_multi)
 *   res      struct with the variables the script saves (:498-510), shaped as the script shapes
 *            them (:104-139): min_z_dlas, max_z_dlas, log_priors_no_dla, log_priors_lls,
 *            log_likelihoods_no_dla, log_likelihoods_lls, log_posteriors_no_dla, log_posteriors_lls,
 *            p_no_dlas, p_dlas, p_lls, all_exceptions [nq x 1]; log_priors_dla, log_likelihoods_dla,
 *            log_posteriors_dla [nq x max_dlas]; sample_log_likelihoods_dla [nq x S x max_dlas];
 *            sample_log_likelihoods_lls [nq x S]; base_sample_inds uint32 [nq x S x (max_dlas - 1)];
 *            MAP_z_dlas, MAP_log_nhis (and MAP_inds, which the script fills, :441, but does not save)
 *            [nq x max_dlas x max_dlas]; model_posteriors [nq x (2 + max_dlas)].
 *
 * Build (MATLAB):  mex process_qsos_multi_gpdla_mex.c -output process_qsos_multi_gpdla -I<repo>/include ...
 *                      -L<repo>/gp_dla_detection_amd/csrc -lgpdla -lamdhip64
 * Only documented mex.h / matrix.h calls are used; tests/test_integration.py compiles this file
 * against declarations of exactly those calls (no MATLAB exists in the build image).  The gateway
 * owns no state: every buffer it allocates is freed before it returns.
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

static size_t at_least_one(size_t n) { return n ? n : 1; } /* mxMalloc(0) may return NULL */

/* a double array of the given dimensions, NaN-filled as multi :104-139 pre-fills its results */
static mxArray *nan_array(size_t d0, size_t d1, size_t d2) {
  mwSize dims[3];
  mxArray *a;
  double *p;
  size_t i, n = d0 * d1 * d2;
  dims[0] = d0;
  dims[1] = d1;
  dims[2] = d2;
  a = mxCreateNumericArray(d2 > 1 ? 3 : 2, dims, mxDOUBLE_CLASS, mxREAL);
  p = mxGetPr(a);
  for (i = 0; i < n; ++i) p[i] = mxGetNaN();
  return a;
}

/* [nq][inner] (quasar slowest, the library's layout) -> MATLAB's [nq x inner] column-major */
static void to_matlab_2d(const double *lib, double *m, size_t nq, size_t inner) {
  size_t q, j;
  for (q = 0; q < nq; ++q)
    for (j = 0; j < inner; ++j) m[q + nq * j] = lib[q * inner + j];
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  static const char *fields[] = {"min_z_dlas", "max_z_dlas", "sample_log_likelihoods_dla", "base_sample_inds",
                                 "log_priors_no_dla", "log_priors_dla", "log_priors_lls", "log_likelihoods_no_dla",
                                 "MAP_z_dlas", "MAP_log_nhis", "log_likelihoods_dla", "log_likelihoods_lls",
                                 "log_posteriors_no_dla", "log_posteriors_dla", "log_posteriors_lls",
                                 "model_posteriors", "p_no_dlas", "p_dlas", "p_lls", "all_exceptions",
                                 "sample_log_likelihoods_lls", "MAP_inds"}; /* multi :498-510 (+ MAP_inds, :441) */
  enum { F_MIN_Z, F_MAX_Z, F_SLL_DLA, F_BASE, F_LP_NO, F_LP_DLA, F_LP_LLS, F_LL_NO, F_MAP_Z, F_MAP_N, F_LL_DLA,
         F_LL_LLS, F_LPOST_NO, F_LPOST_DLA, F_LPOST_LLS, F_POST, F_P_NO, F_P_DLA, F_P_LLS, F_EXC, F_SLL_LLS, F_MAP_I,
         NUM_FIELDS };
  const mxArray *m_model, *m_samples, *m_spectra, *m_prior, *m_params, *m_base;
  const mxArray *c_wl, *c_flux, *c_nv, *c_mask, *prior_flags;
  gpdla_model model;
  gpdla_samples samples;
  gpdla_spectra_cells spectra;
  gpdla_config cfg;
  gpdla_results_multi res;
  mxArray *out[NUM_FIELDS];
  size_t nq, S, G, n_prior, count, q, i, j, total, md, nb;
  int64_t *npix;
  const double **p_wl, **p_flux, **p_nv;
  const uint8_t **p_mask;
  double *lp_no, *lp_lls, *lp_dla, *lp_dla_lib;
  double *sll_dla, *ll_dla, *lpost_dla, *post, *map_z, *map_n, *map_i;
  uint32_t *base_lib, *base_in_lib = NULL;
  int32_t *status;
  uint8_t *mask;
  const double *prior_z;
  double z_increase, Z_lls, Z_dla;
  int device_id, rc, f;

  if (nrhs < 6 || nrhs > 8)
    mexErrMsgIdAndTxt("gpdla:arg", "usage: res = process_qsos_multi_gpdla(model, samples, spectra, prior, Z_lls, Z_dla "
                                   "[, params [, base_sample_inds]])");
  if (nlhs > 1) mexErrMsgIdAndTxt("gpdla:arg", "one output");
  m_model = prhs[0];
  m_samples = prhs[1];
  m_spectra = prhs[2];
  m_prior = prhs[3];
  if (mxGetNumberOfElements(prhs[4]) != 1 || mxGetNumberOfElements(prhs[5]) != 1)
    mexErrMsgIdAndTxt("gpdla:arg", "Z_lls and Z_dla must be scalars (set_lls_parameters.m:69-71)");
  Z_lls = mxGetScalar(prhs[4]);
  Z_dla = mxGetScalar(prhs[5]);
  m_params = nrhs > 6 ? prhs[6] : NULL;
  m_base = nrhs > 7 && mxGetNumberOfElements(prhs[7]) > 0 ? prhs[7] : NULL;

  /* ---- the learned mean-flux model ---- */
  memset(&model, 0, sizeof model);
  model.rest_wavelengths = need_doubles(m_model, "model", "rest_wavelengths", &G);
  model.mu = need_doubles(m_model, "model", "mu", &count);
  if (count != G) mexErrMsgIdAndTxt("gpdla:arg", "model.mu must have numel(rest_wavelengths) entries");
  model.log_omega = need_doubles(m_model, "model", "log_omega", &count);
  if (count != G) mexErrMsgIdAndTxt("gpdla:arg", "model.log_omega must have numel(rest_wavelengths) entries");
  model.M = need_doubles(m_model, "model", "M", &count);
  if (mxGetM(need_field(m_model, "model", "M")) != G)
    mexErrMsgIdAndTxt("gpdla:arg", "model.M must have numel(rest_wavelengths) rows");
  model.num_rest_pixels = (int32_t)G;
  model.k = (int32_t)mxGetN(need_field(m_model, "model", "M"));
  model.log_c_0 = need_scalar(m_model, "model", "log_c_0");
  model.log_tau_0 = need_scalar(m_model, "model", "log_tau_0");
  model.log_beta = need_scalar(m_model, "model", "log_beta");

  /* ---- the DLA and sub-DLA parameter samples ---- */
  memset(&samples, 0, sizeof samples);
  samples.offset_samples = need_doubles(m_samples, "samples", "offset_samples", &S);
  samples.nhi_samples = need_doubles(m_samples, "samples", "nhi_samples", &count);
  if (count != S) mexErrMsgIdAndTxt("gpdla:arg", "samples.nhi_samples must match offset_samples");
  samples.log_nhi_samples = need_doubles(m_samples, "samples", "log_nhi_samples", &count);
  if (count != S) mexErrMsgIdAndTxt("gpdla:arg", "samples.log_nhi_samples must match offset_samples");
  samples.lls_nhi_samples = need_doubles(m_samples, "samples", "lls_nhi_samples", &count);
  if (count != S) mexErrMsgIdAndTxt("gpdla:arg", "samples.lls_nhi_samples must match offset_samples");
  samples.num_dla_samples = (int64_t)S;

  /* ---- set_parameters_multi.m / multi :31-37 ---- */
  gpdla_default_config(&cfg);
  cfg.num_lines = (int32_t)optional_scalar(m_params, "num_lines", cfg.num_lines);
  cfg.max_dlas = (int32_t)optional_scalar(m_params, "max_dlas", cfg.max_dlas);                     /* :32 */
  cfg.min_z_separation = optional_scalar(m_params, "min_z_separation", cfg.min_z_separation);      /* :33 */
  cfg.prev_tau_0 = optional_scalar(m_params, "prev_tau_0", cfg.prev_tau_0);                        /* :36 */
  cfg.prev_beta = optional_scalar(m_params, "prev_beta", cfg.prev_beta);                           /* :37 */
  cfg.num_forest_lines = (int32_t)optional_scalar(m_params, "num_forest_lines", cfg.num_forest_lines);
  cfg.min_lambda = optional_scalar(m_params, "min_lambda", cfg.min_lambda);
  cfg.max_lambda = optional_scalar(m_params, "max_lambda", cfg.max_lambda);
  cfg.lya_wavelength = optional_scalar(m_params, "lya_wavelength", cfg.lya_wavelength);
  cfg.lyman_limit = optional_scalar(m_params, "lyman_limit", cfg.lyman_limit);
  cfg.pixel_spacing = optional_scalar(m_params, "pixel_spacing", cfg.pixel_spacing);
  cfg.max_z_cut = optional_scalar(m_params, "max_z_cut", cfg.max_z_cut);
  cfg.min_z_cut = optional_scalar(m_params, "min_z_cut", cfg.min_z_cut);
  if (optional_scalar(m_params, "rng_seed", -1.0) >= 0.0) /* (a double holds 53 bits; absent: the library's default) */
    cfg.rng_seed = (uint64_t)optional_scalar(m_params, "rng_seed", 0.0);
  device_id = (int)optional_scalar(m_params, "device_id", 0.0);
  if (cfg.max_dlas < 1 || cfg.max_dlas > 4) mexErrMsgIdAndTxt("gpdla:arg", "max_dlas must be 1..4");
  md = (size_t)cfg.max_dlas;
  nb = md - 1;

  /* ---- the ragged cell arrays of preloaded_qsos.mat, handed over cell by cell ---- */
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
  if (m_base != NULL) {
    if (!mxIsUint32(m_base) || mxGetNumberOfElements(m_base) != nq * S * nb)
      mexErrMsgIdAndTxt("gpdla:arg", "base_sample_inds must be uint32 [nq x S x (max_dlas - 1)] (multi :116)");
  }
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

  /* ---- result arrays, shaped and pre-filled as multi :104-139 ---- */
  out[F_MIN_Z] = nan_array(nq, 1, 1);
  out[F_MAX_Z] = nan_array(nq, 1, 1);
  out[F_SLL_DLA] = nan_array(nq, S, md);
  {
    mwSize dims[3];
    dims[0] = nq;
    dims[1] = S;
    dims[2] = nb;
    out[F_BASE] = mxCreateNumericArray(nb > 1 ? 3 : 2, dims, mxUINT32_CLASS, mxREAL); /* zeros, :116 */
  }
  out[F_LP_NO] = nan_array(nq, 1, 1);
  out[F_LP_DLA] = nan_array(nq, md, 1);
  out[F_LP_LLS] = nan_array(nq, 1, 1);
  out[F_LL_NO] = nan_array(nq, 1, 1);
  out[F_MAP_Z] = nan_array(nq, md, md);
  out[F_MAP_N] = nan_array(nq, md, md);
  out[F_LL_DLA] = nan_array(nq, md, 1);
  out[F_LL_LLS] = nan_array(nq, 1, 1);
  out[F_LPOST_NO] = nan_array(nq, 1, 1);
  out[F_LPOST_DLA] = nan_array(nq, md, 1);
  out[F_LPOST_LLS] = nan_array(nq, 1, 1);
  out[F_POST] = nan_array(nq, 2 + md, 1);
  out[F_P_NO] = nan_array(nq, 1, 1);
  out[F_P_DLA] = nan_array(nq, 1, 1);
  out[F_P_LLS] = nan_array(nq, 1, 1);
  out[F_EXC] = nan_array(nq, 1, 1);
  out[F_SLL_LLS] = nan_array(nq, S, 1);
  out[F_MAP_I] = nan_array(nq, md, md);

  /* ---- the model priors, multi :189-216 ---- */
  prior_z = need_doubles(m_prior, "prior", "z_qsos", &n_prior);
  prior_flags = need_field(m_prior, "prior", "dla_ind");
  if (mxGetNumberOfElements(prior_flags) != n_prior || !(mxIsLogical(prior_flags) || mxIsDouble(prior_flags)))
    mexErrMsgIdAndTxt("gpdla:arg", "prior.dla_ind must be a logical vector the size of prior.z_qsos");
  z_increase = optional_scalar(m_params, "prior_z_qso_increase", 30000.0 / 299792.458);
  lp_no = mxGetPr(out[F_LP_NO]);
  lp_lls = mxGetPr(out[F_LP_LLS]);
  lp_dla = mxGetPr(out[F_LP_DLA]);
  lp_dla_lib = (double *)mxMalloc(at_least_one(nq * md) * sizeof(double));
  for (q = 0; q < nq; ++q) {
    double num_quasars = 0.0, num_dlas = 0.0, p[4], ratio;
    const double limit = spectra.z_qsos[q] + z_increase; /* :189 */
    for (i = 0; i < n_prior; ++i) {
      if (prior_z[i] < limit) {
        const int is_dla = mxIsLogical(prior_flags) ? (mxGetLogicals(prior_flags)[i] != 0) : (mxGetPr(prior_flags)[i] != 0.0);
        num_quasars += 1.0;             /* :192 */
        num_dlas += is_dla ? 1.0 : 0.0; /* :191 */
      }
    }
    ratio = num_dlas / num_quasars;
    for (j = 0; j < md; ++j) p[j] = pow(ratio, (double)(j + 1));   /* :193 */
    for (j = 0; j + 1 < md; ++j) p[j] = p[j] - p[j + 1];           /* :196-198 */
    for (j = 0; j < md; ++j) {
      lp_dla[q + nq * j] = log(p[j]);                              /* :204 */
      lp_dla_lib[q * md + j] = lp_dla[q + nq * j];
    }
    lp_lls[q] = log(num_dlas) - log(num_quasars) + log(Z_lls) - log(Z_dla);                  /* :208-210 */
    lp_no[q] = log(num_quasars - num_dlas - Z_lls * num_dlas / Z_dla) - log(num_quasars);    /* :214-216 */
  }

  spectra.num_quasars = (int64_t)nq;
  spectra.num_pixels = npix;
  spectra.wavelengths = p_wl;
  spectra.flux = p_flux;
  spectra.noise_variance = p_nv;
  spectra.pixel_mask = p_mask;
  spectra.log_priors_no_dla = lp_no;
  spectra.log_priors_dla = lp_dla_lib;
  spectra.log_priors_lls = lp_lls;

  /* ---- the loop, :141-495.  The library's tables have the quasar slowest (row-major, [nq][model][S]
   * and so on); MATLAB's are column-major with the quasar fastest: they are transposed on the way
   * in (base_sample_inds) and out. ---- */
  sll_dla = (double *)mxMalloc(at_least_one(nq * md * S) * sizeof(double));
  ll_dla = (double *)mxMalloc(at_least_one(nq * md) * sizeof(double));
  lpost_dla = (double *)mxMalloc(at_least_one(nq * md) * sizeof(double));
  post = (double *)mxMalloc(at_least_one(nq * (2 + md)) * sizeof(double));
  map_z = (double *)mxMalloc(at_least_one(nq * md * md) * sizeof(double));
  map_n = (double *)mxMalloc(at_least_one(nq * md * md) * sizeof(double));
  map_i = (double *)mxMalloc(at_least_one(nq * md * md) * sizeof(double));
  base_lib = (uint32_t *)mxMalloc(at_least_one(nq * nb * S) * sizeof(uint32_t));
  status = (int32_t *)mxMalloc(at_least_one(nq) * sizeof(int32_t));
  if (m_base != NULL) {
    const uint32_t *b = (const uint32_t *)mxGetData(m_base); /* (quasar, sample, model) -> [quasar][model][sample] */
    base_in_lib = (uint32_t *)mxMalloc(at_least_one(nq * nb * S) * sizeof(uint32_t));
    for (q = 0; q < nq; ++q)
      for (j = 0; j < nb; ++j)
        for (i = 0; i < S; ++i) base_in_lib[(q * nb + j) * S + i] = b[q + nq * (i + S * j)];
  }
  memset(&res, 0, sizeof res);
  res.min_z_dlas = mxGetPr(out[F_MIN_Z]);
  res.max_z_dlas = mxGetPr(out[F_MAX_Z]);
  res.log_likelihoods_no_dla = mxGetPr(out[F_LL_NO]);
  res.sample_log_likelihoods_dla = sll_dla;
  res.sample_log_likelihoods_lls = NULL; /* set below: [nq][S] needs the transpose as well */
  res.log_likelihoods_dla = ll_dla;
  res.log_likelihoods_lls = mxGetPr(out[F_LL_LLS]);
  res.log_posteriors_no_dla = mxGetPr(out[F_LPOST_NO]);
  res.log_posteriors_lls = mxGetPr(out[F_LPOST_LLS]);
  res.log_posteriors_dla = lpost_dla;
  res.model_posteriors = post;
  res.p_no_dlas = mxGetPr(out[F_P_NO]);
  res.p_lls = mxGetPr(out[F_P_LLS]);
  res.p_dlas = mxGetPr(out[F_P_DLA]);
  res.MAP_z_dlas = map_z;
  res.MAP_log_nhis = map_n;
  res.MAP_inds = map_i;
  res.base_sample_inds = base_lib;
  res.status = status;
  {
    double *sll_lls = (double *)mxMalloc(at_least_one(nq * S) * sizeof(double));
    res.sample_log_likelihoods_lls = sll_lls;
    rc = nq ? gpdla_process_cells_multi(&model, &samples, &spectra, base_in_lib, &cfg, &res, device_id) : GPDLA_OK;
    if (rc == GPDLA_OK) {
      double *m = mxGetPr(out[F_SLL_DLA]), *exc = mxGetPr(out[F_EXC]);
      uint32_t *bm = (uint32_t *)mxGetData(out[F_BASE]);
      for (q = 0; q < nq; ++q) {
        for (j = 0; j < md; ++j)
          for (i = 0; i < S; ++i) m[q + nq * (i + S * j)] = sll_dla[(q * md + j) * S + i]; /* (quasar_ind, :, :), :477 */
        for (j = 0; j < nb; ++j)
          for (i = 0; i < S; ++i) bm[q + nq * (i + S * j)] = base_lib[(q * nb + j) * S + i];  /* :476 */
        if (status[q] == 1) exc[q] = 1.0; /* all_exceptions, :139, :232 */
      }
      to_matlab_2d(sll_lls, mxGetPr(out[F_SLL_LLS]), nq, S);
      to_matlab_2d(ll_dla, mxGetPr(out[F_LL_DLA]), nq, md);
      to_matlab_2d(lpost_dla, mxGetPr(out[F_LPOST_DLA]), nq, md);
      to_matlab_2d(post, mxGetPr(out[F_POST]), nq, 2 + md);
      /* [nq][model][slot] -> (quasar, model, slot) */
      to_matlab_2d(map_z, mxGetPr(out[F_MAP_Z]), nq, md * md);
      to_matlab_2d(map_n, mxGetPr(out[F_MAP_N]), nq, md * md);
      to_matlab_2d(map_i, mxGetPr(out[F_MAP_I]), nq, md * md);
      if (md > 1) { /* to_matlab_2d put entry model * md + slot at column model * md + slot: swap to model + md * slot */
        double *maps[3];
        size_t a, b, t;
        maps[0] = mxGetPr(out[F_MAP_Z]);
        maps[1] = mxGetPr(out[F_MAP_N]);
        maps[2] = mxGetPr(out[F_MAP_I]);
        for (t = 0; t < 3; ++t)
          for (q = 0; q < nq; ++q)
            for (a = 0; a < md; ++a)
              for (b = a + 1; b < md; ++b) {
                const double x = maps[t][q + nq * (a * md + b)];
                maps[t][q + nq * (a * md + b)] = maps[t][q + nq * (b * md + a)];
                maps[t][q + nq * (b * md + a)] = x;
              }
      }
    }
    mxFree(sll_lls);
  }
  if (base_in_lib) mxFree(base_in_lib);
  mxFree(status);
  mxFree(base_lib);
  mxFree(map_i);
  mxFree(map_n);
  mxFree(map_z);
  mxFree(post);
  mxFree(lpost_dla);
  mxFree(ll_dla);
  mxFree(sll_dla);
  mxFree(lp_dla_lib);
  mxFree(mask);
  mxFree((void *)p_mask);
  mxFree((void *)p_nv);
  mxFree((void *)p_flux);
  mxFree((void *)p_wl);
  mxFree(npix);
  if (rc != GPDLA_OK) {
    for (f = 0; f < NUM_FIELDS; ++f) mxDestroyArray(out[f]);
    mexErrMsgIdAndTxt("gpdla:process_qsos_multi", "%s", gpdla_last_error());
  }
  plhs[0] = mxCreateStructMatrix(1, 1, NUM_FIELDS, fields);
  for (f = 0; f < NUM_FIELDS; ++f) mxSetFieldByNumber(plhs[0], 0, f, out[f]);
}
