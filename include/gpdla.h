/* gpdla.h -- C-ABI of the MI355X-native GP/DLA inference sweep (libgpdla.so).
 *
 * Drop-in boundary for ONE hot path of jibanCat/gp_dla_detection: the per-spectrum GP
 * marginal-likelihood sweep.  The reference has no plugin registry; the path sits behind three
 * plain call surfaces, and each entry point below replaces one of them (paths relative to the
 * reference tree):
 *
 *   gpdla_voigt                  <-  voigt.c:253-304          MEX gateway  voigt(lambdas, z, N[, num_lines])
 *   gpdla_log_mvnpdf_low_rank    <-  log_mvnpdf_low_rank.m:5  log_p = log_mvnpdf_low_rank(y, mu, M, d)
 *   gpdla_process_batch          <-  process_qsos.m:88-233    the per-quasar loop + posteriors
 *   gpdla_process_batch_multi    <-  multi_dlas/process_qsos_multiple_dlas_meanflux.m:141-495
 *
 * plus a resident-data form (context + uploaded batch) so that a caller that keeps spectra in HBM
 * -- the production case, and what bench.py times -- pays no PCIe inside the sweep.
 *
 * Conventions: plain pointers and sizes only, caller owns every buffer,
 * no global state, all arithmetic IEEE fp64, MATLAB matrices are column-major, every function
 * returns 0 or a negative gpdla_status.  All compute runs in hand-written HIP kernels for gfx950;
 * there is NO CPU fallback: without a usable GPU every compute entry point returns
 * GPDLA_ERR_NO_DEVICE.
 */
#ifndef GPDLA_H
#define GPDLA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  GPDLA_OK = 0,
  GPDLA_ERR_INVALID_ARGUMENT = -1, /* null pointer, n_padded <= 6, num_lines outside [1,31], ... */
  GPDLA_ERR_NO_DEVICE = -2,        /* no HIP device / device_id out of range */
  GPDLA_ERR_HIP = -3,              /* a HIP runtime call failed; see gpdla_last_error() */
  GPDLA_ERR_NOT_POSITIVE_DEFINITE = -4, /* chol(B) would throw, log_mvnpdf_low_rank.m:24 */
  GPDLA_ERR_UNSUPPORTED = -5,      /* e.g. k > GPDLA_MAX_K */
  GPDLA_ERR_HOST = -6              /* host side: out of memory, a stage thread could not be started, or an
                                      unexpected C++ exception -- none ever crosses this boundary */
} gpdla_status;

#define GPDLA_MAX_K 40
#define GPDLA_ABI_VERSION 6

int gpdla_abi_version(void);
/* Human-readable text of the most recent error on this thread (never NULL). */
const char *gpdla_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Stateless single-call surfaces (host pointers; device chosen by device_id).
 * ------------------------------------------------------------------------------------------- */

/* voigt.c:253-304.  profile_out has n_padded - 6 entries (numel(lambdas) - 2*width, :271).
 * num_lines in [1, 31] (the MEX default when the 4th argument is omitted is 31, :16, :266).
 * The reference validates nothing; here bad arguments return GPDLA_ERR_INVALID_ARGUMENT. */
int gpdla_voigt(const double *lambdas, int64_t n_padded, double z, double N, int num_lines,
                double *profile_out, int device_id);

/* log_mvnpdf_low_rank.m:5-34.  y, mu, d: n;  M: n x k column-major (ld = n).  A non-PD
 * B = I + M' D^-1 M returns GPDLA_ERR_NOT_POSITIVE_DEFINITE and *log_p = NaN (MATLAB: chol throws).
 * The MATLAB function takes any k; this stand-alone surface accepts k <= 256 (the batch sweep,
 * whose accumulators live in registers, is the one limited to GPDLA_MAX_K). */
int gpdla_log_mvnpdf_low_rank(const double *y, const double *mu, const double *M, const double *d,
                              int64_t n, int k, double *log_p, int device_id);

/* ---------------------------------------------------------------------------------------------
 * Batch surface: process_qsos.m.
 * ------------------------------------------------------------------------------------------- */

/* learned_qso_model_*.mat, process_qsos.m:30-35 (written by learn_qso_model.m:113-123). */
typedef struct {
  int32_t num_rest_pixels;          /* G = numel(rest_wavelengths) */
  int32_t k;                        /* columns of M (set_parameters.m:36) */
  const double *rest_wavelengths;   /* [G] ascending */
  const double *mu;                 /* [G] */
  const double *M;                  /* [G x k] column-major */
  const double *log_omega;          /* [G] */
  double log_c_0, log_tau_0, log_beta;
} gpdla_model;

/* dla_samples.mat, process_qsos.m:38-40 (+ lls_nhi_samples of set_lls_parameters.m:59-63 for the
 * multi-DLA driver; may be NULL otherwise). */
typedef struct {
  int64_t num_dla_samples;          /* S (set_parameters.m:48) */
  const double *offset_samples;     /* [S] in [0,1) */
  const double *log_nhi_samples;    /* [S]  (only the multi-DLA MAP bookkeeping reads it) */
  const double *nhi_samples;        /* [S] */
  const double *lls_nhi_samples;    /* [S] or NULL */
} gpdla_samples;

/* preloaded_qsos.mat's ragged cell arrays (preload_qsos.m:64-79) flattened CSR-style, after the
 * test_ind subset of process_qsos.m:56-61, plus the per-quasar scalars the loop reads. */
typedef struct {
  int64_t num_quasars;
  const int64_t *offsets;           /* [num_quasars + 1] into the four pixel arrays */
  const double *wavelengths;        /* observed, Angstrom */
  const double *flux;
  const double *noise_variance;
  const uint8_t *pixel_mask;        /* nonzero = masked */
  const double *z_qsos;             /* [num_quasars] */
  const double *log_priors_no_dla;  /* [num_quasars]  process_qsos.m:130-131 (host logic) */
  const double *log_priors_dla;     /* [num_quasars]  process_qsos.m:128-129; multi: [nq][max_dlas] (multi :204) */
  const double *log_priors_lls;     /* multi only (:208-210), else NULL */
} gpdla_spectra;

/* The set_parameters.m values the loop reads (process_qsos.m:104-105, 118, 159-176, 188). */
typedef struct {
  double min_lambda, max_lambda;    /* set_parameters.m:33-34 */
  double lya_wavelength;            /* :5  */
  double lyman_limit;               /* :7  */
  double pixel_spacing;             /* :60 */
  double max_z_cut, min_z_cut;      /* :65, :69 */
  int32_t width;                    /* :59 -- must be 3 (voigt.c:229 hard-codes it) */
  int32_t num_lines;                /* :63 */
  /* multi-DLA only (process_qsos_multiple_dlas_meanflux.m:32-37, set_parameters_multi.m:75) */
  int32_t max_dlas;
  int32_t num_forest_lines;
  double min_z_separation;
  double prev_tau_0, prev_beta;
  /* weighted resampling of the multi-DLA driver when base_sample_inds is not supplied: the draw
   * for (quasar, model, index) depends only on rng_seed and first_quasar_index + local quasar
   * index, so shards of one run pass their global offset here and agree with an unsharded run */
  uint64_t rng_seed;
  int64_t first_quasar_index;
  /* 0 (default): the [W|U]*[P|M] contraction in fp64 (parity-grade).  1: BASELINE config 5's study
   * variant -- contraction on the fp32 matrix cores, everything else (Voigt profile, weights,
   * quadratic form, log-determinant, Cholesky) in fp64.  Not parity-grade; single-DLA sweep only. */
  int32_t contraction_precision;
  /* multi-DLA driver: bytes of HBM the per-quasar Voigt profile table (2 S rows per quasar) may take
   * at a time; quasars are swept in sub-batches that fit.  0 = default (16 GiB).  The table is
   * scratch of one gpdla_batch_process_multi call and belongs to the CONTEXT: it is allocated on
   * first use, grows only, is shared by all batches of the context and freed with it (the calls of
   * one context are serialized on its stream; two threads' calls take turns). */
  int64_t multi_profile_bytes;
  /* single-DLA sweep: bytes of HBM the per-K-step records of a batch may take at a time.  A batch
   * whose records exceed it is swept in groups of quasars (records built, then swept, group after
   * group into the same pool), so that the resident size of a batch is its spectra and results, not
   * its records: 0.9 KB per K-step for k <= 20, but 29 KB for 20 < k <= 40 -- 228 GB for a DR12Q
   * shard.  Results do not depend on it.  0 = default (16 GiB). */
  int64_t record_pool_bytes;
  /* one-shot entries (gpdla_process_batch, gpdla_process_batch_multi) only: the quasars of a call are
   * swept in HBM-resident batches of at most max_quasars_per_batch through pipeline_slots batch slots
   * (upload of batch i+1 / sweep of batch i / download of batch i-1 overlap).  0 = defaults: 3 slots,
   * gpdla_default_batch_quasars() quasars.  Results do not depend on either. */
  int32_t pipeline_slots;
  int64_t max_quasars_per_batch;
} gpdla_config;

/* Fills a gpdla_config with the reference's defaults (set_parameters.m / set_parameters_multi.m). */
void gpdla_default_config(gpdla_config *cfg);

/* Outputs of process_qsos.m:74-82, 224-233 (field names of :236-244).  Caller-owned; any pointer
 * may be NULL to skip that field.  Entries of spectra that cannot be processed (no pixel survives
 * the selection) are set to NaN, exactly as the reference's NaN pre-fill leaves them.
 * sample_log_likelihoods_dla is [num_quasars][S] with the quasar index slowest (element
 * (quasar_ind, i) of the MATLAB array at [quasar_ind * S + i]). */
typedef struct {
  double *min_z_dlas;                 /* [nq] */
  double *max_z_dlas;                 /* [nq] */
  double *log_likelihoods_no_dla;     /* [nq] */
  double *sample_log_likelihoods_dla; /* [nq][S] */
  double *log_likelihoods_dla;        /* [nq] */
  double *log_posteriors_no_dla;      /* [nq] */
  double *log_posteriors_dla;         /* [nq] */
  double *model_posteriors;           /* [nq][2] = (no DLA, DLA) */
  double *p_no_dlas;                  /* [nq] */
  double *p_dlas;                     /* [nq] */
  int32_t *status;                    /* [nq] 0 ok, 1 = empty spectrum (multi: all_exceptions, :232),
                                         3 = a kept pixel with noise variance <= 0 or NaN (skipped) */
  /* generate_ascii_catalog.m:73-80, found by the evidence kernel while it walks the table anyway:
   * [~, map_ind] = nanmax(sample_log_likelihoods_dla(i, :)) (1-based, first index on ties; 1 for an
   * all-NaN row, as MATLAB returns), map_z_dla = min_z + (max_z - min_z) * offset_samples(map_ind),
   * log_nhi_samples(map_ind) (log10 of nhi_samples(map_ind) when log_nhi_samples was not given). */
  double *MAP_inds;                   /* [nq] */
  double *MAP_z_dlas;                 /* [nq] */
  double *MAP_log_nhis;               /* [nq] */
} gpdla_results;

/* One-shot: host buffers in, host buffers out -- the loop of process_qsos.m:88 over ALL quasars of the
 * call.  Inside, the quasars are cut into blocks (config->max_quasars_per_batch) that go through
 * config->pipeline_slots batch slots in HBM: one library thread uploads block i+1 (the CSR arrays are
 * sliced in place, nothing is copied on the host) and another downloads block i-1 straight into the
 * caller's arrays while the calling thread has block i swept, so the PCIe copies hide behind the
 * sweeps (INTEGRATION.md section 3 states the measured rate against the resident form).  Host memory
 * beyond the caller's arrays: per-quasar bookkeeping of `slots` blocks.  The call owns a context for
 * its duration (streams, model, samples, slots) and returns when every result is in the caller's
 * arrays; on an error no result array is meaningful.  Results are bit-identical for every batching. */
int gpdla_process_batch(const gpdla_model *model, const gpdla_samples *samples,
                        const gpdla_spectra *spectra, const gpdla_config *config,
                        gpdla_results *results, int device_id);

/* The same loop with the spectra as preloaded_qsos.mat holds them (preload_qsos.m:64-79): ONE ARRAY PER
 * QUASAR -- the cells of all_wavelengths / all_flux / all_noise_variance / all_pixel_mask (after the
 * test_ind subset of process_qsos.m:56-61), or a Python list of NumPy arrays.  Nothing is flattened
 * up front: the library's upload thread copies block i+1 of the cells into the staging vectors of
 * its batch slot while block i is swept, so the host copy hides behind the sweeps like the PCIe
 * copies do (host memory beyond the caller's arrays: `slots` blocks of spectra).  pixel_mask cells
 * are one byte per pixel, nonzero = masked (mxLogical and numpy.bool_ as they are). */
typedef struct {
  int64_t num_quasars;
  const int64_t *num_pixels;              /* [nq] entries of each of the four cells of quasar q */
  const double *const *wavelengths;       /* [nq] pointers */
  const double *const *flux;
  const double *const *noise_variance;
  const uint8_t *const *pixel_mask;
  const double *z_qsos;                   /* [nq] */
  const double *log_priors_no_dla;        /* [nq] */
  const double *log_priors_dla;           /* [nq]; multi: [nq][max_dlas] */
  const double *log_priors_lls;           /* multi only, else NULL */
} gpdla_spectra_cells;
int gpdla_process_cells(const gpdla_model *model, const gpdla_samples *samples,
                        const gpdla_spectra_cells *spectra, const gpdla_config *config,
                        gpdla_results *results, int device_id);

/* Quasars per batch the one-shot entries (and the Python file pipeline) use by default: small enough
 * that `slots` batches of quasars of `longest_spectrum` pixels fit budget_bytes of HBM (0 = 96 GiB)
 * next to the record pool, and that a run has ~8 batches to overlap, at least 128 so that a launch
 * fills the 256 CUs many times over, at most 4096.  multi_models = max_dlas + 1 for the multi-DLA
 * driver (its batches also hold 2 x models sample tables and all their records), else 0.  No GPU. */
int64_t gpdla_default_batch_quasars(int64_t num_quasars, int64_t longest_spectrum, int k,
                                    int64_t num_dla_samples, int slots, int64_t budget_bytes,
                                    int multi_models);

/* ---------------------------------------------------------------------------------------------
 * Resident form: a context owns a device, a stream, the replicated model + samples and scratch;
 * a batch owns spectra in HBM.  Nothing here synchronises the device except where stated.
 * ------------------------------------------------------------------------------------------- */
typedef struct gpdla_context gpdla_context;
typedef struct gpdla_batch gpdla_batch;

int gpdla_context_create(int device_id, gpdla_context **ctx);
void gpdla_context_destroy(gpdla_context *ctx);
/* Use the caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) for all launches;
 * NULL restores the context's own stream. */
int gpdla_context_set_stream(gpdla_context *ctx, void *hip_stream);
int gpdla_context_set_model(gpdla_context *ctx, const gpdla_model *model);      /* H2D copy */
int gpdla_context_set_samples(gpdla_context *ctx, const gpdla_samples *samples);/* H2D copy */
int gpdla_context_set_config(gpdla_context *ctx, const gpdla_config *config);
/* Sets config.first_quasar_index alone.  gpdla_context_set_config replaces the whole configuration and
 * must not run while another thread uploads or re-fills a batch of this context (the upload reads
 * the configuration); this call touches one field no upload reads, so the thread that launches the
 * sweeps of a host pipeline may call it per batch while its upload thread is busy (the value is
 * read by the next gpdla_batch_process_multi on the calling thread). */
int gpdla_context_set_first_quasar_index(gpdla_context *ctx, int64_t first_quasar_index);
int gpdla_context_synchronize(gpdla_context *ctx);

/* Copies a CSR batch of spectra to HBM and allocates its result table there.  With
 * spectra->log_priors_lls != NULL the batch is a multi-DLA batch: log_priors_dla is then
 * [nq][max_dlas] (max_dlas of the context's config at this call) and the batch is processed with
 * gpdla_batch_process_multi. */
int gpdla_batch_upload(gpdla_context *ctx, const gpdla_spectra *spectra, gpdla_batch **batch);
/* Re-fills an existing batch with another set of spectra (any size, same or other kind), reusing
 * its device allocations where they are large enough: the batch slots of a host pipeline -- the
 * loop process_qsos.m:88 runs serially -- do no allocation in the steady state.  The batch's
 * previous results must have been downloaded (or be no longer wanted). */
int gpdla_batch_reload(gpdla_context *ctx, gpdla_batch *batch, const gpdla_spectra *spectra);
/* Safe in either order with gpdla_context_destroy (a batch whose context went first only frees
 * its memory). */
void gpdla_batch_destroy(gpdla_batch *batch);

/* The hot path: selection + interpolation (process_qsos.m:102-146), null evidence (:149-151),
 * S-sample Voigt/low-rank sweep (:185-199), evidence + posteriors (:203-213, :224-233) for every
 * quasar of the batch.  Asynchronous on the context's stream; results stay in HBM. */
int gpdla_batch_process(gpdla_context *ctx, gpdla_batch *batch);

/* D2H copy of the batch's results; returns when they are in the caller's arrays.  Uploads,
 * reloads and downloads run on the context's own copy streams, ordered against the batch's own
 * sweep by events: they do not wait for a sweep of ANOTHER batch in flight on the compute stream,
 * so a host pipeline can upload batch i+1 and download batch i-1 while batch i is swept.  The
 * entry points of one context may be called from several threads as long as each batch is used
 * by one thread at a time. */
int gpdla_batch_download(gpdla_context *ctx, gpdla_batch *batch, gpdla_results *results);

/* Device pointer to the per-quasar summary table of the batch, [nq][GPDLA_SUMMARY_COLS] doubles:
 * min_z_dla, max_z_dla, log_prior_no_dla, log_prior_dla, log_likelihood_no_dla, log_likelihood_dla,
 * log_posterior_no_dla, log_posterior_dla, model_posterior[0], model_posterior[1], p_no_dla, p_dla,
 * MAP_ind (1-based), MAP_z_dla, MAP_log_nhi (generate_ascii_catalog.m:73-80).
 * This is the row a multi-GPU run all-gathers. */
#define GPDLA_SUMMARY_COLS 15
int gpdla_batch_summary_device_ptr(gpdla_batch *batch, double **table, int64_t *num_quasars);
/* Device pointer to sample_log_likelihoods_dla [nq][S] of the batch. */
int gpdla_batch_samples_device_ptr(gpdla_batch *batch, double **table, int64_t *num_quasars,
                                   int64_t *num_samples);

/* Profiling aid for bench.py: duration in ms of the most recent sweep-kernel launch of this
 * context, measured with hipEvents on the launch stream (synchronises).  Negative if none. */
double gpdla_context_last_sweep_ms(gpdla_context *ctx);
/* Enables/disables the hipEvent bracketing above (off by default: zero overhead). */
int gpdla_context_set_timing(gpdla_context *ctx, int enabled);

/* ---------------------------------------------------------------------------------------------
 * Multi-DLA driver: multi_dlas/process_qsos_multiple_dlas_meanflux.m.
 * base_sample_inds: [nq][max_dlas-1][S] uint32, 1-BASED as in the reference's output file (:116,
 * :476).  MATLAB's rng('default') + randsample stream (:143, :471-472) cannot be reproduced outside
 * MATLAB, so either the caller supplies the indices (replaying a
 * reference output, or for parity tests), or passes NULL and they are drawn on the GPU with a
 * documented counter-based generator (Philox4x32-10, inverse-CDF sampling; config->rng_seed).
 * Either way the indices used are returned in results->base_sample_inds.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  double *min_z_dlas, *max_z_dlas;        /* [nq] */
  double *log_likelihoods_no_dla;         /* [nq] */
  double *sample_log_likelihoods_dla;     /* [nq][max_dlas][S] */
  double *sample_log_likelihoods_lls;     /* [nq][S] */
  double *log_likelihoods_dla;            /* [nq][max_dlas] */
  double *log_likelihoods_lls;            /* [nq] */
  double *log_posteriors_no_dla;          /* [nq] */
  double *log_posteriors_lls;             /* [nq] */
  double *log_posteriors_dla;             /* [nq][max_dlas] */
  double *model_posteriors;               /* [nq][2 + max_dlas] = (no DLA, LLS, 1..max_dlas DLAs) */
  double *p_no_dlas, *p_lls, *p_dlas;     /* [nq] */
  double *MAP_z_dlas, *MAP_log_nhis, *MAP_inds; /* [nq][max_dlas(model)][max_dlas(slot)], NaN unused */
  uint32_t *base_sample_inds;             /* [nq][max_dlas-1][S], 1-based: the indices used (:476) */
  int32_t *status;                        /* [nq] 1 = all_exceptions (:232) */
} gpdla_results_multi;

/* One-shot form, pipelined like gpdla_process_batch (blocks of quasars through batch slots; block b's
 * draws are keyed by config->first_quasar_index + its first quasar, so the batching does not change
 * them). */
int gpdla_process_batch_multi(const gpdla_model *model, const gpdla_samples *samples,
                              const gpdla_spectra *spectra, const uint32_t *base_sample_inds,
                              const gpdla_config *config, gpdla_results_multi *results,
                              int device_id);

/* ... and with one array per quasar (gpdla_spectra_cells, see gpdla_process_cells). */
int gpdla_process_cells_multi(const gpdla_model *model, const gpdla_samples *samples,
                              const gpdla_spectra_cells *spectra, const uint32_t *base_sample_inds,
                              const gpdla_config *config, gpdla_results_multi *results,
                              int device_id);

/* Resident form of the same driver: the batch was uploaded with log_priors_lls (see
 * gpdla_batch_upload).  base_sample_inds: HOST pointer [nq][max_dlas-1][S] (1-based; an entry 0
 * means "never drawn", as in the rows the reference leaves zero after its early exit, :116,
 * :460-464 -- a sample that would consume it gets NaN; entries > S are rejected) or NULL to draw
 * on the GPU.  Asynchronous on the context's stream apart from that one H2D copy; every result
 * stays in HBM until gpdla_batch_download_multi. */
int gpdla_batch_process_multi(gpdla_context *ctx, gpdla_batch *batch,
                              const uint32_t *base_sample_inds);
int gpdla_batch_download_multi(gpdla_context *ctx, gpdla_batch *batch, gpdla_results_multi *results);

/* Per-quasar summary row of a multi-DLA batch -- every saved variable of multi :498-510 that is not
 * a per-sample array; the row a multi-GPU run all-gathers.  Layout, md =
 * max_dlas: min_z_dla, max_z_dla | log_prior_no_dla, log_prior_lls, log_prior_dla[md] |
 * log_likelihood_no_dla, log_likelihood_lls, log_likelihood_dla[md] | log_posterior_no_dla,
 * log_posterior_lls, log_posterior_dla[md] | model_posteriors[2+md] | p_no_dla, p_lls, p_dla |
 * MAP_z_dlas[md][md], MAP_log_nhis[md][md], MAP_inds[md][md] ([model][slot]) | all_exceptions
 * (1 or NaN, :139, :232).  78 columns for max_dlas = 4. */
#define GPDLA_SUMMARY_COLS_MULTI(md) (14 + 4 * (md) + 3 * (md) * (md))
int gpdla_batch_summary_multi_device_ptr(gpdla_batch *batch, double **table, int64_t *num_quasars,
                                         int32_t *num_cols);
/* Device pointers to the per-sample tables of a multi-DLA batch (valid after the first
 * gpdla_batch_process_multi): sample_log_likelihoods_dla [nq][max_dlas][S],
 * sample_log_likelihoods_lls [nq][S], base_sample_inds [nq][max_dlas-1][S].  Any may be NULL. */
int gpdla_batch_samples_multi_device_ptr(gpdla_batch *batch, double **sample_ll_dla,
                                         double **sample_ll_lls, uint32_t **base_sample_inds);

/* ---------------------------------------------------------------------------------------------
 * Training objective: objective.m:12-75 over spectrum_loss.m:14-76.
 * The training set stays resident in HBM; each call evaluates f(x) and g(x) = df/dx for
 * x = [vec M (G x k, column-major); log omega (G); log c0; log tau0; log beta] (objective.m:5),
 * including the Kim et al. priors the reference adds to the gradient (:59-71).
 * The three data matrices are [num_quasars x num_pixels] column-major as MATLAB holds them, NaN =
 * missing pixel (:42).  Returns GPDLA_ERR_NOT_POSITIVE_DEFINITE where chol would throw (:42).
 * ------------------------------------------------------------------------------------------- */
typedef struct gpdla_training gpdla_training;
int gpdla_training_create(int device_id, int64_t num_quasars, int64_t num_pixels,
                          const double *centered_rest_fluxes, const double *lya_1pzs,
                          const double *rest_noise_variances, gpdla_training **out);
int gpdla_training_objective(gpdla_training *t, const double *x, int k, double *f, double *g);
/* Switches the training set to the mean-flux model's objective, multi_dlas/objective_lyseries.m:12-78
 * over multi_dlas/spectrum_loss_lyseries.m:14-93 (what multi_dlas/learn_qso_model_meanflux.m:140-142
 * minimises): the optical depth of a pixel sums the first num_forest_lines Lyman lines, each counted
 * only where its redshift does not exceed the quasar's (zqso + 1 = the quasar's last lya_1pz,
 * objective_lyseries.m:46); everything else is objective.m.  all_transition_wavelengths (any unit,
 * decreasing) / all_oscillator_strengths: num_forest_lines entries each, or both NULL for the table
 * of set_parameters_multi.m:76-143 (include/gpdla_lyman_series.h).  num_forest_lines <= 1 switches
 * back to objective.m.  Takes effect from the next gpdla_training_objective call. */
int gpdla_training_set_lyseries(gpdla_training *t, int num_forest_lines, const double *all_transition_wavelengths,
                                const double *all_oscillator_strengths);
void gpdla_training_destroy(gpdla_training *t);

/* ---------------------------------------------------------------------------------------------
 * Diagnostics.  The sweep kernel evaluates the Voigt function within 30 Doppler widths of a line
 * centre from per-line piecewise polynomials of Re w(x + i y_line) (what voigt.c:288 gets from
 * libcerf's voigt()).  This returns the HOST evaluation of the table of Lyman line `line`
 * (0 = Ly-alpha) at |x| < 32, and the line's damping parameter y = gamma/(sqrt2 sigma) in *y_out
 * (may be NULL).  Needs no GPU.
 * ------------------------------------------------------------------------------------------- */
int gpdla_debug_near_poly(int line, double x, double *value_out, double *y_out);

/* Test hook: runs only the preparation kernel of a batch (pixel selection, GP interpolation, noise
 * scaling; with multi != 0 the Lyman-series scaling and mean-flux suppression of
 * process_qsos_multiple_dlas_meanflux.m:245-293) and copies out, for quasar `quasar`, its rows on
 * the unmasked-range grid: rows_out[4 i + (0..3)] = (y, mu, omega2, nu) of pixel i (masked pixels:
 * 0, 0, 0, 1), for i < *num_rows_out <= capacity_rows.  Lets a test hold the GPU's mean-flux factor
 * to the numbers the reference's own QSOLoader.total_scale_factor produced (tests/golden/mean_flux.npz). */
int gpdla_debug_prepared_rows(gpdla_context *ctx, gpdla_batch *batch, int multi, int64_t quasar,
                              double *rows_out, int64_t capacity_rows, int64_t *num_rows_out);

/* The counter-based generator behind the multi-DLA resampling (Philox4x32-10 of Salmon et al.,
 * SC'11), evaluated on the HOST by the same function the kernel compiles: out = philox(ctr, key).
 * For known-answer tests against the Random123 vectors.  Needs no GPU. */
void gpdla_debug_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* Test hook for the boundary itself: throws, inside an entry point's body, std::bad_alloc (kind 1),
 * std::runtime_error (2) or a non-standard exception (3); kind 0 returns GPDLA_OK.  Every
 * int-returning entry point ends in the same handlers: the caller gets GPDLA_ERR_HOST and a message,
 * never a C++ exception (which would end MATLAB / Python).  Needs no GPU. */
int gpdla_debug_throw(int kind);

#ifdef __cplusplus
}
#endif
#endif
