// gpdla.hip -- host side of libgpdla.so: the C-ABI of include/gpdla.h over the HIP kernels in
// sweep_kernels.hpp.  No torch types, no CPU compute path: if the device is missing every compute
// entry point fails with GPDLA_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <stdexcept>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "../../include/gpdla.h"
#include "../../include/gpdla_lyman_series.h"
#include "multi_kernels.hpp"
#include "sweep_multi_slim_kernel.hpp"
#include "sweep_slim_kernel.hpp"
#include "sweep_split_kernel.hpp"
#include "sweep_split_slim_kernel.hpp"
#include "training_kernels.hpp"
#include "training_mfma_kernels.hpp"

using namespace gpdla;

// Superseded kernels and the environment switches that select them live in a SECOND library only:
// libgpdla_legacy.so = this file built with -DGPDLA_WITH_LEGACY (gp_dla_detection_amd/_lib.py:
// build_legacy), loaded through GPDLA_LIB_PATH by tests/test_gpu_record_classes.py, tools/ab*.sh and
// tools/check_training_legacy.py for bit-identity tests and A/B timing.  The product library reads no
// environment variable: no stray variable can select a slower kernel, and the superseded kernels
// are not in its code object.
//   GPDLA_EXPANDED_RECORDS  the sweeps on pre-expanded records (k_sweep / k_sweep_split / k_sweep_multi*)
//   GPDLA_SPLIT_LEGACY      20 < k <= 40 with every wave of a group repeating the Voigt / weight arithmetic
//   GPDLA_TRAIN_SPLITS      "H,H2,GS": the three splits of the training objective
//   GPDLA_TRAIN_FACTOR_LDS  k_train_factor<40> (LDS broadcasts) instead of k_train_factor16<40>
//   GPDLA_TRAIN_LEGACY      the round-1 training kernel (one block per slot of quasars)
#ifdef GPDLA_WITH_LEGACY
#define GPDLA_LEGACY_SWITCH(name, var) static const bool name = std::getenv(var) != nullptr
#else
#define GPDLA_LEGACY_SWITCH(name, var) constexpr bool name = false
#endif

namespace {

thread_local std::string t_error = "";

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  t_error = buf;
  return code;
}

// No C++ exception leaves the library: one that reaches an extern "C" frame ends the host process
// (MATLAB through a MEX gateway, Python through ctypes).  Every int-returning entry point is a
// function-try-block closed by this: std::bad_alloc / std::length_error of a host container and
// anything else unexpected become GPDLA_ERR_HOST with a message.
#define GPDLA_NO_THROW                                                                                    \
  catch (const std::bad_alloc &) { return fail(GPDLA_ERR_HOST, "out of host memory"); }                   \
  catch (const std::exception &e) { return fail(GPDLA_ERR_HOST, "unexpected C++ exception: %s", e.what()); } \
  catch (...) { return fail(GPDLA_ERR_HOST, "unexpected C++ exception"); }

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(GPDLA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                  __FILE__, __LINE__);                                                     \
  } while (0)

int select_device(int device_id) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(GPDLA_ERR_NO_DEVICE, "no HIP device available (%s); libgpdla has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= n)
    return fail(GPDLA_ERR_NO_DEVICE, "device_id %d out of range [0, %d)", device_id, n);
  HIP_TRY(hipSetDevice(device_id));
  return GPDLA_OK;
}

// Lyman-series tables -> __constant__ memory, once per device.
std::mutex g_table_mutex;
bool g_table_loaded[64] = {false};

// Damping parameters y_j = gamma_j / (sqrt2 sigma) and the accurate-tier polynomial tables built from
// them (near_tables.hpp); host copy, built once per process.
std::vector<double> g_near_host;
double g_line_y[kMaxLines];

void ensure_near_host() {  // caller holds g_table_mutex
  if (!g_near_host.empty()) return;
#define GP_GAM0(i, wl, f, G, lead, gam) gam,
  const double gam[] = {GPDLA_LYMAN_SERIES(GP_GAM0)};
#undef GP_GAM0
  const double sigma = GPDLA_GAUSS_SIGMA_CGS;
  for (int i = 0; i < kMaxLines; ++i) g_line_y[i] = gam[i] / std::sqrt(2.0) / sigma;
  build_near_tables(g_line_y, kMaxLines, g_near_host);
}

int ensure_line_table(int device_id) {
  std::lock_guard<std::mutex> lock(g_table_mutex);
  if (device_id < 64 && g_table_loaded[device_id]) return GPDLA_OK;
  ensure_near_host();
  double *d_near = nullptr;  // lives as long as the process (one per device)
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_near), g_near_host.size() * sizeof(double)));
  HIP_TRY(hipMemcpy(d_near, g_near_host.data(), g_near_host.size() * sizeof(double), hipMemcpyHostToDevice));
  LineTable t;
  t.near_poly = d_near;
#define GP_WL(i, wl, f, G, lead, gam) wl,
#define GP_LEAD(i, wl, f, G, lead, gam) lead,
#define GP_GAM(i, wl, f, G, lead, gam) gam,
#define GP_OSC(i, wl, f, G, lead, gam) f,
  const double wl[] = {GPDLA_LYMAN_SERIES(GP_WL)};
  const double lead[] = {GPDLA_LYMAN_SERIES(GP_LEAD)};
  const double gam[] = {GPDLA_LYMAN_SERIES(GP_GAM)};
  const double osc[] = {GPDLA_LYMAN_SERIES(GP_OSC)};
  const double taps[] = GPDLA_INSTRUMENT_PROFILE;
  const double sigma = GPDLA_GAUSS_SIGMA_CGS;
  for (int i = 0; i < kMaxLines; ++i) {
    t.wavelength_cm[i] = wl[i];
    t.leading[i] = lead[i];
    t.osc[i] = osc[i];
    t.y[i] = gam[i] / std::sqrt(2.0) / sigma;
    t.y2[i] = t.y[i] * t.y[i];
    t.cwing[i] = lead[i] * t.y[i];
    t.t2[i] = kE2 - 2.0 * t.y2[i];
    t.wing[i] = {GPDLA_SPEED_OF_LIGHT_CGS / wl[i] / 1e8 / (std::sqrt(2.0) * sigma), t.y2[i], t.cwing[i], 0.0};
  }
  for (int i = 0; i < 7; ++i) t.taps[i] = taps[i];
  t.c = GPDLA_SPEED_OF_LIGHT_CGS;
  t.inv_sqrt2_sigma = 1.0 / (std::sqrt(2.0) * sigma);
  t.inv_sqrt2pi_sigma = 1.0 / (std::sqrt(2.0 * 3.14159265358979323846) * sigma);
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_lines), &t, sizeof(t)));
  if (device_id < 64) g_table_loaded[device_id] = true;
  return GPDLA_OK;
}

template <typename T>
int dev_alloc(T **p, size_t count) {
  *p = nullptr;
  if (count == 0) count = 1;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)));
  return GPDLA_OK;
}

template <typename T>
int upload(T **p, const T *host, size_t count, hipStream_t st) {
  int rc = dev_alloc(p, count);
  if (rc) return rc;
  if (count) HIP_TRY(hipMemcpyAsync(*p, host, count * sizeof(T), hipMemcpyHostToDevice, st));
  return GPDLA_OK;
}

void dev_free(void *p) {
  if (p) (void)hipFree(p);
}

// Drains a stream when it leaves scope.  Declared AFTER the host buffers that asynchronous copies on
// that stream write into, so that on every exit path -- the error returns included -- the copies have
// finished before those buffers are destroyed (a copy engine still writing into a freed std::vector
// would corrupt the heap).
struct StreamDrain {
  hipStream_t stream;
  ~StreamDrain() { (void)hipStreamSynchronize(stream); }
};

}  // namespace

struct gpdla_context {
  int device_id = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // Copy streams: uploads and downloads run beside a sweep in flight on `stream`, so a host
  // pipeline can upload batch i+1 and download batch i-1 while batch i is swept (one thread each:
  // the entry points of ONE context may be called concurrently as long as each batch is touched
  // by one thread at a time).
  hipStream_t up_stream = nullptr, down_stream = nullptr;
  std::mutex mu;  // guards `batches`
  // model
  bool has_model = false;
  ModelDev model{};
  double *d_rest = nullptr, *d_mu = nullptr, *d_M = nullptr, *d_log_omega = nullptr;
  // samples
  bool has_samples = false;
  int64_t S = 0;
  double *d_offset = nullptr, *d_nhi = nullptr, *d_log_nhi = nullptr, *d_lls_nhi = nullptr;
  int32_t *d_perm = nullptr;
  gpdla_config cfg{};
  // timing
  bool timing = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool have_timing = false;
  // batches uploaded through this context and not yet destroyed.  A batch points back at its
  // context; destroying the context first orphans them (ctx = nullptr) instead of leaving that
  // pointer dangling, so gpdla_batch_destroy is safe in either order.
  std::vector<gpdla_batch *> batches;
  // The multi-DLA profile table (k_profiles -> k_sweep_multi, up to cfg.multi_profile_bytes, 16 GiB
  // by default) is scratch of one process call: it belongs to the context, is allocated once and
  // grows only.  multi_mu keeps two threads' gpdla_batch_process_multi calls on this context from
  // interleaving their launches (the launches of one call are ordered on `stream`).
  double *d_prof = nullptr;
  size_t prof_capacity = 0;  // doubles
  std::mutex multi_mu;
};

struct gpdla_batch {
  gpdla_context *ctx = nullptr;
  int device_id = 0;
  hipEvent_t ev_done = nullptr;  // recorded on the compute stream behind the last kernel of a process call
  // capacities (elements) of the device arrays below: gpdla_batch_reload re-fills a batch in
  // place and reallocates only what has grown, so a pipeline's batch slots do no hipMalloc/hipFree
  // (hipFree waits for the whole device) in the steady state
  // Every array below except the record pool is carved out of ONE device allocation (arena): a
  // batch slot costs two hipMalloc / hipFree in its life, not eighteen (a hipFree waits for the
  // whole device; on the PCIe-inclusive path the frees of three slots were 1.5 % of a 2048-quasar run)
  void *arena = nullptr;
  struct {
    size_t arena = 0, records = 0;  // bytes; elements
  } cap;
  // Record plan (plan_records): the K-step records of the batch's quasars live in ONE pool of at most
  // cfg.record_pool_bytes; quasars are taken in dealing order (h_order: decreasing length) and cut
  // into groups whose records fit, each group built and swept in turn.
  std::vector<int32_t> h_order;                            // host copy of d_order
  std::vector<int64_t> h_recs;                             // records a quasar occupies (K-steps + 1), by quasar
  std::vector<int64_t> h_rec_off;                          // planned pool offset (in records), by quasar
  std::vector<std::pair<int64_t, int64_t>> groups;         // [g0, g1) ranges of h_order
  int64_t *d_rec_off = nullptr;
  int64_t plan_per_step = 0, plan_budget = -1, plan_pool_records = 0;  // what the current plan was made for
  int64_t nq = 0, S = 0, total_pix = 0;
  int64_t *d_offsets = nullptr;
  double *d_wl = nullptr, *d_flux = nullptr, *d_nv = nullptr, *d_z = nullptr;
  uint8_t *d_mask = nullptr;
  double *d_lp_no = nullptr, *d_lp_dla = nullptr;
  QuasarMeta *d_meta = nullptr;
  int32_t *d_order = nullptr;  // quasar indices by decreasing pixel count (dealing order of k_sweep)
  PixelRow *d_pix = nullptr;
  double *d_Mi = nullptr, *d_lam = nullptr, *d_records = nullptr;
  double *d_sample_ll = nullptr, *d_ll_no = nullptr, *d_summary = nullptr;
  int64_t pool_rows = 0, max_pix = 0;
  int32_t k = 0, tiles_w = 0, ntiles = 0;
  // multi-DLA batch (uploaded with log_priors_lls): result tables, allocated by the first
  // gpdla_batch_process_multi and kept for the life of the batch
  int32_t md = 0;  // max_dlas the priors were uploaded for; 0 = single-DLA batch
  struct MultiBuffers *mb = nullptr;
};

struct MultiBuffers {
  double *sll_dla = nullptr, *sll_lls = nullptr, *ll_no = nullptr, *ll_dla = nullptr, *ll_lls = nullptr;
  double *map_z = nullptr, *map_n = nullptr, *map_i = nullptr;
  double *lp_lls = nullptr, *lp_dla = nullptr;
  double *post = nullptr, *scal = nullptr;  // scal: lpost_no, lpost_lls, p_no, p_lls, p_dla [5][nq]; lpost_dla after
  double *summary = nullptr;                // [nq][GPDLA_SUMMARY_COLS_MULTI(md)]
  uint32_t *base = nullptr;
  int32_t *alive = nullptr;
  // what the result tables / the prior arrays were allocated for: a re-filled batch slot keeps them
  // while it does not grow (the tables are indexed per quasar, so spare rows behind nq are unused)
  int64_t cap_nq = 0, cap_S = 0, lp_cap_nq = 0;
  int cap_md = 0, lp_cap_md = 0;
  int64_t prof_quasars = 0, prof_stride = 0;  // sub-batching of the context's profile table for this batch
  bool processed = false;
  void free_tables() {
    for (void **p : {(void **)&sll_dla, (void **)&sll_lls, (void **)&ll_no, (void **)&ll_dla, (void **)&ll_lls,
                     (void **)&map_z, (void **)&map_n, (void **)&map_i, (void **)&post, (void **)&scal,
                     (void **)&summary, (void **)&base, (void **)&alive}) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
    cap_nq = cap_S = 0;
    cap_md = 0;
  }
  ~MultiBuffers() {
    free_tables();
    if (lp_lls) (void)hipFree(lp_lls);
    if (lp_dla) (void)hipFree(lp_dla);
  }
};

extern "C" {

int gpdla_abi_version(void) { return GPDLA_ABI_VERSION; }

const char *gpdla_last_error(void) { return t_error.c_str(); }

void gpdla_default_config(gpdla_config *cfg) {
  if (!cfg) return;
  const double kms = 1000.0 / 299792458.0;  // set_parameters.m:8, :11
  cfg->min_lambda = 911.75;                 // :33
  cfg->max_lambda = 1215.75;                // :34
  cfg->lya_wavelength = 1215.6701;          // :5
  cfg->lyman_limit = 911.7633;              // :7
  cfg->pixel_spacing = 1e-4;                // :60
  cfg->max_z_cut = 3000 * kms;              // :65
  cfg->min_z_cut = 3000 * kms;              // :69
  cfg->width = 3;                           // :59
  cfg->num_lines = 3;                       // :63
  cfg->max_dlas = 4;                        // process_qsos_multiple_dlas_meanflux.m:32
  cfg->num_forest_lines = 31;               // set_parameters_multi.m:75
  cfg->min_z_separation = 3000 * kms;       // multi :33
  cfg->prev_tau_0 = 0.0023;                 // multi :36
  cfg->prev_beta = 3.65;                    // multi :37
  cfg->rng_seed = 0x9E3779B97F4A7C15ull;
  cfg->first_quasar_index = 0;
  cfg->contraction_precision = 0;
  cfg->multi_profile_bytes = 0;
  cfg->record_pool_bytes = 0;
  cfg->pipeline_slots = 0;
  cfg->max_quasars_per_batch = 0;
}

/* ------------------------------ context ------------------------------ */

int gpdla_context_create(int device_id, gpdla_context **out) try {
  if (!out) return fail(GPDLA_ERR_INVALID_ARGUMENT, "ctx out pointer is null");
  *out = nullptr;
  int rc = select_device(device_id);
  if (rc) return rc;
  rc = ensure_line_table(device_id);
  if (rc) return rc;
  gpdla_context *c = new gpdla_context();
  c->device_id = device_id;
  hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    gpdla_context_destroy(c);
    return fail(GPDLA_ERR_HIP, "hipStreamCreateWithFlags failed: %s", hipGetErrorString(e));
  }
  c->stream = c->own_stream;
  gpdla_default_config(&c->cfg);
  *out = c;
  return GPDLA_OK;
} GPDLA_NO_THROW

void gpdla_context_destroy(gpdla_context *c) {
  if (!c) return;
  (void)hipSetDevice(c->device_id);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->up_stream) (void)hipStreamSynchronize(c->up_stream);
  if (c->down_stream) (void)hipStreamSynchronize(c->down_stream);
  {
    std::lock_guard<std::mutex> lock(c->mu);
    for (gpdla_batch *b : c->batches) b->ctx = nullptr;  // orphaned: they only free their memory now
    c->batches.clear();
  }
  dev_free(c->d_rest);
  dev_free(c->d_mu);
  dev_free(c->d_M);
  dev_free(c->d_log_omega);
  dev_free(c->d_offset);
  dev_free(c->d_nhi);
  dev_free(c->d_log_nhi);
  dev_free(c->d_lls_nhi);
  dev_free(c->d_perm);
  dev_free(c->d_prof);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
  if (c->down_stream) (void)hipStreamDestroy(c->down_stream);
  delete c;
}

int gpdla_context_set_stream(gpdla_context *c, void *hip_stream) try {
  if (!c) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null context");
  hipStream_t next = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->own_stream;
  if (next != c->stream && c->d_prof) {
    // work queued on the old stream may still use the context's profile table, which the next
    // multi-DLA call (on the new stream) overwrites
    HIP_TRY(hipSetDevice(c->device_id));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  c->stream = next;
  return GPDLA_OK;
} GPDLA_NO_THROW

int gpdla_context_set_config(gpdla_context *c, const gpdla_config *cfg) try {
  if (!c || !cfg) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null context/config");
  if (cfg->width != 3)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "width must be 3 (voigt.c:229 hard-codes the 7-tap profile)");
  if (cfg->num_lines < 1 || cfg->num_lines > kMaxLines)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "num_lines %d outside [1, 31]", cfg->num_lines);
  if (cfg->contraction_precision != 0 && cfg->contraction_precision != 1)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "contraction_precision must be 0 (fp64) or 1 (fp32 study)");
  c->cfg = *cfg;
  return GPDLA_OK;
} GPDLA_NO_THROW

int gpdla_context_set_first_quasar_index(gpdla_context *c, int64_t first_quasar_index) try {
  if (!c) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null context");
  c->cfg.first_quasar_index = first_quasar_index;  // (no upload path reads this field)
  return GPDLA_OK;
} GPDLA_NO_THROW

int gpdla_context_synchronize(gpdla_context *c) try {
  if (!c) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null context");
  HIP_TRY(hipSetDevice(c->device_id));
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipStreamSynchronize(c->up_stream));
  HIP_TRY(hipStreamSynchronize(c->down_stream));
  return GPDLA_OK;
} GPDLA_NO_THROW

int gpdla_context_set_model(gpdla_context *c, const gpdla_model *m) try {
  if (!c || !m || !m->rest_wavelengths || !m->mu || !m->M || !m->log_omega)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "null model field");
  if (m->num_rest_pixels < 2 || m->k < 1)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "model needs >= 2 grid points and k >= 1");
  if (m->k > GPDLA_MAX_K) return fail(GPDLA_ERR_UNSUPPORTED, "k = %d > %d", m->k, GPDLA_MAX_K);
  HIP_TRY(hipSetDevice(c->device_id));
  HIP_TRY(hipStreamSynchronize(c->stream));
  dev_free(c->d_rest);
  dev_free(c->d_mu);
  dev_free(c->d_M);
  dev_free(c->d_log_omega);
  const size_t G = (size_t)m->num_rest_pixels;
  int rc;
  if ((rc = upload(&c->d_rest, m->rest_wavelengths, G, c->stream))) return rc;
  if ((rc = upload(&c->d_mu, m->mu, G, c->stream))) return rc;
  if ((rc = upload(&c->d_M, m->M, G * m->k, c->stream))) return rc;
  if ((rc = upload(&c->d_log_omega, m->log_omega, G, c->stream))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->model.G = m->num_rest_pixels;
  c->model.k = m->k;
  c->model.rest = c->d_rest;
  c->model.mu = c->d_mu;
  c->model.M = c->d_M;
  c->model.log_omega = c->d_log_omega;
  c->model.c_0 = std::exp(m->log_c_0);      // process_qsos.m:84-86
  c->model.tau_0 = std::exp(m->log_tau_0);
  c->model.beta = std::exp(m->log_beta);
  c->has_model = true;
  return GPDLA_OK;
} GPDLA_NO_THROW

int gpdla_context_set_samples(gpdla_context *c, const gpdla_samples *s) try {
  if (!c || !s || !s->offset_samples || !s->nhi_samples || s->num_dla_samples < 1)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "null/empty samples");
  HIP_TRY(hipSetDevice(c->device_id));
  HIP_TRY(hipStreamSynchronize(c->stream));
  dev_free(c->d_offset);
  dev_free(c->d_nhi);
  dev_free(c->d_log_nhi);
  dev_free(c->d_lls_nhi);
  dev_free(c->d_perm);
  c->d_log_nhi = c->d_lls_nhi = nullptr;
  const size_t S = (size_t)s->num_dla_samples;
  // visit samples in ascending z_DLA order: z = min + (max - min) * offset is monotone in offset
  // for every quasar, so one permutation serves the whole run
  std::vector<int32_t> perm(S);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) {
    return s->offset_samples[a] < s->offset_samples[b];
  });
  int rc;
  if ((rc = upload(&c->d_offset, s->offset_samples, S, c->stream))) return rc;
  if ((rc = upload(&c->d_nhi, s->nhi_samples, S, c->stream))) return rc;
  if (s->log_nhi_samples && (rc = upload(&c->d_log_nhi, s->log_nhi_samples, S, c->stream))) return rc;
  if (s->lls_nhi_samples && (rc = upload(&c->d_lls_nhi, s->lls_nhi_samples, S, c->stream))) return rc;
  if ((rc = upload(&c->d_perm, perm.data(), S, c->stream))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->S = (int64_t)S;
  c->has_samples = true;
  return GPDLA_OK;
} GPDLA_NO_THROW

int gpdla_context_set_timing(gpdla_context *c, int enabled) try {
  if (!c) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null context");
  HIP_TRY(hipSetDevice(c->device_id));
  if (enabled && !c->ev0) {
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
  }
  c->timing = enabled != 0;
  c->have_timing = false;
  return GPDLA_OK;
} GPDLA_NO_THROW

double gpdla_context_last_sweep_ms(gpdla_context *c) {
  if (!c || !c->have_timing) return -1.0;
  (void)hipSetDevice(c->device_id);
  if (hipEventSynchronize(c->ev1) != hipSuccess) return -1.0;
  float ms = -1.f;
  if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.0;
  return (double)ms;
}

/* ------------------------------ batch ------------------------------ */

void gpdla_batch_destroy(gpdla_batch *b) {
  if (!b) return;
  (void)hipSetDevice(b->device_id);
  if (b->ctx) {
    (void)hipStreamSynchronize(b->ctx->stream);
    (void)hipStreamSynchronize(b->ctx->up_stream);
    (void)hipStreamSynchronize(b->ctx->down_stream);
    std::lock_guard<std::mutex> lock(b->ctx->mu);
    auto &v = b->ctx->batches;
    v.erase(std::remove(v.begin(), v.end(), b), v.end());
  } else {
    (void)hipDeviceSynchronize();  // the context (and its streams) went first
  }
  if (b->ev_done) (void)hipEventDestroy(b->ev_done);
  dev_free(b->arena);
  dev_free(b->d_records);
  delete b->mb;
  delete b;
}

}  // extern "C"

namespace {

// (re)allocate *p for `count` elements unless its capacity already suffices
template <typename T>
int reserve(T **p, size_t *cap, size_t count) {
  if (count == 0) count = 1;
  if (*p && *cap >= count) return GPDLA_OK;
  dev_free(*p);
  *p = nullptr;
  *cap = 0;
  int rc = dev_alloc(p, count);
  if (!rc) *cap = count;
  return rc;
}

int validate_spectra(gpdla_context *c, const gpdla_spectra *sp, int *md_out) {
  if (!c->has_model || !c->has_samples)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "set the model and the samples before uploading spectra");
  if (sp->num_quasars < 1 || !sp->offsets || !sp->wavelengths || !sp->flux || !sp->noise_variance ||
      !sp->pixel_mask || !sp->z_qsos || !sp->log_priors_no_dla || !sp->log_priors_dla)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "null/empty spectra field");
  const int md = sp->log_priors_lls ? c->cfg.max_dlas : 0;
  if (sp->log_priors_lls && (md < 1 || md > 4))
    return fail(GPDLA_ERR_UNSUPPORTED, "max_dlas = %d outside [1, 4]", md);
  for (int64_t q = 0; q < sp->num_quasars; ++q)
    if (sp->offsets[q + 1] < sp->offsets[q])
      return fail(GPDLA_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing (quasar %lld)", (long long)q);
  *md_out = md;
  return GPDLA_OK;
}

// Fill batch b (new or being reloaded) from host spectra: H2D on the context's upload stream, which
// is drained before returning (the caller's buffers and the host vectors here are consumed).
int batch_fill(gpdla_context *c, gpdla_batch *b, const gpdla_spectra *sp, int md) {
  const int64_t nq = sp->num_quasars;
  b->nq = nq;
  b->S = c->S;
  b->k = c->model.k;
  // k <= 20: 13 w-tiles + 1 u-tile (+ 2 + 4 columns on the VALU); k <= 40: 52 + 4 tiles
  b->tiles_w = b->k <= 20 ? 13 : 52;
  b->ntiles = b->k <= 20 ? kCompactTiles : 56;
  const int64_t base = sp->offsets[0];
  b->total_pix = sp->offsets[nq] - base;
  b->max_pix = 0;
  std::vector<int64_t> off(nq + 1);
  std::vector<QuasarMeta> meta(nq);
  int64_t rows = 0, lam = 0;
  for (int64_t q = 0; q <= nq; ++q) off[q] = sp->offsets[q] - base;
  for (int64_t q = 0; q < nq; ++q) {
    const int64_t npix = off[q + 1] - off[q];
    std::memset(&meta[q], 0, sizeof(QuasarMeta));
    meta[q].status = 1;
    meta[q].pix_off = rows;
    meta[q].lam_off = lam;
    rows += 4 * ((npix + 3) / 4) + 4;
    lam += ((npix + 6 + 1) / 2) * 2 + 2;
    b->max_pix = std::max(b->max_pix, npix);
  }
  b->pool_rows = rows;
  b->h_recs.resize((size_t)nq);
  for (int64_t q = 0; q < nq; ++q) b->h_recs[q] = (off[q + 1] - off[q] + 3) / 4 + 1;
  b->plan_budget = -1;  // the record plan is remade by the next process call
  if (b->md != md) {  // (reload with a different kind of batch)
    delete b->mb;
    b->mb = nullptr;
  }
  b->md = md;
  hipStream_t st = c->up_stream;
  StreamDrain drain{st};  // on every exit: nothing still reads off / meta / order / the caller's arrays
  int rc = GPDLA_OK;
  auto chk = [&](int r) { if (r && !rc) rc = r; };
  std::vector<int32_t> order((size_t)nq);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
    return off[x + 1] - off[x] > off[y + 1] - off[y];
  });
  b->h_order = order;
  // lay the arrays out in the arena (256-byte aligned), growing it when this fill needs more
  size_t need = 0;
  auto take = [&](size_t bytes) {
    const size_t at = need;
    need += (std::max<size_t>(bytes, 8) + 255) & ~(size_t)255;
    return at;
  };
  const size_t npx = (size_t)b->total_pix, nqs = (size_t)nq;
  const size_t o_offsets = take((nqs + 1) * 8), o_wl = take(npx * 8), o_flux = take(npx * 8), o_nv = take(npx * 8),
               o_mask = take(npx), o_z = take(nqs * 8), o_lp_no = take(nqs * 8), o_lp_dla = take(md ? 8 : nqs * 8),
               o_meta = take(nqs * sizeof(QuasarMeta)), o_order = take(nqs * 4), o_rec_off = take(nqs * 8),
               o_pix = take((size_t)rows * sizeof(PixelRow)), o_Mi = take((size_t)rows * b->k * 8),
               o_lam = take((size_t)lam * 8), o_sll = take(md ? 8 : nqs * b->S * 8), o_ll_no = take(md ? 8 : nqs * 8),
               o_summary = take(md ? 8 : nqs * GPDLA_SUMMARY_COLS * 8);
  if (!b->arena || b->cap.arena < need) {
    dev_free(b->arena);
    b->arena = nullptr;
    b->cap.arena = 0;
    void *p = nullptr;
#ifdef ONESHOT_EXP_TIMING
    const auto t_malloc = std::chrono::steady_clock::now();
#endif
    if (hipMalloc(&p, need) != hipSuccess) return fail(GPDLA_ERR_HIP, "hipMalloc of %zu bytes for a batch failed", need);
#ifdef ONESHOT_EXP_TIMING
    std::fprintf(stderr, "[batch] arena of %.1f MB: hipMalloc %.2f ms\n", (double)need / 1e6,
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_malloc).count());
#endif
    b->arena = p;
    b->cap.arena = need;
  }
  char *base_p = static_cast<char *>(b->arena);
  auto at = [&](size_t o) { return static_cast<void *>(base_p + o); };
  b->d_offsets = static_cast<int64_t *>(at(o_offsets));
  b->d_wl = static_cast<double *>(at(o_wl));
  b->d_flux = static_cast<double *>(at(o_flux));
  b->d_nv = static_cast<double *>(at(o_nv));
  b->d_mask = static_cast<uint8_t *>(at(o_mask));
  b->d_z = static_cast<double *>(at(o_z));
  b->d_lp_no = static_cast<double *>(at(o_lp_no));
  b->d_lp_dla = static_cast<double *>(at(o_lp_dla));
  b->d_meta = static_cast<QuasarMeta *>(at(o_meta));
  b->d_order = static_cast<int32_t *>(at(o_order));
  b->d_rec_off = static_cast<int64_t *>(at(o_rec_off));
  b->d_pix = static_cast<PixelRow *>(at(o_pix));
  b->d_Mi = static_cast<double *>(at(o_Mi));
  b->d_lam = static_cast<double *>(at(o_lam));
  b->d_sample_ll = static_cast<double *>(at(o_sll));
  b->d_ll_no = static_cast<double *>(at(o_ll_no));
  b->d_summary = static_cast<double *>(at(o_summary));
  auto put = [&](void *dst, const void *src, size_t bytes) -> int {
    if (bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
    return GPDLA_OK;
  };
  chk(put(b->d_offsets, off.data(), (nqs + 1) * 8));
  chk(put(b->d_wl, sp->wavelengths + base, npx * 8));
  chk(put(b->d_flux, sp->flux + base, npx * 8));
  chk(put(b->d_nv, sp->noise_variance + base, npx * 8));
  chk(put(b->d_mask, sp->pixel_mask + base, npx));
  chk(put(b->d_z, sp->z_qsos, nqs * 8));
  chk(put(b->d_lp_no, sp->log_priors_no_dla, nqs * 8));
  if (!md) {
    chk(put(b->d_lp_dla, sp->log_priors_dla, nqs * 8));
  } else {  // multi-DLA batch: [nq][max_dlas] DLA priors + the sub-DLA prior (multi :204-210)
    if (!b->mb) b->mb = new MultiBuffers();
    MultiBuffers &mb = *b->mb;
    mb.processed = false;  // (the result tables are kept: gpdla_batch_process_multi regrows them if needed)
    if (mb.lp_cap_nq < nq || mb.lp_cap_md != md) {
      dev_free(mb.lp_dla);
      dev_free(mb.lp_lls);
      mb.lp_dla = mb.lp_lls = nullptr;
      mb.lp_cap_nq = 0;
      chk(dev_alloc(&mb.lp_dla, nqs * md));
      chk(dev_alloc(&mb.lp_lls, nqs));
      if (!rc) {
        mb.lp_cap_nq = nq;
        mb.lp_cap_md = md;
      }
    }
    if (!rc) {
      chk(put(mb.lp_dla, sp->log_priors_dla, nqs * md * 8));
      chk(put(mb.lp_lls, sp->log_priors_lls, nqs * 8));
    }
  }
  chk(put(b->d_meta, meta.data(), nqs * sizeof(QuasarMeta)));
  chk(put(b->d_order, order.data(), nqs * 4));
  if (rc) return rc;
  if (hipStreamSynchronize(st) != hipSuccess) return fail(GPDLA_ERR_HIP, "upload synchronize failed");
  return GPDLA_OK;
}

}  // namespace

extern "C" {

int gpdla_batch_upload(gpdla_context *c, const gpdla_spectra *sp, gpdla_batch **out) try {
  if (!c || !sp || !out) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  int md = 0;
  int rc = validate_spectra(c, sp, &md);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device_id));
  gpdla_batch *b = new gpdla_batch();
  b->ctx = c;
  b->device_id = c->device_id;
  {
    std::lock_guard<std::mutex> lock(c->mu);
    c->batches.push_back(b);
  }
  if (hipEventCreateWithFlags(&b->ev_done, hipEventDisableTiming) != hipSuccess) {
    gpdla_batch_destroy(b);
    return fail(GPDLA_ERR_HIP, "hipEventCreateWithFlags failed");
  }
  if ((rc = batch_fill(c, b, sp, md))) {
    gpdla_batch_destroy(b);
    return rc;
  }
  *out = b;
  return GPDLA_OK;
} GPDLA_NO_THROW

int gpdla_batch_reload(gpdla_context *c, gpdla_batch *b, const gpdla_spectra *sp) try {
  if (!c || !b || !sp || b->ctx != c) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null/mismatched context or batch");
  int md = 0;
  int rc = validate_spectra(c, sp, &md);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device_id));
  // the batch's previous sweep (if any) must have finished reading what is overwritten here; its
  // download is the caller's to have completed (gpdla.h)
  HIP_TRY(hipEventSynchronize(b->ev_done));
  HIP_TRY(hipStreamSynchronize(c->down_stream));
  return batch_fill(c, b, sp, md);  // on failure the batch stays valid to destroy, not to process
} GPDLA_NO_THROW

}  // extern "C"

namespace {

template <typename T, int WAVES, int NTW, int TS, int CH, int TW, int LINES>
int launch_sweep(gpdla_context *c, gpdla_batch *b, SweepArgs args) {
  constexpr int groups = WAVES / TS;
  const int L = args.num_lines;
  const size_t RD = (size_t)record_doubles(b->ntiles, sizeof(T) == 4);
  const size_t stage_doubles = 2 * (size_t)CH * RD;
  const size_t epi_doubles = (size_t)groups * EpilogueShape<TW, TS>::SPP * EpilogueShape<TW, TS>::stride(logical_tiles(b->ntiles));
  // the epilogue reuses the whole dynamic array (stage buffers, then rings etc.: all dead by then)
  const size_t loop_doubles = stage_doubles + (size_t)WAVES * kSamplesPerWave * kRing2 + kExpTab +
                              (size_t)groups * kSamplesPerWave * L;
  const size_t lds = std::max(loop_doubles, epi_doubles + kExpTab) * sizeof(double);  // (the epilogue rows start after the exp table)
  if (lds > 160 * 1024) return fail(GPDLA_ERR_UNSUPPORTED, "sweep needs %zu B of LDS", lds);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweep<T, WAVES, NTW, TS, CH, TW, LINES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  args.blocks_per_quasar = (int32_t)((b->S + 1 + groups * kSamplesPerWave - 1) / (groups * kSamplesPerWave));
  const int64_t nblocks = 8 * ((b->nq + 7) / 8) * (int64_t)args.blocks_per_quasar;
  if (nblocks > 2147483647LL) return fail(GPDLA_ERR_UNSUPPORTED, "batch too large for one launch");
  if (c->timing) HIP_TRY(hipEventRecord(c->ev0, c->stream));
  hipLaunchKernelGGL((k_sweep<T, WAVES, NTW, TS, CH, TW, LINES>), dim3((unsigned)nblocks), dim3(WAVES * 64), lds, c->stream, args);
  HIP_TRY(hipGetLastError());
  if (c->timing) {
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    c->have_timing = true;
  }
  return GPDLA_OK;
}

// k_sweep_slim: k <= 20, fp64, slim records (LINES = 3, or 0: the line count of the configuration)
template <int LINES>
int launch_sweep_slim(gpdla_context *c, gpdla_batch *b, SweepArgs args) {
  const size_t lds = (size_t)kSlimLdsDoubles * sizeof(double);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweep_slim<LINES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  args.blocks_per_quasar = (int32_t)((b->S + 1 + kSweepWaves * kSamplesPerWave - 1) / (kSweepWaves * kSamplesPerWave));
  const int64_t nblocks = 8 * ((b->nq + 7) / 8) * (int64_t)args.blocks_per_quasar;
  if (nblocks > 2147483647LL) return fail(GPDLA_ERR_UNSUPPORTED, "batch too large for one launch");
  if (c->timing) HIP_TRY(hipEventRecord(c->ev0, c->stream));
  hipLaunchKernelGGL(k_sweep_slim<LINES>, dim3((unsigned)nblocks), dim3(kSweepWaves * 64), lds, c->stream, args);
  HIP_TRY(hipGetLastError());
  if (c->timing) {
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    c->have_timing = true;
  }
  return GPDLA_OK;
}

// k_sweep_split: 20 < k <= 40 in fp64 (shared Voigt/weight pipeline across the four tile-split waves)
template <int LINES>
int launch_sweep_split(gpdla_context *c, gpdla_batch *b, SweepArgs args) {
  const size_t loop_doubles = sweep_split_lds_doubles(LINES > 0 ? 0 : args.num_lines);
  using ES = EpilogueShape<52, 4>;
  const size_t epi_doubles = kExpTab + (size_t)2 * ES::SPP * ES::stride(56);
  const size_t lds = std::max(loop_doubles, epi_doubles) * sizeof(double);
  if (lds > 160 * 1024) return fail(GPDLA_ERR_UNSUPPORTED, "split sweep needs %zu B of LDS", lds);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweep_split<LINES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  args.blocks_per_quasar = (int32_t)((b->S + 1 + 2 * kSamplesPerWave - 1) / (2 * kSamplesPerWave));
  const int64_t nblocks = 8 * ((b->nq + 7) / 8) * (int64_t)args.blocks_per_quasar;
  if (nblocks > 2147483647LL) return fail(GPDLA_ERR_UNSUPPORTED, "batch too large for one launch");
  if (c->timing) HIP_TRY(hipEventRecord(c->ev0, c->stream));
  hipLaunchKernelGGL((k_sweep_split<LINES>), dim3((unsigned)nblocks), dim3(512), lds, c->stream, args);
  HIP_TRY(hipGetLastError());
  if (c->timing) {
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    c->have_timing = true;
  }
  return GPDLA_OK;
}

// k_sweep_split_slim: 20 < k <= 40 in fp64 on slim records (tile split over eight waves, B operands
// formed in registers)
template <int LINES>
int launch_sweep_split_slim(gpdla_context *c, gpdla_batch *b, SweepArgs args) {
  const size_t loop_doubles = sweep_split_slim_lds_doubles(false);
  using ES = EpilogueShape<52, 4>;
  const size_t epi_doubles = kExpTab + (size_t)2 * ES::SPP * ES::stride(56);
  const size_t lds = std::max(loop_doubles, epi_doubles) * sizeof(double);
  if (lds > 160 * 1024) return fail(GPDLA_ERR_UNSUPPORTED, "split sweep needs %zu B of LDS", lds);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweep_split_slim<LINES, 0, SweepArgs>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  args.blocks_per_quasar = (int32_t)((b->S + 1 + 2 * kSamplesPerWave - 1) / (2 * kSamplesPerWave));
  const int64_t nblocks = 8 * ((args.nq + 7) / 8) * (int64_t)args.blocks_per_quasar;
  if (nblocks > 2147483647LL) return fail(GPDLA_ERR_UNSUPPORTED, "batch too large for one launch");
  if (c->timing) HIP_TRY(hipEventRecord(c->ev0, c->stream));
  hipLaunchKernelGGL((k_sweep_split_slim<LINES, 0, SweepArgs>), dim3((unsigned)nblocks), dim3(512), lds, c->stream, args);
  HIP_TRY(hipGetLastError());
  if (c->timing) {
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    c->have_timing = true;
  }
  return GPDLA_OK;
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {

// Plan the record pool of a batch for records of `per_step` doubles: offsets per quasar, groups of
// quasars (in dealing order) whose records fit the pool budget, the pool itself.  Remade only when
// the record class, the budget or the batch's contents changed.
int plan_records(gpdla_context *c, gpdla_batch *b, int64_t per_step, bool single_group) {
  const int64_t budget_bytes = c->cfg.record_pool_bytes > 0 ? c->cfg.record_pool_bytes : (int64_t)16 << 30;
  const int64_t budget = single_group ? INT64_MAX : std::max<int64_t>(1, budget_bytes / (per_step * 8));
  if (b->plan_per_step != per_step || b->plan_budget != budget) {
    const int64_t nq = b->nq;
    // h_rec_off is the source of an asynchronous copy enqueued by the previous plan, in front of
    // that process call's kernels: it may be rewritten once they have run (a no-op after a reload,
    // which has waited for the same event)
    HIP_TRY(hipEventSynchronize(b->ev_done));
    b->h_rec_off.assign((size_t)nq, 0);
    b->groups.clear();
    int64_t cur = 0, g0 = 0, most = 0;
    for (int64_t i = 0; i < nq; ++i) {
      const int64_t q = b->h_order[(size_t)i], n = b->h_recs[(size_t)q];
      if (cur > 0 && cur + n > budget) {
        b->groups.emplace_back(g0, i);
        most = std::max(most, cur);
        g0 = i;
        cur = 0;
      }
      b->h_rec_off[(size_t)q] = cur;
      cur += n;
    }
    b->groups.emplace_back(g0, nq);
    most = std::max(most, cur);
    b->plan_pool_records = most + kRecordPoolPad;
    b->plan_per_step = per_step;
    b->plan_budget = budget;
    // (h_rec_off lives as long as the batch: the copy may complete after this call returns)
    HIP_TRY(hipMemcpyAsync(b->d_rec_off, b->h_rec_off.data(), (size_t)nq * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
  }
  return reserve(&b->d_records, &b->cap.records, (size_t)b->plan_pool_records * (size_t)per_step);
}

// k_prepare for a batch (multi: the mean-flux / Lyman-series variant)
int launch_prepare(gpdla_context *c, gpdla_batch *b, bool multi) {
  hipStream_t st = c->stream;
  Config cfg;
  cfg.min_lambda = c->cfg.min_lambda;
  cfg.max_lambda = c->cfg.max_lambda;
  cfg.lya_wavelength = c->cfg.lya_wavelength;
  cfg.lyman_limit = c->cfg.lyman_limit;
  cfg.pixel_spacing = c->cfg.pixel_spacing;
  cfg.max_z_cut = c->cfg.max_z_cut;
  cfg.min_z_cut = c->cfg.min_z_cut;
  cfg.num_lines = c->cfg.num_lines;
  PrepareArgs pa;
  pa.nq = b->nq;
  pa.offsets = b->d_offsets;
  pa.wavelengths = b->d_wl;
  pa.flux = b->d_flux;
  pa.noise_variance = b->d_nv;
  pa.pixel_mask = b->d_mask;
  pa.z_qsos = b->d_z;
  pa.model = c->model;
  pa.cfg = cfg;
  pa.meta = b->d_meta;
  pa.pix = b->d_pix;
  pa.Mi = b->d_Mi;
  pa.lam_pad = b->d_lam;
  pa.rec_off = b->d_rec_off;
  pa.multi = multi ? 1 : 0;
  pa.num_forest_lines = c->cfg.num_forest_lines;
  pa.prev_tau_0 = c->cfg.prev_tau_0;
  pa.prev_beta = c->cfg.prev_beta;
  hipLaunchKernelGGL(k_prepare, dim3((unsigned)b->nq), dim3(256), 0, st, pa);
  HIP_TRY(hipGetLastError());
  return GPDLA_OK;
}

// The K-step records of the quasars h_order[g0 .. g1) into the pool, in one of three classes:
// pre-expanded MFMA tiles (k_sweep and the legacy / diagnostic paths), the 896-byte records of the
// k <= 20 slim sweeps, the 1536-byte records of the k <= 40 slim sweeps.
enum RecordClass { kRecExpanded = 0, kRecSlim20 = 1, kRecSlim40 = 2 };
int64_t record_class_doubles(RecordClass rc, int ntiles, bool f32_tiles) {
  return rc == kRecSlim20 ? kSlimRec : rc == kRecSlim40 ? kS40Rec : record_doubles(ntiles, f32_tiles ? 1 : 0);
}
int launch_build_records(gpdla_context *c, gpdla_batch *b, int64_t g0, int64_t g1, bool f32_tiles, RecordClass cls) {
  const bool slim = cls != kRecExpanded;
  BuildRecordsArgs ba;
  ba.meta = b->d_meta;
  ba.pix = b->d_pix;
  ba.Mi = b->d_Mi;
  ba.lam_pad = b->d_lam;
  ba.records = b->d_records;
  ba.k = b->k;
  ba.tiles_w = b->tiles_w;
  ba.ntiles = b->ntiles;
  ba.blocks_per_quasar = slim ? 4 : 16;
  ba.f32_tiles = f32_tiles ? 1 : 0;
  ba.order = b->d_order + g0;
  const unsigned grid = (unsigned)((g1 - g0) * ba.blocks_per_quasar);
  if (cls == kRecSlim20)
    hipLaunchKernelGGL(k_build_slim_records, dim3(grid), dim3(256), 0, c->stream, ba);
  else if (cls == kRecSlim40)
    hipLaunchKernelGGL(k_build_slim40_records, dim3(grid), dim3(256), 0, c->stream, ba);
  else
    hipLaunchKernelGGL(k_build_records, dim3(grid), dim3(256), 0, c->stream, ba);
  HIP_TRY(hipGetLastError());
  return GPDLA_OK;
}

}  // namespace

extern "C" {

int gpdla_batch_process(gpdla_context *c, gpdla_batch *b) try {
  if (!c || !b || b->ctx != c) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null/mismatched context or batch");
  if (b->S != c->S || b->k != c->model.k)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "model/samples changed after the batch was uploaded");
  if (b->md) return fail(GPDLA_ERR_INVALID_ARGUMENT, "multi-DLA batch: use gpdla_batch_process_multi");
  HIP_TRY(hipSetDevice(c->device_id));
  hipStream_t st = c->stream;
  const int num_lines = c->cfg.num_lines;
  // k <= 20, fp64: slim step records, vech(m m') formed inside the sweep (k_sweep_slim; three lines at
  // compile time, any other count at run time).
  // GPDLA_EXPANDED_RECORDS=1 (diagnostic): the pre-expanded records of k_sweep, for A/B timing.
  GPDLA_LEGACY_SWITCH(expanded, "GPDLA_EXPANDED_RECORDS");
  const bool f32 = c->cfg.contraction_precision == 1;
  // GPDLA_SPLIT_LEGACY=1 (diagnostic): the k_sweep form of 20 < k <= 40 in which every wave of a group
  // repeats the Voigt/weight arithmetic, for A/B timing against k_sweep_split
  GPDLA_LEGACY_SWITCH(legacy, "GPDLA_SPLIT_LEGACY");
  const bool slim = b->k <= 20 && !f32 && !expanded;
  // 20 < k <= 40, fp64: slim records as well (k_sweep_split_slim); GPDLA_EXPANDED_RECORDS=1 keeps
  // k_sweep_split on the pre-expanded 29-KiB records
  const bool slim40 = b->k > 20 && !f32 && !expanded && !legacy;
  const RecordClass cls = slim ? kRecSlim20 : slim40 ? kRecSlim40 : kRecExpanded;
  if (b->k > 40) return fail(GPDLA_ERR_UNSUPPORTED, "k = %d needs %d B tiles (max 56)", b->k, b->ntiles);
  int rc = plan_records(c, b, record_class_doubles(cls, b->ntiles, false), false);
  if (rc) return rc;
  if ((rc = launch_prepare(c, b, false))) return rc;

  // NaN pre-fill, as process_qsos.m:74-82 does for quasars that are skipped
  HIP_TRY(hipMemsetAsync(b->d_sample_ll, 0xFF, (size_t)b->nq * b->S * sizeof(double), st));
  HIP_TRY(hipMemsetAsync(b->d_ll_no, 0xFF, (size_t)b->nq * sizeof(double), st));

  SweepArgs sa;
  sa.meta = b->d_meta;
  sa.records = b->d_records;
  sa.lam_pad = b->d_lam;
  sa.offset_samples = c->d_offset;
  sa.nhi_samples = c->d_nhi;
  sa.perm = c->d_perm;
  sa.pix = b->d_pix;
  sa.S = b->S;
  sa.k = b->k;
  sa.tiles_w = b->tiles_w;
  sa.ntiles = b->ntiles;
  sa.num_lines = num_lines;
  sa.sample_ll = b->d_sample_ll;
  sa.ll_no_dla = b->d_ll_no;
  sa.blocks_per_quasar = 0;  // set by launch_sweep
  const bool three = num_lines == 3;
  // the timed region of gpdla_context_last_sweep_ms spans the sweeps of all groups (one group unless
  // the records exceed cfg.record_pool_bytes)
  const bool timing = c->timing;
  if (timing) HIP_TRY(hipEventRecord(c->ev0, st));
  c->timing = false;
  for (const auto &g : b->groups) {
    if ((rc = launch_build_records(c, b, g.first, g.second, f32, cls))) break;
    sa.order = b->d_order + g.first;
    sa.nq = g.second - g.first;
    if (slim) {
      rc = three ? launch_sweep_slim<3>(c, b, sa) : launch_sweep_slim<0>(c, b, sa);
    } else if (b->k <= 20) {  // compact class: 13 w-tiles + 1 u-tile on the matrix cores, 2 + 4 columns on the VALU
      if (!f32) {  // (fp64 takes k_sweep_slim above: the pre-expanded fp64 forms are in libgpdla_legacy.so only)
#ifdef GPDLA_WITH_LEGACY
        rc = three ? launch_sweep<double, 8, 14, 1, 8, 13, 3>(c, b, sa) : launch_sweep<double, 8, 14, 1, 4, 13, 0>(c, b, sa);
#else
        rc = fail(GPDLA_ERR_UNSUPPORTED, "pre-expanded fp64 records are in libgpdla_legacy.so only");
#endif
      }
      else rc = three ? launch_sweep<float, 8, 14, 1, 8, 13, 3>(c, b, sa) : launch_sweep<float, 8, 14, 1, 4, 13, 0>(c, b, sa);
    } else if (slim40) {  // 52 w-tiles (<= 820 columns) + 3 u-tiles split over the 8 waves of a block
      rc = three ? launch_sweep_split_slim<3>(c, b, sa) : launch_sweep_split_slim<0>(c, b, sa);
    } else if (!f32) {  // (legacy library only) the same tiles pre-expanded in the records, split over the 4 waves of a sample group
#ifdef GPDLA_WITH_LEGACY
      if (legacy)
        rc = three ? launch_sweep<double, 8, 14, 4, 2, 52, 3>(c, b, sa) : launch_sweep<double, 8, 14, 4, 1, 52, 0>(c, b, sa);
      else
        rc = three ? launch_sweep_split<3>(c, b, sa) : launch_sweep_split<0>(c, b, sa);
#else
      rc = fail(GPDLA_ERR_UNSUPPORTED, "pre-expanded records at 20 < k <= 40 are in libgpdla_legacy.so only");
#endif
    } else {  // fp32: 224 accumulator registers fit one wave (4-wave blocks, one wave per SIMD)
      rc = three ? launch_sweep<float, 4, 56, 1, 4, 52, 3>(c, b, sa) : launch_sweep<float, 4, 56, 1, 4, 52, 0>(c, b, sa);
    }
    if (rc) break;
  }
  c->timing = timing;
  if (rc) return rc;
  if (timing) {
    HIP_TRY(hipEventRecord(c->ev1, st));
    c->have_timing = true;
  }

  EvidenceArgs ea;
  ea.meta = b->d_meta;
  ea.sample_ll = b->d_sample_ll;
  ea.ll_no_dla = b->d_ll_no;
  ea.log_prior_no_dla = b->d_lp_no;
  ea.log_prior_dla = b->d_lp_dla;
  ea.offset_samples = c->d_offset;
  ea.nhi_samples = c->d_nhi;
  ea.log_nhi_samples = c->d_log_nhi;
  ea.S = b->S;
  ea.summary = b->d_summary;
  hipLaunchKernelGGL(k_evidence, dim3((unsigned)b->nq), dim3(256), 0, st, ea);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(b->ev_done, st));
  return GPDLA_OK;
} GPDLA_NO_THROW

int gpdla_batch_summary_device_ptr(gpdla_batch *b, double **table, int64_t *nq) try {
  if (!b || !table) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null argument");
  if (b->md) return fail(GPDLA_ERR_INVALID_ARGUMENT, "multi-DLA batch: use gpdla_batch_summary_multi_device_ptr");
  *table = b->d_summary;
  if (nq) *nq = b->nq;
  return GPDLA_OK;
} GPDLA_NO_THROW

int gpdla_batch_samples_device_ptr(gpdla_batch *b, double **table, int64_t *nq, int64_t *S) try {
  if (!b || !table) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null argument");
  if (b->md) return fail(GPDLA_ERR_INVALID_ARGUMENT, "multi-DLA batch: use gpdla_batch_samples_multi_device_ptr");
  *table = b->d_sample_ll;
  if (nq) *nq = b->nq;
  if (S) *S = b->S;
  return GPDLA_OK;
} GPDLA_NO_THROW

int gpdla_batch_download(gpdla_context *c, gpdla_batch *b, gpdla_results *r) try {
  if (!c || !b || !r || b->ctx != c) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null/mismatched argument");
  if (b->md) return fail(GPDLA_ERR_INVALID_ARGUMENT, "multi-DLA batch: use gpdla_batch_download_multi");
  HIP_TRY(hipSetDevice(c->device_id));
  const size_t nq = (size_t)b->nq;
  std::vector<double> summary(nq * GPDLA_SUMMARY_COLS);
  std::vector<QuasarMeta> meta(nq);
  // on the download stream, behind this batch's last kernel: a sweep of ANOTHER batch that is in
  // flight on the compute stream is not waited for
  hipStream_t ds = c->down_stream;
  StreamDrain drain{ds};
  HIP_TRY(hipStreamWaitEvent(ds, b->ev_done, 0));
  HIP_TRY(hipMemcpyAsync(summary.data(), b->d_summary, summary.size() * sizeof(double),
                         hipMemcpyDeviceToHost, ds));
  HIP_TRY(hipMemcpyAsync(meta.data(), b->d_meta, nq * sizeof(QuasarMeta), hipMemcpyDeviceToHost, ds));
  if (r->sample_log_likelihoods_dla)
    HIP_TRY(hipMemcpyAsync(r->sample_log_likelihoods_dla, b->d_sample_ll, nq * b->S * sizeof(double),
                           hipMemcpyDeviceToHost, ds));
  HIP_TRY(hipStreamSynchronize(ds));
  for (size_t q = 0; q < nq; ++q) {
    const double *s = &summary[q * GPDLA_SUMMARY_COLS];
    if (r->min_z_dlas) r->min_z_dlas[q] = s[0];
    if (r->max_z_dlas) r->max_z_dlas[q] = s[1];
    if (r->log_likelihoods_no_dla) r->log_likelihoods_no_dla[q] = s[4];
    if (r->log_likelihoods_dla) r->log_likelihoods_dla[q] = s[5];
    if (r->log_posteriors_no_dla) r->log_posteriors_no_dla[q] = s[6];
    if (r->log_posteriors_dla) r->log_posteriors_dla[q] = s[7];
    if (r->model_posteriors) {
      r->model_posteriors[2 * q] = s[8];
      r->model_posteriors[2 * q + 1] = s[9];
    }
    if (r->p_no_dlas) r->p_no_dlas[q] = s[10];
    if (r->p_dlas) r->p_dlas[q] = s[11];
    if (r->status) r->status[q] = meta[q].status;
    if (r->MAP_inds) r->MAP_inds[q] = s[12];
    if (r->MAP_z_dlas) r->MAP_z_dlas[q] = s[13];
    if (r->MAP_log_nhis) r->MAP_log_nhis[q] = s[14];
  }
  return GPDLA_OK;
} GPDLA_NO_THROW

/* ------------------------------ stand-alone surfaces ------------------------------ */

int gpdla_voigt(const double *lambdas, int64_t n_padded, double z, double N, int num_lines,
                double *profile_out, int device_id) try {
  if (!lambdas || !profile_out) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null pointer");
  if (n_padded <= 6) return fail(GPDLA_ERR_INVALID_ARGUMENT, "n_padded = %lld must exceed 2*width = 6", (long long)n_padded);
  if (num_lines < 1 || num_lines > kMaxLines)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "num_lines %d outside [1, 31]", num_lines);
  int rc = select_device(device_id);
  if (rc) return rc;
  if ((rc = ensure_line_table(device_id))) return rc;
  double *d_lam = nullptr, *d_raw = nullptr, *d_prof = nullptr;
  const int64_t n_out = n_padded - 6;
  auto cleanup = [&]() {
    dev_free(d_lam);
    dev_free(d_raw);
    dev_free(d_prof);
  };
  if ((rc = dev_alloc(&d_lam, (size_t)n_padded)) || (rc = dev_alloc(&d_raw, (size_t)n_padded)) ||
      (rc = dev_alloc(&d_prof, (size_t)n_out))) {
    cleanup();
    return rc;
  }
  hipError_t e = hipMemcpy(d_lam, lambdas, (size_t)n_padded * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_voigt_raw, dim3((unsigned)((n_padded + 255) / 256)), dim3(256), 0, 0, d_lam,
                       n_padded, z, N, num_lines, d_raw);
    hipLaunchKernelGGL(k_voigt_broaden, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, 0, d_raw,
                       n_out, d_prof);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipMemcpy(profile_out, d_prof, (size_t)n_out * sizeof(double), hipMemcpyDeviceToHost);
  cleanup();
  if (e != hipSuccess) return fail(GPDLA_ERR_HIP, "gpdla_voigt: %s", hipGetErrorString(e));
  return GPDLA_OK;
} GPDLA_NO_THROW

int gpdla_log_mvnpdf_low_rank(const double *y, const double *mu, const double *M, const double *d,
                              int64_t n, int k, double *log_p, int device_id) try {
  if (!y || !mu || !M || !d || !log_p) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null pointer");
  if (n < 1 || k < 1) return fail(GPDLA_ERR_INVALID_ARGUMENT, "n and k must be positive");
  if (k > 256) return fail(GPDLA_ERR_UNSUPPORTED, "k = %d too large", k);
  int rc = select_device(device_id);
  if (rc) return rc;
  // one packed upload (y | mu | d | M), one packed download (log_p | status)
  double *buf = nullptr;
  const size_t nn = (size_t)n, ws = (size_t)k * (k + 1) / 2 + k + 2;
  const size_t n_in = 3 * nn + nn * k, total = n_in + ws + 2;
  if ((rc = dev_alloc(&buf, total))) return rc;
  std::vector<double> host(n_in);
  std::memcpy(host.data(), y, nn * sizeof(double));
  std::memcpy(host.data() + nn, mu, nn * sizeof(double));
  std::memcpy(host.data() + 2 * nn, d, nn * sizeof(double));
  std::memcpy(host.data() + 3 * nn, M, nn * k * sizeof(double));
  double *dy = buf, *dmu = dy + nn, *dd = dmu + nn, *dM = dd + nn, *dws = dM + nn * k, *dlp = dws + ws;
  int *d_status = reinterpret_cast<int *>(dlp + 1);
  hipError_t e = hipMemcpy(buf, host.data(), n_in * sizeof(double), hipMemcpyHostToDevice);
  double back[2] = {NAN, 0.0};
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_lowrank_single, dim3(1), dim3(256), 0, 0, dy, dmu, dM, dd, n, k, dws, dlp, d_status);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(back, dlp, 2 * sizeof(double), hipMemcpyDeviceToHost);
  dev_free(buf);
  if (e != hipSuccess) return fail(GPDLA_ERR_HIP, "gpdla_log_mvnpdf_low_rank: %s", hipGetErrorString(e));
  int status;
  std::memcpy(&status, &back[1], sizeof(int));
  *log_p = back[0];
  if (status) {
    *log_p = NAN;
    return fail(GPDLA_ERR_NOT_POSITIVE_DEFINITE, "B = I + M' D^-1 M is not positive definite");
  }
  return GPDLA_OK;
} GPDLA_NO_THROW

}  // extern "C"

namespace {

template <int NTW, int TS, int CH, int TW, int ND>
int launch_sweep_multi_nd(gpdla_context *c, gpdla_batch *b, SweepMultiArgs args) {
  constexpr int groups = kSweepWaves / TS;
  const size_t RD = (size_t)record_doubles(b->ntiles, 0);
  // stage buffers during the loop; the epilogue reuses the array for its factorisation rows
  const size_t lds = std::max(2 * (size_t)CH * RD,
                              (size_t)groups * EpilogueShape<TW, TS>::SPP * EpilogueShape<TW, TS>::stride(logical_tiles(b->ntiles))) * sizeof(double);
  if (lds > 160 * 1024) return fail(GPDLA_ERR_UNSUPPORTED, "multi sweep needs %zu B of LDS", lds);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweep_multi<NTW, TS, CH, TW, ND>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  args.blocks_per_quasar = (int32_t)((args.S + 1 + groups * kSamplesPerWave - 1) / (groups * kSamplesPerWave));
  const int64_t nblocks = 8 * (((int64_t)args.nq_sub + 7) / 8) * (int64_t)args.blocks_per_quasar;
  if (nblocks > 2147483647LL) return fail(GPDLA_ERR_UNSUPPORTED, "sub-batch too large for one launch");
  hipLaunchKernelGGL((k_sweep_multi<NTW, TS, CH, TW, ND>), dim3((unsigned)nblocks), dim3(512), lds, c->stream, args);
  HIP_TRY(hipGetLastError());
  return GPDLA_OK;
}

template <int NTW, int TS, int CH, int TW>
int launch_sweep_multi(gpdla_context *c, gpdla_batch *b, const SweepMultiArgs &args) {
  switch (args.mode == 0 ? 1 : args.mode) {  // profiles multiplied per sample
    case 1: return launch_sweep_multi_nd<NTW, TS, CH, TW, 1>(c, b, args);
    case 2: return launch_sweep_multi_nd<NTW, TS, CH, TW, 2>(c, b, args);
    case 3: return launch_sweep_multi_nd<NTW, TS, CH, TW, 3>(c, b, args);
    case 4: return launch_sweep_multi_nd<NTW, TS, CH, TW, 4>(c, b, args);
    default: return fail(GPDLA_ERR_UNSUPPORTED, "max_dlas = %d > 4", args.mode);
  }
}

template <int ND>
int launch_sweep_multi_split_nd(gpdla_context *c, SweepMultiArgs args) {
  using ES = EpilogueShape<52, 4>;
  const size_t lds = std::max(sweep_multi_split_lds_doubles(), (size_t)2 * ES::SPP * ES::stride(56)) * sizeof(double);
  if (lds > 160 * 1024) return fail(GPDLA_ERR_UNSUPPORTED, "multi split sweep needs %zu B of LDS", lds);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweep_multi_split<ND>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  args.blocks_per_quasar = (int32_t)((args.S + 1 + 2 * kSamplesPerWave - 1) / (2 * kSamplesPerWave));
  const int64_t nblocks = 8 * (((int64_t)args.nq_sub + 7) / 8) * (int64_t)args.blocks_per_quasar;
  if (nblocks > 2147483647LL) return fail(GPDLA_ERR_UNSUPPORTED, "sub-batch too large for one launch");
  hipLaunchKernelGGL(k_sweep_multi_split<ND>, dim3((unsigned)nblocks), dim3(512), lds, c->stream, args);
  HIP_TRY(hipGetLastError());
  return GPDLA_OK;
}

template <int ND>
int launch_sweep_multi_slim_nd(gpdla_context *c, SweepMultiArgs args) {
  const size_t lds = sweep_multi_slim_lds_doubles() * sizeof(double);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweep_multi_slim<ND>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  args.blocks_per_quasar = (int32_t)((args.S + 1 + kSweepWaves * kSamplesPerWave - 1) / (kSweepWaves * kSamplesPerWave));
  const int64_t nblocks = 8 * (((int64_t)args.nq_sub + 7) / 8) * (int64_t)args.blocks_per_quasar;
  if (nblocks > 2147483647LL) return fail(GPDLA_ERR_UNSUPPORTED, "sub-batch too large for one launch");
  hipLaunchKernelGGL(k_sweep_multi_slim<ND>, dim3((unsigned)nblocks), dim3(512), lds, c->stream, args);
  HIP_TRY(hipGetLastError());
  return GPDLA_OK;
}

// k <= 20 on slim records (k_sweep_multi with k_sweep_slim's in-sweep vech expansion)
int launch_sweep_multi_slim(gpdla_context *c, const SweepMultiArgs &args) {
  switch (args.mode == 0 ? 1 : args.mode) {
    case 1: return launch_sweep_multi_slim_nd<1>(c, args);
    case 2: return launch_sweep_multi_slim_nd<2>(c, args);
    case 3: return launch_sweep_multi_slim_nd<3>(c, args);
    case 4: return launch_sweep_multi_slim_nd<4>(c, args);
    default: return fail(GPDLA_ERR_UNSUPPORTED, "max_dlas = %d > 4", args.mode);
  }
}

template <int ND>
int launch_sweep_multi_split_slim_nd(gpdla_context *c, SweepMultiArgs args) {
  using ES = EpilogueShape<52, 4>;
  const size_t lds = std::max(sweep_split_slim_lds_doubles(true), kExpTab + (size_t)2 * ES::SPP * ES::stride(56)) * sizeof(double);
  if (lds > 160 * 1024) return fail(GPDLA_ERR_UNSUPPORTED, "multi split sweep needs %zu B of LDS", lds);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweep_split_slim<0, ND, SweepMultiArgs>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  args.blocks_per_quasar = (int32_t)((args.S + 1 + 2 * kSamplesPerWave - 1) / (2 * kSamplesPerWave));
  const int64_t nblocks = 8 * (((int64_t)args.nq_sub + 7) / 8) * (int64_t)args.blocks_per_quasar;
  if (nblocks > 2147483647LL) return fail(GPDLA_ERR_UNSUPPORTED, "sub-batch too large for one launch");
  hipLaunchKernelGGL((k_sweep_split_slim<0, ND, SweepMultiArgs>), dim3((unsigned)nblocks), dim3(512), lds, c->stream, args);
  HIP_TRY(hipGetLastError());
  return GPDLA_OK;
}

// 20 < k <= 40 on slim records (k_sweep_split_slim with gathers in place of the Voigt stages)
int launch_sweep_multi_split_slim(gpdla_context *c, const SweepMultiArgs &args) {
  switch (args.mode == 0 ? 1 : args.mode) {
    case 1: return launch_sweep_multi_split_slim_nd<1>(c, args);
    case 2: return launch_sweep_multi_split_slim_nd<2>(c, args);
    case 3: return launch_sweep_multi_split_slim_nd<3>(c, args);
    case 4: return launch_sweep_multi_split_slim_nd<4>(c, args);
    default: return fail(GPDLA_ERR_UNSUPPORTED, "max_dlas = %d > 4", args.mode);
  }
}

#ifdef GPDLA_WITH_LEGACY
// 20 < k <= 40: the roles of a sample group share the gathers and weights (k_sweep_multi_split)
int launch_sweep_multi_split(gpdla_context *c, const SweepMultiArgs &args) {
  switch (args.mode == 0 ? 1 : args.mode) {
    case 1: return launch_sweep_multi_split_nd<1>(c, args);
    case 2: return launch_sweep_multi_split_nd<2>(c, args);
    case 3: return launch_sweep_multi_split_nd<3>(c, args);
    case 4: return launch_sweep_multi_split_nd<4>(c, args);
    default: return fail(GPDLA_ERR_UNSUPPORTED, "max_dlas = %d > 4", args.mode);
  }
}
#endif

// Result tables of a multi-DLA batch (allocated on first use, kept while the batch does not grow)
// and the context's profile table.
int multi_alloc(gpdla_batch *b) {
  MultiBuffers &mb = *b->mb;
  gpdla_context *c = b->ctx;
  const size_t nqs = (size_t)b->nq, S = (size_t)b->S;
  const int md = b->md;
  int rc = GPDLA_OK;
  auto chk = [&](int x) { if (x && !rc) rc = x; };
  if (mb.sll_dla && (b->nq > mb.cap_nq || b->S != mb.cap_S || md != mb.cap_md)) mb.free_tables();
  if (!mb.sll_dla) {
    chk(dev_alloc(&mb.sll_dla, nqs * md * S));
    chk(dev_alloc(&mb.sll_lls, nqs * S));
    chk(dev_alloc(&mb.ll_no, nqs));
    chk(dev_alloc(&mb.ll_dla, nqs * md));
    chk(dev_alloc(&mb.ll_lls, nqs));
    chk(dev_alloc(&mb.map_z, nqs * md * md));
    chk(dev_alloc(&mb.map_n, nqs * md * md));
    chk(dev_alloc(&mb.map_i, nqs * md * md));
    chk(dev_alloc(&mb.base, nqs * (md > 1 ? md - 1 : 1) * S));
    chk(dev_alloc(&mb.alive, nqs));
    chk(dev_alloc(&mb.post, nqs * (2 + md)));
    chk(dev_alloc(&mb.scal, nqs * (5 + md)));
    chk(dev_alloc(&mb.summary, nqs * GPDLA_SUMMARY_COLS_MULTI(md)));
    if (rc) {
      mb.free_tables();
      return rc;
    }
    mb.cap_nq = b->nq;
    mb.cap_S = b->S;
    mb.cap_md = md;
  }
  // profile table: rows of `stride` doubles, 2 S rows per quasar, sub-batches sized to the budget
  const int64_t stride = ((4 * ((b->max_pix + 3) / 4) + 4 + 15) / 16) * 16;
  const double per_q = 2.0 * (double)S * (double)stride * sizeof(double);
  const double budget = c->cfg.multi_profile_bytes > 0 ? (double)c->cfg.multi_profile_bytes : 16.0 * 1073741824.0;
  int64_t nq_sub = (int64_t)std::max(1.0, std::floor(budget / per_q));
  nq_sub = std::min(nq_sub, b->nq);
  const size_t need = (size_t)nq_sub * 2 * S * stride;
  if (c->prof_capacity < need) {
    HIP_TRY(hipStreamSynchronize(c->stream));  // an earlier call's sweeps may still read the old table
    dev_free(c->d_prof);
    c->d_prof = nullptr;
    c->prof_capacity = 0;
    if ((rc = dev_alloc(&c->d_prof, need))) return rc;
    c->prof_capacity = need;
    // (Touching the fresh table once here -- a 15 GB hipMemsetAsync -- was tried in round 4 against the
    // slower first k_profiles launch into never-written memory: no change in the call, 46.35 vs 46.34 ms,
    // k_profiles still 4.2-5.1 ms; the memset costs what it saves.  Dropped.)
  }
  mb.prof_quasars = nq_sub;
  mb.prof_stride = stride;
  return GPDLA_OK;
}

}  // namespace

extern "C" {

int gpdla_batch_process_multi(gpdla_context *c, gpdla_batch *b, const uint32_t *base_in) try {
  if (!c || !b || b->ctx != c) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null/mismatched context or batch");
  if (!b->md) return fail(GPDLA_ERR_INVALID_ARGUMENT, "not a multi-DLA batch (upload it with log_priors_lls)");
  if (b->S != c->S || b->k != c->model.k)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "model/samples changed after the batch was uploaded");
  const int64_t nq = b->nq, S = b->S;
  const int md = b->md;
  if (md != c->cfg.max_dlas)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "max_dlas changed after the batch was uploaded (%d -> %d)", md, c->cfg.max_dlas);
  if (!c->d_lls_nhi || !c->d_log_nhi)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "multi-DLA needs lls_nhi_samples and log_nhi_samples");
  const size_t nqs = (size_t)nq;
  const size_t nbase = nqs * (md > 1 ? md - 1 : 0) * S;
  if (base_in)  // 0 = never drawn (the sample is NaN); anything above S cannot be an index
    for (size_t e = 0; e < nbase; ++e)
      if (base_in[e] > (uint64_t)S)
        return fail(GPDLA_ERR_INVALID_ARGUMENT, "base_sample_inds[%zu] = %u exceeds num_dla_samples = %lld",
                    e, base_in[e], (long long)S);
  HIP_TRY(hipSetDevice(c->device_id));
  std::lock_guard<std::mutex> multi_lock(c->multi_mu);  // the profile table is the context's: one call's launches at a time
  hipStream_t st = c->stream;
  int rc = multi_alloc(b);
  if (rc) return rc;
  MultiBuffers &mb = *b->mb;
  if (c->timing) HIP_TRY(hipEventRecord(c->ev0, st));
  // (the multi-DLA sweeps walk the batch in profile-table sub-batches of their own: all records
  // are built up front, one group)
  // GPDLA_SPLIT_LEGACY=1 (diagnostic): the k <= 40 form in which every wave of a group gathers and
  // weighs for itself; GPDLA_EXPANDED_RECORDS=1 (diagnostic): the sweeps on pre-expanded records
  GPDLA_LEGACY_SWITCH(legacy, "GPDLA_SPLIT_LEGACY");
  GPDLA_LEGACY_SWITCH(expanded, "GPDLA_EXPANDED_RECORDS");
  const RecordClass cls = expanded ? kRecExpanded : b->k <= 20 ? kRecSlim20 : !legacy ? kRecSlim40 : kRecExpanded;
  if ((rc = plan_records(c, b, record_class_doubles(cls, b->ntiles, false), true))) return rc;
  if ((rc = launch_prepare(c, b, true))) return rc;
  if ((rc = launch_build_records(c, b, 0, b->nq, false, cls))) return rc;
  // NaN pre-fill (multi :110-131); alive != 0; base = 0 (multi :116) or the caller's indices
  HIP_TRY(hipMemsetAsync(mb.sll_dla, 0xFF, nqs * md * S * sizeof(double), st));
  HIP_TRY(hipMemsetAsync(mb.sll_lls, 0xFF, nqs * S * sizeof(double), st));
  HIP_TRY(hipMemsetAsync(mb.ll_no, 0xFF, nqs * sizeof(double), st));
  HIP_TRY(hipMemsetAsync(mb.ll_dla, 0xFF, nqs * md * sizeof(double), st));
  HIP_TRY(hipMemsetAsync(mb.ll_lls, 0xFF, nqs * sizeof(double), st));
  HIP_TRY(hipMemsetAsync(mb.map_z, 0xFF, nqs * md * md * sizeof(double), st));
  HIP_TRY(hipMemsetAsync(mb.map_n, 0xFF, nqs * md * md * sizeof(double), st));
  HIP_TRY(hipMemsetAsync(mb.map_i, 0xFF, nqs * md * md * sizeof(double), st));
  HIP_TRY(hipMemsetAsync(mb.alive, 0x01, nqs * sizeof(int32_t), st));
  if (base_in && nbase) {
    // the caller's buffer is consumed before this call returns (gpdla.h): a pageable source may
    // otherwise still be read by the copy engine after the caller has freed it
    HIP_TRY(hipMemcpyAsync(mb.base, base_in, nbase * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
  } else
    HIP_TRY(hipMemsetAsync(mb.base, 0, (nbase ? nbase : 1) * sizeof(uint32_t), st));

  const int64_t nq_sub = mb.prof_quasars, stride = mb.prof_stride;
  const double log_S = std::log((double)S);
  for (int64_t q0 = 0; q0 < nq; q0 += nq_sub) {
    const int32_t nsub = (int32_t)std::min(nq_sub, nq - q0);
    ProfilesArgs pa;
    pa.meta = b->d_meta;
    pa.lam_pad = b->d_lam;
    pa.offset_samples = c->d_offset;
    pa.nhi_samples = c->d_nhi;
    pa.lls_nhi_samples = c->d_lls_nhi;
    pa.perm = c->d_perm;
    pa.S = S;
    pa.num_lines = c->cfg.num_lines;
    pa.q0 = q0;
    pa.nq_sub = nsub;
    pa.stride = stride;
    pa.prof = c->d_prof;
    const int64_t waves = (int64_t)nsub * ((S + 63) / 64);  // one wave per 64 samples, both kinds
    hipLaunchKernelGGL(k_profiles, dim3((unsigned)((waves + kProfWaves - 1) / kProfWaves)), dim3(kProfWaves * 64), 0, st, pa);
    HIP_TRY(hipGetLastError());
    for (int mode = 1; mode <= md; ++mode) {
      for (int pass = (mode == 1 ? 0 : 1); pass < 2; ++pass) {  // the LLS pass (mode 0) rides with model 1
        SweepMultiArgs sa;
        sa.meta = b->d_meta;
        sa.records = b->d_records;
        sa.prof = c->d_prof;
        sa.base_inds = mb.base;
        sa.alive = mb.alive;
        sa.S = S;
        sa.q0 = q0;
        sa.stride = stride;
        sa.nq_sub = nsub;
        sa.blocks_per_quasar = 0;
        sa.k = b->k;
        sa.mode = pass == 0 ? 0 : mode;
        sa.max_dlas = md;
        sa.log_S = log_S;
        sa.sample_ll_dla = mb.sll_dla;
        sa.sample_ll_lls = mb.sll_lls;
        sa.ll_no_dla = mb.ll_no;
        sa.pix = b->d_pix;
#ifdef GPDLA_WITH_LEGACY
        rc = cls == kRecSlim20 ? launch_sweep_multi_slim(c, sa)
             : b->k <= 20 ? launch_sweep_multi<14, 1, 8, 13>(c, b, sa)
             : legacy ? launch_sweep_multi<14, 4, 1, 52>(c, b, sa)
             : cls == kRecSlim40 ? launch_sweep_multi_split_slim(c, sa) : launch_sweep_multi_split(c, sa);
#else
        rc = cls == kRecSlim20 ? launch_sweep_multi_slim(c, sa) : launch_sweep_multi_split_slim(c, sa);
#endif
        if (rc) return rc;
      }
      // evidence, MAP, early-exit flags for the quasars of this sub-batch
      MultiEvidenceArgs ea;
      ea.meta = b->d_meta + q0;
      ea.offset_samples = c->d_offset;
      ea.log_nhi_samples = c->d_log_nhi;
      ea.base_inds = mb.base + (size_t)q0 * (md > 1 ? md - 1 : 0) * S;
      ea.alive = mb.alive + q0;
      ea.S = S;
      ea.nd = mode;
      ea.max_dlas = md;
      ea.min_z_separation = c->cfg.min_z_separation;
      ea.log_S = log_S;
      ea.sample_ll_dla = mb.sll_dla + (size_t)q0 * md * S;
      ea.sample_ll_lls = mb.sll_lls + (size_t)q0 * S;
      ea.ll_dla = mb.ll_dla + (size_t)q0 * md;
      ea.ll_lls = mb.ll_lls + q0;
      ea.map_z = mb.map_z + (size_t)q0 * md * md;
      ea.map_lognhi = mb.map_n + (size_t)q0 * md * md;
      ea.map_ind = mb.map_i + (size_t)q0 * md * md;
      hipLaunchKernelGGL(k_multi_evidence, dim3((unsigned)nsub), dim3(256), 0, st, ea);
      HIP_TRY(hipGetLastError());
      if (mode < md && !base_in) {  // multi :467-472
        MultiResampleArgs ra;
        ra.meta = b->d_meta + q0;
        ra.alive = mb.alive + q0;
        ra.sample_ll_dla = mb.sll_dla + (size_t)q0 * md * S;
        ra.S = S;
        ra.first_quasar_index = c->cfg.first_quasar_index + q0;
        ra.seed = c->cfg.rng_seed;
        ra.nd = mode;
        ra.max_dlas = md;
        ra.base_inds = mb.base + (size_t)q0 * (md - 1) * S;
        const size_t lds = (size_t)S * sizeof(double);
        if (lds > 150 * 1024) return fail(GPDLA_ERR_UNSUPPORTED, "resampling supports S <= 19200");
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_multi_resample),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_multi_resample, dim3((unsigned)nsub), dim3(256), lds, st, ra);
        HIP_TRY(hipGetLastError());
      }
    }
  }
  // posteriors over (no DLA, LLS, 1..max_dlas DLAs) + the summary row
  MultiPostArgs pp;
  pp.meta = b->d_meta;
  pp.nq = nq;
  pp.max_dlas = md;
  pp.lp_no = b->d_lp_no;
  pp.lp_lls = mb.lp_lls;
  pp.lp_dla = mb.lp_dla;
  pp.ll_no = mb.ll_no;
  pp.ll_lls = mb.ll_lls;
  pp.ll_dla = mb.ll_dla;
  pp.lpost_no = mb.scal;
  pp.lpost_lls = mb.scal + nqs;
  pp.p_no = mb.scal + 2 * nqs;
  pp.p_lls = mb.scal + 3 * nqs;
  pp.p_dla = mb.scal + 4 * nqs;
  pp.lpost_dla = mb.scal + 5 * nqs;
  pp.post = mb.post;
  pp.map_z = mb.map_z;
  pp.map_lognhi = mb.map_n;
  pp.map_ind = mb.map_i;
  pp.summary = mb.summary;
  hipLaunchKernelGGL(k_multi_posteriors, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, pp);
  HIP_TRY(hipGetLastError());
  if (c->timing) {
    HIP_TRY(hipEventRecord(c->ev1, st));
    c->have_timing = true;
  }
  HIP_TRY(hipEventRecord(b->ev_done, st));
  mb.processed = true;
  return GPDLA_OK;
} GPDLA_NO_THROW

int gpdla_batch_download_multi(gpdla_context *c, gpdla_batch *b, gpdla_results_multi *r) try {
  if (!c || !b || !r || b->ctx != c) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null/mismatched argument");
  if (!b->md || !b->mb || !b->mb->processed)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "no multi-DLA results: call gpdla_batch_process_multi first");
  HIP_TRY(hipSetDevice(c->device_id));
  MultiBuffers &mb = *b->mb;
  hipStream_t st = c->down_stream;  // behind this batch's last kernel, beside other batches' sweeps
  const size_t nqs = (size_t)b->nq, S = (size_t)b->S;
  const int md = b->md;
  const size_t nbase = nqs * (md > 1 ? md - 1 : 0) * S;
  int rc = GPDLA_OK;
  auto chk = [&](int x) { if (x && !rc) rc = x; };
  std::vector<QuasarMeta> meta(nqs);
  StreamDrain drain{st};  // (also covers the caller's arrays: nothing is in flight once this returns)
  HIP_TRY(hipStreamWaitEvent(st, b->ev_done, 0));
  HIP_TRY(hipMemcpyAsync(meta.data(), b->d_meta, nqs * sizeof(QuasarMeta), hipMemcpyDeviceToHost, st));
  auto dl = [&](void *dst, const void *src, size_t bytes) -> int {
    if (dst && bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
    return GPDLA_OK;
  };
  chk(dl(r->log_likelihoods_no_dla, mb.ll_no, nqs * 8));
  chk(dl(r->sample_log_likelihoods_dla, mb.sll_dla, nqs * md * S * 8));
  chk(dl(r->sample_log_likelihoods_lls, mb.sll_lls, nqs * S * 8));
  chk(dl(r->log_likelihoods_dla, mb.ll_dla, nqs * md * 8));
  chk(dl(r->log_likelihoods_lls, mb.ll_lls, nqs * 8));
  chk(dl(r->log_posteriors_no_dla, mb.scal, nqs * 8));
  chk(dl(r->log_posteriors_lls, mb.scal + nqs, nqs * 8));
  chk(dl(r->log_posteriors_dla, mb.scal + 5 * nqs, nqs * md * 8));
  chk(dl(r->model_posteriors, mb.post, nqs * (2 + md) * 8));
  chk(dl(r->p_no_dlas, mb.scal + 2 * nqs, nqs * 8));
  chk(dl(r->p_lls, mb.scal + 3 * nqs, nqs * 8));
  chk(dl(r->p_dlas, mb.scal + 4 * nqs, nqs * 8));
  chk(dl(r->MAP_z_dlas, mb.map_z, nqs * md * md * 8));
  chk(dl(r->MAP_log_nhis, mb.map_n, nqs * md * md * 8));
  chk(dl(r->MAP_inds, mb.map_i, nqs * md * md * 8));
  chk(dl(r->base_sample_inds, mb.base, nbase * sizeof(uint32_t)));
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(st));
  for (size_t q = 0; q < nqs; ++q) {
    if (r->min_z_dlas) r->min_z_dlas[q] = meta[q].min_z_dla;
    if (r->max_z_dlas) r->max_z_dlas[q] = meta[q].max_z_dla;
    if (r->status) r->status[q] = meta[q].status;
  }
  return GPDLA_OK;
} GPDLA_NO_THROW

int gpdla_batch_summary_multi_device_ptr(gpdla_batch *b, double **table, int64_t *nq, int32_t *cols) try {
  if (!b || !table) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null argument");
  if (!b->md || !b->mb || !b->mb->summary)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "no multi-DLA results: call gpdla_batch_process_multi first");
  *table = b->mb->summary;
  if (nq) *nq = b->nq;
  if (cols) *cols = GPDLA_SUMMARY_COLS_MULTI(b->md);
  return GPDLA_OK;
} GPDLA_NO_THROW

int gpdla_batch_samples_multi_device_ptr(gpdla_batch *b, double **sll_dla, double **sll_lls,
                                         uint32_t **base) try {
  if (!b) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null argument");
  if (!b->md || !b->mb || !b->mb->sll_dla)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "no multi-DLA results: call gpdla_batch_process_multi first");
  if (sll_dla) *sll_dla = b->mb->sll_dla;
  if (sll_lls) *sll_lls = b->mb->sll_lls;
  if (base) *base = b->mb->base;
  return GPDLA_OK;
} GPDLA_NO_THROW

}  // extern "C"

/* ------------------------------ one-shot entries: the host pipeline ------------------------------ */

namespace {

// Touch every page of a caller-owned output array without changing its contents, so that the
// device-to-host copies into it do not run at page-fault speed (4 GB/s measured into untouched
// pageable memory, 10+ once the pages exist).  MADV_POPULATE_WRITE where the kernel has it.
void prefault_pages(void *p, size_t bytes) {
  if (!p || !bytes) return;
  const uintptr_t page = 4096, lo = ((uintptr_t)p + page - 1) & ~(page - 1), hi = ((uintptr_t)p + bytes) & ~(page - 1);
  if (hi <= lo) return;
#ifdef MADV_POPULATE_WRITE
  if (madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_POPULATE_WRITE) == 0) return;
#endif
  for (uintptr_t a = lo; a < hi; a += page) {
    volatile char *c = reinterpret_cast<volatile char *>(a);
    *c = *c;
  }
}

// Three stages over `nblocks` blocks of quasars and `slots` HBM-resident batch slots, the loop of
// process_qsos.m:88 as a pipeline: an upload thread fills slot i % slots with block i (once the slot's
// previous results are on the host), the calling thread launches the sweeps in order, a download
// thread copies block i's results into the caller's arrays.  The library's copy streams run beside
// the compute stream (gpdla.h, gpdla_batch_download), so while block i is swept block i+1 is
// uploaded and block i-1 downloaded.  The first error of any stage stops all three; its message
// becomes the calling thread's gpdla_last_error().
struct HostPipeline {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<char> uploaded, processed, downloaded;
  int err = GPDLA_OK;
  std::string msg;

  explicit HostPipeline(size_t n) : uploaded(n, 0), processed(n, 0), downloaded(n, 0) {}
  void raise(int rc) {  // called on the failing thread: t_error is that thread's message
    std::lock_guard<std::mutex> lock(mu);
    if (!err) {
      err = rc;
      msg = t_error;
    }
    cv.notify_all();
  }
  bool wait(const std::vector<char> &flag, size_t i) {
    std::unique_lock<std::mutex> lock(mu);
    cv.wait(lock, [&] { return err || flag[i]; });
    return !err;
  }
  void set(std::vector<char> &flag, size_t i) {
    std::lock_guard<std::mutex> lock(mu);
    flag[i] = 1;
    cv.notify_all();
  }
};

template <class Up, class Proc, class Down, class Warm>
int run_host_pipeline(size_t nblocks, size_t slots, Up up, Proc proc, Down down, Warm warm) {
  HostPipeline ps(nblocks);
  auto guarded = [&](auto &&body) {
    try {
      body();
    } catch (const std::bad_alloc &) {
      fail(GPDLA_ERR_HOST, "host pipeline: out of host memory");
      ps.raise(GPDLA_ERR_HOST);
    } catch (const std::exception &e) {
      fail(GPDLA_ERR_HOST, "host pipeline: %s", e.what());
      ps.raise(GPDLA_ERR_HOST);
    } catch (...) {  // (a stage thread that lets an exception escape ends the process)
      fail(GPDLA_ERR_HOST, "host pipeline: unexpected C++ exception");
      ps.raise(GPDLA_ERR_HOST);
    }
  };
  auto upload_stage = [&] {
    guarded([&] {
      for (size_t i = 0; i < nblocks; ++i) {
        if (i >= slots && !ps.wait(ps.downloaded, i - slots)) return;
        if (int rc = up(i, i % slots)) return ps.raise(rc);
        ps.set(ps.uploaded, i);
      }
    });
  };
  auto download_stage = [&] {
    guarded([&] {
      warm();
      for (size_t i = 0; i < nblocks; ++i) {
        if (!ps.wait(ps.processed, i)) return;
        if (int rc = down(i, i % slots)) return ps.raise(rc);
        ps.set(ps.downloaded, i);
      }
    });
  };
  // (a thread that cannot be started -- std::system_error -- must not leave the other one running, nor
  // an exception cross the C boundary: the stages that did start are told to stop and joined)
  std::thread uploader, downloader;
  try {
    uploader = std::thread(upload_stage);
    downloader = std::thread(download_stage);
  } catch (const std::exception &e) {
    fail(GPDLA_ERR_HOST, "host pipeline: cannot start a thread: %s", e.what());
    ps.raise(GPDLA_ERR_HOST);
  }
  guarded([&] {
    for (size_t i = 0; i < nblocks; ++i) {
      if (!ps.wait(ps.uploaded, i)) return;
      if (int rc = proc(i, i % slots)) return ps.raise(rc);
      ps.set(ps.processed, i);
    }
  });
  if (uploader.joinable()) uploader.join();
  if (downloader.joinable()) downloader.join();
  if (ps.err) return fail(ps.err, "%s", ps.msg.c_str());
  return GPDLA_OK;
}

// api.record_bytes_per_quasar / resident_bytes_per_quasar: what a quasar of `npix` stored pixels
// occupies in a resident batch
int64_t batch_bytes_per_quasar(int64_t npix, int k, int64_t S, int multi_models) {
  const double rows = (double)(npix + 8) * (k + 4 + 1 + 3.2) * 8.0;
  int64_t per_q = (int64_t)(rows + 8.0 * (double)S * std::max(1, 2 * multi_models));
  if (multi_models)  // the multi-DLA sweeps build all records of a batch up front
    per_q += (int64_t)(((double)npix / 4.0 + 2.0) * (k <= 20 ? 896 : 1536));
  return per_q;
}

struct BlockPlan {
  size_t slots = 1;
  std::vector<std::pair<int64_t, int64_t>> blocks;
};

int plan_blocks(int64_t nq, int64_t longest, const gpdla_config &cfg, int k, int64_t S, int multi_models, BlockPlan *plan) {
  if (cfg.pipeline_slots < 0 || cfg.max_quasars_per_batch < 0)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "pipeline_slots and max_quasars_per_batch must be >= 0");
  const int slots = cfg.pipeline_slots > 0 ? cfg.pipeline_slots : 3;
  const int64_t per_batch = cfg.max_quasars_per_batch > 0
                                ? cfg.max_quasars_per_batch
                                : gpdla_default_batch_quasars(nq, longest, k, S, slots, 0, multi_models);
  for (int64_t lo = 0; lo < nq; lo += per_batch) plan->blocks.emplace_back(lo, std::min(lo + per_batch, nq));
  // (Measured and not kept, profiles/r05_one_shot_timing.txt: a short last block -- an eighth of a block, so
  // that the one download nothing overlaps is small.  The call ends ~4 ms behind its last sweep either
  // way: that tail is the latency of the stage hand-offs and of the copies' synchronisation, not bytes.)
  plan->slots = std::min<size_t>((size_t)slots, plan->blocks.size());
  return GPDLA_OK;
}

// Where a one-shot call's spectra come from: CSR arrays (a block is a pointer shift, nothing is
// copied on the host) or one array per quasar, as preloaded_qsos.mat's cell arrays hold them (a
// block is flattened into its batch slot's staging vectors by the upload thread, beside the sweeps).
struct CsrSource {
  const gpdla_spectra *sp;
  int md;
  int validate(int64_t *longest) const {
    if (sp->num_quasars < 1 || !sp->offsets || !sp->z_qsos || !sp->log_priors_no_dla || !sp->log_priors_dla)
      return fail(GPDLA_ERR_INVALID_ARGUMENT, "null/empty spectra field");
    *longest = 1;
    for (int64_t q = 0; q < sp->num_quasars; ++q) {
      if (sp->offsets[q + 1] < sp->offsets[q])
        return fail(GPDLA_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing (quasar %lld)", (long long)q);
      *longest = std::max(*longest, sp->offsets[q + 1] - sp->offsets[q]);
    }
    return GPDLA_OK;
  }
  int64_t num_quasars() const { return sp->num_quasars; }
  int block(int64_t lo, int64_t hi, size_t, gpdla_spectra *out) const {
    *out = *sp;  // the pixel arrays are indexed through offsets
    out->num_quasars = hi - lo;
    out->offsets = sp->offsets + lo;
    out->z_qsos = sp->z_qsos + lo;
    out->log_priors_no_dla = sp->log_priors_no_dla + lo;
    out->log_priors_dla = sp->log_priors_dla + lo * (md ? md : 1);
    if (sp->log_priors_lls) out->log_priors_lls = sp->log_priors_lls + lo;
    return GPDLA_OK;
  }
};

struct CellSource {
  const gpdla_spectra_cells *sp;
  int md;
  struct Staging {
    std::vector<int64_t> offsets;
    std::vector<double> wl, flux, nv;
    std::vector<uint8_t> mask;
  };
  mutable std::vector<Staging> staging;  // one per batch slot; touched by the upload thread only
  int validate(int64_t *longest) const {
    if (sp->num_quasars < 1 || !sp->num_pixels || !sp->wavelengths || !sp->flux || !sp->noise_variance || !sp->pixel_mask ||
        !sp->z_qsos || !sp->log_priors_no_dla || !sp->log_priors_dla)
      return fail(GPDLA_ERR_INVALID_ARGUMENT, "null/empty spectra field");
    *longest = 1;
    for (int64_t q = 0; q < sp->num_quasars; ++q) {
      const int64_t n = sp->num_pixels[q];
      if (n < 0) return fail(GPDLA_ERR_INVALID_ARGUMENT, "num_pixels[%lld] is negative", (long long)q);
      if (n > 0 && (!sp->wavelengths[q] || !sp->flux[q] || !sp->noise_variance[q] || !sp->pixel_mask[q]))
        return fail(GPDLA_ERR_INVALID_ARGUMENT, "quasar %lld: null cell", (long long)q);
      *longest = std::max(*longest, n);
    }
    return GPDLA_OK;
  }
  int64_t num_quasars() const { return sp->num_quasars; }
  int block(int64_t lo, int64_t hi, size_t slot, gpdla_spectra *out) const {
    Staging &st = staging[slot];
    const size_t nq = (size_t)(hi - lo);
    st.offsets.resize(nq + 1);
    st.offsets[0] = 0;
    for (size_t q = 0; q < nq; ++q) st.offsets[q + 1] = st.offsets[q] + sp->num_pixels[lo + (int64_t)q];
    const size_t total = (size_t)st.offsets[nq];
    st.wl.resize(total);
    st.flux.resize(total);
    st.nv.resize(total);
    st.mask.resize(total);
    for (size_t q = 0; q < nq; ++q) {
      const size_t at = (size_t)st.offsets[q], n = (size_t)sp->num_pixels[lo + (int64_t)q];
      if (!n) continue;
      std::memcpy(st.wl.data() + at, sp->wavelengths[lo + (int64_t)q], n * sizeof(double));
      std::memcpy(st.flux.data() + at, sp->flux[lo + (int64_t)q], n * sizeof(double));
      std::memcpy(st.nv.data() + at, sp->noise_variance[lo + (int64_t)q], n * sizeof(double));
      std::memcpy(st.mask.data() + at, sp->pixel_mask[lo + (int64_t)q], n);
    }
    std::memset(out, 0, sizeof *out);
    out->num_quasars = (int64_t)nq;
    out->offsets = st.offsets.data();
    out->wavelengths = st.wl.data();
    out->flux = st.flux.data();
    out->noise_variance = st.nv.data();
    out->pixel_mask = st.mask.data();
    out->z_qsos = sp->z_qsos + lo;
    out->log_priors_no_dla = sp->log_priors_no_dla + lo;
    out->log_priors_dla = sp->log_priors_dla + lo * (md ? md : 1);
    out->log_priors_lls = sp->log_priors_lls ? sp->log_priors_lls + lo : nullptr;
    return GPDLA_OK;
  }
};

template <typename T>
T *shifted(T *p, int64_t rows, int64_t width) {
  return p ? p + rows * width : nullptr;
}

struct OneShot {  // context + batch slots of a one-shot call, released on every exit path
  gpdla_context *c = nullptr;
  std::vector<gpdla_batch *> batches;
  ~OneShot() {
    for (gpdla_batch *b : batches) gpdla_batch_destroy(b);
    gpdla_context_destroy(c);
  }
  int open(const gpdla_model *model, const gpdla_samples *samples, const gpdla_config &cfg, int device_id) {
    int rc = gpdla_context_create(device_id, &c);
    if (!rc) rc = gpdla_context_set_config(c, &cfg);
    if (!rc) rc = gpdla_context_set_model(c, model);
    if (!rc) rc = gpdla_context_set_samples(c, samples);
    return rc;
  }
};

// process_qsos.m:88-233 for every quasar of `src`, pipelined (run_host_pipeline)
template <class Source>
int one_shot_single(const gpdla_model *model, const gpdla_samples *samples, const Source &src, const gpdla_config *config,
                    gpdla_results *results, int device_id) {
  gpdla_config cfg;
  gpdla_default_config(&cfg);
  if (config) cfg = *config;
  int64_t longest = 1;
  int rc = src.validate(&longest);
  if (rc) return rc;
  BlockPlan plan;
  if ((rc = plan_blocks(src.num_quasars(), longest, cfg, model->k, samples->num_dla_samples, 0, &plan))) return rc;
#ifdef ONESHOT_EXP_TIMING  // (diagnostic build: where a one-shot call spends what the sweeps do not; stderr)
  using clk = std::chrono::steady_clock;
  const auto t_in = clk::now();
  auto ms_since = [&](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
  struct Report {
    clk::time_point t_in;
    double open = 0, staged0 = 0, up0 = 0, proc0 = 0, proc_last = 0, down_last = 0, pipeline = 0;
    ~Report() {
      std::fprintf(stderr, "[one-shot] context open %.2f ms | block 0 staged at %.2f, uploaded at %.2f, launched at %.2f | last launch at %.2f, "
                   "last download done at %.2f, pipeline returned at %.2f, context closed at %.2f\n", open, staged0, up0, proc0, proc_last,
                   down_last, pipeline, std::chrono::duration<double, std::milli>(clk::now() - t_in).count());
    }
  } report;  // (declared in front of `os`: destroyed behind it)
  report.t_in = t_in;
#endif
  OneShot os;
  if ((rc = os.open(model, samples, cfg, device_id))) return rc;
#ifdef ONESHOT_EXP_TIMING
  report.open = ms_since(t_in);
#endif
  os.batches.assign(plan.slots, nullptr);
  const int64_t S = samples->num_dla_samples;
  auto up = [&](size_t i, size_t slot) {
    gpdla_spectra sp;
    if (int r = src.block(plan.blocks[i].first, plan.blocks[i].second, slot, &sp)) return r;
#ifdef ONESHOT_EXP_TIMING
    if (i == 0) report.staged0 = ms_since(t_in);
#endif
    const int r = os.batches[slot] ? gpdla_batch_reload(os.c, os.batches[slot], &sp) : gpdla_batch_upload(os.c, &sp, &os.batches[slot]);
#ifdef ONESHOT_EXP_TIMING
    if (i == 0) report.up0 = ms_since(t_in);
#endif
    return r;
  };
  auto proc = [&](size_t i, size_t slot) {
    const int r = gpdla_batch_process(os.c, os.batches[slot]);
#ifdef ONESHOT_EXP_TIMING
    if (i == 0) report.proc0 = ms_since(t_in);
    if (i + 1 == plan.blocks.size()) report.proc_last = ms_since(t_in);
#else
    (void)i;
#endif
    return r;
  };
  auto down = [&](size_t i, size_t slot) {
    const int64_t lo = plan.blocks[i].first;
    gpdla_results r;
    r.min_z_dlas = shifted(results->min_z_dlas, lo, 1);
    r.max_z_dlas = shifted(results->max_z_dlas, lo, 1);
    r.log_likelihoods_no_dla = shifted(results->log_likelihoods_no_dla, lo, 1);
    r.sample_log_likelihoods_dla = shifted(results->sample_log_likelihoods_dla, lo, S);
    r.log_likelihoods_dla = shifted(results->log_likelihoods_dla, lo, 1);
    r.log_posteriors_no_dla = shifted(results->log_posteriors_no_dla, lo, 1);
    r.log_posteriors_dla = shifted(results->log_posteriors_dla, lo, 1);
    r.model_posteriors = shifted(results->model_posteriors, lo, 2);
    r.p_no_dlas = shifted(results->p_no_dlas, lo, 1);
    r.p_dlas = shifted(results->p_dlas, lo, 1);
    r.status = shifted(results->status, lo, 1);
    r.MAP_inds = shifted(results->MAP_inds, lo, 1);
    r.MAP_z_dlas = shifted(results->MAP_z_dlas, lo, 1);
    r.MAP_log_nhis = shifted(results->MAP_log_nhis, lo, 1);
    const int rd = gpdla_batch_download(os.c, os.batches[slot], &r);
#ifdef ONESHOT_EXP_TIMING
    if (i + 1 == plan.blocks.size()) report.down_last = ms_since(t_in);
#endif
    return rd;
  };
  auto warm = [&] { prefault_pages(results->sample_log_likelihoods_dla, (size_t)src.num_quasars() * S * sizeof(double)); };
  rc = run_host_pipeline(plan.blocks.size(), plan.slots, up, proc, down, warm);
#ifdef ONESHOT_EXP_TIMING
  report.pipeline = ms_since(t_in);
#endif
  return rc;
}

// multi_dlas/process_qsos_multiple_dlas_meanflux.m:141-495 for every quasar of `src`, pipelined
template <class Source>
int one_shot_multi(const gpdla_model *model, const gpdla_samples *samples, const Source &src, const uint32_t *base_sample_inds,
                   const gpdla_config *config, gpdla_results_multi *results, int device_id) {
  gpdla_config cfg;
  gpdla_default_config(&cfg);
  if (config) cfg = *config;
  const int md = cfg.max_dlas;
  if (md < 1 || md > 4) return fail(GPDLA_ERR_UNSUPPORTED, "max_dlas = %d outside [1, 4]", md);
  int64_t longest = 1;
  int rc = src.validate(&longest);
  if (rc) return rc;
  BlockPlan plan;
  if ((rc = plan_blocks(src.num_quasars(), longest, cfg, model->k, samples->num_dla_samples, md + 1, &plan))) return rc;
  OneShot os;
  if ((rc = os.open(model, samples, cfg, device_id))) return rc;
  os.batches.assign(plan.slots, nullptr);
  const int64_t S = samples->num_dla_samples, nbase_row = (int64_t)(md > 1 ? md - 1 : 0) * S;
  auto up = [&](size_t i, size_t slot) {
    gpdla_spectra sp;
    if (int r = src.block(plan.blocks[i].first, plan.blocks[i].second, slot, &sp)) return r;
    return os.batches[slot] ? gpdla_batch_reload(os.c, os.batches[slot], &sp) : gpdla_batch_upload(os.c, &sp, &os.batches[slot]);
  };
  auto proc = [&](size_t i, size_t slot) {
    const int64_t lo = plan.blocks[i].first;
    // the draws of the resampling are keyed by the quasar's index in the whole call (multi :467-472)
    int rc2 = gpdla_context_set_first_quasar_index(os.c, cfg.first_quasar_index + lo);
    if (rc2) return rc2;
    return gpdla_batch_process_multi(os.c, os.batches[slot], base_sample_inds ? base_sample_inds + lo * nbase_row : nullptr);
  };
  auto down = [&](size_t i, size_t slot) {
    const int64_t lo = plan.blocks[i].first;
    gpdla_results_multi r;
    r.min_z_dlas = shifted(results->min_z_dlas, lo, 1);
    r.max_z_dlas = shifted(results->max_z_dlas, lo, 1);
    r.log_likelihoods_no_dla = shifted(results->log_likelihoods_no_dla, lo, 1);
    r.sample_log_likelihoods_dla = shifted(results->sample_log_likelihoods_dla, lo, (int64_t)md * S);
    r.sample_log_likelihoods_lls = shifted(results->sample_log_likelihoods_lls, lo, S);
    r.log_likelihoods_dla = shifted(results->log_likelihoods_dla, lo, md);
    r.log_likelihoods_lls = shifted(results->log_likelihoods_lls, lo, 1);
    r.log_posteriors_no_dla = shifted(results->log_posteriors_no_dla, lo, 1);
    r.log_posteriors_lls = shifted(results->log_posteriors_lls, lo, 1);
    r.log_posteriors_dla = shifted(results->log_posteriors_dla, lo, md);
    r.model_posteriors = shifted(results->model_posteriors, lo, 2 + md);
    r.p_no_dlas = shifted(results->p_no_dlas, lo, 1);
    r.p_lls = shifted(results->p_lls, lo, 1);
    r.p_dlas = shifted(results->p_dlas, lo, 1);
    r.MAP_z_dlas = shifted(results->MAP_z_dlas, lo, (int64_t)md * md);
    r.MAP_log_nhis = shifted(results->MAP_log_nhis, lo, (int64_t)md * md);
    r.MAP_inds = shifted(results->MAP_inds, lo, (int64_t)md * md);
    r.base_sample_inds = shifted(results->base_sample_inds, lo, nbase_row);
    r.status = shifted(results->status, lo, 1);
    return gpdla_batch_download_multi(os.c, os.batches[slot], &r);
  };
  auto warm = [&] {
    const size_t nq = (size_t)src.num_quasars();
    prefault_pages(results->sample_log_likelihoods_dla, nq * md * S * sizeof(double));
    prefault_pages(results->sample_log_likelihoods_lls, nq * S * sizeof(double));
    prefault_pages(results->base_sample_inds, nq * nbase_row * sizeof(uint32_t));
  };
  return run_host_pipeline(plan.blocks.size(), plan.slots, up, proc, down, warm);
}

}  // namespace

extern "C" {

int64_t gpdla_default_batch_quasars(int64_t num_quasars, int64_t longest_spectrum, int k, int64_t num_dla_samples,
                                    int slots, int64_t budget_bytes, int multi_models) {
  const double budget = budget_bytes > 0 ? (double)budget_bytes : 96.0 * 1073741824.0;
  const int64_t per_q = batch_bytes_per_quasar(std::max<int64_t>(longest_spectrum, 1), k, num_dla_samples, multi_models);
  const int64_t cap = std::max<int64_t>(1, (int64_t)(budget / std::max(slots, 1) / (double)per_q));
  const int64_t want = std::max<int64_t>(128, (num_quasars + 7) / 8);
  return std::max<int64_t>(1, std::min({cap, want, (int64_t)4096}));
}

int gpdla_process_batch(const gpdla_model *model, const gpdla_samples *samples,
                        const gpdla_spectra *spectra, const gpdla_config *config,
                        gpdla_results *results, int device_id) try {
  if (!model || !samples || !spectra || !results)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "null argument");
  if (spectra->log_priors_lls)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "log_priors_lls given: use gpdla_process_batch_multi");
  return one_shot_single(model, samples, CsrSource{spectra, 0}, config, results, device_id);
} GPDLA_NO_THROW

int gpdla_process_cells(const gpdla_model *model, const gpdla_samples *samples,
                        const gpdla_spectra_cells *spectra, const gpdla_config *config,
                        gpdla_results *results, int device_id) try {
  if (!model || !samples || !spectra || !results)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "null argument");
  if (spectra->log_priors_lls)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "log_priors_lls given: use gpdla_process_cells_multi");
  CellSource src{spectra, 0, {}};
  src.staging.resize(config && config->pipeline_slots > 0 ? (size_t)config->pipeline_slots : 3);
  return one_shot_single(model, samples, src, config, results, device_id);
} GPDLA_NO_THROW

int gpdla_process_batch_multi(const gpdla_model *model, const gpdla_samples *samples,
                              const gpdla_spectra *spectra, const uint32_t *base_sample_inds,
                              const gpdla_config *config, gpdla_results_multi *results,
                              int device_id) try {
  if (!model || !samples || !spectra || !results)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "null argument");
  if (!spectra->log_priors_lls) return fail(GPDLA_ERR_INVALID_ARGUMENT, "multi-DLA needs log_priors_lls");
  gpdla_config cfg;
  gpdla_default_config(&cfg);
  if (config) cfg = *config;
  return one_shot_multi(model, samples, CsrSource{spectra, cfg.max_dlas}, base_sample_inds, config, results, device_id);
} GPDLA_NO_THROW

int gpdla_process_cells_multi(const gpdla_model *model, const gpdla_samples *samples,
                              const gpdla_spectra_cells *spectra, const uint32_t *base_sample_inds,
                              const gpdla_config *config, gpdla_results_multi *results,
                              int device_id) try {
  if (!model || !samples || !spectra || !results)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "null argument");
  if (!spectra->log_priors_lls) return fail(GPDLA_ERR_INVALID_ARGUMENT, "multi-DLA needs log_priors_lls");
  gpdla_config cfg;
  gpdla_default_config(&cfg);
  if (config) cfg = *config;
  CellSource src{spectra, cfg.max_dlas, {}};
  src.staging.resize(cfg.pipeline_slots > 0 ? (size_t)cfg.pipeline_slots : 3);
  return one_shot_multi(model, samples, src, base_sample_inds, config, results, device_id);
} GPDLA_NO_THROW

void gpdla_debug_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

int gpdla_debug_throw(int kind) try {
  if (kind == 1) throw std::bad_alloc();
  if (kind == 2) throw std::runtime_error("thrown on request");
  if (kind == 3) throw 42;
  return GPDLA_OK;
} GPDLA_NO_THROW

/* ------------------------------ training objective (N3) ------------------------------ */

}  // extern "C"

struct gpdla_training {
  int device_id = 0;
  int64_t nq = 0, G = 0, ld = 0;  // ld: row stride of the training arrays (G rounded up to 16)
  // mean-flux model's objective (gpdla_training_set_lyseries): lines.nfl > 1, d_nl = active lines per pixel
  TrainLines lines{};
  uint8_t *d_nl = nullptr;
  double *d_flux = nullptr, *d_lya = nullptr, *d_noise = nullptr, *d_loglya = nullptr;
  double *d_x = nullptr, *d_g = nullptr, *d_omega2 = nullptr, *d_f = nullptr;
  int32_t *d_flag = nullptr;
  int64_t x_capacity = 0;
  // one-block-per-slot path (k > 20, GPDLA_TRAIN_LEGACY): per-slot copies of [g | f], summed in order
  double *d_slots = nullptr;
  int64_t slots_capacity = 0;
  // workspace of the matrix-core path (training_mfma_kernels.hpp): its sizes do not depend on k.
  // ws_ready is set only after every allocation, the stream and the kernel attributes succeeded.
  bool ws_ready = false;
  int ws_class = 0;  // rank class the workspace was sized for (20 or 40)
  double *h_stage = nullptr;  // pinned host staging for x (in) and [g | f | flag] (out)
  int64_t stage_capacity = 0;
  // one evaluation = H2D of x, six kernels, D2H of [g | f | flag]: captured once per k into a
  // hipGraph and replayed (no kernel argument changes between evaluations)
  hipStream_t stream = nullptr;
  hipGraphExec_t graph = nullptr;
  int graph_k = 0;
  double *d_wB = nullptr, *d_uB = nullptr, *d_part1 = nullptr;
  double *d_recM = nullptr, *d_recP = nullptr, *d_partB = nullptr, *d_recD = nullptr, *d_recE = nullptr;
  double *d_nlogp = nullptr, *d_partD = nullptr, *d_partcol = nullptr, *d_partsc = nullptr;
};

extern "C" {

}  // extern "C"

namespace {

// The captured graph holds raw pointers to d_x, d_g, h_stage and the workspace: it is destroyed
// BEFORE any of them is freed or replaced, never after.
void training_drop_graph(gpdla_training *t) {
  if (t->stream) (void)hipStreamSynchronize(t->stream);
  if (t->graph) (void)hipGraphExecDestroy(t->graph);
  t->graph = nullptr;
  t->graph_k = 0;
}

void training_free_workspace(gpdla_training *t) {
  training_drop_graph(t);
  for (double **p : {&t->d_wB, &t->d_uB, &t->d_part1, &t->d_recM, &t->d_recP, &t->d_partB,
                     &t->d_recD, &t->d_recE, &t->d_nlogp, &t->d_partD, &t->d_partcol, &t->d_partsc}) {
    dev_free(*p);
    *p = nullptr;
  }
  if (t->stream) (void)hipStreamDestroy(t->stream);
  t->stream = nullptr;
  t->ws_ready = false;
}

}  // namespace

extern "C" {

void gpdla_training_destroy(gpdla_training *t) {
  if (!t) return;
  (void)hipSetDevice(t->device_id);
  (void)hipDeviceSynchronize();
  training_free_workspace(t);  // graph, then stream, then the buffers the graph pointed at
  for (void *p : {(void *)t->d_flux, (void *)t->d_lya, (void *)t->d_noise, (void *)t->d_x, (void *)t->d_g,
                  (void *)t->d_omega2, (void *)t->d_f, (void *)t->d_flag, (void *)t->d_loglya, (void *)t->d_slots,
                  (void *)t->d_nl})
    dev_free(p);
  if (t->h_stage) (void)hipHostFree(t->h_stage);
  delete t;
}

int gpdla_training_create(int device_id, int64_t nq, int64_t G, const double *flux, const double *lya,
                          const double *noise, gpdla_training **out) try {
  if (!out || !flux || !lya || !noise) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  if (nq < 1 || G < 1) return fail(GPDLA_ERR_INVALID_ARGUMENT, "empty training set");
  int rc = select_device(device_id);
  if (rc) return rc;
  gpdla_training *t = new gpdla_training();
  t->device_id = device_id;
  t->nq = nq;
  t->G = G;
  // MATLAB column-major [nq x G] -> quasar-major [nq][ld], ld = G rounded up to 16 pixels: the rows
  // start 128-byte aligned and end in missing pixels (NaN flux, 1 + z = 1, unit noise), so the
  // matrix-core kernels read whole 16-pixel chunks without bounds checks
  const int64_t ld = 16 * ((G + 15) / 16);
  t->ld = ld;
  const size_t n = (size_t)nq * ld;
  std::vector<double> tmp(n);
  auto up = [&](const double *src, double **dst, double pad, bool take_log) -> int {
    for (int64_t i = 0; i < nq; ++i) {
      for (int64_t p = 0; p < G; ++p) tmp[(size_t)i * ld + p] = take_log ? std::log(src[i + p * nq]) : src[i + p * nq];
      for (int64_t p = G; p < ld; ++p) tmp[(size_t)i * ld + p] = pad;
    }
    int r = dev_alloc(dst, n);
    if (r) return r;
    HIP_TRY(hipMemcpy(*dst, tmp.data(), n * sizeof(double), hipMemcpyHostToDevice));
    return GPDLA_OK;
  };
  if ((rc = up(flux, &t->d_flux, std::nan(""), false)) || (rc = up(lya, &t->d_lya, 1.0, false)) ||
      (rc = up(noise, &t->d_noise, 1.0, false)) ||
      (rc = up(lya, &t->d_loglya, 0.0, true)) ||  // log(1 + z): data, taken once
      (rc = dev_alloc(&t->d_omega2, (size_t)ld)) || (rc = dev_alloc(&t->d_f, 1)) ||
      (rc = dev_alloc(&t->d_flag, 1))) {
    gpdla_training_destroy(t);
    return rc;
  }
  *out = t;
  return GPDLA_OK;
} GPDLA_NO_THROW

}  // extern "C"

namespace {

TrainDims train_dims(const gpdla_training *t, int k) {
  TrainDims d;
  d.nq = t->nq;
  d.G = t->G;
  d.k = k;
  d.NQ16 = (t->nq + 15) / 16;
  d.PG = (t->G + 15) / 16;
  d.T = 4 * d.PG;
  d.TQ = 4 * d.NQ16;
  d.ld = 16 * d.PG;
  d.H = 6;    // 79 row blocks x 6 = 474 blocks of 4 waves for 5000 quasars (two per CU)
  d.H2 = 24;  // 20 row blocks x 24 = 480
  d.GS = 24;  // 20 pixel blocks x 24 = 480 blocks of 4 waves (59 KiB of LDS each: two per CU)
  if (k > 20) {
    d.GS = 24 * kTrWidePB;  // k_train_core_wide: ceil(77 / PB) pixel-group blocks x GS / 4 = 468 blocks at PB = 2
    // four tile groups make the contraction grids four times larger, so they need fewer splits to fill
    // the chip -- and every split is a copy of the partial sums through HBM (246 MB at H = 6, 242 MB at
    // H2 = 24): 79 x 4 x 3 = 948 and 20 x 4 x 12 = 960 blocks.  Measured (tools/train_knobs.sh 40, two
    // rounds): 6,24 -> 1.122 ms; 3,24 -> 1.075; 6,12 -> 1.077; 3,12 -> 1.03; 3,8 / 3,6 the same; 2,x worse.
    d.H = 3;
    d.H2 = 12;
  }
  // (diagnostic: GPDLA_TRAIN_SPLITS="H,H2,GS" overrides the three splits)
#ifdef GPDLA_WITH_LEGACY
  static const char *splits = std::getenv("GPDLA_TRAIN_SPLITS");
#else
  constexpr const char *splits = nullptr;
#endif
  int h = 0, h2 = 0, gs = 0;
  if (splits && std::sscanf(splits, "%d,%d,%d", &h, &h2, &gs) == 3 && h > 0 && h2 > 0 && gs > 0 && h <= 64 &&
      h2 <= 256 && gs <= 256) {
    d.H = h;
    d.H2 = h2;
    d.GS = gs;
  }
  d.H = (int32_t)std::max<int64_t>(d.H, (d.PG + kTrBuildMaxChunks - 1) / kTrBuildMaxChunks);  // a split's omega2 table fits its LDS
  d.GS = (d.GS + 3) / 4 * 4;  // k_train_core_wide: four splits per block
  return d;
}

// One evaluation of objective.m:12-75 on the matrix cores, enqueued on `st`: H2D of x from the pinned
// staging buffer, the kernels, D2H of [g | f | flag] into it.  KMAX: rank class (20 or 40).
template <int KMAX>
int training_enqueue_mfma(gpdla_training *t, int k, hipStream_t st) {
  using K = TrC<KMAX>;
  const TrainDims d = train_dims(t, k);
  const int64_t strideM = (d.T + kTrChunk) * kTrGroupD, strideD = (d.TQ + kTrChunk) * kTrGroupD;
  const int64_t G = t->G, nx = G * (k + 1) + 3;
  HIP_TRY(hipMemcpyAsync(t->d_x, t->h_stage, (size_t)nx * sizeof(double), hipMemcpyHostToDevice, st));
  TrainRecordsArgs ra;
  ra.d = d;
  ra.M = t->d_x;
  ra.recM = t->d_recM;
  ra.recP = t->d_recP;
  ra.group_stride = strideM;
  ra.not_pd = t->d_flag;
  ra.omega2 = t->d_omega2;
  hipLaunchKernelGGL(k_train_records<KMAX>, dim3(1024), dim3(256), 0, st, ra);
  TrainBuildArgs ba;  // B_q, t_q: rows = quasars, steps over pixels, w and u made on the fly
  ba.d = d;
  ba.flux = t->d_flux;
  ba.log_lya_1pz = t->d_loglya;
  ba.noise = t->d_noise;
  ba.omega2 = t->d_omega2;
  ba.x = t->d_x;
  ba.nl = t->d_nl;
  ba.lines = t->lines;
  ba.Brec = t->d_recM;
  ba.groups = K::Groups;
  ba.w_tiles = K::W;
  ba.cols = K::Cols;
  ba.group_stride = strideM;
  ba.out = t->d_partB;
  ba.part1 = t->d_part1;
  const bool ly = t->lines.nfl > 1;
  const dim3 build_grid((unsigned)(((d.NQ16 + kTrCWaves - 1) / kTrCWaves) * d.H * K::Groups));
  if (ly) hipLaunchKernelGGL(k_train_build<true>, build_grid, dim3(kTrCWaves * 64), kTrBuildLds, st, ba);
  else hipLaunchKernelGGL(k_train_build<false>, build_grid, dim3(kTrCWaves * 64), kTrBuildLds, st, ba);
  TrainFactorArgs fa;
  fa.d = d;
  fa.partB = t->d_partB;
  fa.part1 = t->d_part1;
  fa.recD = t->d_recD;
  fa.recE = t->d_recE;
  fa.nlogp = t->d_nlogp;
  fa.not_pd = t->d_flag;
  fa.group_stride = strideD;
  // k <= 40: the per-quasar algebra in registers (k_train_factor16); GPDLA_TRAIN_FACTOR_LDS=1 (diagnostic): the
  // round-3 kernel, which stays the k <= 20 form
  GPDLA_LEGACY_SWITCH(factor_lds, "GPDLA_TRAIN_FACTOR_LDS");
  const dim3 factor_grid((unsigned)((d.NQ16 * 16 + TrF<KMAX>::FQ - 1) / TrF<KMAX>::FQ));
  if constexpr (KMAX == 40) {
#ifdef GPDLA_WITH_LEGACY
    if (factor_lds) hipLaunchKernelGGL(k_train_factor<KMAX>, factor_grid, dim3(256), 0, st, fa);
    else
#endif
      hipLaunchKernelGGL(k_train_factor16<KMAX>, factor_grid, dim3(kTrF16Threads), 0, st, fa);
    (void)factor_lds;
  } else {
    hipLaunchKernelGGL(k_train_factor<KMAX>, factor_grid, dim3(256), 0, st, fa);
  }
  TrainCoreArgs co;
  co.d = d;
  co.recP = t->d_recP;
  co.recE = t->d_recE;
  co.flux = t->d_flux;
  co.log_lya_1pz = t->d_loglya;
  co.noise = t->d_noise;
  co.x = t->d_x;
  co.nl = t->d_nl;
  co.lines = t->lines;
  co.wB = t->d_wB;
  co.uB = t->d_uB;
  co.partcol = t->d_partcol;
  co.partsc = t->d_partsc;
  const dim3 core_grid((unsigned)(((d.PG + 3) / 4) * d.GS));
  if (KMAX <= 20) {
    if (ly) hipLaunchKernelGGL(k_train_core<true>, core_grid, dim3(256), kTrCoreLds, st, co);
    else hipLaunchKernelGGL(k_train_core<false>, core_grid, dim3(256), kTrCoreLds, st, co);
  } else {
    const dim3 wide_grid((unsigned)(((d.PG + kTrWidePB - 1) / kTrWidePB) * (d.GS / 4)));  // kTrWidePB pixel groups per block, four splits (train_dims keeps GS % 4 == 0)
    if (ly) hipLaunchKernelGGL(k_train_core_wide<true>, wide_grid, dim3(256), 0, st, co);
    else hipLaunchKernelGGL(k_train_core_wide<false>, wide_grid, dim3(256), 0, st, co);
  }
  TrainContractArgs ca;  // dM: rows = pixels, steps over quasars
  ca.Aw = t->d_wB;
  ca.Au = t->d_uB;
  ca.groups = K::Groups;
  ca.w_tiles = K::W;
  ca.cols = K::Cols;
  ca.Brec = t->d_recD;
  ca.R = d.PG;
  ca.steps = d.TQ;
  ca.nsplit = d.H2;
  ca.group_stride = strideD;
  ca.out = t->d_partD;
  hipLaunchKernelGGL(k_train_contract, dim3((unsigned)(((d.PG + kTrCWaves - 1) / kTrCWaves) * d.H2 * K::Groups)), dim3(kTrCWaves * 64), kTrContractLds, st, ca);
  TrainFinishArgs fi;
  fi.d = d;
  fi.M = t->d_x;
  fi.partD = t->d_partD;
  fi.partcol = t->d_partcol;
  fi.partsc = t->d_partsc;
  fi.nlogp = t->d_nlogp;
  fi.f = t->d_g + nx;        // f and the not-PD flag ride behind g: one copy back
  fi.flag_in = t->d_flag;
  fi.flag_out = t->d_g + nx + 1;
  fi.x = t->d_x;
  fi.g = t->d_g;
  hipLaunchKernelGGL(k_train_finish<KMAX>, dim3((unsigned)(G + 1)), dim3(256), 0, st, fi);
  HIP_TRY(hipMemcpyAsync(t->h_stage, t->d_g, (size_t)(nx + 2) * sizeof(double), hipMemcpyDeviceToHost, st));
  return GPDLA_OK;
}

// objective.m:12-75 on the matrix cores: value and gradient, deterministic.  x is in the pinned
// staging buffer on entry; [g | f | flag] is there on return.  The workspace is sized by the rank
// class (k <= 20: one tile group; k <= 40: four) and rebuilt when the class changes.
int training_objective_mfma(gpdla_training *t, int k, double *f, double *g) {
  const TrainDims d = train_dims(t, k);
  const int64_t G = t->G, nx = G * (k + 1) + 3;
  const int kc = k <= 20 ? 20 : 40;
  const int groups = kc == 20 ? TrC<20>::Groups : TrC<40>::Groups, cols = kc == 20 ? TrC<20>::Cols : TrC<40>::Cols,
            ks = kc == 20 ? TrC<20>::Ks : TrC<40>::Ks;
  int rc;
  if (!t->ws_ready || t->ws_class != kc) {
    training_free_workspace(t);  // another class's workspace, or what an earlier, failed attempt left behind
    auto setup = [&]() -> int {
    if ((rc = dev_alloc(&t->d_wB, (size_t)d.PG * d.TQ * 64)) || (rc = dev_alloc(&t->d_uB, (size_t)d.PG * d.TQ * 64)) ||
        (rc = dev_alloc(&t->d_part1, (size_t)d.NQ16 * 16 * d.H * 3)) ||
        (rc = dev_alloc(&t->d_recM, (size_t)groups * (d.T + kTrChunk) * kTrGroupD)) || (rc = dev_alloc(&t->d_recP, (size_t)d.PG * ks * 64)) ||
        (rc = dev_alloc(&t->d_partB, (size_t)d.NQ16 * d.H * 16 * cols)) ||
        (rc = dev_alloc(&t->d_recD, (size_t)groups * (d.TQ + kTrChunk) * kTrGroupD)) || (rc = dev_alloc(&t->d_recE, (size_t)d.NQ16 * ks * 64)) ||
        (rc = dev_alloc(&t->d_nlogp, (size_t)d.NQ16 * 16)) ||
        (rc = dev_alloc(&t->d_partD, (size_t)d.PG * d.H2 * 16 * cols)) ||
        (rc = dev_alloc(&t->d_partcol, (size_t)d.PG * d.GS * 16)) || (rc = dev_alloc(&t->d_partsc, (size_t)d.PG * d.GS * 3)))
      return rc;
    HIP_TRY(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_train_contract),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTrContractLds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_train_build<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTrBuildLds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_train_build<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTrBuildLds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_train_core<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTrCoreLds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_train_core<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTrCoreLds));
    return GPDLA_OK;
    };
    if ((rc = setup())) {
      training_free_workspace(t);
      return rc;
    }
    // the chunk padding behind each tile group of recM / recD is read (never used) by the last chunk copy
    HIP_TRY(hipMemset(t->d_recM, 0, (size_t)groups * (d.T + kTrChunk) * kTrGroupD * sizeof(double)));
    HIP_TRY(hipMemset(t->d_recD, 0, (size_t)groups * (d.TQ + kTrChunk) * kTrGroupD * sizeof(double)));
    // the columns of the padding tiles (k <= 40: 9 tiles of the last group) are never written by the contractions
    HIP_TRY(hipMemset(t->d_partB, 0, (size_t)d.NQ16 * d.H * 16 * cols * sizeof(double)));
    HIP_TRY(hipMemset(t->d_partD, 0, (size_t)d.PG * d.H2 * 16 * cols * sizeof(double)));
    t->ws_ready = true;
    t->ws_class = kc;
  }
  if (!t->graph || t->graph_k != k) {  // capture the evaluation once per k
    training_drop_graph(t);
    hipGraph_t graph = nullptr;
    HIP_TRY(hipStreamBeginCapture(t->stream, hipStreamCaptureModeThreadLocal));
    rc = kc == 20 ? training_enqueue_mfma<20>(t, k, t->stream) : training_enqueue_mfma<40>(t, k, t->stream);
    hipError_t e = hipStreamEndCapture(t->stream, &graph);
    if (rc) {
      if (graph) (void)hipGraphDestroy(graph);
      return rc;
    }
    if (e != hipSuccess) return fail(GPDLA_ERR_HIP, "training graph capture failed: %s", hipGetErrorString(e));
    e = hipGraphInstantiate(&t->graph, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(GPDLA_ERR_HIP, "training graph instantiation failed: %s", hipGetErrorString(e));
    t->graph_k = k;
  }
#ifdef TR_EXP_TIMING
  static double acc_l = 0, acc_s = 0, acc_m = 0, acc_gap = 0;
  static int n_calls = 0;
  static std::chrono::steady_clock::time_point last_end;
  auto c0 = std::chrono::steady_clock::now();
  if (n_calls) acc_gap += std::chrono::duration<double, std::micro>(c0 - last_end).count();
#endif
  HIP_TRY(hipGraphLaunch(t->graph, t->stream));
#ifdef TR_EXP_TIMING
  auto c1 = std::chrono::steady_clock::now();
#endif
  HIP_TRY(hipStreamSynchronize(t->stream));
#ifdef TR_EXP_TIMING
  auto c2 = std::chrono::steady_clock::now();
#endif
  std::memcpy(g, t->h_stage, (size_t)nx * sizeof(double));
#ifdef TR_EXP_TIMING
  auto c3 = std::chrono::steady_clock::now();
  last_end = c3;
  acc_l += std::chrono::duration<double, std::micro>(c1 - c0).count();
  acc_s += std::chrono::duration<double, std::micro>(c2 - c1).count();
  acc_m += std::chrono::duration<double, std::micro>(c3 - c2).count();
  if (++n_calls % 6 == 0) {
    std::fprintf(stderr, "[timing] launch %.1f us, sync %.1f us, memcpy-out %.1f us, between calls (python + memcpy-in) %.1f us\n",
                 acc_l / 6, acc_s / 6, acc_m / 6, acc_gap / 6);
    acc_l = acc_s = acc_m = acc_gap = 0;
  }
#endif
  *f = t->h_stage[nx];
  if (t->h_stage[nx + 1] != 0.0)
    return fail(GPDLA_ERR_NOT_POSITIVE_DEFINITE, "B = I + M' D^-1 M not positive definite for some quasar");
  return GPDLA_OK;
}

}  // namespace

extern "C" {

int gpdla_training_set_lyseries(gpdla_training *t, int num_forest_lines, const double *all_transition_wavelengths,
                                const double *all_oscillator_strengths) try {
  if (!t) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null training set");
  if (num_forest_lines < 0 || num_forest_lines > kTrMaxLines)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "num_forest_lines = %d outside [0, %d]", num_forest_lines, kTrMaxLines);
  HIP_TRY(hipSetDevice(t->device_id));
  training_drop_graph(t);  // the captured kernel arguments carry the line table
  TrainLines L{};
  if (num_forest_lines <= 1) {  // back to objective.m / spectrum_loss.m
    t->lines = L;
    return GPDLA_OK;
  }
  const double *wl = all_transition_wavelengths, *fs = all_oscillator_strengths;
  // default: the Lyman series of voigt.c:20-182 (include/gpdla_lyman_series.h) = set_parameters_multi.m:76-143,
  // wavelengths in Angstrom
#define GPDLA_LINE_WL(i, wl_cm, f, rate, lead, width) wl_cm * 1e8,
#define GPDLA_LINE_FS(i, wl_cm, f, rate, lead, width) f,
  static const double wl_default[] = {GPDLA_LYMAN_SERIES(GPDLA_LINE_WL)};
  static const double fs_default[] = {GPDLA_LYMAN_SERIES(GPDLA_LINE_FS)};
#undef GPDLA_LINE_WL
#undef GPDLA_LINE_FS
  static_assert(sizeof wl_default / sizeof wl_default[0] == kTrMaxLines, "31 Lyman lines");
  if (!wl || !fs) {
    wl = wl_default;
    fs = fs_default;
  }
  for (int l = 0; l < num_forest_lines; ++l) {
    if (!(wl[l] > 0.0) || !(fs[l] > 0.0) || (l && !(wl[l] < wl[l - 1])))
      return fail(GPDLA_ERR_INVALID_ARGUMENT, "line %d: wavelengths must be positive and decreasing, strengths positive", l + 1);
    L.coef[l] = wl[l] * fs[l] / (wl[0] * fs[0]);  // spectrum_loss_lyseries.m:34-35
    L.logr[l] = std::log(wl[0] / wl[l]);
  }
  L.nfl = num_forest_lines;
  int rc;
  if (!t->d_nl && (rc = dev_alloc(&t->d_nl, (size_t)t->nq * t->ld))) return rc;
  HIP_TRY(hipMemset(t->d_flag, 0, sizeof(int32_t)));
  TrainLinesArgs la;
  la.nq = t->nq;
  la.G = t->G;
  la.ld = t->ld;
  la.nfl = num_forest_lines;
  for (int l = 0; l < kTrMaxLines; ++l) la.wl[l] = l < num_forest_lines ? wl[l] : 1.0;
  la.lya_1pz = t->d_lya;
  la.nl = t->d_nl;
  la.not_prefix = t->d_flag;
  const int64_t n = t->nq * t->ld;
  hipLaunchKernelGGL(k_train_lines, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, la);
  HIP_TRY(hipGetLastError());
  int32_t bad = 0;
  HIP_TRY(hipMemcpy(&bad, t->d_flag, sizeof bad, hipMemcpyDeviceToHost));
  if (bad) return fail(GPDLA_ERR_UNSUPPORTED, "the active Lyman lines of some pixel are not a prefix of the series");
  t->lines = L;
  return GPDLA_OK;
} GPDLA_NO_THROW

int gpdla_training_objective(gpdla_training *t, const double *x, int k, double *f, double *g) try {
  if (!t || !x || !f || !g) return fail(GPDLA_ERR_INVALID_ARGUMENT, "null argument");
  if (k < 1 || k > GPDLA_MAX_K) return fail(GPDLA_ERR_UNSUPPORTED, "k = %d outside [1, %d]", k, GPDLA_MAX_K);
  HIP_TRY(hipSetDevice(t->device_id));
  const int64_t G = t->G;
  const int64_t nx = G * (k + 1) + 3;
  if (nx > t->x_capacity) {
    training_drop_graph(t);  // it points at the buffers replaced below
    dev_free(t->d_x);
    dev_free(t->d_g);
    t->d_x = t->d_g = nullptr;
    t->x_capacity = 0;
    if (t->h_stage) (void)hipHostFree(t->h_stage);
    t->h_stage = nullptr;
    int rc;
    if ((rc = dev_alloc(&t->d_x, (size_t)nx)) || (rc = dev_alloc(&t->d_g, (size_t)nx + 2))) return rc;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&t->h_stage), (size_t)(nx + 2) * sizeof(double), hipHostMallocDefault));
    t->x_capacity = nx;
  }
  // The three contractions on the matrix cores, ordered (deterministic) sums, one graph launch per
  // evaluation (k <= 20: one 16-tile group per contraction step; 20 < k <= 40: four).
  // GPDLA_TRAIN_LEGACY=1 (diagnostic cross-check): one block per slot of quasars, each slot adding
  // into its own copy of g, slots summed in order -- deterministic too (round 1 used fp64 atomics).
  GPDLA_LEGACY_SWITCH(legacy, "GPDLA_TRAIN_LEGACY");
  std::memcpy(t->h_stage, x, (size_t)nx * sizeof(double));
  if (!legacy) return training_objective_mfma(t, k, f, g);
#ifdef GPDLA_WITH_LEGACY
  if (t->lines.nfl > 1) return fail(GPDLA_ERR_UNSUPPORTED, "GPDLA_TRAIN_LEGACY has no Lyman-series objective");
  const int num_slots = (int)std::min<int64_t>(t->nq, 512);
  const int64_t slot_n = nx + 1;  // [g | f]
  if ((int64_t)num_slots * slot_n > t->slots_capacity) {
    dev_free(t->d_slots);
    t->d_slots = nullptr;
    t->slots_capacity = 0;
    int rc = dev_alloc(&t->d_slots, (size_t)num_slots * slot_n);
    if (rc) return rc;
    t->slots_capacity = (int64_t)num_slots * slot_n;
  }
  HIP_TRY(hipMemcpy(t->d_x, t->h_stage, (size_t)nx * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(t->d_slots, 0, (size_t)num_slots * slot_n * sizeof(double)));
  HIP_TRY(hipMemset(t->d_flag, 0, sizeof(int32_t)));
  hipLaunchKernelGGL(k_training_omega2, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, 0,
                     t->d_x + G * k, G, t->d_omega2);
  TrainingArgs a;
  a.nq = t->nq;
  a.G = G;
  a.ld = t->ld;
  a.k = k;
  a.flux = t->d_flux;
  a.lya_1pz = t->d_lya;
  a.noise = t->d_noise;
  a.M = t->d_x;
  a.omega2 = t->d_omega2;
  a.c_0 = std::exp(x[G * (k + 1)]);       // objective.m:30-32
  a.tau_0 = std::exp(x[G * (k + 1) + 1]);
  a.beta = std::exp(x[G * (k + 1) + 2]);
  a.slots = t->d_slots;
  a.not_pd = t->d_flag;
  const size_t lds = training_lds_doubles(G, k) * sizeof(double);
  if (lds > 160 * 1024) return fail(GPDLA_ERR_UNSUPPORTED, "training kernel needs %zu B of LDS", lds);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_training_loss),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_training_loss, dim3((unsigned)num_slots), dim3(256), lds, 0, a);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_training_reduce, dim3((unsigned)((slot_n + 255) / 256)), dim3(256), 0, 0, t->d_slots,
                     num_slots, slot_n, t->d_g);
  HIP_TRY(hipGetLastError());
  int32_t flag = 0;
  HIP_TRY(hipMemcpy(g, t->d_g, (size_t)nx * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(f, t->d_g + nx, sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&flag, t->d_flag, sizeof(int32_t), hipMemcpyDeviceToHost));
  if (flag) return fail(GPDLA_ERR_NOT_POSITIVE_DEFINITE, "B = I + M' D^-1 M not positive definite for some quasar");
  // priors of Kim et al. (2007) on tau0 and beta, gradient only (objective.m:59-71)
  const double tau_0_mu = 0.0023, tau_0_sigma = 0.0007, beta_mu = 3.65, beta_sigma = 0.21;
  g[G * (k + 1) + 1] += a.tau_0 * (a.tau_0 - tau_0_mu) / (tau_0_sigma * tau_0_sigma);
  g[G * (k + 1) + 2] += a.beta * (a.beta - beta_mu) / (beta_sigma * beta_sigma);
  return GPDLA_OK;
#else
  return GPDLA_OK;  // (not reached: `legacy` is false in the product library)
#endif
} GPDLA_NO_THROW

}  // extern "C"

// Host evaluation of the accurate-tier table of one Lyman line (0-based) at |x| < 32: what the sweep
// kernel computes for Re w(x + i y_line).  Needs no GPU; tests/test_near_tables.py checks it
// against mpmath.  *y_out (optional) receives the line's damping parameter.
extern "C" int gpdla_debug_near_poly(int line, double x, double *value_out, double *y_out) {
  if (line < 0 || line >= kMaxLines || !value_out || !(std::fabs(x) < 32.0))
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "gpdla_debug_near_poly: line %d, x %g", line, x);
  {
    std::lock_guard<std::mutex> lock(g_table_mutex);
    ensure_near_host();
  }
  *value_out = near_poly_host(g_near_host.data() + (size_t)line * kNearLineDoubles, std::fabs(x));
  if (y_out) *y_out = g_line_y[line];
  return GPDLA_OK;
}

// Test hook (gpdla.h): k_prepare alone, then the rows of one quasar.
extern "C" int gpdla_debug_prepared_rows(gpdla_context *c, gpdla_batch *b, int multi, int64_t quasar,
                                         double *rows_out, int64_t capacity_rows, int64_t *num_rows_out) {
  if (!c || !b || b->ctx != c || !rows_out || !num_rows_out)
    return fail(GPDLA_ERR_INVALID_ARGUMENT, "null/mismatched argument");
  if (quasar < 0 || quasar >= b->nq) return fail(GPDLA_ERR_INVALID_ARGUMENT, "quasar %lld outside the batch", (long long)quasar);
  HIP_TRY(hipSetDevice(c->device_id));
  int rc = plan_records(c, b, b->k <= 20 ? kSlimRec : record_doubles(b->ntiles, 0), true);
  if (rc) return rc;
  if ((rc = launch_prepare(c, b, multi != 0))) return rc;
  QuasarMeta m;
  HIP_TRY(hipMemcpyAsync(&m, b->d_meta + quasar, sizeof(m), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const int64_t n = std::min<int64_t>(m.n_u, capacity_rows);
  static_assert(sizeof(PixelRow) == 4 * sizeof(double), "rows_out is [n][4] doubles");
  if (n > 0) {
    HIP_TRY(hipMemcpyAsync(rows_out, b->d_pix + m.pix_off, (size_t)n * sizeof(PixelRow), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  *num_rows_out = n;
  return GPDLA_OK;
}

#ifdef GPDLA_STAMP
// Diagnostic build only (tools/stamps.sh): read and clear the per-segment wave-cycle sums.
extern "C" int gpdla_debug_stamps(unsigned long long *out) {
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gpdla::g_stamps), sizeof(zero)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(gpdla::g_stamps), zero, sizeof(zero)) != hipSuccess) return -1;
  return 0;
}
#endif
