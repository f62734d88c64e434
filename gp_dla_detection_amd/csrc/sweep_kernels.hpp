// sweep_kernels.hpp -- device side of the GP/DLA inference sweep for gfx950 (MI355X).
//
// Kernels (reference lines they replace; paths relative to the reference tree):
//   k_prepare   process_qsos.m:102-119, 138-146, 159-176   per-quasar pixel selection, GP
//                                                          interpolation, noise scaling, padding
//   k_build_records (no reference counterpart)              packs vech(m m') | m per pixel into MFMA
//                                                          B-operand tiles + the per-pixel vectors
//   k_sweep     process_qsos.m:149-151 and 185-199          fused Voigt profile -> scaled
//               + voigt.c:278-299 + log_mvnpdf_low_rank.m   low-rank Gaussian log-pdf, one
//                                                          sample per MFMA row
//   k_evidence  process_qsos.m:203-213, 224-233             log-mean-exp, posteriors
//
// Algebra of the sweep.  For one quasar and one sample with absorption a (n pixels):
//   r = y - a mu,  d = omega2 a^2 + nu,  w = a^2/d,  u = a r/d
//   B = I + Sum_p w_p m_p m_p',  v = Sum_p u_p m_p,
//   log N = -1/2 [ Sum r^2/d - v' B^-1 v + Sum log d + 2 Sum log L_jj + n log 2pi ],  B = L L'.
// This is log_mvnpdf_low_rank.m:11-32 with the k x n matrix C of :26 eliminated.  Over the S
// samples of a quasar, B and v are ONE dense contraction  [W | U] (S x n) * [P | M] (n x (k(k+1)/2
// + k))  with P_p = vech(m_p m_p') depending on the quasar only: that runs on the fp64 matrix
// cores (v_mfma_f64_16x16x4_f64), while the Voigt profile that produces W and U runs on the VALU
// and never touches HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "faddeeva.hpp"
#include "near_tables.hpp"

namespace gpdla {

typedef double d4 __attribute__((ext_vector_type(4)));

typedef float f4 __attribute__((ext_vector_type(4)));

// Precision of the [W | U] * [P | M] contraction.  double: v_mfma_f64_16x16x4_f64 (the shipped,
// parity-grade path).  float: v_mfma_f32_16x16x4_f32 on the XDL matrix cores -- BASELINE config 5's
// "fp32 mixed-precision variant with fp64 log-det accumulation": weights, quadratic form and log
// determinant stay fp64, only B and v are accumulated in fp32 (a speed/accuracy study, not parity).
// Both instructions take A[row l&15][k l>>4], B[k l>>4][col l&15]; the result maps differ:
// f64: row = (l>>4) + 4 reg;  f32: row = 4 (l>>4) + reg  (col = l&15 in both).
template <typename T> struct Mat;
template <> struct Mat<double> {
  using acc_t = d4;
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int sample_of(int jj, int r) { return jj + 4 * r; }
  static __device__ __forceinline__ int jj_of(int sample) { return sample & 3; }
  static __device__ __forceinline__ int reg_of(int sample) { return sample >> 2; }
};
template <> struct Mat<float> {
  using acc_t = f4;
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int sample_of(int jj, int r) { return 4 * jj + r; }
  static __device__ __forceinline__ int jj_of(int sample) { return sample >> 2; }
  static __device__ __forceinline__ int reg_of(int sample) { return sample & 3; }
};

constexpr int kMaxLines = 31;
constexpr int kSamplesPerWave = 16;   // rows of the 16x16x4 MFMA
constexpr double kLog2Pi = 1.83787706640934534;  // log_mvnpdf_low_rank.m:7

// Lyman-series constants in device constant memory (filled once per process from
// include/gpdla_lyman_series.h).
struct LineTable {
  double wavelength_cm[kMaxLines];  // voigt.c:31
  double leading[kMaxLines];        // voigt.c:151
  double osc[kMaxLines];            // oscillator strengths, voigt.c:66 (multi-DLA mean-flux model)
  double y[kMaxLines];              // gamma_j / (sqrt2 sigma): damping parameter of w(z)
  double y2[kMaxLines];             // y_j^2
  double cwing[kMaxLines];          // leading_j * y_j            (wing formula prefactor)
  double t2[kMaxLines];             // kE2 - 2 y_j^2: rho^2 coefficient of T(rho) - 2 y^2 rho^2 (wing formula)
  double taps[7];                   // voigt.c:242-251
  double c;                         // voigt.c:22
  double inv_sqrt2_sigma;           // 1/(sqrt2 sigma)
  double inv_sqrt2pi_sigma;         // 1/(sqrt(2 pi) sigma)
  const double *near_poly;          // [31][kNearIntervals][kNearCoef], near_tables.hpp (device memory)
  // wing tier of a run-time line count, one 32-byte entry per line (one scalar load each):
  // x_j = lambda / (1 + z) kms_j - c / (sqrt2 sigma) with kms_j = c / (wavelength_j 1e8) / (sqrt2 sigma).
  struct WingLine {
    double kms, y2, cwing, pad;
  } wing[kMaxLines];
};
__constant__ LineTable g_lines;

// Per-quasar metadata produced by k_prepare.
struct QuasarMeta {
  int32_t n_u;       // pixels in the modelled rest range (process_qsos.m:104-108)
  int32_t n_kept;    // of those, not masked (:110)
  int32_t steps;     // ceil(n_u / 4): K-steps of the contraction
  int32_t status;    // 0 ok, 1 empty, 3 a kept pixel with noise variance <= 0 or NaN (skipped)
  double min_z_dla;  // :159
  double max_z_dla;  // :160
  int64_t pix_off;   // first row of this quasar in the pixel pools (multiple of 4)
  int64_t lam_off;   // first entry in the padded-wavelength pool
  int64_t rec_off;   // first K-step record in the record pool (records of a group of quasars share the
                     // pool: PrepareArgs::rec_off; without groups it is pix_off / 4)
  // 0, or -inf when a kept pixel has noise variance +inf (a zero inverse variance the mask missed):
  // log_mvnpdf_low_rank.m:30 then sums log(inf) into the log-determinant and every log-likelihood of
  // the quasar is -inf.  The sweeps treat that pixel as neutral and add this to each result, so the
  // K-loop's fast reciprocal never sees an infinite d (it would return NaN where exact division
  // gives 0).
  double ll_bias;
};

struct PixelRow {  // one row of the per-pixel pool, on the unmasked-range grid
  double y, mu, omega2, nu;
};

struct Config {
  double min_lambda, max_lambda, lya_wavelength, lyman_limit, pixel_spacing, max_z_cut, min_z_cut;
  int32_t num_lines;
};

struct ModelDev {
  int32_t G, k;
  const double *rest, *mu, *M, *log_omega;
  double c_0, tau_0, beta;
};

// ------------------------------------------------------------------------------------------
// k_prepare: one 256-thread block per quasar.
// ------------------------------------------------------------------------------------------
struct PrepareArgs {
  int64_t nq;
  const int64_t *offsets;
  const double *wavelengths, *flux, *noise_variance;
  const uint8_t *pixel_mask;
  const double *z_qsos;
  ModelDev model;
  Config cfg;
  QuasarMeta *meta;        // [nq]  pix_off / lam_off pre-filled by the host
  PixelRow *pix;           // pool
  double *Mi;              // pool [row][k] interpolated (and zeroed for masked rows) M
  double *lam_pad;         // pool
  const int64_t *rec_off;  // [nq] record-pool offsets planned by the host (gpdla.hip plan_records)
  // multi-DLA driver only (process_qsos_multiple_dlas_meanflux.m:245-293): Lyman-series noise
  // scaling and mean-flux suppression of mu, M, omega2
  int32_t multi;
  int32_t num_forest_lines;
  double prev_tau_0, prev_beta;
};

__device__ __forceinline__ double block_reduce_minmax(double v, bool is_min, double *sh) {
  for (int o = 32; o > 0; o >>= 1) {
    double other = __shfl_xor(v, o);
    v = is_min ? fmin(v, other) : fmax(v, other);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double r = sh[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = is_min ? fmin(r, sh[w]) : fmax(r, sh[w]);
  return r;
}

__global__ __launch_bounds__(256) void k_prepare(PrepareArgs a) {
  const int q = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int s_cnt[4];
  __shared__ int s_base;
  __shared__ double s_red[4];
  const int64_t base = a.offsets[q];
  const int npix = (int)(a.offsets[q + 1] - base);
  const double z_qso = a.z_qsos[q];
  QuasarMeta m = a.meta[q];
  PixelRow *pix = a.pix + m.pix_off;
  double *Mi = a.Mi + m.pix_off * a.model.k;
  double *lam = a.lam_pad + m.lam_off;
  const int k = a.model.k, G = a.model.G;
  if (tid == 0) s_base = 0;
  __syncthreads();
  double kept_min = INFINITY, kept_max = -INFINITY, un_min = INFINITY, un_max = -INFINITY;
  int kept_count = 0;
  int nv_flags = 0;  // 1: a kept pixel with noise variance +inf; 2: one with variance <= 0 or NaN
  for (int tile = 0; tile < npix; tile += 256) {
    const int i = tile + tid;
    double wl = 0.0, rest = 0.0;
    bool in_range = false;
    if (i < npix) {
      wl = a.wavelengths[base + i];
      rest = wl / (1 + z_qso);                                             // process_qsos.m:102
      in_range = (rest >= a.cfg.min_lambda) && (rest <= a.cfg.max_lambda); // :104-105
    }
    const unsigned long long bal = __ballot(in_range);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_cnt[wave] = __popcll(bal);
    __syncthreads();
    int wbase = s_base, total = 0;
    for (int w = 0; w < 4; ++w) {
      if (w < wave) wbase += s_cnt[w];
      total += s_cnt[w];
    }
    if (in_range) {
      const int u = wbase + pre;  // order-preserving index on the unmasked grid (:108)
      lam[3 + u] = wl;
      un_min = fmin(un_min, wl);
      un_max = fmax(un_max, wl);
      const bool keep = a.pixel_mask[base + i] == 0;                       // :110, :181
      PixelRow row = {0.0, 0.0, 0.0, 1.0};  // masked: contributes r = 0, d = 1, zero B-operand row
      double mf = 1.0;                       // mean-flux factor applied to mu and M (multi only)
      // bracket rest in the model grid (griddedInterpolant 'linear', :66-71)
      int lo = 0, hi = G - 1;
      if (rest >= a.model.rest[G - 1]) lo = G - 2;
      else if (rest > a.model.rest[0]) {
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (a.model.rest[mid] <= rest) lo = mid; else hi = mid;
        }
      }
      const double t = (rest - a.model.rest[lo]) / (a.model.rest[lo + 1] - a.model.rest[lo]);
      if (keep) {
        kept_count++;
        kept_min = fmin(kept_min, wl);
        kept_max = fmax(kept_max, wl);
        row.y = a.flux[base + i];
        row.nu = a.noise_variance[base + i];
        if (!(row.nu > 0.0)) nv_flags |= 2;  // the reference's result is undefined (log of d <= 0, 0/0)
        else if (row.nu == INFINITY) nv_flags |= 1;
        row.mu = a.model.mu[lo] + (a.model.mu[lo + 1] - a.model.mu[lo]) * t;          // :138
        const double lo_om = a.model.log_omega[lo] +
                             (a.model.log_omega[lo + 1] - a.model.log_omega[lo]) * t;  // :141
        const double omega2 = exp(2 * lo_om);                                          // :142
        const double lya_z = (wl - a.cfg.lya_wavelength) / a.cfg.lya_wavelength;       // :117-119
        if (!a.multi) {
          const double sc = 1 - exp(-a.model.tau_0 * pow(1 + lya_z, a.model.beta)) + a.model.c_0; // :144
          row.omega2 = omega2 * (sc * sc);                                             // :146
        } else {
          // multi :245-263: effective optical depth of the whole Lyman series in the noise model
          const double wl_1 = g_lines.wavelength_cm[0] * 1e8, f_1 = g_lines.osc[0];
          double depth = a.model.tau_0 * pow(1 + lya_z, a.model.beta);
          for (int l = 1; l < a.num_forest_lines; ++l) {
            const double wl_l = g_lines.wavelength_cm[l] * 1e8;
            double one_pz = wl_1 * (1 + lya_z) / wl_l;                                 // multi :248-249
            one_pz = one_pz * ((one_pz <= (1 + z_qso)) ? 1.0 : 0.0);                   // multi :252-253
            const double tau = a.model.tau_0 * wl_l * g_lines.osc[l] / (wl_1 * f_1);   // multi :255-256
            depth = depth + tau * pow(one_pz, a.model.beta);                           // multi :258
          }
          const double sc = 1 - exp(-depth) + a.model.c_0;                             // multi :261
          double om = omega2 * (sc * sc);                                              // multi :263
          // multi :267-285: mean-flux suppression exp(-Sum tau_l (1+z_l)^beta) with Kim's priors
          double total = 0.0;
          for (int l = 0; l < a.num_forest_lines; ++l) {
            const double wl_l = g_lines.wavelength_cm[l] * 1e8;
            const double z_l = (wl - wl_l) / wl_l;                                     // multi :184-186
            const double tau_l = a.prev_tau_0 * g_lines.osc[l] / f_1 * wl_l / a.cfg.lya_wavelength;
            const double od = tau_l * pow(1 + z_l, a.prev_beta);                       // multi :275-276
            if (l > 0 && z_l > z_qso) continue;                                        // multi :279-282
            total += od;
          }
          mf = exp(-total);                                                            // multi :285
          row.mu = row.mu * mf;                                                        // multi :287
          row.omega2 = om * (mf * mf);                                                 // multi :293
        }
      }
      // a kept pixel of infinite variance weighs nothing (1/d = 0) but log d = inf: it is swept as
      // a neutral row and its effect re-enters through ll_bias (unless its flux is NaN, which the
      // reference propagates into every result: then the row is left as it is)
      const bool inf_nv = keep && row.nu == INFINITY && row.y == row.y;
      if (inf_nv) row = PixelRow{0.0, 0.0, 0.0, 1.0};
      pix[u] = row;
      for (int c = 0; c < k; ++c) {                                                    // :139
        const double m0 = a.model.M[lo + (int64_t)c * G], m1 = a.model.M[lo + 1 + (int64_t)c * G];
        Mi[(int64_t)u * k + c] = (keep && !inf_nv) ? (m0 + (m1 - m0) * t) * mf : 0.0;  // multi :288
      }
    }
    __syncthreads();
    if (tid == 0) s_base += total;
    __syncthreads();
  }
  const int n_u = s_base;
  // block reductions
  int kc = kept_count;
  for (int o = 32; o > 0; o >>= 1) kc += __shfl_xor(kc, o);
  __syncthreads();
  if (lane == 0) s_cnt[wave] = kc;
  __syncthreads();
  const int n_kept = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
  kept_min = block_reduce_minmax(kept_min, true, s_red);
  kept_max = block_reduce_minmax(kept_max, false, s_red);
  un_min = block_reduce_minmax(un_min, true, s_red);
  un_max = block_reduce_minmax(un_max, false, s_red);
  nv_flags = __syncthreads_or(nv_flags & 1) | (__syncthreads_or(nv_flags & 2) ? 2 : 0);
  const int steps = (n_u + 3) >> 2;
  // pad rows up to 4*(steps+1): neutral pixels (the last 4 feed the neutral trailing record)
  for (int u = n_u + tid; u < 4 * steps + 4; u += 256) {
    pix[u] = PixelRow{0.0, 0.0, 0.0, 1.0};
    for (int c = 0; c < k; ++c) Mi[(int64_t)u * k + c] = 0.0;
  }
  if (tid == 0) {
    m.n_u = n_u;
    m.n_kept = n_kept;
    m.steps = steps;
    m.status = (n_kept > 0) ? ((nv_flags & 2) ? 3 : 0) : 1;
    m.rec_off = a.rec_off[q];
    m.ll_bias = (nv_flags & 1) ? -INFINITY : 0.0;
    if (n_kept > 0) {
      // set_parameters.m:65-73 on the kept-pixel wavelengths (process_qsos.m:159-160)
      m.max_z_dla = (kept_max / a.cfg.lya_wavelength - 1) - a.cfg.max_z_cut;
      const double za = kept_min / a.cfg.lya_wavelength - 1;
      const double zb = a.cfg.lyman_limit * (1 + z_qso) / a.cfg.lya_wavelength - 1 + a.cfg.min_z_cut;
      m.min_z_dla = fmax(za, zb);
      // process_qsos.m:168-176: logspace(a, b, 3) = 10.^[a, a + (b-a)/2, b]
      const double lo = log10(un_min), hi = log10(un_max), ps = a.cfg.pixel_spacing;
      const double a0 = lo - 3 * ps, b0 = lo - ps, a1 = hi + ps, b1 = hi + 3 * ps;
      lam[0] = pow(10.0, a0);
      lam[1] = pow(10.0, a0 + 1.0 * (b0 - a0) / 2.0);
      lam[2] = pow(10.0, b0);
      lam[3 + n_u] = pow(10.0, a1);
      lam[4 + n_u] = pow(10.0, a1 + 1.0 * (b1 - a1) / 2.0);
      lam[5 + n_u] = pow(10.0, b1);
    } else {
      m.min_z_dla = m.max_z_dla = NAN;
    }
    a.meta[q] = m;
  }
}

// ------------------------------------------------------------------------------------------
// k_build_records: everything one K-step of the sweep reads, packed contiguously so that a
// chunk of steps is ONE linear global->LDS DMA (global_load_lds_dwordx4).
//
// record(q, t) = RD doubles, RD = ntiles*64 + 32:
//   [0, ntiles*64)   B-operand tiles.  Tile c, lane l = 16*jj + col holds, for pixel 4t + jj,
//                    column 16c + col of [vech(m m') | m]  (row-wise lower triangle:
//                    idx(i, j) = i(i+1)/2 + j, j <= i).  The first tiles_w = ceil(k(k+1)/2 / 16)
//                    tiles take the weight w, the following ceil(k/16) tiles take u.
//   [+0, +16)        PixelRow (y, mu, omega2, nu) of pixels 4t .. 4t+3
//   [+16, +20)       padded wavelengths 4(t+3) .. 4(t+3)+3 (the raw profile runs three steps ahead)
// Record `steps` (one past the last K-step) is neutral: zero tiles, (y, mu, omega2, nu) = (0,0,0,1).
//   [+20, +32)       unused (keeps records 16-byte granular and 256-byte aligned)
// compact class (record_extras == 64) instead: one 16-double block per pixel jj at +16 jj, so that
// a lane of the sweep reaches all of its operands from ONE address (see k_sweep on LDS offsets):
//   +0 .. +3   PixelRow (y, mu, omega2, nu) of pixel 4t + jj
//   +4 .. +7   m columns 16 .. 19 of that pixel
//   +8, +9     vech columns 208, 209 of that pixel
//   +10        padded wavelength 4(t+3) + jj
//   +11 .. +15 unused
// ------------------------------------------------------------------------------------------
struct BuildRecordsArgs {
  const QuasarMeta *meta;
  const PixelRow *pix;
  const double *Mi;
  const double *lam_pad;
  double *records;        // pool, record index = pix_off/4 + step
  int32_t k, tiles_w, ntiles;
  int32_t blocks_per_quasar;
  int32_t f32_tiles;      // 1: tiles stored as float (the fp32-contraction study), 0: double
  const int32_t *order;   // the records of quasars order[0 .. grid / blocks_per_quasar) are built
};

// Tile classes.  k <= 40: ntiles = 52 w-tiles + 4 u-tiles.  k <= 20 ("compact", ntiles == 14): 13
// FULL w-tiles (vech columns 0..207) + 1 full u-tile (m columns 0..15) on the matrix cores; the
// remaining 2 vech columns (208, 209) and 4 m columns (16..19) would each occupy a tile that is
// 7/8 resp. 3/4 empty -- 128 of the 1024 MFMA cycles of a K-step -- and are accumulated with 6
// FMAs per lane and K-step instead (kXW / kXU, operands in the record's extras).
constexpr int kCompactTiles = 14, kXW = 2, kXU = 4, kXWColumn = 208, kXUColumn = 16;
__host__ __device__ constexpr bool tiles_compact(int ntiles) { return ntiles == kCompactTiles; }
// doubles of per-step extras behind the tiles of a record
// (compact: 4 x 11 used of 64, which also makes a record a whole number of 512-byte tiles)
// (other classes: 20 used of 128, which makes a record a whole number of KiB for the chunk copy)
__host__ __device__ constexpr int record_extras(int ntiles) { return tiles_compact(ntiles) ? 64 : 128; }
// offsets into the extras of pixel jj's PixelRow and of its padded wavelength; compact class: of
// its m columns 16..19 and vech columns 208, 209 relative to the PixelRow
__host__ __device__ constexpr int extras_row(int ntiles, int jj) { return tiles_compact(ntiles) ? 16 * jj : 4 * jj; }
__host__ __device__ constexpr int extras_lam(int ntiles, int jj) { return tiles_compact(ntiles) ? 16 * jj + 10 : 16 + jj; }
__host__ __device__ constexpr int extras_lam_stride(int ntiles) { return extras_lam(ntiles, 1) - extras_lam(ntiles, 0); }
constexpr int kExtrasXU = 4, kExtrasXW = 8;
// tiles' worth of columns a sample's [vech(B) | v] occupies in the epilogue's LDS rows
__host__ __device__ constexpr int logical_tiles(int ntiles) { return tiles_compact(ntiles) ? 16 : ntiles; }
// doubles per record: tiles (64 elements each, as double or float) + the extras
__host__ __device__ constexpr int record_doubles(int ntiles, int f32_tiles) {
  return ntiles * (f32_tiles ? 32 : 64) + record_extras(ntiles);
}

__global__ __launch_bounds__(256) void k_build_records(BuildRecordsArgs a) {
  // One block builds whole records (steps bq, bq + blocks_per_quasar, ...): the four M rows of a
  // step are staged in LDS, the (i, j) of every vech column comes from a table built once per
  // block, and every thread writes 16 bytes at a time -- the kernel is a pure HBM write stream.
  constexpr int SB = 8;  // steps staged per pair of barriers
  __shared__ double s_rows[SB * 4][GPDLA_MAX_K];
  __shared__ uint8_t s_vi[52 * 16], s_vj[52 * 16];
  const int q = a.order[blockIdx.x / a.blocks_per_quasar];
  const int bq = blockIdx.x % a.blocks_per_quasar;
  const QuasarMeta m = a.meta[q];
  const int tid = threadIdx.x;
  const int RD = record_doubles(a.ntiles, a.f32_tiles);
  const int xtra = record_extras(a.ntiles);
  const int ncol_w = a.k * (a.k + 1) / 2;
  const int n_pad = m.n_u + 6;
  const int k = a.k;
  for (int c = tid; c < a.tiles_w * 16; c += 256) {
    int i = 0, j = 0;
    if (c < ncol_w) {
      i = (int)((sqrt(8.0 * c + 1.0) - 1.0) * 0.5);
      while ((i + 1) * (i + 2) / 2 <= c) ++i;
      while (i * (i + 1) / 2 > c) --i;
      j = c - i * (i + 1) / 2;
    }
    s_vi[c] = (uint8_t)i;
    s_vj[c] = (uint8_t)j;
  }
  double *out = a.records + m.rec_off * (int64_t)RD;
  const int nrec = m.steps + 1;  // record `steps` is the neutral trailing one
  for (int step0 = bq * SB; step0 < nrec; step0 += a.blocks_per_quasar * SB) {
    const int nst = min(SB, nrec - step0);
    __syncthreads();  // previous rows consumed (and, first time, the table complete)
    for (int e = tid; e < nst * 4 * k; e += 256)  // rows 4 step0 .. of Mi are contiguous
      s_rows[e / k][e % k] = a.Mi[(m.pix_off + 4 * (int64_t)step0) * k + e];
    __syncthreads();
    const int per = a.ntiles * 32;  // 16-byte units of tiles per record
    for (int u = tid; u < nst * per; u += 256) {
      const int ls = u / per, rem = 2 * (u - ls * per);  // two neighbouring columns of one pixel
      const int tile = rem >> 6, l = rem & 63;
      const int jj = l >> 4, col = l & 15;
      const double *row = s_rows[ls * 4 + jj];
      double v0 = 0.0, v1 = 0.0;
      if (tile < a.tiles_w) {
        const int c = tile * 16 + col;
        if (c < ncol_w) v0 = row[s_vi[c]] * row[s_vj[c]];
        if (c + 1 < ncol_w) v1 = row[s_vi[c + 1]] * row[s_vj[c + 1]];
      } else {
        const int c = (tile - a.tiles_w) * 16 + col;
        if (c < k) v0 = row[c];
        if (c + 1 < k) v1 = row[c + 1];
      }
      double *rec = out + (int64_t)(step0 + ls) * RD;
      if (a.f32_tiles) *reinterpret_cast<float2 *>(reinterpret_cast<float *>(rec) + rem) = make_float2((float)v0, (float)v1);
      else *reinterpret_cast<double2 *>(rec + rem) = make_double2(v0, v1);
    }
    for (int e = tid; e < nst * xtra; e += 256) {
      const int ls = e / xtra, r2 = e - ls * xtra;
      const int step = step0 + ls;
      double v = 0.0;
      // (pixel, field): field 0..3 PixelRow, 4 wavelength, 5..8 m columns 16.., 9, 10 vech 208, 209
      int jj = -1, f = 0;
      if (tiles_compact(a.ntiles)) {  // compact class: per-pixel blocks
        const int o = r2 & 15;
        jj = r2 >> 4;
        f = o < 4 ? o : o < 4 + kXU ? 5 + (o - 4) : o < 4 + kXU + kXW ? 9 + (o - 4 - kXU) : o == 10 ? 4 : -1;
        static_assert(kExtrasXU == 4 && kExtrasXW == 4 + kXU && 4 + kXU + kXW == 10, "block layout");
      } else if (r2 < 16) {
        jj = r2 >> 2;
        f = r2 & 3;
      } else if (r2 < 20) {
        jj = r2 - 16;
        f = 4;
      }
      if (jj >= 0 && f >= 0) {
        if (f < 4) {
          const PixelRow px = a.pix[m.pix_off + 4 * (int64_t)step + jj];
          v = f == 0 ? px.y : f == 1 ? px.mu : f == 2 ? px.omega2 : px.nu;
        } else if (f == 4) {
          int P = 4 * (step + 3) + jj;
          if (P > n_pad - 1) P = n_pad - 1;
          v = a.lam_pad[m.lam_off + P];
        } else {  // compact class: the columns kept off the matrix cores
          const double *row = s_rows[ls * 4 + jj];
          if (f >= 9) {  // vech index 208 + x = (19, 18 + x)
            const int x = f - 9;
            if (kXWColumn + x < ncol_w) v = row[19] * row[18 + x];
          } else if (kXUColumn + (f - 5) < k) {
            v = row[kXUColumn + (f - 5)];
          }
        }
      }
      out[(int64_t)step * RD + RD - xtra + r2] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_sweep.
//
// Grid: 8 * ceil(nq/8) * blocks_per_quasar blocks of 512 threads = 8 waves (two per SIMD, so one
// wave's Voigt VALU work overlaps the other's MFMAs), flattened in x.  Workgroups are dealt
// round-robin over the 8 XCDs, so block i works on quasar 8*(i/8/bpq) + i%8: every XCD streams ONE
// quasar's records at a time and keeps them in its own 4 MB L2.
//
// A wave owns 16 sample slots (MFMA rows); lane l = 16*jj + s works on sample slot s and, at
// K-step t, on pixel 4t + jj.  Slot S (one past the last sample) is the null model (a = 1):
// process_qsos.m:149-151.  Samples are visited in ascending z_DLA order (perm), so the lanes of a
// wave sit within a few pixels of each other relative to every line centre and the accurate
// tier of the Voigt function is taken by whole waves, one line at a time.
//
// TS ("tile split"): number of waves that share one group of 16 samples and split the B tiles
// between them (1 for k <= 20; 4 for k <= 40, where 55 tiles of accumulators do not fit one wave).
//
// LDS (one dynamic array): exp table | two chunk buffers of kChunkSteps records (global_load_lds
// double buffering) | raw-profile rings | per-sample line multipliers.  After the loop everything
// behind the exp table is reused by the Cholesky epilogue.
// ------------------------------------------------------------------------------------------
constexpr int kSweepWaves = 8;  // waves per block of the fp64 sweeps

struct SweepArgs {
  const QuasarMeta *meta;
  const double *records;
  const double *lam_pad;
  const double *offset_samples;   // [S]
  const double *nhi_samples;      // [S]
  const int32_t *perm;            // [S] sample indices in ascending offset (= z_DLA) order
  const int32_t *order;           // [nq] quasar indices in order of decreasing pixel count
  const PixelRow *pix;            // per-pixel pool (k_sweep_split reads rows directly)
  int64_t S;
  int64_t nq;
  int32_t blocks_per_quasar;
  int32_t k, tiles_w, ntiles, num_lines;
  double *sample_ll;              // [nq][S]   process_qsos.m:196
  double *ll_no_dla;              // [nq]      process_qsos.m:149
};

// 1/a to 2.2e-15 relative: v_rcp_f64 seed (measured 4.6e-8, tools/rcp_accuracy_probe.hip) + ONE
// Newton step.  a must be finite, non-zero and normal (NaN otherwise): k_prepare keeps non-finite
// and non-positive noise variances out of the sweeps (QuasarMeta::ll_bias, status 3).  A second step would make it correctly rounded at
// two more fp64 instructions per use -- two uses per K-step of the sweep, 4 % of its VALU work.
// What 2e-15 costs: the optical depth moves by 2e-15 relative (absorption by <= 8e-16 absolute),
// Sum r^2/d by <= 2e-15 |Sum r^2/d| ~ 1e-11..1e-10, log det B by <= k 2e-15: two orders below the
// 1e-8 parity tolerance (GPU tests observe <= 3e-10 against the oracle and the 50-digit values).
__device__ __forceinline__ double fast_rcp(double a) {
  const double r = __builtin_amdgcn_rcp(a);
  const double e = fma(-a, r, 1.0);
  return fma(r, e, r);
}

// exp(x) for x <= 0 (optical depths): x = n ln2 + r, |r| <= ln2/2, degree-12 Taylor, ldexp.
__device__ __forceinline__ double exp_nonpos(double x) {
  x = fmax(x, -800.0);  // exp(-800) = 0 in fp64; keeps n in int range
  const double n = rint(x * 1.4426950408889634);
  double r = fma(-n, 0.6931471803691238, x);      // ln2 high part (trailing bits zero)
  r = fma(-n, 1.9082149292705877e-10, r);         // ln2 low part
  double p = 2.08767569878681e-09;                // 1/12!
  p = fma(p, r, 2.505210838544172e-08);           // 1/11!
  p = fma(p, r, 2.755731922398589e-07);
  p = fma(p, r, 2.755731922398589e-06);
  p = fma(p, r, 2.48015873015873e-05);
  p = fma(p, r, 0.0001984126984126984);
  p = fma(p, r, 0.001388888888888889);
  p = fma(p, r, 0.008333333333333333);
  p = fma(p, r, 0.041666666666666664);
  p = fma(p, r, 0.16666666666666666);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)n);
}

// Re w(x+iy) sqrt(pi)/y for |x| >= 30 given x2 = x^2 (see faddeeva.hpp rew_wing), with the fast
// reciprocal.  Returns the value with the y/sqrt(pi) factor left out.
// The wing series T(rho) = Sum_{m<=6} (2m+1)!!/2^m rho^m economised to degree 5 on the interval the
// wing tier uses, 0 <= rho <= 1/900 (|x| >= 30): rho^6 replaced by its shifted-Chebyshev remainder,
// max change 1.9e-18 (derivation: 50-digit mpmath).  One FMA per line and K-step less.
constexpr double kE1 = 0x1.8000000000236p+0, kE2 = 0x1.dffffffd2a557p+1, kE3 = 0x1.a4000aa13fb7fp+3,
                 kE4 = 0x1.d86dfb645a1cbp+5, kE5 = 0x1.4be1ccccccccdp+8;

__device__ __forceinline__ double wing_core(double x2, double y2) {
  const double rho = fast_rcp(x2 + y2);
  double t = fma(kE5, rho, kE4);
  t = fma(t, rho, kE3);
  t = fma(t, rho, kE2);
  t = fma(t, rho, kE1);
  t = fma(t, rho, 1.0);
  t = fma(-2.0 * y2 * rho, rho, t);
  return rho * t;
}

// Asynchronous global -> LDS copy, 16 bytes per lane (lane l lands at lds_wave_base + 16 l).
// Issued as inline assembly on purpose: with __builtin_amdgcn_global_load_lds the compiler cannot
// tell which LDS bytes the DMA writes and puts s_waitcnt vmcnt(0) in front of EVERY later LDS read,
// i.e. the first K-step of a chunk waits for the prefetch of the next one (measured: 20 % of the
// sweep).  The kernels order these copies themselves: glds_wait() + __syncthreads() before a stage
// buffer is read, and a barrier after its last read before it is refilled.
__device__ __forceinline__ void glds16(const double *gsrc, double *lds_wave_base) {
  const uint32_t lds = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds_wave_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :
               : "v"(gsrc), "s"(lds)
               : "memory", "m0");
}
// s_waitcnt vmcnt(0) only (gfx9 encoding: expcnt and lgkmcnt fields at their maxima).  The builtin,
// not inline assembly: the compiler's own wait-count bookkeeping sees it and keeps its
// conservative vmcnt waits out of the loops that follow.
__device__ __forceinline__ void glds_wait() {
  __builtin_amdgcn_s_waitcnt(0x0F70);
  asm volatile("" ::: "memory");
}

// A whole chunk, KIB KiB-blocks long, copied global -> LDS by the WAVES waves of a block.  One
// global_load_lds_dwordx4 moves 1 KiB (lane l: 16 bytes at +16 l on both sides), and the
// instruction's immediate offset moves BOTH addresses (tools/glds_offset_probe.hip), so four
// instructions share one M0 and one address register; each wave takes a contiguous span of PER
// blocks.  Per chunk and wave that is two or three vector instructions where a loop over
// (unit < units ? glds16 : skip) costs eight per block -- and non-arithmetic VALU instructions are
// MFMA time in these kernels.  The copy always moves the whole chunk: a chunk that runs past a
// quasar's last record reads the next quasar's records or the pool's padding (gpdla.hip allocates
// kRecordPoolPad records behind the pool), which no K-step consumes.
constexpr int kRecordPoolPad = 8;
__device__ __forceinline__ void glds_quad(const double *gsrc, uint32_t lds, int count) {
  if (count >= 4)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
                 "global_load_lds_dwordx4 %0, off offset:1024\n\tglobal_load_lds_dwordx4 %0, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %0, off offset:3072"
                 :
                 : "v"(gsrc), "s"(lds)
                 : "memory", "m0");
  else if (count == 3)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
                 "global_load_lds_dwordx4 %0, off offset:1024\n\tglobal_load_lds_dwordx4 %0, off offset:2048"
                 :
                 : "v"(gsrc), "s"(lds)
                 : "memory", "m0");
  else if (count == 2)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
                 "global_load_lds_dwordx4 %0, off offset:1024"
                 :
                 : "v"(gsrc), "s"(lds)
                 : "memory", "m0");
  else if (count == 1)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds) : "memory", "m0");
}
// src: the chunk's first double in global memory; lds: byte address of its LDS buffer; wave: this
// wave's index within the block -- all three wave-uniform (scalar registers).
template <int KIB, int WAVES>
__device__ __forceinline__ void glds_chunk(const double *src, uint32_t lds, int wave, int lane) {
  constexpr int PER = (KIB + WAVES - 1) / WAVES;
  const int first = wave * PER;
  const int count = min(PER, KIB - first);  // <= 0 for waves behind the last block
  const double *p = src + (size_t)first * 128 + 2 * lane;
  const uint32_t l = lds + (uint32_t)first * 1024u;
#pragma unroll
  for (int g = 0; g < (PER + 3) / 4; ++g)
    if (count > 4 * g) glds_quad(p + g * 512, l + (uint32_t)g * 4096u, count - 4 * g);
}
// LDS byte address of a pointer into the dynamic shared array
__device__ __forceinline__ uint32_t lds_address(const void *p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

// Epilogue shared by the sweep kernels: one round (MFMA result register r) of the per-sample
// factorisation.  The 16 lanes of row jj hold, in register r of every tile, the 16*NT columns of
// sample jj + 4r: they spill them to LDS (e = [16*NT] doubles for this row: packed lower triangle
// of B, then v at voff) and factor the augmented (k+1) x (k+1) matrix [[I + B, v], [v', .]] column
// by column (left-looking).  Row k of the augmented matrix is v, so its factor row is z = L^-1 v.
//
// Lane s of the 16 owns rows s, s+16 (, s+32) for the whole factorisation and keeps their running
// diagonals  dd_i = A_ii - Sum_{m<j} l_im^2  in registers.  Column j then needs no dot product for
// its pivot: the owner of row j broadcasts dd_j, every lane takes ONE reciprocal square root
// (v_rsq_f64 + two Newton steps: no sqrt, no division, :24 is only ever used through 1/L_jj and
// log L_jj = log(d_j)/2), and the row dot products -- which do not depend on the pivot -- run
// alongside.  -dd of row k ends as z'z.
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);  // ~2^-26
  const double h = 0.5 * x;
  double err = fma(-h * y, y, 0.5);
  y = fma(y, err, y);
  err = fma(-h * y, y, 0.5);
  return fma(y, err, y);
}

// The factorisation proper, on one sample's columns already in LDS, by LPS lanes (s = 0..LPS-1);
// ROWS = ceil((k+1)/LPS) rows per lane at most.  Returns log N(y; a mu, ...) of that sample
// (log_mvnpdf_low_rank.m:30-32) in every lane of the group.  q_s = Sum r^2/d, ld_s = Sum log d.
template <int ROWS, int LPS>
__device__ __forceinline__ double factor_lds(double *e, int s, int k, int voff, double q_s, double ld_s,
                                          int n_kept) {
  // Columns are taken in panels of PW: the part of their dot products that involves earlier
  // panels (the bulk) is formed for the whole panel at once, so each own-row entry read from LDS
  // feeds PW multiply-adds instead of one; only the few in-panel terms follow the column-by-column
  // dependency chain.  The summation order per entry is unchanged (columns ascending).
  constexpr int PW = ROWS <= 3 ? 8 : 4;
  int ro[ROWS];
  double dd[ROWS];
#pragma unroll
  for (int a = 0; a < ROWS; ++a) {
    const int i = s + LPS * a;
    ro[a] = i < k ? i * (i + 1) / 2 : voff;        // rows beyond k alias v and are never written
    dd[a] = i < k ? e[ro[a] + i] + 1.0 : 0.0;      // log_mvnpdf_low_rank.m:22-23
  }
  double lprod = 1.0;
  int lexp = 0;
  bool pd = true;
  for (int j0 = 0; j0 < k; j0 += PW) {
    double t[ROWS][PW];
    int rp[PW];  // pivot rows of the panel (clamped in the last, partial panel: those columns are unused)
#pragma unroll
    for (int c = 0; c < PW; ++c) {
      const int j = min(j0 + c, k - 1);
      rp[c] = j * (j + 1) / 2;
#pragma unroll
      for (int a = 0; a < ROWS; ++a) t[a][c] = e[ro[a] + j0 + c];
    }
#pragma unroll 2
    for (int mm = 0; mm < j0; ++mm) {
      double own[ROWS], piv[PW];
#pragma unroll
      for (int a = 0; a < ROWS; ++a) own[a] = e[ro[a] + mm];
#pragma unroll
      for (int c = 0; c < PW; ++c) piv[c] = e[rp[c] + mm];
#pragma unroll
      for (int a = 0; a < ROWS; ++a)
#pragma unroll
        for (int c = 0; c < PW; ++c) t[a][c] = fma(-own[a], piv[c], t[a][c]);
    }
#pragma unroll
    for (int c = 0; c < PW; ++c) {
      const int j = j0 + c;
      if (j < k) {  // block-uniform
#pragma unroll
        for (int cp = 0; cp < c; ++cp) {  // in-panel terms: own entries are still in registers
          const double pv = e[rp[c] + j0 + cp];
#pragma unroll
          for (int a = 0; a < ROWS; ++a) t[a][c] = fma(-t[a][cp], pv, t[a][c]);
        }
        double dsel = dd[0];
#pragma unroll
        for (int a = 1; a < ROWS; ++a)
          if (j >= LPS * a) dsel = dd[a];
        const double dj = __shfl(dsel, j & (LPS - 1), LPS);  // pivot, from the lane that owns row j
        pd = pd && (dj > 0.0);                         // chol would throw here (:24)
        const double inv = rsqrt_nr(dj);
        lprod *= dj;                                   // 2 Sum log L_jj = log Prod d_j (:30)
        lexp += __builtin_amdgcn_frexp_exp(lprod);
        lprod = __builtin_amdgcn_frexp_mant(lprod);
#pragma unroll
        for (int a = 0; a < ROWS; ++a) {
          const int i = s + LPS * a;
          t[a][c] *= inv;
          if (i > j && i <= k) {
            e[ro[a] + j] = t[a][c];
            dd[a] = fma(-t[a][c], t[a][c], dd[a]);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  double zsel = dd[0];
#pragma unroll
  for (int a = 1; a < ROWS; ++a)
    if (k >= LPS * a) zsel = dd[a];
  const double zz = -__shfl(zsel, k & (LPS - 1), LPS);  // z'z with z = L^-1 v
  const double log_det = ld_s + log(lprod) + (double)lexp * 0.6931471805599453;  // :30
  const double ll = -0.5 * ((q_s - zz) + log_det + (double)n_kept * kLog2Pi);    // :32
  return pd ? ll : NAN;
}

// The same factorisation with each lane's own rows held in registers (k <= KMAX known at compile
// time, column loop fully unrolled).  The LDS version reads, for every column, the lane's three
// rows AND the pivot row -- 4 streams, 300 KB per wave and pass, which made the epilogue LDS
// bandwidth-bound; here only the pivot row (a broadcast within the sample's lanes) comes from LDS,
// at compile-time offsets.  Same operation order, so results are bit-identical to factor_lds.
template <int ROWS, int LPS, int KMAX>
__device__ __forceinline__ double factor_rows(double *e, int s, int k, int voff, double q_s, double ld_s,
                                           int n_kept) {
  int ro[ROWS];
  double dd[ROWS];
  double rc[ROWS][KMAX];  // rc[a][m] = entry (row s + LPS a, column m): A before column m, L after
#pragma unroll
  for (int a = 0; a < ROWS; ++a) {
    const int i = s + LPS * a;
    ro[a] = i < k ? i * (i + 1) / 2 : voff;        // rows beyond k alias v and are never written
    dd[a] = i < k ? e[ro[a] + i] + 1.0 : 0.0;      // log_mvnpdf_low_rank.m:22-23
#pragma unroll
    for (int m = 0; m < KMAX; ++m) rc[a][m] = e[ro[a] + m];  // (columns >= the row index: unused)
  }
  double lprod = 1.0;
  int lexp = 0;
  bool pd = true;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (j < k) {  // block-uniform
      const int rj = j * (j + 1) / 2;
      double dsel = dd[0];
#pragma unroll
      for (int a = 1; a < ROWS; ++a)
        if (j >= LPS * a) dsel = dd[a];
      const double dj = __shfl(dsel, j & (LPS - 1), LPS);  // pivot, from the lane that owns row j
      double t[ROWS];
#pragma unroll
      for (int a = 0; a < ROWS; ++a) t[a] = rc[a][j];
#pragma unroll
      for (int mm = 0; mm < j; ++mm) {
        const double c = e[rj + mm];
#pragma unroll
        for (int a = 0; a < ROWS; ++a) t[a] = fma(-rc[a][mm], c, t[a]);
      }
      pd = pd && (dj > 0.0);                         // chol would throw here (:24)
      const double inv = rsqrt_nr(dj);
      lprod *= dj;                                   // 2 Sum log L_jj = log Prod d_j (:30)
      lexp += __builtin_amdgcn_frexp_exp(lprod);
      lprod = __builtin_amdgcn_frexp_mant(lprod);
#pragma unroll
      for (int a = 0; a < ROWS; ++a) {
        const int i = s + LPS * a;
        t[a] *= inv;
        rc[a][j] = t[a];
        if (i > j && i <= k) {
          e[ro[a] + j] = t[a];                       // row i becomes a pivot row later
          dd[a] = fma(-t[a], t[a], dd[a]);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  double zsel = dd[0];
#pragma unroll
  for (int a = 1; a < ROWS; ++a)
    if (k >= LPS * a) zsel = dd[a];
  const double zz = -__shfl(zsel, k & (LPS - 1), LPS);  // z'z with z = L^-1 v
  const double log_det = ld_s + log(lprod) + (double)lexp * 0.6931471805599453;  // :30
  const double ll = -0.5 * ((q_s - zz) + log_det + (double)n_kept * kLog2Pi);    // :32
  return pd ? ll : NAN;
}

// t += (-p[lane n of this lane's 16-lane row]) * r in one instruction: the DPP form of the fp64
// multiply-add broadcasts its first source across a row of 16 lanes (row_newbcast, the only DPP
// control the 64-bit ALU knows).  Bit for bit fma(-p_n, r, t) and the same issue cost as the plain
// instruction (tools/dpp_f64_probe.hip).  p must not have been written by a VALU instruction in
// the two instructions before (DPP hazard; nothing pads inline assembly): here p always comes
// straight from an LDS read, and tools/check_dpp_hazard.py checks the ISA of a build.
template <int N, bool PAD = false>
__device__ __forceinline__ void fmac_bcast(double &t, double p, double r) {
  if constexpr (PAD)  // p may have been written by the instruction before: two wait states by hand
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(p), "v"(r), "n"(N));
  else
    asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(p), "v"(r), "n"(N));
}

// p[lane N of this lane's 16-lane row] (same hazard rule)
template <int N, bool PAD = false>
__device__ __forceinline__ double mov_bcast(double p) {
  double d;
  if constexpr (PAD)
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(p), "n"(N));
  else
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(p), "n"(N));
  return d;
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant
// expression in the body (the DPP lane above is an instruction field)
template <int... I, typename F>
__device__ __forceinline__ void static_for_seq(std::integer_sequence<int, I...>, F &&f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_seq(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f));
}

// The factorisation of the 20 < k <= 40 classes: 32 lanes per sample, as factor_lds<2, 32>, but
//   * own rows in registers (as factor_rows), and the k + 1 rows dealt so that the SHORT rows share
//     lanes with long ones: lane s owns row R0 + s (R0 = KMAX - 31 = 9), and lanes 0 .. R0 - 1 own
//     rows 0 .. R0 - 1 as well.  factor_lds deals rows s and s + 32: its second slot holds a row in 9 of
//     32 lanes only, and every column costs two dot products of the column's length whichever rows
//     are already finished.  A row is finished once the column index reaches it, so here the second
//     slot is dead in EVERY lane from column R0 - 1 on and its dot products are not issued there:
//     808 multiply-adds per pass at k = 40 against 1640 (a first version paired rows s and k - s:
//     970);
//   * the pivot row is not read as a broadcast.  The epilogue of these kernels is bound by the LDS
//     return path, not by arithmetic: a ds_read2_b64 in which all lanes of a sample read the same
//     16 bytes still returns 1 KiB, and eight waves issuing them get 171 B/clk, 48 cycles per
//     instruction and wave (tools/dpp_f64_probe.hip).  Here lane l of each 16-lane row reads
//     entries l, 16 + l, 32 + l of the pivot row -- one to three ds_read_b64 per pivot row instead
//     of one read per two entries -- and the multiply-add broadcasts the entry it needs from the
//     row's registers (fmac_bcast).
// Columns are taken in panels of PW as in factor_lds (the part of a panel's dot products that
// involves earlier panels needs nothing of the panel's own columns).  Entry by entry the operations
// and their order are those of factor_lds, so results are bit-identical.  (The name is the first
// version's.)
template <int KMAX, int PW>
__device__ __forceinline__ double factor_paired(double *e, int s, int k, int voff, double q_s, double ld_s,
                                                int n_kept) {
  static_assert(KMAX % PW == 0 && PW <= 16 && KMAX <= 48, "whole panels; three registers per pivot row");
  constexpr int R0 = KMAX + 1 - 32;                   // rows 0 .. R0 - 1 are the second rows of lanes 0 .. R0 - 1
  constexpr int K0 = (R0 - 1 + PW - 1) / PW * PW;     // ... and need columns < R0 - 1, in whole panels
  static_assert(R0 >= 1 && R0 <= 32, "two rows per lane at most");
  const bool has0 = s < R0 && s < k, has1 = R0 + s <= k;
  const int i0 = has0 ? s : -1, i1 = has1 ? R0 + s : -1;  // (-1: no row; its reads alias v, it is never written)
  const int ro0 = has0 ? i0 * (i0 + 1) / 2 : voff, ro1 = has1 && i1 < k ? i1 * (i1 + 1) / 2 : voff;
  // Where the lane keeps the running diagonal dd_i = A_ii + 1 - Sum_{m<j} l_im^2 of each of its rows
  // for the others to see: the row's own diagonal slot.  It is stored after every column, finished
  // or not -- nobody reads a finished row's slot -- so the pivot of column j is simply entry j of
  // row j when the column's step reads that row (no shuffle, no select between the two slots).  Row k
  // (v) has no diagonal slot and is nobody's pivot.
  const bool pub0 = has0, pub1 = has1 && i1 < k;
  const int dg0 = ro0 + i0, dg1 = ro1 + i1;
  double dd0 = pub0 ? e[dg0] + 1.0 : 0.0;   // log_mvnpdf_low_rank.m:22-23
  double dd1 = pub1 ? e[dg1] + 1.0 : 0.0;
  double rc0[K0], rc1[KMAX];  // entry (row, column m): A before column m, L after
#pragma unroll
  for (int m = 0; m < K0; ++m) rc0[m] = e[ro0 + m];
#pragma unroll
  for (int m = 0; m < KMAX; ++m) rc1[m] = e[ro1 + m];  // (columns >= the row index: unused)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();  // (every lane has read its A_ii before anybody overwrites one)
  if (pub0) e[dg0] = dd0;
  if (pub1) e[dg1] = dd1;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // this lane's window on a pivot row: entry (s & 15) + 16 r of row j is el[j (j + 1) / 2 + 16 r]
  // (immediate offsets of one address register; a row's tail reads into the next rows, in bounds)
  const __attribute__((address_space(3))) double *el =
      (const __attribute__((address_space(3))) double *)(uintptr_t)lds_address(e + (s & 15));
  double lprod = 1.0;
  int lexp = 0;
  bool pd = true;
  static_for<KMAX / PW>([&](auto P_) __attribute__((always_inline)) {
    constexpr int j0 = decltype(P_)::value * PW;
    if (j0 < k) {  // block-uniform
      // (panels j0 < K0 carry both slots)
      constexpr bool two = j0 < K0;
      constexpr int NR = (j0 + 15) / 16;  // registers of a pivot row that hold columns < j0
      double pb[PW][3];
#pragma unroll
      for (int c = 0; c < PW; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) pb[c][r] = r < NR ? el[(j0 + c) * (j0 + c + 1) / 2 + 16 * r] : 0.0;
      static_for<j0>([&](auto M_) __attribute__((always_inline)) {
        constexpr int mm = decltype(M_)::value;
        static_for<PW>([&](auto C_) __attribute__((always_inline)) {
          constexpr int c = decltype(C_)::value;
          if constexpr (two) fmac_bcast<mm % 16>(rc0[two ? j0 + c : 0], pb[c][mm / 16], rc0[mm < K0 ? mm : 0]);
          fmac_bcast<mm % 16>(rc1[j0 + c], pb[c][mm / 16], rc1[mm]);
        });
      });
      static_for<PW>([&](auto C_) __attribute__((always_inline)) {
        constexpr int c = decltype(C_)::value, j = j0 + c;
        if (j < k) {  // block-uniform
          // columns j0 .. j0 + 15 of row j: the entries the previous column steps of this panel
          // stored (the in-panel terms), and at column j the row's running diagonal, i.e. the pivot
          const double pin = el[j * (j + 1) / 2 + j0];
          static_for<c>([&](auto Q_) __attribute__((always_inline)) {
            constexpr int cp = decltype(Q_)::value;
            fmac_bcast<cp>(rc1[j], pin, rc1[j0 + cp]);
            if constexpr (two) fmac_bcast<cp>(rc0[two ? j : 0], pin, rc0[two ? j0 + cp : 0]);
          });
          const double dj = mov_bcast<c>(pin);
          pd = pd && (dj > 0.0);                         // chol would throw here (:24)
          const double inv = rsqrt_nr(dj);
          lprod *= dj;                                   // 2 Sum log L_jj = log Prod d_j (:30)
          if constexpr (c == PW - 1) {  // (scaling by powers of two is exact: once per panel is the same product)
            lexp += __builtin_amdgcn_frexp_exp(lprod);
            lprod = __builtin_amdgcn_frexp_mant(lprod);
          }
          int jc = j;  // (opaque: or all k compares are formed up front and their masks spilled)
          asm volatile("" : "+s"(jc)::"memory");
          rc1[j] *= inv;
          if (i1 > jc) e[ro1 + j] = rc1[j];              // row i1 becomes a pivot row later
          dd1 = fma(-rc1[j], rc1[j], dd1);               // (a finished row's dd is dead)
          if (pub1) e[dg1] = dd1;
          if constexpr (two) {
            rc0[two ? j : 0] *= inv;
            if (i0 > jc) e[ro0 + j] = rc0[two ? j : 0];
            dd0 = fma(-rc0[two ? j : 0], rc0[two ? j : 0], dd0);
            if (pub0) e[dg0] = dd0;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      });
    }
  });
  lexp += __builtin_amdgcn_frexp_exp(lprod);  // (k need not end a panel)
  lprod = __builtin_amdgcn_frexp_mant(lprod);
  const double zz = -__shfl(dd1, k - R0, 32);  // row k is the first row of lane k - R0: -dd ends as z'z with z = L^-1 v
  const double log_det = ld_s + log(lprod) + (double)lexp * 0.6931471805599453;  // :30
  const double ll = -0.5 * ((q_s - zz) + log_det + (double)n_kept * kLog2Pi);    // :32
  return pd ? ll : NAN;
}

// The factorisation of the 20 < k <= 40 classes with NOTHING in LDS: one sample on the 16 lanes of
// one DPP row, all its k + 1 rows in registers.  Lane l owns rows l (l < RA = KMAX - 31 = 9), RA + l
// and RA + 16 + l -- three slots whose rows need 8, 24 and 40 columns: 72 entries per lane.  Every
// row a column step needs from another lane -- the pivot row's entries, the pivot itself -- lives
// in a register of the SAME 16-lane row, so the DPP broadcast takes it straight from its owner's
// register (fmac_bcast / mov_bcast with the owner's lane as a compile-time field): no LDS traffic,
// no masked stores, no panels, and a slot's multiply-adds stop once its rows are finished
// (1084 per column sweep at k = 40 for FOUR samples per wave; factor_paired: 808 for two).  A
// block's 32 samples therefore factor in ONE round, four per wave; LDS only transposes the
// accumulators into rows (load_rows16 reads what the spill laid down, 16 samples at a time).
// Entry by entry the operations and their order are those of factor_lds: bit-identical results.
template <int KMAX> struct Rows16 {
  static constexpr int RA = KMAX + 1 - 32;
  static constexpr int NA = RA - 1, NB = RA + 15, NC = KMAX;  // columns the rows of a slot need
  static_assert(RA >= 2 && RA <= 16, "three slots of at most 16 rows");
  double a[NA], b[NB], c[NC];  // entry (row, column m): A before column m, L after
  double da, db, dc;           // running diagonals dd_i = A_ii + 1 - Sum_{m<j} l_im^2 (row k: -z'z)
};

// e: the sample's spilled columns (packed lower triangle of B, then v at voff), l = 0..15
template <int KMAX>
__device__ __forceinline__ void load_rows16(Rows16<KMAX> &R, const double *e, int l, int k, int voff) {
  using RR = Rows16<KMAX>;
  const int ia = l, ib = RR::RA + l, ic = RR::RA + 16 + l;
  // rows beyond k do not exist and row k is v: both read v's storage (the former compute a copy of
  // z nobody looks at)
  const bool fa = l < RR::RA && ia < k, fb = ib < k, fc = ic < k;
  const int roa = fa ? ia * (ia + 1) / 2 : voff, rob = fb ? ib * (ib + 1) / 2 : voff, roc = fc ? ic * (ic + 1) / 2 : voff;
  R.da = fa ? e[roa + ia] + 1.0 : 0.0;  // log_mvnpdf_low_rank.m:22-23
  R.db = fb ? e[rob + ib] + 1.0 : 0.0;
  R.dc = fc ? e[roc + ic] + 1.0 : 0.0;
#pragma unroll
  for (int m = 0; m < RR::NA; ++m) R.a[m] = e[roa + m];
#pragma unroll
  for (int m = 0; m < RR::NB; ++m) R.b[m] = e[rob + m];
#pragma unroll
  for (int m = 0; m < RR::NC; ++m) R.c[m] = e[roc + m];  // (columns >= the row index: unused)
}

// Returns log N(y; a mu, ...) of the sample (log_mvnpdf_low_rank.m:30-32) in each of its 16 lanes.
template <int KMAX>
__device__ __forceinline__ double factor_rows16(Rows16<KMAX> &R, int k, double q_s, double ld_s, int n_kept) {
  using RR = Rows16<KMAX>;
  double lprod = 1.0;
  int lexp = 0;
  bool pd = true;
  static_for<KMAX>([&](auto J_) __attribute__((always_inline)) {
    constexpr int j = decltype(J_)::value;
    if (j < k) {  // block-uniform
      constexpr int slot = j < RR::RA ? 0 : j < RR::RA + 16 ? 1 : 2;         // where row j lives ...
      constexpr int ol = j - (slot == 0 ? 0 : slot == 1 ? RR::RA : RR::RA + 16);  // ... and in which lane
      constexpr bool act_a = j < RR::NA, act_b = j < RR::NB;  // some row of the slot is still below row j
      static_for<j>([&](auto M_) __attribute__((always_inline)) {
        constexpr int mm = decltype(M_)::value;
        constexpr bool fresh = mm == j - 1;  // the pivot row's newest entry was scaled one column step ago
        double piv;
        if constexpr (slot == 0) piv = R.a[mm];
        else if constexpr (slot == 1) piv = R.b[mm];
        else piv = R.c[mm];
        if constexpr (act_a) fmac_bcast<ol, fresh>(R.a[j], piv, R.a[mm]);
        if constexpr (act_b) fmac_bcast<ol, fresh>(R.b[j], piv, R.b[mm]);
        fmac_bcast<ol, fresh>(R.c[j], piv, R.c[mm]);
      });
      double dsel;
      if constexpr (slot == 0) dsel = R.da;
      else if constexpr (slot == 1) dsel = R.db;
      else dsel = R.dc;
      const double dj = mov_bcast<ol, true>(dsel);     // pivot
      pd = pd && (dj > 0.0);                           // chol would throw here (:24)
      const double inv = rsqrt_nr(dj);
      lprod *= dj;                                     // 2 Sum log L_jj = log Prod d_j (:30)
      if constexpr ((j & 3) == 3) {  // (scaling by powers of two is exact: every fourth column gives the same bits)
        lexp += __builtin_amdgcn_frexp_exp(lprod);
        lprod = __builtin_amdgcn_frexp_mant(lprod);
      }
      if constexpr (act_a) {
        R.a[j] *= inv;
        R.da = fma(-R.a[j], R.a[j], R.da);             // (a finished row's dd is dead)
      }
      if constexpr (act_b) {
        R.b[j] *= inv;
        R.db = fma(-R.b[j], R.b[j], R.db);
      }
      R.c[j] *= inv;
      R.dc = fma(-R.c[j], R.c[j], R.dc);
    }
  });
  lexp += __builtin_amdgcn_frexp_exp(lprod);  // (k need not be a multiple of four)
  lprod = __builtin_amdgcn_frexp_mant(lprod);
  // row k: -dd ends as z'z with z = L^-1 v
  const double zz = -(k >= RR::RA + 16 ? __shfl(R.dc, k - RR::RA - 16, 16) : __shfl(R.db, k - RR::RA, 16));
  const double log_det = ld_s + log(lprod) + (double)lexp * 0.6931471805599453;  // :30
  const double ll = -0.5 * ((q_s - zz) + log_det + (double)n_kept * kLog2Pi);    // :32
  return pd ? ll : NAN;
}

// Samples factored per pass by one wave (and LDS rows of 16*NT doubles it needs for them): 8 lanes
// per sample, i.e. 8 samples = two MFMA result registers per pass, with three rows per lane for
// k <= 20 (TW = 14) and six for k <= 40 (TW = 52).  The factorisation is a latency chain of k
// column steps per pass, so fewer, fatter passes win.
template <int TW, int TS> struct EpilogueShape {
  // (k <= 40 without a tile split -- the fp32 study kernel, one wave per sample group -- keeps 16
  // lanes per sample: 8 samples x 56 tiles per wave would not fit the LDS)
  static constexpr int LPS = (TW > 30 && TS == 1) ? 16 : 8;  // lanes per sample
  static constexpr int RPP = 16 / LPS;               // MFMA result registers per pass
  static constexpr int PASSES = 4 / RPP;
  static constexpr int SPP = 4 * RPP;                // samples per pass
  static constexpr int ROWS = TW > 30 ? 48 / LPS : 3;  // ceil(41/LPS); ceil(21/8)
  // LDS doubles per sample: its 16*NT columns + 4.  Without the pad every sample's copy of a row
  // starts on the same bank (16*NT*8 bytes is a multiple of the 256-byte bank cycle) and the SPP
  // samples a wave factors at once conflict SPP-way on every read; +32 bytes staggers them.
  static constexpr int stride(int nt) { return nt * 16 + 4; }
};

// One epilogue pass: spill the result registers RPP*p .. of every tile (the 16 lanes of row jj
// hold, in register r, the 16*NT columns of sample sample_of(jj, r)) to LDS and factor them.
// Returns the sample's log-likelihood; *sigma_out = its index among the wave's 16, *writer = this
// lane is the one that stores it.
// xw / xu: the compact class's columns accumulated off the matrix cores (sample s = lane & 15 of
// the wave, already summed over the four pixel phases); ignored otherwise.
template <typename T, int NTW, int TS, int TW, typename ACC>
__device__ __forceinline__ double factor_pass(const ACC (&acc)[NTW], const double (&xw)[kXW],
                                              const double (&xu)[kXU], int p, double *Eg, int lane,
                                              int role, int tile0, int k, double quad_sum,
                                              double logd_sum, int n_kept, int *sigma_out,
                                              bool *writer) {
  using ES = EpilogueShape<TW, TS>;
  constexpr bool compact = tiles_compact(NTW * TS);
  constexpr int voff = (TW + (compact ? 1 : 0)) * 16;  // v behind the vech columns (210 <= 224 when compact)
  constexpr int ncols = ES::stride(logical_tiles(NTW * TS));
  const int s = lane & 15, jj = lane >> 4;
  const int half = ES::RPP == 2 ? (s >> 3) : 0;
  const int sigma = Mat<T>::sample_of(jj, ES::RPP * p + half);
  // the sample's scalar sums live in the lanes whose s equals sigma
  const double q_s = __shfl(quad_sum, sigma + 16 * jj);
  const double ld_s = __shfl(logd_sum, sigma + 16 * jj);
  double *e = Eg + (size_t)(jj * ES::RPP) * ncols;
  if (TS > 1) __syncthreads();
#pragma unroll
  for (int cc = 0; cc < NTW; ++cc) {
    const int tile = tile0 + cc;
    const int col0 = tile < TW ? tile * 16 : voff + (tile - TW) * 16;  // w-tiles, then u-tiles at voff
#pragma unroll
    for (int h = 0; h < ES::RPP; ++h) e[h * ncols + col0 + s] = (double)acc[cc][ES::RPP * p + h];
  }
  if (compact) {
    // lane (s, jj = 0) holds sample s of the wave; it is factored in pass reg/RPP from the LDS row
    // of lane group jj_of(s), half reg % RPP
    const int r = Mat<T>::reg_of(s);
    if (jj == 0 && r / ES::RPP == p) {
      double *es = Eg + (size_t)(Mat<T>::jj_of(s) * ES::RPP + r % ES::RPP) * ncols;
#pragma unroll
      for (int x = 0; x < kXW; ++x) es[kXWColumn + x] = xw[x];
#pragma unroll
      for (int x = 0; x < kXU; ++x) es[voff + kXUColumn + x] = xu[x];
    }
  }
  if (TS > 1) __syncthreads();
  else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  *sigma_out = sigma;
  *writer = role == 0 && (s & (ES::LPS - 1)) == 0;
#ifdef GPDLA_ABLATE_NOEPI
  return q_s + ld_s + e[0];
#else
  if (role != 0) return NAN;
  if (TW <= 14)  // k <= 20: own rows in registers
    return factor_rows<ES::ROWS, ES::LPS, 20>(e + half * ncols, s & (ES::LPS - 1), k, voff, q_s, ld_s, n_kept);
#ifndef GPDLA_EXP_LDSEPI
  if constexpr (TW > 30 && ES::LPS == 16) {
    // k <= 40 without a tile split (the fp32 study kernel): a sample is on the 16 lanes of one DPP
    // row already -- all its rows in registers, nothing read from LDS after the spill (factor_rows16)
    Rows16<40> R;
    load_rows16<40>(R, e, s, k, voff);
    return factor_rows16<40>(R, k, q_s, ld_s, n_kept);
  }
#endif
  return factor_lds<ES::ROWS, ES::LPS>(e + half * ncols, s & (ES::LPS - 1), k, voff, q_s, ld_s, n_kept);
#endif
}

// Accurate tier of the raw profile: Re w(|x| + i y_j) for |x| < 32 from line j's piecewise
// polynomial (near_tables.hpp; same index arithmetic and Horner order as near_poly_host).
__device__ __forceinline__ double near_poly(const double *line_tab, double ax) {
  const bool core = ax < 8.0;
  const double u = core ? ax * 8.0 : (ax - 8.0) * 2.0;
  const int i = (int)u;
  const double t = (u - (double)i) - 0.5;
  const double *c = line_tab + (size_t)((core ? 0 : kNearCore) + i) * kNearCoef;
  double cf[kNearCoef];
#pragma unroll
  for (int k = 0; k < kNearCoef; ++k) cf[k] = c[k];  // 12 independent loads, then the Horner chain
  double p = cf[kNearCoef - 1];
#pragma unroll
  for (int k = kNearCoef - 2; k >= 0; --k) p = fma(p, t, cf[k]);
  return p;
}

// Template parameters: NTW B tiles per wave, TS tile split, kChunkSteps records per LDS chunk,
// TW tiles that take the weight w (the rest take u; TS*NTW tiles in all, zero-padded), LINES
// number of Lyman lines when known at compile time (0: read num_lines at run time).
__device__ __forceinline__ uint32_t hi_word(double v) { return (uint32_t)__double2hiint(v); }

// Wing-tier optical-depth sum for the three-line case (Ly-alpha, beta, gamma): FMA-form
// velocities, ONE reciprocal for the three lines (1/(sa sb sc), then peeled), 6-term series.
// Returns Sum_j lead_j y_j [Re w_j sqrt(pi)/y_j]; *near: some line within 30 Doppler widths.
__device__ __forceinline__ double wing_sum3(double lamP, double msa, double msb, double msc, double cs,
                                            bool *near) {
  const double xa = fma(lamP, msa, -cs), xb = fma(lamP, msb, -cs), xc = fma(lamP, msc, -cs);
  // x^2 + y^2 in one FMA; "near" is then tested on it (a threshold shift of y^2 <= 3e-7 in x^2,
  // immaterial: both tiers are accurate on either side of |x| = 30)
  const double sa = fma(xa, xa, g_lines.y2[0]), sb = fma(xb, xb, g_lines.y2[1]), sc = fma(xc, xc, g_lines.y2[2]);
  // some s < 900: on the high words, as unsigned integers -- positive doubles order like their bit
  // patterns and 900.0 has a zero low word, so s < 900.0 <=> hi(s) < hi(900.0); one three-way
  // integer minimum and one compare instead of two fp64 minima and a compare.
  static_assert(__builtin_bit_cast(unsigned long long, 900.0) == 0x408C200000000000ull, "bits of 900.0");
  *near = min(min(hi_word(sa), hi_word(sb)), hi_word(sc)) < 0x408C2000u;
  const double pab = sa * sb, pbc = sb * sc, pac = sa * sc;
  const double rinv = fast_rcp(pab * sc);
  const double ra = rinv * pbc, rb = rinv * pac, rc = rinv * pab;
  // T(rho) - 2 y^2 rho^2 by Horner; the -2 y_j^2 correction rides in the rho^2 coefficient (t2[j]).
  // The leading step as a three-address v_fma_f64 with kE4 in a vector register (the compiler
  // would pick v_fmac_f64 and copy the constant in front of each: two extra moves per line).
  double ta = ra * kE5 + kE4, tb = rb * kE5 + kE4, tc = rc * kE5 + kE4;
  ta = fma(ta, ra, kE3); tb = fma(tb, rb, kE3); tc = fma(tc, rc, kE3);
  ta = fma(ta, ra, g_lines.t2[0]); tb = fma(tb, rb, g_lines.t2[1]); tc = fma(tc, rc, g_lines.t2[2]);
  ta = fma(ta, ra, kE1); tb = fma(tb, rb, kE1); tc = fma(tc, rc, kE1);
  ta = fma(ta, ra, 1.0); tb = fma(tb, rb, 1.0); tc = fma(tc, rc, 1.0);
  return fma(g_lines.cwing[2], rc * tc, fma(g_lines.cwing[1], rb * tb, g_lines.cwing[0] * (ra * ta)));
}

// wing_sum3 with the K-step's OTHER reciprocal riding along: 1/d of the weights (process_qsos.m:192-198
// folded into log_mvnpdf_low_rank.m:13-15; d is finite and positive, k_prepare sees to that) comes out
// of the same v_rcp_f64 as the three lines' 1/s_j -- prefix products, one reciprocal of s_a s_b s_c d,
// peeled: 9 multiplies and one fast_rcp instead of 7 multiplies and two (v_rcp_f64 issues in 17 cycles,
// a multiply in 5.4: tools/valu_rate_probe.hip).  s_j in [2e-7, 2e9] and d in (0, 1e300): the product
// stays normal.  Each quotient carries two or three more roundings than fast_rcp's 2.2e-15.
__device__ __forceinline__ double wing_sum3_rcp4(double lamP, double msa, double msb, double msc, double cs,
                                                 bool *near, double d, double *inv_d) {
  const double xa = fma(lamP, msa, -cs), xb = fma(lamP, msb, -cs), xc = fma(lamP, msc, -cs);
  const double sa = fma(xa, xa, g_lines.y2[0]), sb = fma(xb, xb, g_lines.y2[1]), sc = fma(xc, xc, g_lines.y2[2]);
  *near = min(min(hi_word(sa), hi_word(sb)), hi_word(sc)) < 0x408C2000u;
  const double pab = sa * sb, pabc = pab * sc;
  const double rinv = fast_rcp(pabc * d);
  *inv_d = rinv * pabc;
  const double r3 = rinv * d;            // 1 / (sa sb sc)
  const double rc = r3 * pab, r2 = r3 * sc;  // 1 / sc, 1 / (sa sb)
  const double ra = r2 * sb, rb = r2 * sa;
  double ta = ra * kE5 + kE4, tb = rb * kE5 + kE4, tc = rc * kE5 + kE4;
  ta = fma(ta, ra, kE3); tb = fma(tb, rb, kE3); tc = fma(tc, rc, kE3);
  ta = fma(ta, ra, g_lines.t2[0]); tb = fma(tb, rb, g_lines.t2[1]); tc = fma(tc, rc, g_lines.t2[2]);
  ta = fma(ta, ra, kE1); tb = fma(tb, rb, kE1); tc = fma(tc, rc, kE1);
  ta = fma(ta, ra, 1.0); tb = fma(tb, rb, 1.0); tc = fma(tc, rc, 1.0);
  return fma(g_lines.cwing[2], rc * tc, fma(g_lines.cwing[1], rb * tb, g_lines.cwing[0] * (ra * ta)));
}

constexpr int kRing2 = 33;   // doubled raw-profile ring: 32 slots + 1 pad per sample
constexpr int kExpTab = 64;  // entries of the 2^(j/64) table behind exp_table()

// exp(x) for x <= 0 through a 64-entry table: x = (64 n + j) ln2/64 + r, |r| <= ln2/128,
// exp(x) = 2^n * 2^(j/64) * P5(r).  Relative error < 2e-16 (r^6/720 < 3.6e-17).
// In two halves so that a caller can request the table entry early (exp_table_begin), put other
// LDS traffic behind it, and finish later (exp_table_end) without waiting for that traffic.
struct ExpState {
  double r, tabv;
  int ni;
};
__device__ __forceinline__ ExpState exp_table_begin(double x, const double *tab) {
  // no clamp: for x << -745 the integer conversion saturates, ni >> 6 is hugely negative and
  // ldexp returns 0; the reduced argument stays tiny (nf is exact to 0.5 up to |x| ~ 1e13)
  ExpState e;
  const double nf = rint(x * 92.33248261689366);            // 64 / ln2
  // saturating conversion, spelled as the instruction: (int)nf is undefined out of range
  asm("v_cvt_i32_f64 %0, %1" : "=v"(e.ni) : "v"(nf));
  e.tabv = tab[e.ni & (kExpTab - 1)];
  e.r = fma(-nf, 0.010830424667801708, x);                    // ln2/64 high part (low 24 bits zero)
  e.r = fma(-nf, 2.8447437476627285e-11, e.r);                // ln2/64 low part
  return e;
}
__device__ __forceinline__ double exp_table_end(const ExpState &e) {
  // The first two steps are spelled as three-address v_fma_f64: the compiler picks the two-address
  // v_fmac_f64 and then has to copy the (register-resident) constant in front of each, 3 moves
  // per value; 0.5 and 1.0 below are inline constants and need none.
  double p, c4 = 0.041666666666666664, c3 = 0.16666666666666666;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "s"(0.008333333333333333), "v"(e.r), "v"(c4));
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(p), "v"(e.r), "v"(c3));
  p = fma(p, e.r, 0.5);
  p = fma(p, e.r, 1.0);
  p = fma(p, e.r, 1.0);
  return ldexp(e.tabv * p, e.ni >> 6);
}
// The same with the argument pre-scaled by the caller: t = x * 64/ln2 (fold the factor into whatever
// x is multiplied by anyway).  delta = t - rint(t) is exact, and the series runs in delta with the
// powers of ln2/64 folded into its coefficients: one multiply and one FMA fewer than the form
// above (which forms x * 64/ln2 and then reduces x in two parts).  Same table, same error: the
// rounding of t is the rounding x itself would have had.
constexpr double kExpScale = 92.33248261689366;  // 64 / ln2
__device__ __forceinline__ ExpState exp_table_begin_scaled(double t, const double *tab) {
  ExpState e;
  const double nf = rint(t);
  asm("v_cvt_i32_f64 %0, %1" : "=v"(e.ni) : "v"(nf));  // saturating (see exp_table_begin)
  e.tabv = tab[e.ni & (kExpTab - 1)];
  e.r = t - nf;
  return e;
}
__device__ __forceinline__ double exp_table_end_scaled(const ExpState &e) {
  // (ln2/64)^k / k!
  constexpr double L = 0.010830424696249145;
  constexpr double c1 = L, c2 = L * L / 2, c3 = L * L * L / 6, c4 = L * L * L * L / 24, c5 = L * L * L * L * L / 120;
  double p;
  const double c4v = c4;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "s"(c5), "v"(e.r), "v"(c4v));
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(p), "v"(e.r), "s"(c3));
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(p), "v"(e.r), "s"(c2));
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(p), "v"(e.r), "s"(c1));
  p = fma(p, e.r, 1.0);
  return ldexp(e.tabv * p, e.ni >> 6);
}
__device__ __forceinline__ double exp_table(double x, const double *tab) {
  return exp_table_end(exp_table_begin(x, tab));
}
__device__ __forceinline__ double exp_table_scaled(double t, const double *tab) {  // t = x * kExpScale
  return exp_table_end_scaled(exp_table_begin_scaled(t, tab));
}

// Optical-depth sum sqrt(pi) Sum_j lead_j Re w_j where some line is within 30 Doppler widths
// (voigt.c:282-289, reference two-rounding velocity): per line either the piecewise polynomial or
// the wing formula, a few dozen instructions inline -- no call, no divergent trapezoid sums.
template <int LINES>
__device__ __forceinline__ double total_near(double lamP, double m0, double m1, double m2,
                                             const double *mult_lds, int L) {
  const double c_light = g_lines.c, inv_s = g_lines.inv_sqrt2_sigma;
  const double *tab = g_lines.near_poly;
  double total = 0.0;  // sqrt(pi) Sum_j lead_j Re w_j, the convention of the wing tier
  if (LINES > 0) {
    const double mm[3] = {m0, m1, m2};
#pragma unroll
    for (int j = 0; j < (LINES > 0 ? LINES : 1); ++j) {
      const double ax = fabs((lamP * mm[j < 3 ? j : 0] - c_light) * inv_s);  // voigt.c:287
      const double f = ax < 30.0
                           ? 1.7724538509055159 * g_lines.leading[j] * near_poly(tab + j * kNearLineDoubles, ax)
                           : g_lines.cwing[j] * wing_core(ax * ax, g_lines.y2[j]);
      total += f;
    }
  } else {
    for (int j = 0; j < L; ++j) {
      const double ax = fabs((lamP * mult_lds[j] - c_light) * inv_s);
      const double f = ax < 30.0
                           ? 1.7724538509055159 * g_lines.leading[j] * near_poly(tab + j * kNearLineDoubles, ax)
                           : g_lines.cwing[j] * wing_core(ax * ax, g_lines.y2[j]);
      total += f;
    }
  }
  return total;
}

// Wing tier for a line count known at run time only: sqrt(pi) Sum_{j<L} lead_j Re w_j by the wing
// formula, from rho = lambda / (1 + z_DLA), kLineUnroll lines at a time -- their scalar loads are
// issued together and their dependent chains (reciprocal, Newton step, degree-5 Horner) interleave;
// line by line, a wave waited for three scalar loads and one ~20-deep chain per line (measured at 31
// lines on 200 x 1500 pixels x 10^4 samples: 115 ms one at a time).  *near: some line j < L is
// within 30 Doppler widths of this lane's pixel (the caller then takes total_near_at).
constexpr int kLineUnroll = 4;
__device__ __forceinline__ double wing_sum_runtime(double rho, double cs, int L, bool *near_out) {
  double total = 0.0;
  bool near = false;
  int j0 = 0;
  for (; j0 + kLineUnroll <= L; j0 += kLineUnroll) {  // whole groups
    double f[kLineUnroll];
#pragma unroll
    for (int u = 0; u < kLineUnroll; ++u) {
      const double x = fma(rho, g_lines.wing[j0 + u].kms, -cs);
      const double x2 = x * x;
      near |= x2 < 900.0;
      f[u] = wing_core(x2, g_lines.wing[j0 + u].y2);
    }
#pragma unroll
    for (int u = 0; u < kLineUnroll; ++u) total = fma(g_lines.wing[j0 + u].cwing, f[u], total);
  }
  for (; j0 < L; ++j0) {  // the last L mod kLineUnroll lines, one at a time (padding the group to
    // four cost 5 ms of 40 at five lines and of 31 at one)
    const double x = fma(rho, g_lines.wing[j0].kms, -cs);
    const double x2 = x * x;
    near |= x2 < 900.0;
    total = fma(g_lines.wing[j0].cwing, wing_core(x2, g_lines.wing[j0].y2), total);
  }
  *near_out = near;
  return total;
}

// The same sum with the reference's multiplier c / (wavelength_j (1 + z)) / 1e8 (voigt.c:278-279) formed
// here, line by line: for kernels that keep no table of it (k_sweep_slim<0>); two divisions per line,
// on the rare pixels within 30 Doppler widths of a line only.
__device__ __forceinline__ double total_near_at(double lamP, double one_plus_z, int L) {
  const double c_light = g_lines.c, inv_s = g_lines.inv_sqrt2_sigma;
  const double *tab = g_lines.near_poly;
  double total = 0.0;
  for (int j = 0; j < L; ++j) {
    const double mult = c_light / (g_lines.wavelength_cm[j] * one_plus_z) / 1e8;
    const double ax = fabs((lamP * mult - c_light) * inv_s);  // voigt.c:287
    const double f = ax < 30.0
                         ? 1.7724538509055159 * g_lines.leading[j] * near_poly(tab + j * kNearLineDoubles, ax)
                         : g_lines.cwing[j] * wing_core(ax * ax, g_lines.y2[j]);
    total += f;
  }
  return total;
}

// Diagnostic build only (-DGPDLA_STAMP, tools/stamps.sh): s_memtime brackets around the segments of
// a K-step, summed per wave in scalar registers and added to g_stamps once per wave.  The stamp
// drains lgkmcnt, so it forbids overlaps the real kernel has: read SHARES, never the run time.
#ifdef GPDLA_STAMP
__device__ unsigned long long g_stamps[8];
#define GPDLA_ST(i)                                                                     \
  {                                                                                     \
    unsigned long long t_;                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    st_sum[i] += (uint32_t)t_ - st_prev; /* a wave's whole sweep is < 2^32 cycles */    \
    st_prev = (uint32_t)t_;                                                             \
  }
#else
#define GPDLA_ST(i)
#endif

// Template parameters: NTW B tiles per wave, TS tile split, kChunkSteps records per LDS chunk
// (4, or 2 with two chunks unrolled: ring slots are then compile-time; 1: run-time slots),
// TW tiles that take the weight w (the rest take u; TS*NTW tiles in all, zero-padded), LINES
// number of Lyman lines when known at compile time (0: read num_lines at run time).
template <typename T, int WAVES, int NTW, int TS, int kChunkSteps, int TW, int LINES>
__global__ __launch_bounds__(WAVES * 64) void k_sweep(SweepArgs a) {
  extern __shared__ double smem[];
  static_assert(kChunkSteps == 8 || kChunkSteps == 4 || kChunkSteps == 2 || kChunkSteps == 1, "chunks of 8, 4, 2 or 1 K-steps");
  // chunks unrolled per loop iteration so that 4 K-steps (one turn of the 16-slot ring) are
  // straight-line code with compile-time ring slots and stage-buffer addresses
  constexpr int UN = kChunkSteps == 2 ? 2 : 1;
  constexpr int GROUPS = WAVES / TS;  // sample groups per block
  constexpr int NT = NTW * TS;
  constexpr int TD = 64 * (int)sizeof(T) / 8;  // doubles occupied by one 64-element tile
  constexpr int RD = NT * TD + record_extras(NT);
  constexpr bool kCompact = tiles_compact(NT);  // 13 + 1 tiles on the matrix cores, 2 + 4 columns on the VALU
  using acc_t = typename Mat<T>::acc_t;
  const int64_t xj = blockIdx.x >> 3;
  const int64_t pos = 8 * (xj / a.blocks_per_quasar) + (blockIdx.x & 7);
  const int bq = (int)(xj % a.blocks_per_quasar);
  if (pos >= a.nq) return;
  // quasars are dealt in order of decreasing length (a.order, built at upload), so the eight
  // quasars in flight -- one per XCD -- are of nearly equal length and the XCDs stay in step
  const int64_t q = a.order[pos];
  const QuasarMeta m = a.meta[q];
  if (m.status != 0) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int group = wave / TS, role = wave % TS;
  const int s = lane & 15, jj = lane >> 4;
  const int L = LINES > 0 ? LINES : a.num_lines;

  // LDS addressing costs VALU instructions (which cost MFMA time, fact 1 of DESIGN.md section 4)
  // whenever a constant does not fit the instruction's offset field: 8 bits x 8 B for
  // ds_read2_b64, 8 bits x 512 B for ds_read2st64_b64, 16 bits for ds_read_b64 / ds_read_b128.  So:
  // the ring, read with ds_read2_b64, sits at address 0 (a lane's 37 slots fit the 2040 B reach);
  // the exp table and the records' extras are read with b64 / b128; records are a multiple of
  // 512 B and the stage buffers 512-B aligned, so the tile reads fold (buffer, step, tile) into
  // the st64 offsets.
  double *ring = smem;                                             // [WAVES][16][33]
  double *exp_tab = ring + WAVES * kSamplesPerWave * kRing2;       // [64]
  double *stage = exp_tab + kExpTab;                               // [2][kChunkSteps][RD]
  double *mult_s = stage + (size_t)2 * kChunkSteps * RD;           // [GROUPS*16][L]

  const int64_t slot0 = (int64_t)bq * (GROUPS * kSamplesPerWave) + group * kSamplesPerWave;
  const int64_t slot = slot0 + s;
  const bool is_sample = slot < a.S;
  const bool is_null = !is_sample;  // slot == S is the null model; slots beyond it are idle copies
  const int32_t sample = is_sample ? a.perm[slot] : 0;
  // process_qsos.m:162-164
  const double z_dla = m.min_z_dla + (m.max_z_dla - m.min_z_dla) * a.offset_samples[sample];
  const double nhi = a.nhi_samples[sample];
  double *my_mult = mult_s + (size_t)(group * kSamplesPerWave + s) * L;
  double mult_r[LINES > 0 ? LINES : 1];
  if (LINES > 0) {
#pragma unroll
    for (int j = 0; j < LINES; ++j)  // voigt.c:278-279
      mult_r[j] = g_lines.c / (g_lines.wavelength_cm[j] * (1 + z_dla)) / 1e8;
  } else if (role == 0 && jj == 0) {
    for (int j = 0; j < L; ++j) my_mult[j] = g_lines.c / (g_lines.wavelength_cm[j] * (1 + z_dla)) / 1e8;
  }
  if (tid < kExpTab) exp_tab[tid] = exp2((double)tid * (1.0 / kExpTab));
  // this lane's ring row, already offset by its pixel phase jj (slots are then compile-time)
  double *my_ring = ring + (size_t)(wave * kSamplesPerWave + s) * kRing2 + jj;
  const double *lam = a.lam_pad + m.lam_off;
  const int n_pad = m.n_u + 6;
  // exp(N * total / (sqrt(2 pi) sigma)) with total = -Sum lead_j Re w_j (voigt.c:288-291)
  // (times 64/ln2: the exp below takes its argument pre-scaled, exp_table_begin_scaled)
  const double nscale64 = -nhi * g_lines.inv_sqrt2pi_sigma * kInvSqrtPi * kExpScale;
  const double *rec_base = a.records + m.rec_off * (int64_t)RD;
  const int nchunks = (m.steps + kChunkSteps - 1) / kChunkSteps;

  // asynchronous global -> LDS copy of one chunk of records (see glds_chunk)
  static_assert((kChunkSteps * RD) % 128 == 0, "a chunk is a whole number of KiB");
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t stage_lds = __builtin_amdgcn_readfirstlane(lds_address(stage));
  auto issue_chunk = [&](int c) {
    glds_chunk<kChunkSteps * RD / 128, WAVES>(rec_base + (size_t)c * kChunkSteps * RD,
                                              stage_lds + (uint32_t)(c & 1) * (uint32_t)(kChunkSteps * RD * 8), wave_s, lane);
  };
  issue_chunk(0);

  const double c_light = g_lines.c, inv_s = g_lines.inv_sqrt2_sigma;
  // wing tier in FMA form: x = lam * (mult_j / (sqrt2 sigma)) - c / (sqrt2 sigma).  It differs
  // from the reference's two-rounding velocity (voigt.c:287, kept in the accurate tier) by
  // <= 3e-12 in x, i.e. <= 2e-13 relative in the optical depth where |x| >= 30.
  double ms_r[LINES > 0 ? LINES : 1];
#pragma unroll
  for (int j = 0; j < (LINES > 0 ? LINES : 0); ++j) ms_r[j] = mult_r[j] * inv_s;
  const double cs = c_light * inv_s;
#define GPDLA_TOTAL_ACCURATE(lamP)                                                        \
  total_near<LINES>((lamP), mult_r[0], mult_r[LINES > 1 ? 1 : 0], mult_r[LINES > 2 ? 2 : 0], \
                    my_mult, L)
#define GPDLA_RAW_ACCURATE(lamP) exp_table_scaled(nscale64 * GPDLA_TOTAL_ACCURATE(lamP), exp_tab)

  __syncthreads();  // multipliers and the exp table visible
  // prime the ring with padded pixels 0..11 (the raw profile runs three K-steps ahead)
  for (int c3 = 0; c3 < 3; ++c3) {
    const double lam0 = lam[min(4 * c3 + jj, n_pad - 1)];
    double v;
    if (LINES == 3) {
      bool near0;
      v = exp_table_scaled(nscale64 * wing_sum3(lam0, ms_r[0], ms_r[LINES > 1 ? 1 : 0], ms_r[LINES > 2 ? 2 : 0],
                                                cs, &near0), exp_tab);
      if (__any(near0)) v = GPDLA_RAW_ACCURATE(lam0);
    } else {
      v = GPDLA_RAW_ACCURATE(lam0);
    }
    my_ring[4 * c3] = v;
    my_ring[4 * c3 + 16] = v;
  }

  acc_t acc[NTW];
#pragma unroll
  for (int c = 0; c < NTW; ++c) acc[c] = acc_t{0, 0, 0, 0};
  double quad_sum = 0.0, dprod = 1.0;
  double xw[kXW] = {0.0, 0.0}, xu[kXU] = {0.0, 0.0, 0.0, 0.0};  // compact class: columns kept off the MFMA
  int dexp = 0;
  const int tile0 = role * NTW;
  [[maybe_unused]] const int nw = TS == 1 ? TW : max(0, min(NTW, TW - tile0));  // w-tiles of this wave (ablation build)
  const double tap0 = g_lines.taps[0], tap1 = g_lines.taps[1], tap2 = g_lines.taps[2],
               tap3 = g_lines.taps[3];

  glds_wait();  // chunk 0 landed
  __syncthreads();

  // The fp64 MFMA does not overlap with VALU work on its SIMD (tools/fp64_mix_probe.hip), so the
  // loop is written for the fewest instructions, not for interleaving: per K-step one raw-profile
  // value three steps ahead (wing tier branch-free; accurate tier under a wave-uniform vote), the
  // 7-tap broadening from the doubled ring (no wrap-around: slots are compile-time), the weights,
  // then 16 MFMAs.  Two waves per SIMD hide LDS and dependent-issue latency.
  // Per K-step: all LDS operands (7 ring taps, pixel row, 16 B fragments) are requested first and
  // land while the raw-profile VALU chain runs; the only read that chain itself needs, the padded
  // wavelength, is fetched one step early (within a chunk), before the previous MFMA burst.
#ifdef GPDLA_STAMP
  uint32_t st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
  {
    unsigned long long t0_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_)::"memory");
    st_prev = (uint32_t)t0_;
  }
#endif
  for (int c0 = 0; c0 < nchunks; c0 += UN) {
#pragma unroll
  for (int hc = 0; hc < UN; ++hc) {
    const int c = c0 + hc;
    if (UN > 1 && c >= nchunks) break;  // block-uniform
    // Nothing of ours is in flight here (drained before the barrier), so this wait is free; it is
    // for the compiler, whose scratch reloads of the loop preheader would otherwise be waited for
    // inside the K-steps -- behind the prefetch issued on the next line.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    GPDLA_ST(4)  // chunk-end drain + barrier
    if (c + 1 < nchunks) issue_chunk(c + 1);  // lands in the other buffer while we compute
    GPDLA_ST(5)  // prefetch issue
    const double *buf = stage + (size_t)(UN > 1 ? hc : (c & 1)) * kChunkSteps * RD;  // UN = 2: c0 is even
    double lam_next = 0.0;
#pragma unroll
    for (int tt = 0; tt < kChunkSteps; ++tt) {
      const int rn = c * kChunkSteps + tt;
      if (rn < m.steps) {
        const double *rec = buf + (size_t)tt * RD;
        const double *extra = rec + NT * TD;
        // ring slot of pixel 4 rn (+ jj, folded into my_ring); compile-time when chunks are 4 long
        const int slot_p = kChunkSteps >= 4 ? (4 * tt) & 15 : (kChunkSteps == 2 ? 4 * (2 * hc + tt) : ((4 * rn) & 15));
        const int slot_w = (slot_p + 12) & 15;
        // lam_next was requested before the previous MFMA burst and has long landed: saying so
        // (s_waitcnt lgkmcnt(0), free) lets the raw chain start under the 15 reads issued next
        // instead of behind all of them.
        if (tt > 0) __builtin_amdgcn_s_waitcnt(0xC07F);
        const double *mine = extra + extras_row(NT, 1) * jj;  // this lane's pixel
        const double lamP = tt == 0 ? extra[extras_lam_stride(NT) * jj + extras_lam(NT, 0)] : lam_next;
        // operands of the broadening / weights / MFMAs of this step
        const double *g = my_ring + slot_p;
        const double g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3], g4 = g[4], g5 = g[5], g6 = g[6];
        const double2 p01 = *reinterpret_cast<const double2 *>(mine);  // ds_read_b128
        const double2 p23 = *reinterpret_cast<const double2 *>(mine + 2);
        const double py = p01.x, pmu = p01.y, pom = p23.x, pnu = p23.y;
        __builtin_amdgcn_sched_barrier(0);  // keep the reads up here (the scheduler sinks them)
        // (2) instrument broadening for pixel 4 rn + jj: voigt.c:297-299 (symmetric taps) -- in front of
        // (1) since round 5: the three-line wing tier takes d along (wing_sum3_rcp4)
        double absorb = fma(g6, tap0, g0 * tap0);
        {
          double ab2 = fma(g5, tap1, g1 * tap1);
          absorb = fma(g2, tap2, absorb);
          ab2 = fma(g4, tap2, ab2);
          absorb = fma(g3, tap3, absorb) + ab2;
        }
        if (is_null) absorb = 1.0;
        const double a2 = absorb * absorb;
        const double d = fma(pom, a2, pnu);
        double inv_d;
        // (1) raw profile three K-steps ahead: voigt.c:282-292
        double total;
        bool near;
        if (LINES == 3) {
          total = wing_sum3_rcp4(lamP, ms_r[0], ms_r[LINES > 1 ? 1 : 0], ms_r[LINES > 2 ? 2 : 0], cs, &near, d, &inv_d);
        } else {
          inv_d = fast_rcp(d);
          total = 0.0;
          near = false;
          for (int j = 0; j < L; ++j) {
            const double x = fma(lamP, my_mult[j] * inv_s, -cs);
            const double x2 = x * x;
            near |= x2 < 900.0;
            total = fma(g_lines.cwing[j], wing_core(x2, g_lines.y2[j]), total);
          }
        }
        GPDLA_ST(0)  // operand requests + wing-tier optical depth
#ifndef GPDLA_ABLATE_NOSLOW
        if (__builtin_expect(__any(near), 0)) total = GPDLA_TOTAL_ACCURATE(lamP);
#endif
        GPDLA_ST(1)  // accurate tier
        // B fragments of this step: requested only now, so that they are not live across the
        // accurate-tier branch above (it would spill to make room); they land during the rest of
        // the step, which is ONE basic block from here on: the exp chain of the raw profile and the
        // broadening / weight chain of this step's pixel are independent, and the scheduler
        // interleaves them, so a wave running alone on its SIMD does not sit out their latencies.
        // (the exp table entry is requested ahead of them: LDS returns in order, and the exp chain
        // then waits for one read instead of nine)
        const ExpState es = exp_table_begin_scaled(nscale64 * total, exp_tab);
        __builtin_amdgcn_sched_barrier(0);
        const T *bt = reinterpret_cast<const T *>(rec) + (size_t)tile0 * 64 + lane;
        T bop[NTW];
#ifdef GPDLA_ABLATE_NOBFRAG
#pragma unroll
        for (int cc = 0; cc < NTW; ++cc) bop[cc] = (T)(lane + cc) * (T)lamP;  // timing experiment: no fragment reads
        (void)bt;
#else
#pragma unroll
        for (int cc = 0; cc < NTW; ++cc) bop[cc] = bt[(size_t)cc * 64];
#endif
        __builtin_amdgcn_sched_barrier(0);
        double raw = exp_table_end_scaled(es);
#ifdef GPDLA_ABLATE_NOVOIGT
        raw = lamP * 1e-4;
#endif
        my_ring[slot_w] = raw;
        my_ring[slot_w + 16] = raw;
        // (3) weights: process_qsos.m:192-198 folded into log_mvnpdf_low_rank.m:11-15
        const double r = fma(-absorb, pmu, py);
        const double w = a2 * inv_d;
        const double ri = r * inv_d;
        const double u = absorb * ri;
        quad_sum = fma(r, ri, quad_sum);
        // Sum log d as the log of a running product, renormalised every second step (the
        // mantissa times two factors stays in range for any d in [1e-150, 1e150])
        dprod *= d;
        if (kChunkSteps == 1 || (tt & 1)) {
          dexp += __builtin_amdgcn_frexp_exp(dprod);
          dprod = __builtin_amdgcn_frexp_mant(dprod);
        }
        // next step's wavelength, in flight during the MFMA burst
        if (tt + 1 < kChunkSteps) {
          lam_next = extra[RD + extras_lam_stride(NT) * jj + extras_lam(NT, 0)];
          __builtin_amdgcn_sched_barrier(0);
        }
        GPDLA_ST(2)  // fragments, ring, broadening, weights
        // (4) rank-4 update of [B | v] on the matrix cores
#ifdef GPDLA_ABLATE_NOMFMA
#pragma unroll
        for (int cc = 0; cc < NTW; ++cc) {
          asm volatile("" ::"v"(bop[cc]));
          if (cc < 2) acc[cc][0] += (T)(cc < nw ? w : u) * bop[cc];
        }
#else
        // With a tile split only the last role's last NT - TW tiles take u; one wave-uniform
        // select per step instead of one per tile.
        constexpr int kTail = TS == 1 ? 0 : NT - TW;
        static_assert(TS == 1 || kTail <= NTW, "u-tiles must sit in the last role's tiles");
        const T a_tail = (T)(role == TS - 1 ? u : w);
#pragma unroll
        for (int cc = 0; cc < NTW; ++cc) {
          const T aop = TS == 1 ? (T)(cc < TW ? w : u) : (cc < NTW - kTail ? (T)w : a_tail);
          acc[cc] = Mat<T>::mfma(aop, bop[cc], acc[cc]);
        }
#endif
        if (kCompact) {  // vech columns 208, 209 and m columns 16..19 of this lane's pixel: 6 FMAs
          static_assert(kXW == 2 && kXU == 4, "three 16-byte reads");
          const double2 xp = *reinterpret_cast<const double2 *>(mine + kExtrasXW);
          const double2 u01 = *reinterpret_cast<const double2 *>(mine + kExtrasXU);
          const double2 u23 = *reinterpret_cast<const double2 *>(mine + kExtrasXU + 2);
          xw[0] = fma(w, xp.x, xw[0]);
          xw[1] = fma(w, xp.y, xw[1]);
          xu[0] = fma(u, u01.x, xu[0]);
          xu[1] = fma(u, u01.y, xu[1]);
          xu[2] = fma(u, u23.x, xu[2]);
          xu[3] = fma(u, u23.y, xu[3]);
        }
        GPDLA_ST(3)  // MFMA burst (issue)
      }
    }
#ifdef GPDLA_ABLATE_NOBARRIER
    __builtin_amdgcn_s_waitcnt(0);  // timing experiment only: races on the stage buffers
#else
    glds_wait();  // the prefetched chunk has landed (this wave's part) ...
    __syncthreads();  // ... and everyone's; all reads of the buffer refilled next are done
#endif
  }
  }
#undef GPDLA_RAW_ACCURATE
#undef GPDLA_TOTAL_ACCURATE
  // per-sample scalar sums: combine the four pixel phases jj of each sample
  double logd_sum = log(dprod) + (double)dexp * 0.6931471805599453;
  quad_sum += __shfl_xor(quad_sum, 16);
  quad_sum += __shfl_xor(quad_sum, 32);
  logd_sum += __shfl_xor(logd_sum, 16);
  logd_sum += __shfl_xor(logd_sum, 32);
  if (kCompact) {
#pragma unroll
    for (int x = 0; x < kXW; ++x) {
      xw[x] += __shfl_xor(xw[x], 16);
      xw[x] += __shfl_xor(xw[x], 32);
    }
#pragma unroll
    for (int x = 0; x < kXU; ++x) {
      xu[x] += __shfl_xor(xu[x], 16);
      xu[x] += __shfl_xor(xu[x], 32);
    }
  }

#undef GPDLA_RAW_ACCURATE
  // ---- epilogue: factor_pass over the MFMA result registers ------------------------------------
  using ES = EpilogueShape<TW, TS>;
  double *Eg = smem + kExpTab + (size_t)group * ES::SPP * ES::stride(logical_tiles(NT));  // [SPP samples][stride] of this group
#pragma unroll
  for (int p = 0; p < ES::PASSES; ++p) {
    int sigma;
    bool writer;
    const double ll = factor_pass<T, NTW, TS, TW>(acc, xw, xu, p, Eg, lane, role, tile0, a.k, quad_sum,
                                                  logd_sum, m.n_kept, &sigma, &writer);
    const int64_t slot_s = slot0 + sigma;
    const int32_t sample_s = __shfl(sample, sigma + 16 * jj);
    if (writer) {
      if (slot_s < a.S) a.sample_ll[(int64_t)q * a.S + sample_s] = ll + m.ll_bias;
      else if (slot_s == a.S) a.ll_no_dla[q] = ll + m.ll_bias;
    }
  }
#ifdef GPDLA_STAMP
  GPDLA_ST(6)  // epilogue (after the last chunk barrier)
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_stamps[i], (unsigned long long)st_sum[i]);
#endif
}

// ------------------------------------------------------------------------------------------
// k_evidence: one block per quasar.  process_qsos.m:203-213 and :224-233.
// summary row: see GPDLA_SUMMARY_COLS in gpdla.h.
// ------------------------------------------------------------------------------------------
struct EvidenceArgs {
  const QuasarMeta *meta;
  const double *sample_ll;   // [nq][S]
  const double *ll_no_dla;   // [nq]
  const double *log_prior_no_dla, *log_prior_dla;
  const double *offset_samples, *nhi_samples, *log_nhi_samples;  // log_nhi_samples may be null
  int64_t S;
  double *summary;           // [nq][kSummaryCols]
};
constexpr int kSummaryCols = 15;

__global__ __launch_bounds__(256) void k_evidence(EvidenceArgs a) {
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ double sh[4];
  __shared__ long long sh_arg[4];
  const QuasarMeta m = a.meta[q];
  double *out = a.summary + (int64_t)q * kSummaryCols;
  if (m.status != 0) {
    if (tid < kSummaryCols) out[tid] = NAN;
    if (tid == 2) out[2] = a.log_prior_no_dla[q];
    if (tid == 3) out[3] = a.log_prior_dla[q];
    return;
  }
  const double *ll = a.sample_ll + (int64_t)q * a.S;
  double mx = -INFINITY;
  long long arg = a.S;  // first index attaining the nanmax (generate_ascii_catalog.m:73)
  for (int64_t i = tid; i < a.S; i += 256) {
    const double v = ll[i];
    if (v > mx || (v == mx && i < arg)) {  // NaN compares false: skipped like MATLAB's max (:203)
      mx = v;
      arg = i;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double om = __shfl_xor(mx, o);
    const long long oa = __shfl_xor(arg, o);
    if (om > mx || (om == mx && oa < arg)) {
      mx = om;
      arg = oa;
    }
  }
  if (lane == 0) {
    sh[wave] = mx;
    sh_arg[wave] = arg;
  }
  __syncthreads();
  mx = sh[0];
  arg = sh_arg[0];
  for (int w = 1; w < 4; ++w)
    if (sh[w] > mx || (sh[w] == mx && sh_arg[w] < arg)) {
      mx = sh[w];
      arg = sh_arg[w];
    }
  if (arg >= a.S) arg = 0;  // all NaN (or all -inf): MATLAB's nanmax returns index 1
  double sum = 0.0;
  for (int64_t i = tid; i < a.S; i += 256) sum += exp(ll[i] - mx);  // :205-207 (NaN propagates)
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  __syncthreads();
  if (lane == 0) sh[wave] = sum;
  __syncthreads();
  if (tid == 0) {
    sum = sh[0] + sh[1] + sh[2] + sh[3];
    const double ll_dla = mx + log(sum / (double)a.S);                   // :209-210
    const double ll_no = a.ll_no_dla[q];
    const double lp_no = a.log_prior_no_dla[q] + ll_no;                  // :153-154
    const double lp_dla = a.log_prior_dla[q] + ll_dla;                   // :212-213
    const double mxp = fmax(lp_no, lp_dla);                              // :224-225
    double p0 = exp(lp_no - mxp), p1 = exp(lp_dla - mxp);                // :227-228
    const double tot = p0 + p1;
    p0 /= tot;                                                           // :230
    p1 /= tot;
    out[0] = m.min_z_dla;
    out[1] = m.max_z_dla;
    out[2] = a.log_prior_no_dla[q];
    out[3] = a.log_prior_dla[q];
    out[4] = ll_no;
    out[5] = ll_dla;
    out[6] = lp_no;
    out[7] = lp_dla;
    out[8] = p0;
    out[9] = p1;
    out[10] = p0;        // p_no_dlas, :232
    out[11] = 1 - p0;    // p_dlas,    :233
    // generate_ascii_catalog.m:73-80
    out[12] = (double)(arg + 1);
    out[13] = m.min_z_dla + (m.max_z_dla - m.min_z_dla) * a.offset_samples[arg];
    out[14] = a.log_nhi_samples ? a.log_nhi_samples[arg] : log10(a.nhi_samples[arg]);
  }
}

// ------------------------------------------------------------------------------------------
// Stand-alone surfaces.
// ------------------------------------------------------------------------------------------

// voigt.c:278-292, one thread per padded pixel.
__global__ void k_voigt_raw(const double *lambdas, int64_t n, double z, double N, int num_lines,
                            double *raw) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double total = 0.0;
  for (int j = 0; j < num_lines; ++j) {
    const double mult = g_lines.c / (g_lines.wavelength_cm[j] * (1 + z)) / 1e8;  // voigt.c:279
    const double velocity = lambdas[i] * mult - g_lines.c;                       // voigt.c:287
    const double v = rew_full(velocity * g_lines.inv_sqrt2_sigma, g_lines.y[j]) * g_lines.inv_sqrt2pi_sigma;
    total += -g_lines.leading[j] * v;                                            // voigt.c:288
  }
  raw[i] = exp(N * total);                                                       // voigt.c:291
}

// voigt.c:297-299, one thread per output pixel.
__global__ void k_voigt_broaden(const double *raw, int64_t n_out, double *profile) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  double acc = 0.0;
  for (int kk = 0; kk < 7; ++kk) acc += raw[i + kk] * g_lines.taps[kk];
  profile[i] = acc;
}

// log_mvnpdf_low_rank.m:5-34 for one (y, mu, M, d): a single 256-thread block.
// ws: k(k+1)/2 + k + 2 doubles of workspace.  status: 0 ok, 1 not PD.
__global__ __launch_bounds__(256) void k_lowrank_single(const double *y, const double *mu,
                                                        const double *M, const double *d, int64_t n,
                                                        int k, double *ws, double *log_p,
                                                        int *status) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ double sh[4];
  const int nb = k * (k + 1) / 2;
  // B (lower triangle) and v = M' D^-1 (y - mu): one output entry per thread, strided
  for (int e = tid; e < nb + k; e += 256) {
    double acc = 0.0;
    if (e < nb) {
      int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
      while ((i + 1) * (i + 2) / 2 <= e) ++i;
      while (i * (i + 1) / 2 > e) --i;
      const int j = e - i * (i + 1) / 2;
      for (int64_t p = 0; p < n; ++p) acc = fma(M[p + i * n] / d[p], M[p + j * n], acc);
      if (i == j) acc += 1.0;
    } else {
      const int i = e - nb;
      for (int64_t p = 0; p < n; ++p) acc = fma(M[p + i * n], (y[p] - mu[p]) / d[p], acc);
    }
    ws[e] = acc;
  }
  double qs = 0.0, ld = 0.0;
  for (int64_t p = tid; p < n; p += 256) {
    const double r = y[p] - mu[p];
    qs = fma(r, r / d[p], qs);
    ld += log(d[p]);
  }
  for (int o = 32; o > 0; o >>= 1) {
    qs += __shfl_xor(qs, o);
    ld += __shfl_xor(ld, o);
  }
  if (lane == 0) sh[wave] = qs;
  __syncthreads();
  qs = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  if (lane == 0) sh[wave] = ld;
  __syncthreads();
  ld = sh[0] + sh[1] + sh[2] + sh[3];
  __threadfence_block();
  __syncthreads();
  // Cholesky of the packed lower triangle, column by column: the pivot by thread 0, the column
  // below it by all threads (left-looking), then z = L^-1 v (log_mvnpdf_low_rank.m:24-28)
  __shared__ double s_piv;
  __shared__ int s_pd;
  if (tid == 0) s_pd = 1;
  double *v = ws + nb;
  for (int j = 0; j < k; ++j) {
    const int rj = j * (j + 1) / 2;
    __syncthreads();
    if (tid == 0) {
      double sum = ws[rj + j];
      for (int mm = 0; mm < j; ++mm) sum = fma(-ws[rj + mm], ws[rj + mm], sum);
      if (!(sum > 0.0)) s_pd = 0;
      const double ljj = sqrt(sum);
      ws[rj + j] = ljj;
      s_piv = ljj;
    }
    __syncthreads();
    const double ljj = s_piv;
    for (int i = j + 1 + tid; i < k; i += 256) {
      const int ri = i * (i + 1) / 2;
      double sum = ws[ri + j];
      for (int mm = 0; mm < j; ++mm) sum = fma(-ws[ri + mm], ws[rj + mm], sum);
      ws[ri + j] = sum / ljj;
    }
  }
  __syncthreads();
  if (tid == 0) {
    double log_diag = 0.0, zz = 0.0;
    for (int i = 0; i < k; ++i) {
      const int ri = i * (i + 1) / 2;
      log_diag += log(ws[ri + i]);
      double zi = v[i];
      for (int mm = 0; mm < i; ++mm) zi = fma(-ws[ri + mm], v[mm], zi);
      zi /= ws[ri + i];
      v[i] = zi;
      zz = fma(zi, zi, zz);
    }
    const bool pd = s_pd != 0;
    *log_p = pd ? -0.5 * ((qs - zz) + ld + 2 * log_diag + (double)n * kLog2Pi) : NAN;
    *status = pd ? 0 : 1;
  }
}

}  // namespace gpdla
