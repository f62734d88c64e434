// sweep_kernels.hpp -- device side of the GP/DLA inference sweep for gfx950 (MI355X).
//
// Kernels (reference lines they replace; paths relative to the reference tree):
//   k_prepare   process_qsos.m:102-119, 138-146, 159-176   per-quasar pixel selection, GP
//                                                          interpolation, noise scaling, padding
//   k_build_pm  (no reference counterpart)                  packs vech(m m') | m per pixel into
//                                                          MFMA B-operand tiles
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

#include "faddeeva.hpp"

namespace gpdla {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int kMaxLines = 31;
constexpr int kWavesPerBlock = 4;
constexpr int kSamplesPerWave = 16;   // rows of the 16x16x4 MFMA
constexpr int kChunkSteps = 4;        // K-steps of B tiles staged in LDS at a time
constexpr int kRingStride = 17;       // doubles per sample row of the raw-profile ring (16 + pad)
constexpr double kLog2Pi = 1.83787706640934534;  // log_mvnpdf_low_rank.m:7

// Lyman-series constants in device constant memory (filled once per process from
// include/gpdla_lyman_series.h).
struct LineTable {
  double wavelength_cm[kMaxLines];  // voigt.c:31
  double leading[kMaxLines];        // voigt.c:151
  double y[kMaxLines];              // gamma_j / (sqrt2 sigma): damping parameter of w(z)
  double taps[7];                   // voigt.c:242-251
  double c;                         // voigt.c:22
  double inv_sqrt2_sigma;           // 1/(sqrt2 sigma)
  double inv_sqrt2pi_sigma;         // 1/(sqrt(2 pi) sigma)
};
__constant__ LineTable g_lines;

// Per-quasar metadata produced by k_prepare.
struct QuasarMeta {
  int32_t n_u;       // pixels in the modelled rest range (process_qsos.m:104-108)
  int32_t n_kept;    // of those, not masked (:110)
  int32_t steps;     // ceil(n_u / 4): K-steps of the contraction
  int32_t status;    // 0 ok, 1 empty
  double min_z_dla;  // :159
  double max_z_dla;  // :160
  int64_t pix_off;   // first row of this quasar in the pixel pools (multiple of 4)
  int64_t lam_off;   // first entry in the padded-wavelength pool
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
        row.mu = a.model.mu[lo] + (a.model.mu[lo + 1] - a.model.mu[lo]) * t;          // :138
        const double lo_om = a.model.log_omega[lo] +
                             (a.model.log_omega[lo + 1] - a.model.log_omega[lo]) * t;  // :141
        const double omega2 = exp(2 * lo_om);                                          // :142
        const double lya_z = (wl - a.cfg.lya_wavelength) / a.cfg.lya_wavelength;       // :117-119
        const double sc = 1 - exp(-a.model.tau_0 * pow(1 + lya_z, a.model.beta)) + a.model.c_0; // :144
        row.omega2 = omega2 * (sc * sc);                                               // :146
      }
      pix[u] = row;
      for (int c = 0; c < k; ++c) {                                                    // :139
        const double m0 = a.model.M[lo + (int64_t)c * G], m1 = a.model.M[lo + 1 + (int64_t)c * G];
        Mi[(int64_t)u * k + c] = keep ? m0 + (m1 - m0) * t : 0.0;
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
  const int steps = (n_u + 3) >> 2;
  // pad rows up to 4*steps: neutral pixels
  for (int u = n_u + tid; u < 4 * steps; u += 256) {
    pix[u] = PixelRow{0.0, 0.0, 0.0, 1.0};
    for (int c = 0; c < k; ++c) Mi[(int64_t)u * k + c] = 0.0;
  }
  if (tid == 0) {
    m.n_u = n_u;
    m.n_kept = n_kept;
    m.steps = steps;
    m.status = (n_kept > 0) ? 0 : 1;
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
// k_build_pm: B-operand tiles.  PM[q][step][tile][64]: lane l = 16*jj + col holds, for pixel
// 4*step + jj, column 16*tile + col of [vech(m m') | m]  (row-wise lower triangle:
// idx(i, j) = i(i+1)/2 + j, j <= i).  tiles_w = ceil(k(k+1)/2 / 16) tiles take the weight w,
// the following tiles_u = ceil(k/16) take u.
// ------------------------------------------------------------------------------------------
struct BuildPmArgs {
  const QuasarMeta *meta;
  const double *Mi;
  double *pm;             // pool, [pix_off/4 + step][ntiles][64]
  int32_t k, tiles_w, ntiles;
  int32_t blocks_per_quasar;
};

__global__ __launch_bounds__(256) void k_build_pm(BuildPmArgs a) {
  const int q = blockIdx.x / a.blocks_per_quasar;
  const int bq = blockIdx.x % a.blocks_per_quasar;
  const QuasarMeta m = a.meta[q];
  const int64_t per_step = (int64_t)a.ntiles * 64;
  const int64_t total = (int64_t)m.steps * per_step;
  const int ncol_w = a.k * (a.k + 1) / 2;
  for (int64_t e = (int64_t)bq * 256 + threadIdx.x; e < total;
       e += (int64_t)a.blocks_per_quasar * 256) {
    const int step = (int)(e / per_step);
    const int rem = (int)(e - (int64_t)step * per_step);
    const int tile = rem >> 6, l = rem & 63;
    const int jj = l >> 4, col = l & 15;
    const double *row = a.Mi + (m.pix_off + 4 * (int64_t)step + jj) * a.k;
    double v = 0.0;
    if (tile < a.tiles_w) {
      const int c = tile * 16 + col;
      if (c < ncol_w) {
        int i = (int)((sqrt(8.0 * c + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= c) ++i;
        while (i * (i + 1) / 2 > c) --i;
        const int j = c - i * (i + 1) / 2;
        v = row[i] * row[j];
      }
    } else {
      const int c = (tile - a.tiles_w) * 16 + col;
      if (c < a.k) v = row[c];
    }
    a.pm[(m.pix_off / 4) * per_step + e] = v;
  }
}

// ------------------------------------------------------------------------------------------
// k_sweep.
//
// Grid: 8 * ceil(nq/8) * blocks_per_quasar blocks of 256 threads = 4 waves, flattened in x.
// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 names the XCD group), so block i
// works on quasar 8*(i/8/bpq) + i%8: every XCD streams ONE quasar's 3 MB of B tiles at a time
// and keeps them in its own 4 MB L2.  A wave owns 16 sample slots
// (MFMA rows); lane l = 16*jj + s works on sample slot s and, at K-step t, on pixel 4t + jj.
// Slot S (one past the last sample) is the null model (a = 1): process_qsos.m:149-151.
// Samples are visited in ascending z_DLA order (perm), so the lanes of a wave sit within a few
// pixels of each other relative to every line centre and the accurate-Faddeeva branch is taken
// by whole waves.
//
// TS ("tile split"): number of waves that share one group of 16 samples and split the B tiles
// between them (1 for k <= 20; 4 for k <= 40, where 55 tiles of accumulators do not fit one wave).
// ------------------------------------------------------------------------------------------
struct SweepArgs {
  const QuasarMeta *meta;
  const PixelRow *pix;
  const double *lam_pad;
  const double *pm;
  const double *offset_samples;   // [S]
  const double *nhi_samples;      // [S]
  const int32_t *perm;            // [S] sample indices in ascending offset (= z_DLA) order
  int64_t S;
  int64_t nq;
  int32_t blocks_per_quasar;
  int32_t k, tiles_w, ntiles, num_lines;
  double *sample_ll;              // [nq][S]   process_qsos.m:196
  double *ll_no_dla;              // [nq]      process_qsos.m:149
};

template <int NTW, int TS>
__global__ __launch_bounds__(256) void k_sweep(SweepArgs a) {
  extern __shared__ double smem[];
  constexpr int GROUPS = kWavesPerBlock / TS;          // sample groups per block
  const int64_t xj = blockIdx.x >> 3;
  const int64_t q = 8 * (xj / a.blocks_per_quasar) + (blockIdx.x & 7);
  const int bq = (int)(xj % a.blocks_per_quasar);
  if (q >= a.nq) return;
  const QuasarMeta m = a.meta[q];
  if (m.status != 0) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int group = wave / TS, role = wave % TS;
  const int s = lane & 15, jj = lane >> 4;
  const int L = a.num_lines;
  const int ntiles = a.ntiles;

  // LDS carve-up.  During the loop: B-tile chunk | raw-profile rings | per-sample multipliers.
  // After the loop the whole region is reused for the Cholesky epilogue.
  double *pm_s = smem;                                                  // [kChunkSteps][ntiles][64]
  double *ring = pm_s + (size_t)kChunkSteps * ntiles * 64;             // [4 waves][16][17]
  double *mult_s = ring + kWavesPerBlock * kSamplesPerWave * kRingStride;  // [GROUPS*16][L]

  const int64_t slot = (int64_t)bq * (GROUPS * kSamplesPerWave) + group * kSamplesPerWave + s;
  const bool is_sample = slot < a.S;
  const bool is_null = !is_sample;  // slot == S is the null model; slots beyond it are idle copies
  const int32_t sample = is_sample ? a.perm[slot] : 0;
  // process_qsos.m:162-164
  const double z_dla = m.min_z_dla + (m.max_z_dla - m.min_z_dla) * a.offset_samples[sample];
  const double nhi = a.nhi_samples[sample];
  double *my_mult = mult_s + (size_t)(group * kSamplesPerWave + s) * L;
  if (role == 0 && jj == 0) {
    for (int j = 0; j < L; ++j)  // voigt.c:278-279
      my_mult[j] = g_lines.c / (g_lines.wavelength_cm[j] * (1 + z_dla)) / 1e8;
  }
  double *my_ring = ring + (size_t)(wave * kSamplesPerWave + s) * kRingStride;
  const double *lam = a.lam_pad + m.lam_off;
  const PixelRow *pix = a.pix + m.pix_off;
  const int n_pad = m.n_u + 6;
  const double nscale = nhi * g_lines.inv_sqrt2pi_sigma;
  __syncthreads();

  // raw (un-broadened) profile at padded pixel P: voigt.c:282-292
  auto raw_at = [&](int P) -> double {
    const double lamP = lam[P < n_pad ? P : n_pad - 1];
    double total = 0.0;
    bool near = false;
    for (int j = 0; j < L; ++j) {
      const double velocity = lamP * my_mult[j] - g_lines.c;  // voigt.c:287 (two roundings)
      const double x = fabs(velocity * g_lines.inv_sqrt2_sigma);
      near |= x < 30.0;
      total = fma(-g_lines.leading[j], rew_wing(x, g_lines.y[j]), total);
    }
    if (__any(near)) {
      total = 0.0;
      for (int j = 0; j < L; ++j) {
        const double velocity = lamP * my_mult[j] - g_lines.c;
        total = fma(-g_lines.leading[j],
                    rew_full(velocity * g_lines.inv_sqrt2_sigma, g_lines.y[j]), total);
      }
    }
    return exp(nscale * total);  // voigt.c:291
  };

  // prime the ring with padded pixels 0..7 (K-steps -2 and -1)
  my_ring[jj] = raw_at(jj);
  my_ring[4 + jj] = raw_at(4 + jj);

  d4 acc[NTW];
#pragma unroll
  for (int c = 0; c < NTW; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
  double quad_sum = 0.0, logd_sum = 0.0;
  const int tile0 = role * NTW;

  for (int c0 = 0; c0 < m.steps; c0 += kChunkSteps) {
    const int csteps = min(kChunkSteps, m.steps - c0);
    __syncthreads();  // previous chunk fully consumed
    {
      const double2 *src =
          reinterpret_cast<const double2 *>(a.pm + ((m.pix_off / 4) + c0) * (int64_t)ntiles * 64);
      double2 *dst = reinterpret_cast<double2 *>(pm_s);
      const int n2 = csteps * ntiles * 32;
      for (int e = tid; e < n2; e += 256) dst[e] = src[e];
    }
    __syncthreads();
    for (int t = c0; t < c0 + csteps; ++t) {
      // (1) raw profile two K-steps ahead -> ring
      const int P = 4 * (t + 2) + jj;
      const double raw = raw_at(P);
      my_ring[P & 15] = raw;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // (2) instrument broadening for pixel p = 4t + jj: voigt.c:297-299
      const int p = 4 * t + jj;
      double absorb = 0.0;
#pragma unroll
      for (int kk = 0; kk < 7; ++kk) absorb = fma(my_ring[(p + kk) & 15], g_lines.taps[kk], absorb);
      __builtin_amdgcn_wave_barrier();
      if (is_null) absorb = 1.0;
      // (3) weights: process_qsos.m:192-198 folded into log_mvnpdf_low_rank.m:11-15
      const PixelRow px = pix[p];
      const double r = fma(-absorb, px.mu, px.y);
      const double a2 = absorb * absorb;
      const double d = fma(px.omega2, a2, px.nu);
      const double inv_d = 1.0 / d;
      const double w = a2 * inv_d;
      const double u = absorb * r * inv_d;
      quad_sum = fma(r * r, inv_d, quad_sum);
      logd_sum += log(d);
      // (4) rank-4 update of [B | v] on the matrix cores
      const double *bt = pm_s + (size_t)(t - c0) * ntiles * 64 + lane;
#pragma unroll
      for (int c = 0; c < NTW; ++c) {
        const int tile = tile0 + c;
        if (tile < ntiles) {
          const double b = bt[(size_t)tile * 64];
          acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(tile < a.tiles_w ? w : u, b, acc[c], 0, 0, 0);
        }
      }
    }
  }
  // per-sample scalar sums: combine the four pixel phases jj of each sample
  quad_sum += __shfl_xor(quad_sum, 16);
  quad_sum += __shfl_xor(quad_sum, 32);
  logd_sum += __shfl_xor(logd_sum, 16);
  logd_sum += __shfl_xor(logd_sum, 32);

  // ---- epilogue: k x k Cholesky + forward solve per sample, in LDS --------------------------
  __syncthreads();  // everyone is done with the loop's LDS
  const int ncols = ntiles * 16;
  constexpr int ES = kSamplesPerWave + 1;  // padded sample stride
  double *E = smem + (size_t)group * ncols * ES;  // [ncols][17]
  // MFMA C/D layout (f64 16x16x4): lane l, register r -> row (l >> 4) + 4 r, column l & 15.
#pragma unroll
  for (int c = 0; c < NTW; ++c) {
    const int tile = tile0 + c;
    if (tile < ntiles) {
#pragma unroll
      for (int r = 0; r < 4; ++r) E[(size_t)(tile * 16 + s) * ES + jj + 4 * r] = acc[c][r];
    }
  }
  __syncthreads();
  if (role == 0 && jj == 0) {
    const int k = a.k;
    double *col = E + s;  // this sample's entries: col[c * ES]
    double *vv = col + (size_t)a.tiles_w * 16 * ES;
    double log_diag = 0.0, zz = 0.0;
    bool pd = true;
    for (int i = 0; i < k; ++i) {
      const int ri = i * (i + 1) / 2;
      for (int j = 0; j <= i; ++j) {
        const int rj = j * (j + 1) / 2;
        double sum = col[(size_t)(ri + j) * ES] + (i == j ? 1.0 : 0.0);  // log_mvnpdf_low_rank.m:22-23
        for (int mm = 0; mm < j; ++mm)
          sum = fma(-col[(size_t)(ri + mm) * ES], col[(size_t)(rj + mm) * ES], sum);
        if (i == j) {                                                    // :24
          pd = pd && (sum > 0.0);
          const double lii = sqrt(sum);
          log_diag += log(lii);
          col[(size_t)(ri + j) * ES] = lii;
        } else {
          col[(size_t)(ri + j) * ES] = sum / col[(size_t)(rj + j) * ES];
        }
      }
      double zi = vv[(size_t)i * ES];  // forward solve L z = v
      for (int mm = 0; mm < i; ++mm) zi = fma(-col[(size_t)(ri + mm) * ES], vv[(size_t)mm * ES], zi);
      zi /= col[(size_t)(ri + i) * ES];
      vv[(size_t)i * ES] = zi;
      zz = fma(zi, zi, zz);
    }
    // log_mvnpdf_low_rank.m:30-32
    const double log_det = logd_sum + 2 * log_diag;
    double ll = -0.5 * ((quad_sum - zz) + log_det + (double)m.n_kept * kLog2Pi);
    if (!pd) ll = NAN;
    if (is_sample) a.sample_ll[(int64_t)q * a.S + sample] = ll;
    else if (slot == a.S) a.ll_no_dla[q] = ll;
  }
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
  int64_t S;
  double *summary;           // [nq][12]
};

__global__ __launch_bounds__(256) void k_evidence(EvidenceArgs a) {
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ double sh[4];
  const QuasarMeta m = a.meta[q];
  double *out = a.summary + (int64_t)q * 12;
  if (m.status != 0) {
    if (tid < 12) out[tid] = NAN;
    if (tid == 2) out[2] = a.log_prior_no_dla[q];
    if (tid == 3) out[3] = a.log_prior_dla[q];
    return;
  }
  const double *ll = a.sample_ll + (int64_t)q * a.S;
  double mx = -INFINITY;
  bool has_nan = false;
  for (int64_t i = tid; i < a.S; i += 256) {
    const double v = ll[i];
    has_nan |= isnan(v);
    mx = fmax(mx, v);  // fmax skips NaN like MATLAB's max (:203)
  }
  mx = block_reduce_minmax(mx, false, sh);
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
  if (tid == 0) {
    double *v = ws + nb;
    double log_diag = 0.0, zz = 0.0;
    bool pd = true;
    for (int i = 0; i < k; ++i) {
      const int ri = i * (i + 1) / 2;
      for (int j = 0; j <= i; ++j) {
        const int rj = j * (j + 1) / 2;
        double sum = ws[ri + j];
        for (int mm = 0; mm < j; ++mm) sum = fma(-ws[ri + mm], ws[rj + mm], sum);
        if (i == j) {
          pd = pd && (sum > 0.0);
          const double lii = sqrt(sum);
          log_diag += log(lii);
          ws[ri + j] = lii;
        } else {
          ws[ri + j] = sum / ws[rj + j];
        }
      }
      double zi = v[i];
      for (int mm = 0; mm < i; ++mm) zi = fma(-ws[ri + mm], v[mm], zi);
      zi /= ws[ri + i];
      v[i] = zi;
      zz = fma(zi, zi, zz);
    }
    *log_p = pd ? -0.5 * ((qs - zz) + ld + 2 * log_diag + (double)n * kLog2Pi) : NAN;
    *status = pd ? 0 : 1;
  }
}

}  // namespace gpdla
