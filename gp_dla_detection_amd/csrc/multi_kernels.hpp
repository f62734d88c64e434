// multi_kernels.hpp -- device side of the multi-DLA driver
// (multi_dlas/process_qsos_multiple_dlas_meanflux.m, "multi :N" below).
//
// In the multi-DLA models sample i of model nd multiplies nd Voigt profiles: its own and those of
// the samples base_sample_inds(1:nd-1, i) (multi :342-351).  Every profile is a function of one
// sample index alone, so the 2 S distinct profiles of a quasar (S DLA + S sub-DLA column densities,
// multi :365-368) are computed ONCE (k_profiles) into HBM -- 288 GB holds them for hundreds of
// quasars at a time -- and the sweeps of the 1 + max_dlas models gather and multiply them
// (k_sweep_multi), feeding the same MFMA contraction and Cholesky epilogue as the single-DLA sweep.
//
//   k_profiles        voigt.c:278-299 for every (quasar, DLA|LLS, sample)
//   k_sweep_multi     multi :340-381   (the parfor body)
//   k_multi_evidence  multi :386-445   separation mask, nanmax/nanmean evidence, Occam terms, MAP
//   k_multi_resample  multi :467-472   weighted resampling (documented counter-based RNG; the
//                                      reference uses MATLAB's rng('default') + randsample)
//   k_multi_posteriors multi :482-495
#pragma once
#include <type_traits>

#include "sweep_kernels.hpp"

namespace gpdla {

// ------------------------------------------------------------------------------------------
// k_profiles: one LANE per sample, BOTH of its profiles (kind 0: the DLA column density, kind 1:
// the sub-DLA one, multi :365-368).  The two share z_DLA and therefore the whole line sum
// Sum_j c_j H(x_j) of voigt.c:282-290 -- only the column density that scales it inside the
// exponential differs (:291) -- so a padded pixel costs one wing / accurate-tier evaluation and two
// exponentials (round 2: one wave per (kind, 64 samples), the line sum taken twice).
// A wave holds 64 samples that are neighbours in z_DLA (perm order) and walks the padded pixels in
// lockstep -- the padded wavelength is a wave-uniform load, the accurate Voigt tier is taken by
// whole waves -- and keeps the seven raw values per kind that the instrument broadening needs
// (voigt.c:297-299) in registers: no cross-lane traffic and no idle lanes (the round-1 form, one
// wave per profile with lanes along the pixels, paid 12 ds_bpermute per value and idled 6 of 64
// lanes).  Outputs are transposed through LDS, 16 pixels at a time, so that every store
// instruction writes whole 128-byte pieces of profile rows.
// prof[((ql * 2 + kind) * S + i) * stride + p], p < stride (>= 4 * steps; entries >= n_u are 1).
// ------------------------------------------------------------------------------------------
struct ProfilesArgs {
  const QuasarMeta *meta;
  const double *lam_pad;
  const double *offset_samples, *nhi_samples, *lls_nhi_samples;
  const int32_t *perm;   // [S] sample indices in ascending offset (= z_DLA) order
  int64_t S;
  int32_t num_lines;
  int64_t q0;        // first quasar of this sub-batch
  int32_t nq_sub;
  int64_t stride;
  double *prof;
};

constexpr int kProfTile = 16;   // pixels per transposed store
constexpr int kProfWaves = 4;   // waves per block (4 x 2 x 64 x 17 doubles of LDS: two blocks per CU)

#ifdef PROF_EXP_RHOTABLE
// Cost probe (round 5, results WRONG by construction): the three-line wing sum as ONE piecewise
// polynomial of rho = lambda / (1 + z_DLA) -- the sum depends on lambda and z_DLA only through rho
// (voigt.c:278-290) -- binned geometrically around the nearest line centre, 16 bins per octave of
// |rho - lambda_j| from 2^-4 to 2^9 Angstrom on either side of each of the three lines, degree 7:
// 1248 rows of 8 coefficients (80 KB).  The table holds zeros: the probe measures what the look-up
// COSTS in k_profiles (index arithmetic, four 16-byte gathers per lane at the addresses a real table
// would be read at, the local coordinate, Horner) against the 42 instructions of wing_sum3.
constexpr int kRhoBinsPerSide = 13 * 16;
__device__ double g_rho_probe[6 * kRhoBinsPerSide * 8];
#endif

__global__ __launch_bounds__(kProfWaves * 64) void k_profiles(ProfilesArgs a) {
  __shared__ double s_exp[kExpTab];  // 2^(j/64), the table behind exp_table()
  __shared__ double s_out[kProfWaves][2][64][kProfTile + 1];

  for (int e = threadIdx.x; e < kExpTab; e += kProfWaves * 64) s_exp[e] = exp2((double)e * (1.0 / kExpTab));
  __syncthreads();  // (before any wave may leave)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t wpq = (a.S + 63) / 64;  // waves per quasar
  const int64_t w_global = (int64_t)blockIdx.x * kProfWaves + wave;
  const int64_t ql = w_global / wpq;
  if (ql >= a.nq_sub) return;
  const int64_t pos0 = (w_global - ql * wpq) * 64;
  const QuasarMeta m = a.meta[a.q0 + ql];
  if (m.status != 0) return;
  const int L = a.num_lines;
  const bool live = pos0 + lane < a.S;
  const int64_t i = a.perm[live ? pos0 + lane : a.S - 1];
  const double z_dla = m.min_z_dla + (m.max_z_dla - m.min_z_dla) * a.offset_samples[i];  // multi :309
  const double c_light = g_lines.c, inv_s = g_lines.inv_sqrt2_sigma;
  double mult[3], ms[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    mult[j] = g_lines.c / (g_lines.wavelength_cm[j] * (1 + z_dla)) / 1e8;  // voigt.c:278-279
    ms[j] = mult[j] * inv_s;
  }
  const double cs = c_light * inv_s;
  const double inv_opz = 1.0 / (1 + z_dla);  // (run-time line counts: wing_sum_runtime)
#ifdef PROF_EXP_RHOTABLE
  const double inv1pz = inv_opz;
#endif
  // (pre-scaled exp, sweep_kernels.hpp)
  const double nscale_a = -a.nhi_samples[i] * g_lines.inv_sqrt2pi_sigma * kInvSqrtPi * kExpScale;
  const double nscale_b = -a.lls_nhi_samples[i] * g_lines.inv_sqrt2pi_sigma * kInvSqrtPi * kExpScale;
  const double *lam = a.lam_pad + m.lam_off;
  const int n_pad = m.n_u + 6;
  double *rows_a = a.prof + ((ql * 2 + 0) * a.S) * a.stride;  // + i * stride + p
  double *rows_b = a.prof + ((ql * 2 + 1) * a.S) * a.stride;
  const double t0 = g_lines.taps[0], t1 = g_lines.taps[1], t2 = g_lines.taps[2], t3 = g_lines.taps[3],
               t4 = g_lines.taps[4], t5 = g_lines.taps[5], t6 = g_lines.taps[6];

  auto line_sum = [&](int P) -> double {  // voigt.c:282-290 for this lane's z_DLA at padded pixel P
    const double lamP = lam[min(P, n_pad - 1)];  // wave-uniform address
    double total = 0.0;
    bool near = false;
    if (L == 3) {
#ifdef PROF_EXP_RHOTABLE
      const double rho = lamP * inv1pz;  // rest wavelength at the absorber, Angstrom
      const bool ab = rho > 1120.6962, bc = rho > 999.12955;  // midpoints of Ly-alpha / beta / gamma
      const double lc = ab ? 1215.6701 : bc ? 1025.7223 : 972.5368;
      const int lsel = ab ? 0 : bc ? 1 : 2;
      const double dd = rho - lc, ad = fabs(dd);
      near = ad < 1.3e-3 * lc;  // |x| < 30: 30 sqrt2 sigma / c of the line's wavelength
      const uint32_t hi = hi_word(ad);
      int bin = (int)(hi >> 16) - (1019 << 4);  // exponent and the top four mantissa bits: 16 bins per octave from 2^-4
      bin = min(max(bin, 0), kRhoBinsPerSide - 1);
      const int row = (lsel * 2 + (dd < 0.0 ? 1 : 0)) * kRhoBinsPerSide + bin;
      const double2 *cp = reinterpret_cast<const double2 *>(g_rho_probe + row * 8);
      const double2 c01 = cp[0], c23 = cp[1], c45 = cp[2], c67 = cp[3];
      // the position inside the bin, [0, 1): the mantissa bits below the bin's
      const double tf = __hiloint2double((int)((hi & 0xFFFFu) | 0x3FF00000u), __double2loint(ad)) - 1.0;
      double t = fma(c67.y, tf, c67.x);
      t = fma(t, tf, c45.y);
      t = fma(t, tf, c45.x);
      t = fma(t, tf, c23.y);
      t = fma(t, tf, c23.x);
      t = fma(t, tf, c01.y);
      total = fma(t, tf, c01.x);
#else
      total = wing_sum3(lamP, ms[0], ms[1], ms[2], cs, &near);
#endif
    } else {  // a run-time line count: the wing tier of k_sweep_slim<0> (four lines at a time from rho = lambda / (1 + z))
      total = wing_sum_runtime(lamP * inv_opz, cs, L, &near);
    }
    if (__any(near)) {  // accurate tier: per-line piecewise polynomials (near_tables.hpp), as in k_sweep
      if (L == 3) {
        total = 0.0;
        for (int j = 0; j < 3; ++j) {
          const double ax = fabs((lamP * mult[j] - c_light) * inv_s);
          total += ax < 30.0 ? 1.7724538509055159 * g_lines.leading[j] *
                                   near_poly(g_lines.near_poly + j * kNearLineDoubles, ax)
                             : g_lines.cwing[j] * wing_core(ax * ax, g_lines.y2[j]);
        }
      } else {
        total = total_near_at(lamP, 1 + z_dla, L);
      }
    }
    return total;
  };
  // the two raw profiles at padded pixel P (voigt.c:291)
#define GPDLA_PROF_RAW(P, ra, rb)                          \
  {                                                        \
    const double total_ = line_sum(P);                     \
    ra = exp_table_scaled(nscale_a * total_, s_exp);       \
    rb = exp_table_scaled(nscale_b * total_, s_exp);       \
  }

  // windows of raw values P .. P+6 for output pixel P (the profile of pixel p uses padded p .. p+6)
  double a0, a1, a2, a3, a4, a5, a6, b0, b1, b2, b3, b4, b5, b6;
  GPDLA_PROF_RAW(0, a0, b0);
  GPDLA_PROF_RAW(1, a1, b1);
  GPDLA_PROF_RAW(2, a2, b2);
  GPDLA_PROF_RAW(3, a3, b3);
  GPDLA_PROF_RAW(4, a4, b4);
  GPDLA_PROF_RAW(5, a5, b5);
  const int64_t nrows = min((int64_t)64, a.S - pos0);
  // the 16 profile rows this lane stores into (store instruction e writes rows 4e .. 4e+3): their
  // offsets once, not a perm look-up and a 64-bit multiply per store
  int64_t roff[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int rl = 4 * e + (lane >> 4);
    roff[e] = rl < nrows ? (int64_t)a.perm[pos0 + rl] * a.stride + (lane & 15) : -1;
  }
  for (int p0 = 0; p0 < m.n_u; p0 += kProfTile) {
#pragma unroll
    for (int tt = 0; tt < kProfTile; ++tt) {
      GPDLA_PROF_RAW(p0 + tt + 6, a6, b6);
      double acc = a0 * t0;  // voigt.c:297-299, taps in ascending order
      acc = fma(a1, t1, acc);
      acc = fma(a2, t2, acc);
      acc = fma(a3, t3, acc);
      acc = fma(a4, t4, acc);
      acc = fma(a5, t5, acc);
      acc = fma(a6, t6, acc);
      s_out[wave][0][lane][tt] = acc;
      acc = b0 * t0;
      acc = fma(b1, t1, acc);
      acc = fma(b2, t2, acc);
      acc = fma(b3, t3, acc);
      acc = fma(b4, t4, acc);
      acc = fma(b5, t5, acc);
      acc = fma(b6, t6, acc);
      s_out[wave][1][lane][tt] = acc;
      a0 = a1; a1 = a2; a2 = a3; a3 = a4; a4 = a5; a5 = a6;
      b0 = b1; b1 = b2; b2 = b3; b3 = b4; b4 = b5; b5 = b6;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // transposed store: instruction e writes pixels p0 .. p0+15 of rows 4e .. 4e+3
    const int tt = lane & 15;
    const bool in_row = p0 + tt < m.n_u;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rl = 4 * e + (lane >> 4);
      if (roff[e] >= 0 && in_row) {
        // (non-temporal: the table is 15 GB per sub-batch and is next read by another kernel)
        __builtin_nontemporal_store(s_out[wave][0][rl][tt], &rows_a[roff[e] + p0]);
        __builtin_nontemporal_store(s_out[wave][1][rl][tt], &rows_b[roff[e] + p0]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
#undef GPDLA_PROF_RAW
  // padding behind the pixels: 1 (no absorption)
  for (int64_t rl = 0; rl < nrows; ++rl) {
    const int64_t ir = a.perm[pos0 + rl];
    for (int64_t p = m.n_u + lane; p < a.stride; p += 64) {
      rows_a[ir * a.stride + p] = 1.0;
      rows_b[ir * a.stride + p] = 1.0;
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_sweep_multi: the parfor body of multi :340-381 for one model.
// mode 0: LLS profiles (multi :365-380); mode nd >= 1: product of nd DLA profiles.
// Slot S of the mode-1 pass is the null model (multi :296-298).
// ------------------------------------------------------------------------------------------
struct SweepMultiArgs {
  const QuasarMeta *meta;
  const double *records;
  const double *prof;
  const PixelRow *pix;         // per-pixel pool (k_sweep_multi_split reads rows directly)
  const uint32_t *base_inds;   // [nq][max_dlas-1][S], 1-based (multi :116, :476)
  const int32_t *alive;        // [nq] 0 once an evidence came out NaN (multi :460-464)
  int64_t S, q0, stride;
  int32_t nq_sub, blocks_per_quasar, k, mode, max_dlas;
  double log_S;
  double *sample_ll_dla;       // [nq][max_dlas][S]
  double *sample_ll_lls;       // [nq][S]
  double *ll_no_dla;           // [nq]
};

// ND: number of profiles a sample multiplies (1 for the LLS model and the one-DLA model), known at
// compile time so that the ND gathers of a K-step are issued back to back ahead of the MFMA burst
// (with a run-time count each gather sat behind a branch and was waited for before the next).
template <int NTW, int TS, int kChunkSteps, int TW, int ND>
__global__ __launch_bounds__(512) void k_sweep_multi(SweepMultiArgs a) {
  extern __shared__ double smem[];
  constexpr int GROUPS = kSweepWaves / TS;
  constexpr int NT = NTW * TS;
  constexpr int RD = NT * 64 + record_extras(NT);
  constexpr bool kCompact = tiles_compact(NT);
  const int64_t xj = blockIdx.x >> 3;
  const int64_t ql = 8 * (xj / a.blocks_per_quasar) + (blockIdx.x & 7);
  const int bq = (int)(xj % a.blocks_per_quasar);
  if (ql >= a.nq_sub) return;
  const int64_t q = a.q0 + ql;
  const QuasarMeta m = a.meta[q];
  if (m.status != 0 || a.alive[q] == 0) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int group = wave / TS, role = wave % TS;
  const int s = lane & 15, jj = lane >> 4;
  double *stage = smem;  // [2][kChunkSteps][RD]

  const int64_t slot0 = (int64_t)bq * (GROUPS * kSamplesPerWave) + group * kSamplesPerWave;
  const int64_t slot = slot0 + s;
  const bool is_sample = slot < a.S;
  const bool is_null = !is_sample;
  const int64_t i = is_sample ? slot : 0;
  constexpr int nd = ND;  // a.mode == 0 ? 1 : a.mode
  // rows of the profile table this lane multiplies (multi :342-351)
  // An index outside [1, S] (0 = "never drawn": the rows the reference leaves zero after its early
  // exit, multi :116, :460-464) is never followed: the sample's likelihood becomes NaN.
  const double *rows[4];
  rows[0] = a.prof + ((ql * 2 + (a.mode == 0 ? 1 : 0)) * a.S + i) * a.stride;
  int chain_ok = 1;
#pragma unroll
  for (int j = 1; j < 4; ++j) {
    int64_t kk = i;
    if (j < nd) {
      kk = (int64_t)a.base_inds[((int64_t)q * (a.max_dlas - 1) + (j - 1)) * a.S + i] - 1;
      if (kk < 0 || kk >= a.S) {
        chain_ok = 0;
        kk = i;
      }
    }
    rows[j] = a.prof + ((ql * 2) * a.S + kk) * a.stride;
  }
  const double *rec_base = a.records + m.rec_off * (int64_t)RD;
  const int nrec = m.steps + 1;
  const int nchunks = (nrec + kChunkSteps - 1) / kChunkSteps;
  static_assert((kChunkSteps * RD) % 128 == 0, "a chunk is a whole number of KiB");
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t stage_lds = __builtin_amdgcn_readfirstlane(lds_address(stage));
  auto issue_chunk = [&](int c) {  // (see glds_chunk in sweep_kernels.hpp)
    glds_chunk<kChunkSteps * RD / 128, kSweepWaves>(rec_base + (size_t)c * kChunkSteps * RD,
                                                    stage_lds + (uint32_t)(c & 1) * (uint32_t)(kChunkSteps * RD * 8), wave_s, lane);
  };
  issue_chunk(0);
  d4 acc[NTW];
#pragma unroll
  for (int c = 0; c < NTW; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
  double quad_sum = 0.0, dprod = 1.0;
  double xw[kXW] = {0.0, 0.0}, xu[kXU] = {0.0, 0.0, 0.0, 0.0};  // compact class: columns kept off the MFMA
  int dexp = 0;
  const int tile0 = role * NTW;

  // absorption of pixel p for this lane's sample: product of the gathered profiles.  The gather
  // reads HBM (the profile table is 243 MB per quasar) and a K-step of this kernel is short -- 14
  // MFMAs and ~35 vector instructions, ~1 us per wave -- so the values are requested kAhead K-steps
  // before they are used: with a single step of lookahead every K-step waited for its own gather
  // (PMC, round 2: the matrix pipe busy 65 % of the time with 16 % of it idle on both waves).
  constexpr int kAhead = 3;
  double raw0_product;
  auto gather = [&](int p, double (&r)[ND]) {
#pragma unroll
    for (int j = 0; j < ND; ++j) r[j] = rows[j][p];
  };
  auto product = [&](const double (&r)[ND]) -> double {
    double v = r[0];
#pragma unroll
    for (int j = 1; j < ND; ++j) v *= r[j];
    return is_null ? 1.0 : v;
  };
  const int p_last = 4 * m.steps + jj;  // rows are padded to 4*(steps+1) entries
  double raw[kAhead][ND];               // raw[i]: gathered for K-step (t + 1 + i) mod kAhead at use
  {
    double r0[ND];
    gather(jj, r0);
#pragma unroll
    for (int i = 0; i < kAhead; ++i) gather(min(4 * (1 + i) + jj, p_last), raw[i]);
    raw0_product = product(r0);
  }
  double a_next = raw0_product;  // step 0
  glds_wait();
  __syncthreads();
  if (nchunks > 1) issue_chunk(1);

  double w_cur, u_cur, bop[NTW];
  double xpr[kXW] = {0.0, 0.0}, upr[kXU] = {0.0, 0.0, 0.0, 0.0};  // compact class: this step's VALU columns
#define GPDLA_MPREP(rec, absorb_in)                                                       \
  {                                                                                       \
    const double *extra_ = (rec) + NT * 64;                                               \
    const double *mine_ = extra_ + extras_row(NT, 1) * jj;                                \
    if (kCompact) {                                                                       \
      _Pragma("unroll") for (int x = 0; x < kXW; ++x) xpr[x] = mine_[kExtrasXW + x];      \
      _Pragma("unroll") for (int x = 0; x < kXU; ++x) upr[x] = mine_[kExtrasXU + x];      \
    }                                                                                     \
    const double absorb_ = (absorb_in);                                                   \
    const double py_ = mine_[0], pmu_ = mine_[1], pom_ = mine_[2], pnu_ = mine_[3];       \
    const double r_ = fma(-absorb_, pmu_, py_);      /* multi :355 */                     \
    const double a2_ = absorb_ * absorb_;                                                 \
    const double d_ = fma(pom_, a2_, pnu_);          /* multi :357, :361 */               \
    const double inv_d_ = fast_rcp(d_);                                                   \
    w_cur = a2_ * inv_d_;                                                                 \
    u_cur = absorb_ * r_ * inv_d_;                                                        \
    quad_sum = fma(r_ * r_, inv_d_, quad_sum);                                            \
    dprod *= d_;                                                                          \
    dexp += __builtin_amdgcn_frexp_exp(dprod);                                            \
    dprod = __builtin_amdgcn_frexp_mant(dprod);                                           \
    const double *bt_ = (rec) + (size_t)tile0 * 64 + lane;                                \
    _Pragma("unroll") for (int cc = 0; cc < NTW; ++cc) bop[cc] = bt_[(size_t)cc * 64];    \
  }
  GPDLA_MPREP(stage, a_next)
  // one K-step; I = t mod kAhead selects the gather slot at compile time (the loop is unrolled
  // kAhead times so that the slots rotate without register moves)
  auto kstep = [&](auto I, int t) {
    constexpr int i = decltype(I)::value;
    const int rn = t + 1;
    const int cn = rn / kChunkSteps;
    if (rn % kChunkSteps == 0) {
      glds_wait();  // chunk cn (issued one chunk ago) has landed
      __syncthreads();
      if (cn + 1 < nchunks) issue_chunk(cn + 1);
    }
    const double *rec = stage + ((size_t)(cn & 1) * kChunkSteps + (rn % kChunkSteps)) * RD;
    a_next = product(raw[i]);                                    // K-step rn, requested kAhead steps ago
    gather(min(4 * (rn + kAhead) + jj, p_last), raw[i]);         // K-step rn + kAhead
    const double wa = w_cur, ua = u_cur;
#pragma unroll
    for (int cc = 0; cc < NTW; ++cc)
    {
      constexpr int kTail = TS == 1 ? 0 : NT - TW;
      const double a_tail = role == TS - 1 ? ua : wa;
      const double aop = TS == 1 ? (cc < TW ? wa : ua) : (cc < NTW - kTail ? wa : a_tail);
      acc[cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop[cc], acc[cc], 0, 0, 0);
    }
    if (kCompact) {
#pragma unroll
      for (int x = 0; x < kXW; ++x) xw[x] = fma(wa, xpr[x], xw[x]);
#pragma unroll
      for (int x = 0; x < kXU; ++x) xu[x] = fma(ua, upr[x], xu[x]);
    }
    GPDLA_MPREP(rec, a_next)
  };
  static_assert(kAhead == 3, "the loop below is unrolled three times");
  for (int t = 0; t < m.steps; t += kAhead) {
    kstep(std::integral_constant<int, 0>{}, t);
    if (t + 1 < m.steps) kstep(std::integral_constant<int, 1>{}, t + 1);
    if (t + 2 < m.steps) kstep(std::integral_constant<int, 2>{}, t + 2);
  }
#undef GPDLA_MPREP
  __syncthreads();
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

  using ES = EpilogueShape<TW, TS>;
  double *Eg = stage + (size_t)group * ES::SPP * ES::stride(logical_tiles(NT));
#pragma unroll
  for (int p = 0; p < ES::PASSES; ++p) {
    int sigma;
    bool writer;
    const double ll = factor_pass<double, NTW, TS, TW>(acc, xw, xu, p, Eg, lane, role, tile0, a.k, quad_sum,
                                                       logd_sum, m.n_kept, &sigma, &writer);
    const int64_t slot_s = slot0 + sigma;
    const bool ok_s = __shfl(chain_ok, sigma) != 0;  // lane sigma (jj = 0) holds sample sigma's flag
    if (writer) {
      if (slot_s < a.S) {
        if (a.mode == 0) a.sample_ll_lls[q * a.S + slot_s] = ll + m.ll_bias - a.log_S;     // multi :376-378
        else a.sample_ll_dla[(q * a.max_dlas + (a.mode - 1)) * a.S + slot_s] = ok_s ? ll + m.ll_bias - a.log_S : NAN;  // :359-361
      } else if (slot_s == a.S && a.mode == 1) {
        a.ll_no_dla[q] = ll + m.ll_bias;                                                    // multi :296-298
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_sweep_multi_split: k_sweep_multi for 20 < k <= 40 with the k_sweep_split pipeline
// (sweep_split_kernel.hpp).  56 accumulator tiles do not fit one wave, so four waves ("roles") share
// a group of 16 samples and take 14 tiles each.  In k_sweep_multi<14, 4, ...> every one of the four
// gathered the group's profile values and formed the weights itself -- the HBM gathers of the
// profile table four times over -- and the block met at a barrier every K-step.  Here role r
// gathers and weighs only the K-steps t = r (mod 4), into a double-buffered (w, u) table in LDS,
// and every role reads (w, u) of each K-step for its 14 MFMAs; one loop iteration is 4 K-steps (two
// 2-step record chunks).  The profile values of step 4 (it + 2) + r are requested at the top of
// iteration it and multiplied at the end of iteration it + 1: two iterations (8 K-steps) of latency
// hiding for the HBM gather.
// ------------------------------------------------------------------------------------------
__host__ __device__ constexpr size_t sweep_multi_split_lds_doubles() {
  constexpr int RD = 56 * 64 + record_extras(56);
  return 2 * 2 * (size_t)RD + 2 * 2 * 2 * 4 * 64 + 2 * 4 * 2 * 16;  // stage | (w, u) | per-role partial sums
}

template <int ND>
__global__ __launch_bounds__(512) void k_sweep_multi_split(SweepMultiArgs a) {
  extern __shared__ double smem[];
  constexpr int WAVES = 8, TS = 4, NTW = 14, NT = 56, TW = 52, CH = 2;
  constexpr int RD = NT * 64 + record_extras(NT);
  const int64_t xj = blockIdx.x >> 3;
  const int64_t ql = 8 * (xj / a.blocks_per_quasar) + (blockIdx.x & 7);
  const int bq = (int)(xj % a.blocks_per_quasar);
  if (ql >= a.nq_sub) return;
  const int64_t q = a.q0 + ql;
  const QuasarMeta m = a.meta[q];
  if (m.status != 0 || a.alive[q] == 0) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int group = wave >> 2, role = wave & 3;
  const int s = lane & 15, jj = lane >> 4;

  double *stage = smem;                                    // [2][CH][RD]
  double *wu = stage + (size_t)2 * CH * RD;                // [2 parity][2 groups][2][4 steps][64]
  double *red = wu + 2 * 2 * 2 * 4 * 64;                   // [2 groups][4 roles][2][16]

  const int64_t slot0 = (int64_t)bq * (2 * kSamplesPerWave) + group * kSamplesPerWave;
  const int64_t slot = slot0 + s;
  const bool is_sample = slot < a.S;
  const bool is_null = !is_sample;
  const int64_t i = is_sample ? slot : 0;
  // rows of the profile table this lane multiplies (multi :342-351); see k_sweep_multi
  const double *rows[4];
  rows[0] = a.prof + ((ql * 2 + (a.mode == 0 ? 1 : 0)) * a.S + i) * a.stride;
  int chain_ok = 1;
#pragma unroll
  for (int j = 1; j < 4; ++j) {
    int64_t kk = i;
    if (j < ND) {
      kk = (int64_t)a.base_inds[((int64_t)q * (a.max_dlas - 1) + (j - 1)) * a.S + i] - 1;
      if (kk < 0 || kk >= a.S) {
        chain_ok = 0;
        kk = i;
      }
    }
    rows[j] = a.prof + ((ql * 2) * a.S + kk) * a.stride;
  }
  const PixelRow *pix = a.pix + m.pix_off;
  const double *rec_base = a.records + m.rec_off * (int64_t)RD;
  const int nchunks = (m.steps + CH - 1) / CH;
  const int niter = (m.steps + 3) / 4;
  static_assert((CH * RD) % 128 == 0, "a chunk is a whole number of KiB");
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t stage_lds = __builtin_amdgcn_readfirstlane(lds_address(stage));
  auto issue_chunk = [&](int c) {
    glds_chunk<CH * RD / 128, WAVES>(rec_base + (size_t)c * CH * RD, stage_lds + (uint32_t)(c & 1) * (uint32_t)(CH * RD * 8),
                                     wave_s, lane);
  };
  issue_chunk(0);

  const int p_last = 4 * m.steps + jj;  // rows are padded to 4 (steps + 1) entries
  auto gather = [&](int t, double (&r)[ND]) {
    const int p = min(4 * t + jj, p_last);
#pragma unroll
    for (int j = 0; j < ND; ++j) r[j] = rows[j][p];
  };
  auto pix_of = [&](int t) -> PixelRow { return pix[4 * min(t, m.steps) + jj]; };  // row `steps` is neutral
  double quad_sum = 0.0, dprod = 1.0;
  int dexp = 0;
  auto weights_of = [&](int t, const PixelRow &px, const double (&r)[ND], double *w_out, double *u_out) {
    double absorb = r[0];
#pragma unroll
    for (int j = 1; j < ND; ++j) absorb *= r[j];
    if (is_null) absorb = 1.0;
    const double rr = fma(-absorb, px.mu, px.y);   // multi :355
    const double a2 = absorb * absorb;
    const double d = fma(px.omega2, a2, px.nu);    // multi :357, :361
    const double inv_d = fast_rcp(d);
    *w_out = a2 * inv_d;
    *u_out = absorb * rr * inv_d;
    if (t < m.steps) {  // (steps beyond the last are never consumed; keep them out of the sums)
      quad_sum = fma(rr * rr, inv_d, quad_sum);
      dprod *= d;
      dexp += __builtin_amdgcn_frexp_exp(dprod);
      dprod = __builtin_amdgcn_frexp_mant(dprod);
    }
  };
  auto wu_slot = [&](int par, int which, int step) -> double * {
    return wu + ((((size_t)par * 2 + group) * 2 + which) * 4 + step) * 64 + lane;
  };

  // prime: (w, u) of K-steps 0..3 (role r: step r); profile values of step 4 + r in flight
  double g_cur[ND], g_next[ND];
  {
    double g0[ND], w0, u0;
    gather(role, g0);
    gather(4 + role, g_cur);
    weights_of(role, pix_of(role), g0, &w0, &u0);
    *wu_slot(0, 0, role) = w0;
    *wu_slot(0, 1, role) = u0;
  }
  d4 acc[NTW];
#pragma unroll
  for (int c = 0; c < NTW; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
  const int tile0 = role * NTW;
  constexpr int kTail = NT - TW;  // the last role's last 4 tiles take u

  glds_wait();  // chunk 0 landed
  __syncthreads();

  for (int it = 0; it < niter; ++it) {
    const int par = it & 1;
    const int t_w = 4 * (it + 1) + role;
    PixelRow px_w;
#pragma unroll
    for (int hc = 0; hc < 2; ++hc) {
      const int c = 2 * it + hc;
      if (c < nchunks) {  // block-uniform
        __builtin_amdgcn_s_waitcnt(0x0F70);  // (see k_sweep: free here, keeps compiler waits out of the K-steps)
        if (c + 1 < nchunks) issue_chunk(c + 1);
        if (hc == 0) {
          px_w = pix_of(t_w);
          gather(t_w + 4, g_next);  // for the W stage of the NEXT iteration
        }
        const double *buf = stage + (size_t)hc * CH * RD;  // chunk c lives in buffer c & 1 = hc
#pragma unroll
        for (int tt = 0; tt < CH; ++tt) {
          const int rn = c * CH + tt;
          if (rn < m.steps) {
            const int st = 2 * hc + tt;  // step within the iteration
            const double w = *wu_slot(par, 0, st), u = *wu_slot(par, 1, st);
            const double *bt = buf + (size_t)tt * RD + (size_t)tile0 * 64 + lane;
            double bop[NTW];
#pragma unroll
            for (int cc = 0; cc < NTW; ++cc) bop[cc] = bt[(size_t)cc * 64];
            const double a_tail = role == TS - 1 ? u : w;
#pragma unroll
            for (int cc = 0; cc < NTW; ++cc)
              acc[cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(cc < NTW - kTail ? w : a_tail, bop[cc], acc[cc], 0, 0, 0);
          }
        }
        if (hc == 1) {  // stage W of this role for the next iteration
          double w1, u1;
          weights_of(t_w, px_w, g_cur, &w1, &u1);
          *wu_slot(par ^ 1, 0, role) = w1;
          *wu_slot(par ^ 1, 1, role) = u1;
#pragma unroll
          for (int j = 0; j < ND; ++j) g_cur[j] = g_next[j];
        }
        glds_wait();
        __syncthreads();
      } else if (hc == 1) {
        __syncthreads();  // odd number of chunks: keep the barrier structure uniform
      }
    }
  }

  double logd_sum = log(dprod) + (double)dexp * 0.6931471805599453;
  quad_sum += __shfl_xor(quad_sum, 16);
  quad_sum += __shfl_xor(quad_sum, 32);
  logd_sum += __shfl_xor(logd_sum, 16);
  logd_sum += __shfl_xor(logd_sum, 32);
  if (jj == 0) {
    red[((group * 4 + role) * 2 + 0) * 16 + s] = quad_sum;
    red[((group * 4 + role) * 2 + 1) * 16 + s] = logd_sum;
  }
  __syncthreads();
  quad_sum = 0.0;
  logd_sum = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    quad_sum += red[((group * 4 + r) * 2 + 0) * 16 + s];
    logd_sum += red[((group * 4 + r) * 2 + 1) * 16 + s];
  }
  __syncthreads();

  // epilogue: all four roles factor, 32 lanes per sample (as k_sweep_split)
  using ES = EpilogueShape<TW, TS>;
  constexpr int ncols = ES::stride(logical_tiles(NT));
  constexpr int voff = TW * 16;
  double *Eg = stage + (size_t)group * ES::SPP * ncols;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    __syncthreads();
    double *e = Eg + (size_t)(jj * 2) * ncols;
#pragma unroll
    for (int cc = 0; cc < NTW; ++cc) {
      const int tile = tile0 + cc;
      const int col0 = tile < TW ? tile * 16 : voff + (tile - TW) * 16;
      e[col0 + s] = acc[cc][2 * p];
      e[ncols + col0 + s] = acc[cc][2 * p + 1];
    }
    __syncthreads();
    const int rho = 2 * role + (lane >> 5);
    const int sigma = (rho >> 1) + 4 * (2 * p + (rho & 1));  // Mat<double>::sample_of(jj, reg)
    const double q_s = __shfl(quad_sum, sigma), ld_s = __shfl(logd_sum, sigma);
    const bool ok_s = __shfl(chain_ok, sigma) != 0;
    const double ll = factor_lds<2, 32>(Eg + (size_t)rho * ncols, lane & 31, a.k, voff, q_s, ld_s, m.n_kept);
    const int64_t slot_s = slot0 + sigma;
    if ((lane & 31) == 0) {
      if (slot_s < a.S) {
        if (a.mode == 0) a.sample_ll_lls[q * a.S + slot_s] = ll + m.ll_bias - a.log_S;     // multi :376-378
        else a.sample_ll_dla[(q * a.max_dlas + (a.mode - 1)) * a.S + slot_s] = ok_s ? ll + m.ll_bias - a.log_S : NAN;  // :359-361
      } else if (slot_s == a.S && a.mode == 1) {
        a.ll_no_dla[q] = ll + m.ll_bias;                                                    // multi :296-298
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Block-wide helpers (256 threads).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double *sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// ------------------------------------------------------------------------------------------
// k_multi_evidence: one block per quasar, for model `nd` (and the LLS model when nd == 1).
// ------------------------------------------------------------------------------------------
struct MultiEvidenceArgs {
  const QuasarMeta *meta;
  const double *offset_samples, *log_nhi_samples;
  const uint32_t *base_inds;
  int32_t *alive;
  int64_t S;
  int32_t nd, max_dlas;
  double min_z_separation, log_S;
  double *sample_ll_dla, *sample_ll_lls;
  double *ll_dla;     // [nq][max_dlas]
  double *ll_lls;     // [nq]
  double *map_z, *map_lognhi, *map_ind;   // [nq][max_dlas][max_dlas]
};

__device__ inline void nan_evidence(const double *ll, int64_t S, double *sh, double *out_max,
                                    double *out_lme, int64_t *out_arg) {
  // nanmax with the FIRST index attaining it, then max + log(nanmean(exp(ll - max)))  (multi :400-408)
  const int tid = threadIdx.x;
  double mx = -INFINITY;
  int64_t arg = S;
  for (int64_t i = tid; i < S; i += 256) {
    const double v = ll[i];
    if (!isnan(v) && (v > mx || (v == mx && i < arg))) {
      mx = v;
      arg = i;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double om = __shfl_xor(mx, o);
    const long long oa = __shfl_xor((long long)arg, o);
    if (om > mx || (om == mx && oa < arg)) {
      mx = om;
      arg = oa;
    }
  }
  __shared__ double s_m[4];
  __shared__ long long s_a[4];
  const int wave = tid >> 6, lane = tid & 63;
  __syncthreads();
  if (lane == 0) {
    s_m[wave] = mx;
    s_a[wave] = arg;
  }
  __syncthreads();
  mx = s_m[0];
  arg = s_a[0];
  for (int w = 1; w < 4; ++w)
    if (s_m[w] > mx || (s_m[w] == mx && s_a[w] < arg)) {
      mx = s_m[w];
      arg = s_a[w];
    }
  const bool any = arg < S;
  double sum = 0.0, cnt = 0.0;
  for (int64_t i = tid; i < S; i += 256) {
    const double v = ll[i];
    if (!isnan(v)) {
      sum += exp(v - mx);
      cnt += 1.0;
    }
  }
  sum = block_sum(sum, sh);
  cnt = block_sum(cnt, sh);
  *out_max = any ? mx : NAN;
  *out_lme = any ? mx + log(sum / cnt) : NAN;
  *out_arg = any ? arg : 0;  // MATLAB's nanmax of an all-NaN column returns index 1
}

__global__ __launch_bounds__(256) void k_multi_evidence(MultiEvidenceArgs a) {
  const int q = blockIdx.x, tid = threadIdx.x;
  __shared__ double sh[4];
  const QuasarMeta m = a.meta[q];
  if (m.status != 0 || a.alive[q] == 0) return;
  const int nd = a.nd, md = a.max_dlas;
  double *col = a.sample_ll_dla + ((int64_t)q * md + (nd - 1)) * a.S;
  const uint32_t *base = a.base_inds + (int64_t)q * (md - 1) * a.S;
  const double zr = m.max_z_dla - m.min_z_dla;
  if (nd > 1) {  // multi :386-392: any two absorbers of the sample closer than min_z_separation
    for (int64_t i = tid; i < a.S; i += 256) {
      double zs[4];
      zs[0] = m.min_z_dla + zr * a.offset_samples[i];
      bool close = false;
      for (int j = 1; j < nd; ++j) {
        const int64_t bj = (int64_t)base[(int64_t)(j - 1) * a.S + i] - 1;
        const bool ok = bj >= 0 && bj < a.S;  // undrawn / out-of-range index: the sample is NaN
        close |= !ok;
        zs[j] = m.min_z_dla + zr * a.offset_samples[ok ? bj : i];
      }
      for (int x = 0; x < nd; ++x)
        for (int y = x + 1; y < nd; ++y) close |= fabs(zs[x] - zs[y]) < a.min_z_separation;
      if (close) col[i] = NAN;
    }
    __syncthreads();
  }
  double mx, lme;
  int64_t arg;
  nan_evidence(col, a.S, sh, &mx, &lme, &arg);
  if (tid == 0) {
    const double ev = lme - a.log_S * (nd - 1);  // multi :407-409
    a.ll_dla[(int64_t)q * md + (nd - 1)] = ev;
    // MAP bookkeeping, multi :439-445 ([model][slot], 1-based indices)
    for (int j = 0; j < nd; ++j) {
      const int64_t idx = j == 0 ? arg : (int64_t)base[(int64_t)(j - 1) * a.S + arg] - 1;
      const int64_t at = ((int64_t)q * md + (nd - 1)) * md + j;
      if (idx < 0 || idx >= a.S) continue;  // undrawn index (all-NaN column): slot stays NaN
      a.map_ind[at] = (double)(idx + 1);
      a.map_z[at] = m.min_z_dla + zr * a.offset_samples[idx];
      a.map_lognhi[at] = a.log_nhi_samples[idx];
    }
    if (isnan(ev)) a.alive[q] = 0;  // multi :460-464
  }
  if (nd == 1) {  // multi :416-430
    double mx2, lme2;
    int64_t arg2;
    nan_evidence(a.sample_ll_lls + (int64_t)q * a.S, a.S, sh, &mx2, &lme2, &arg2);
    if (tid == 0) a.ll_lls[q] = lme2;
  }
}

// ------------------------------------------------------------------------------------------
// k_multi_resample: base_sample_inds(nd, :) ~ randsample(S, S, true, W), W = exp(ll - max) with
// NaN -> 0 (multi :467-472).  MATLAB's generator cannot be reproduced; this one is Philox4x32-10
// keyed by (seed, global quasar index) with counter (draw j, model nd), inverse-CDF sampling on the
// prefix sums of W taken in sample order.
// ------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t *out) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct MultiResampleArgs {
  const QuasarMeta *meta;
  const int32_t *alive;
  const double *sample_ll_dla;
  int64_t S, first_quasar_index;
  uint64_t seed;
  int32_t nd, max_dlas;
  uint32_t *base_inds;
};

__global__ __launch_bounds__(256) void k_multi_resample(MultiResampleArgs a) {
  extern __shared__ double cdf[];  // [S]
  __shared__ double sh[4];
  __shared__ double s_part[256];
  const int q = blockIdx.x, tid = threadIdx.x;
  const QuasarMeta m = a.meta[q];
  if (m.status != 0 || a.alive[q] == 0) return;
  const double *col = a.sample_ll_dla + ((int64_t)q * a.max_dlas + (a.nd - 1)) * a.S;
  double mx = -INFINITY;
  for (int64_t i = tid; i < a.S; i += 256) {
    const double v = col[i];
    if (!isnan(v)) mx = fmax(mx, v);
  }
  mx = block_reduce_minmax(mx, false, sh);
  // chunked inclusive scan: thread t owns [t*chunk, (t+1)*chunk)
  const int64_t chunk = (a.S + 255) / 256;
  const int64_t lo = (int64_t)tid * chunk, hi = lo + chunk < a.S ? lo + chunk : a.S;
  double run = 0.0;
  for (int64_t i = lo; i < hi; ++i) {
    const double v = col[i];
    run += isnan(v) ? 0.0 : exp(v - mx);
    cdf[i] = run;
  }
  s_part[tid] = run;
  __syncthreads();
  if (tid == 0) {
    double acc = 0.0;
    for (int t = 0; t < 256; ++t) {
      const double v = s_part[t];
      s_part[t] = acc;
      acc += v;
    }
  }
  __syncthreads();
  const double off = s_part[tid];
  for (int64_t i = lo; i < hi; ++i) cdf[i] += off;
  __syncthreads();
  const double total = cdf[a.S - 1];
  const uint64_t qid = (uint64_t)(a.first_quasar_index + q);
  uint32_t *out = a.base_inds + ((int64_t)q * (a.max_dlas - 1) + (a.nd - 1)) * a.S;
  for (int64_t j = tid; j < a.S; j += 256) {
    uint32_t r[4];
    philox4x32_10((uint32_t)j, (uint32_t)(j >> 32), (uint32_t)a.nd, 0u,
                  (uint32_t)(a.seed ^ qid), (uint32_t)((a.seed >> 32) ^ (qid >> 32) ^ 0x5851F42Du), r);
    const double u = ((double)(r[0] >> 5) * 67108864.0 + (double)(r[1] >> 6)) * (1.0 / 9007199254740992.0);
    const double target = u * total;
    int64_t lo2 = 0, hi2 = a.S - 1;  // first index with cdf > target
    while (lo2 < hi2) {
      const int64_t mid = (lo2 + hi2) >> 1;
      if (cdf[mid] > target) hi2 = mid; else lo2 = mid + 1;
    }
    out[j] = (uint32_t)(lo2 + 1);
  }
}

// ------------------------------------------------------------------------------------------
// k_multi_posteriors: multi :300-301, :411-413, :428-430, :482-495.  One thread per quasar.
// post: [nq][2 + max_dlas] = (no DLA, LLS, 1..max_dlas DLAs).
// ------------------------------------------------------------------------------------------
struct MultiPostArgs {
  const QuasarMeta *meta;
  int64_t nq;
  int32_t max_dlas;
  const double *lp_no, *lp_lls, *lp_dla;   // log priors: [nq], [nq], [nq][max_dlas]
  const double *ll_no, *ll_lls, *ll_dla;
  double *lpost_no, *lpost_lls, *lpost_dla, *post, *p_no, *p_lls, *p_dla;
  const double *map_z, *map_lognhi, *map_ind;   // [nq][max_dlas][max_dlas]
  double *summary;   // [nq][14 + 4 md + 3 md^2], layout: GPDLA_SUMMARY_COLS_MULTI in gpdla.h
};

__global__ void k_multi_posteriors(MultiPostArgs a) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= a.nq) return;
  const int md = a.max_dlas, nm = 2 + md;
  double lp[2 + 8];
  lp[0] = a.lp_no[q] + a.ll_no[q];
  lp[1] = a.lp_lls[q] + a.ll_lls[q];
  for (int j = 0; j < md; ++j) lp[2 + j] = a.lp_dla[q * md + j] + a.ll_dla[q * md + j];
  a.lpost_no[q] = lp[0];
  a.lpost_lls[q] = lp[1];
  for (int j = 0; j < md; ++j) a.lpost_dla[q * md + j] = lp[2 + j];
  double mx = -INFINITY;  // MATLAB max skips NaN (:482-483)
  bool any = false;
  for (int j = 0; j < nm; ++j)
    if (!isnan(lp[j])) {
      mx = fmax(mx, lp[j]);
      any = true;
    }
  if (!any) mx = NAN;
  double e[2 + 8], sum = 0.0;
  for (int j = 0; j < nm; ++j) {
    e[j] = exp(lp[j] - mx);  // :485-488
    sum += e[j];             // NaN entries propagate, as sum() does (:491)
  }
  for (int j = 0; j < nm; ++j) a.post[q * nm + j] = e[j] * (1.0 / sum);
  a.p_no[q] = a.post[q * nm];
  a.p_lls[q] = a.post[q * nm + 1];
  a.p_dla[q] = 1 - a.p_no[q] - a.p_lls[q];  // :493-495
  // the row a multi-GPU run gathers: every saved variable of multi :498-510 that is not per-sample
  const QuasarMeta m = a.meta[q];
  double *o = a.summary + q * (14 + 4 * md + 3 * md * md);
  *o++ = m.min_z_dla;
  *o++ = m.max_z_dla;
  *o++ = a.lp_no[q];
  *o++ = a.lp_lls[q];
  for (int j = 0; j < md; ++j) *o++ = a.lp_dla[q * md + j];
  *o++ = a.ll_no[q];
  *o++ = a.ll_lls[q];
  for (int j = 0; j < md; ++j) *o++ = a.ll_dla[q * md + j];
  *o++ = lp[0];
  *o++ = lp[1];
  for (int j = 0; j < md; ++j) *o++ = lp[2 + j];
  for (int j = 0; j < nm; ++j) *o++ = a.post[q * nm + j];
  *o++ = a.p_no[q];
  *o++ = a.p_lls[q];
  *o++ = a.p_dla[q];
  for (int j = 0; j < md * md; ++j) *o++ = a.map_z[q * md * md + j];
  for (int j = 0; j < md * md; ++j) *o++ = a.map_lognhi[q * md * md + j];
  for (int j = 0; j < md * md; ++j) *o++ = a.map_ind[q * md * md + j];
  *o++ = m.status == 1 ? 1.0 : NAN;  // all_exceptions, multi :139, :232
}

}  // namespace gpdla
