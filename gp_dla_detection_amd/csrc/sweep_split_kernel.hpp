// sweep_split_kernel.hpp -- the fp64 sweep for 20 < k <= 40 (process_qsos.m:185-199).
//
// 52 + 4 B tiles of accumulators do not fit one wave, so four waves ("roles") share a group of 16
// samples and split the tiles 14 each (k_sweep's TS = 4 form).  In k_sweep every one of the four
// then repeats the whole per-(sample, pixel) arithmetic -- raw Voigt profile, broadening, weights
// -- which made that case VALU-bound at 39 % of the fp64 MFMA peak.  Here the four roles SHARE it
// through LDS, as a two-stage pipeline that runs ahead of the contraction:
//
//   stage R  role r computes the raw profile (voigt.c:282-292) of padded pixels 4t'+jj for the raw
//            steps t' = r (mod 4) only, into a per-group ring of 64 slots per sample;
//   stage W  role r computes, for the K-steps t = r (mod 4) only, the broadened absorption
//            (voigt.c:297-299) from 7 ring taps and the weights w, u (process_qsos.m:192-198 folded
//            into log_mvnpdf_low_rank.m:11-15), into a double-buffered (w, u) table, and keeps the
//            partial sums of r^2/d and log d of its own steps;
//   stage C  every role reads (w, u) of each K-step -- two LDS reads -- and issues its 14 MFMAs.
//
// One loop iteration is 4 K-steps (two 2-step record chunks, a block barrier after each, as in
// k_sweep).  In iteration i a role runs C for steps 4i..4i+3, then W for step 4(i+1)+r and R for
// raw step 4(i+3)+r.  W(t) reads raw steps t..t+2, all produced at least one iteration (hence one
// barrier) earlier; C reads the (w, u) buffer the previous iteration filled.  Per wave and K-step
// the VALU work drops from ~96 instructions to ~25.
#pragma once
#include "sweep_kernels.hpp"

namespace gpdla {

constexpr int kSplitRing = 65;  // 64 ring slots per sample + 1 pad (bank spread)

// LDS doubles of k_sweep_split (loop phase; the epilogue reuses the stage region)
__host__ __device__ constexpr size_t sweep_split_lds_doubles(int num_lines_runtime) {
  constexpr int RD = 56 * 64 + record_extras(56);
  return kExpTab + 2 * 2 * (size_t)RD            // exp table, two chunks of two records
         + 2 * 16 * kSplitRing                   // raw ring per group
         + 2 * 2 * 2 * 4 * 64                    // (w, u): parity x group x {w, u} x step x lane
         + 2 * 4 * 2 * 16                        // per-role partial sums at the end
         + (size_t)2 * 16 * num_lines_runtime;   // per-sample line multipliers (run-time line count)
}

template <int LINES>
__global__ __launch_bounds__(512) void k_sweep_split(SweepArgs a) {
  extern __shared__ double smem[];
  constexpr int WAVES = 8, TS = 4, NTW = 14, NT = 56, TW = 52, CH = 2;
  constexpr int RD = NT * 64 + record_extras(NT);
  const int64_t xj = blockIdx.x >> 3;
  const int64_t pos = 8 * (xj / a.blocks_per_quasar) + (blockIdx.x & 7);
  const int bq = (int)(xj % a.blocks_per_quasar);
  if (pos >= a.nq) return;
  const int64_t q = a.order[pos];
  const QuasarMeta m = a.meta[q];
  if (m.status != 0) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int group = wave >> 2, role = wave & 3;
  const int s = lane & 15, jj = lane >> 4;
  const int L = LINES > 0 ? LINES : a.num_lines;

  double *exp_tab = smem;                                  // [64]
  double *stage = exp_tab + kExpTab;                       // [2][CH][RD]
  double *ring = stage + (size_t)2 * CH * RD;              // [2 groups][16][kSplitRing]
  double *wu = ring + 2 * 16 * kSplitRing;                 // [2 parity][2 groups][2][4 steps][64]
  double *red = wu + 2 * 2 * 2 * 4 * 64;                   // [2 groups][4 roles][2][16]
  double *mult_s = red + 2 * 4 * 2 * 16;                   // [2*16][L] (run-time L only)

  const int64_t slot0 = (int64_t)bq * (2 * kSamplesPerWave) + group * kSamplesPerWave;
  const int64_t slot = slot0 + s;
  const bool is_sample = slot < a.S;
  const bool is_null = !is_sample;  // slot == S is the null model; slots beyond it are idle copies
  const int32_t sample = is_sample ? a.perm[slot] : 0;
  const double z_dla = m.min_z_dla + (m.max_z_dla - m.min_z_dla) * a.offset_samples[sample];  // :162-164
  const double nhi = a.nhi_samples[sample];
  double *my_mult = mult_s + (size_t)(group * kSamplesPerWave + s) * L;
  double mult_r[LINES > 0 ? LINES : 1];
  if (LINES > 0) {
#pragma unroll
    for (int j = 0; j < LINES; ++j) mult_r[j] = g_lines.c / (g_lines.wavelength_cm[j] * (1 + z_dla)) / 1e8;  // voigt.c:278-279
  } else if (role == 0 && jj == 0) {
    for (int j = 0; j < L; ++j) my_mult[j] = g_lines.c / (g_lines.wavelength_cm[j] * (1 + z_dla)) / 1e8;
  }
  if (tid < kExpTab) exp_tab[tid] = exp2((double)tid * (1.0 / kExpTab));
  const double *lam = a.lam_pad + m.lam_off;
  const PixelRow *pix = a.pix + m.pix_off;
  const int n_pad = m.n_u + 6;
  const double nscale64 = -nhi * g_lines.inv_sqrt2pi_sigma * kInvSqrtPi * kExpScale;  // (pre-scaled exp, sweep_kernels.hpp)
  const double *rec_base = a.records + m.rec_off * (int64_t)RD;
  const int nchunks = (m.steps + CH - 1) / CH;
  const int niter = (m.steps + 3) / 4;
  const double c_light = g_lines.c, inv_s = g_lines.inv_sqrt2_sigma;
  double ms_r[LINES > 0 ? LINES : 1];
#pragma unroll
  for (int j = 0; j < (LINES > 0 ? LINES : 0); ++j) ms_r[j] = mult_r[j] * inv_s;
  const double cs = c_light * inv_s;
  const double tap0 = g_lines.taps[0], tap1 = g_lines.taps[1], tap2 = g_lines.taps[2], tap3 = g_lines.taps[3];

  static_assert((CH * RD) % 128 == 0, "a chunk is a whole number of KiB");
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t stage_lds = __builtin_amdgcn_readfirstlane(lds_address(stage));
  auto issue_chunk = [&](int c) {  // (see glds_chunk in sweep_kernels.hpp)
    glds_chunk<CH * RD / 128, WAVES>(rec_base + (size_t)c * CH * RD, stage_lds + (uint32_t)(c & 1) * (uint32_t)(CH * RD * 8),
                                     wave_s, lane);
  };
  issue_chunk(0);

  // stage R for one raw step: this lane's padded pixel 4 t' + jj
  double *my_ring = ring + (size_t)(group * 16 + s) * kSplitRing;
  auto raw_of = [&](double lamP) -> double {
    double total;
    bool near;
    if (LINES == 3) {
      total = wing_sum3(lamP, ms_r[0], ms_r[LINES > 1 ? 1 : 0], ms_r[LINES > 2 ? 2 : 0], cs, &near);
    } else {
      total = 0.0;
      near = false;
      for (int j = 0; j < L; ++j) {
        const double x = fma(lamP, my_mult[j] * inv_s, -cs);
        const double x2 = x * x;
        near |= x2 < 900.0;
        total = fma(g_lines.cwing[j], wing_core(x2, g_lines.y2[j]), total);
      }
    }
    if (__builtin_expect(__any(near), 0))
      total = total_near<LINES>(lamP, mult_r[0], mult_r[LINES > 1 ? 1 : 0], mult_r[LINES > 2 ? 2 : 0], my_mult, L);
    return exp_table_scaled(nscale64 * total, exp_tab);
  };
  auto lam_of = [&](int tr) -> double { return lam[min(4 * tr + jj, n_pad - 1)]; };
  auto pix_of = [&](int t) -> PixelRow { return pix[4 * min(t, m.steps) + jj]; };  // row `steps` is neutral

  // stage W for one K-step t: absorption of pixel 4 t + jj, weights, partial sums
  double quad_sum = 0.0, dprod = 1.0;
  int dexp = 0;
  auto weights_of = [&](int t, const PixelRow &px, double *w_out, double *u_out) {
    const int p0 = 4 * t + jj;
    const double g0 = my_ring[p0 & 63], g1 = my_ring[(p0 + 1) & 63], g2 = my_ring[(p0 + 2) & 63],
                 g3 = my_ring[(p0 + 3) & 63], g4 = my_ring[(p0 + 4) & 63], g5 = my_ring[(p0 + 5) & 63],
                 g6 = my_ring[(p0 + 6) & 63];
    double absorb = fma(g6, tap0, g0 * tap0);  // voigt.c:297-299 (symmetric taps), as in k_sweep
    double ab2 = fma(g5, tap1, g1 * tap1);
    absorb = fma(g2, tap2, absorb);
    ab2 = fma(g4, tap2, ab2);
    absorb = fma(g3, tap3, absorb) + ab2;
    if (is_null) absorb = 1.0;
    const double r = fma(-absorb, px.mu, px.y);
    const double a2 = absorb * absorb;
    const double d = fma(px.omega2, a2, px.nu);
    const double inv_d = fast_rcp(d);
    const double ri = r * inv_d;
    *w_out = a2 * inv_d;
    *u_out = absorb * ri;
    if (t < m.steps) {  // (steps beyond the last are never consumed; keep them out of the sums)
      quad_sum = fma(r, ri, quad_sum);
      dprod *= d;
      dexp += __builtin_amdgcn_frexp_exp(dprod);
      dprod = __builtin_amdgcn_frexp_mant(dprod);
    }
  };
  auto wu_slot = [&](int par, int which, int step) -> double * {
    return wu + ((((size_t)par * 2 + group) * 2 + which) * 4 + step) * 64 + lane;
  };

  __syncthreads();  // multipliers and the exp table visible
  // prime: raw steps 0..11 (role r: r, 4 + r, 8 + r), then (w, u) of steps 0..3 (role r: step r)
#pragma unroll
  for (int c3 = 0; c3 < 3; ++c3) {
    const int tr = 4 * c3 + role;
    my_ring[(4 * tr + jj) & 63] = raw_of(lam_of(tr));
  }
  __syncthreads();
  {
    double w0, u0;
    weights_of(role, pix_of(role), &w0, &u0);
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
    const int t_w = 4 * (it + 1) + role, t_r = 4 * (it + 3) + role;
    PixelRow px_w;
    double lam_r;
#pragma unroll
    for (int hc = 0; hc < 2; ++hc) {
      const int c = 2 * it + hc;
      if (c < nchunks) {  // block-uniform
        __builtin_amdgcn_s_waitcnt(0x0F70);  // (see k_sweep: free here, keeps compiler waits out of the K-steps)
#ifndef SPLIT_EXP_NODMA
        if (c + 1 < nchunks) issue_chunk(c + 1);
#endif
        if (hc == 0) {
          // operands of this iteration's stages W and R: requested now (behind the vmcnt drain
          // above, so it does not wait for them), used after the K-steps of the second chunk
          px_w = pix_of(t_w);
          lam_r = lam_of(t_r);
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
#ifndef SPLIT_EXP_NOWR
        if (hc == 1) {
          // stages W and R of this role, for the next iteration and the one after the next two
          double w1, u1;
          weights_of(t_w, px_w, &w1, &u1);
          *wu_slot(par ^ 1, 0, role) = w1;
          *wu_slot(par ^ 1, 1, role) = u1;
          my_ring[(4 * t_r + jj) & 63] = raw_of(lam_r);
        }
#endif
        glds_wait();
#ifndef SPLIT_EXP_NOBAR
        __syncthreads();
#endif
      } else if (hc == 1) {
        // odd number of chunks: the last iteration has no second chunk, but stages W/R and the
        // barrier structure must stay uniform (nothing consumes them; skip the work, keep the sync)
        __syncthreads();
      }
    }
  }

  // per-sample scalar sums: the four pixel phases jj, then the four roles (each holds its own steps)
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
  __syncthreads();  // red is read; the epilogue may now overwrite the stage region (it does not reach red)

  // ---- epilogue: all four roles factor.  Per pass the group's waves spill MFMA result registers
  // 2p, 2p+1 of their tiles -- the columns of 8 samples -- to LDS (row rho = 2 jj + h holds sample
  // jj + 4 (2p + h)); role r then factors rows 2r and 2r+1, 32 lanes per sample (k_sweep's form
  // leaves the factorisation to role 0 with 8 lanes per sample while three waves wait).
  using ES = EpilogueShape<TW, TS>;
  constexpr int ncols = ES::stride(logical_tiles(NT));
  constexpr int voff = TW * 16;
  double *Eg = stage + (size_t)group * ES::SPP * ncols;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    __syncthreads();  // previous pass factored (and, for p = 0, the stage buffers are dead)
    double *e = Eg + (size_t)(jj * 2) * ncols;
#pragma unroll
    for (int cc = 0; cc < NTW; ++cc) {
      const int tile = tile0 + cc;
      const int col0 = tile < TW ? tile * 16 : voff + (tile - TW) * 16;  // w-tiles, then u-tiles at voff
      e[col0 + s] = acc[cc][2 * p];
      e[ncols + col0 + s] = acc[cc][2 * p + 1];
    }
    __syncthreads();
    const int rho = 2 * role + (lane >> 5);
    const int sigma = (rho >> 1) + 4 * (2 * p + (rho & 1));  // Mat<double>::sample_of(jj, reg)
    const double q_s = __shfl(quad_sum, sigma), ld_s = __shfl(logd_sum, sigma);
    const int32_t sample_s = __shfl(sample, sigma);
    const double ll = factor_lds<2, 32>(Eg + (size_t)rho * ncols, lane & 31, a.k, voff, q_s, ld_s, m.n_kept);
    const int64_t slot_s = slot0 + sigma;
    if ((lane & 31) == 0) {
      if (slot_s < a.S) a.sample_ll[(int64_t)q * a.S + sample_s] = ll + m.ll_bias;
      else if (slot_s == a.S) a.ll_no_dla[q] = ll + m.ll_bias;
    }
  }
}

}  // namespace gpdla
