// sweep_split_slim_kernel.hpp -- the fp64 sweeps for 20 < k <= 40 on SLIM step records: the B-operand
// tiles vech(m m') are formed inside the sweep (process_qsos.m:185-199; multi-DLA driver
// multi_dlas/process_qsos_multiple_dlas_meanflux.m:340-381).
//
// k_sweep_split / k_sweep_multi_split (sweep_split_kernel.hpp, multi_kernels.hpp) stream, per K-step of
// 4 pixels, a pre-expanded record of 56 MFMA B-operand tiles: 29 696 B of HBM per step, of which
// 4 x 40 doubles are information (r03 PMC: 331 x the algorithmic traffic), in 2-step chunks because
// two of them fill the LDS -- one block barrier per 2 K-steps and one chunk of copy lead, which the
// round-3 ablation priced at ~13 % of the kernel.  Here a step record is the four interpolated M rows
// alone (4 x 48 doubles = 1536 B: m[0..39], six zeros, a one, a zero) and every B operand is formed
// in registers as the product of two LDS reads:
//
//   * Tile split over EIGHT waves.  The four-role form gave each wave 14 tiles of ONE sample group.
//     Here a wave owns 7 tiles of BOTH sample groups of the block (2 x 7 accumulator tiles, the same
//     112 registers): a B operand it forms feeds two MFMAs, so a K-step costs a wave 7 multiplies
//     and 14 LDS operand reads for its 14 MFMAs (the pre-expanded form: 14 reads, no multiply).
//   * Column map.  The MFMA does not care which (i, j) a tile column holds, only the epilogue does.
//     With A = columns 0..15, B = 16..31, C = 32..39 the 820 pairs of the lower triangle are dealt
//     as: 8 + 8 circulant tiles of A x A and B x B (m[c] m[(c + n) mod 16], n = 0..7), one tile
//     holding both blocks' n = 8 half diagonals, 16 tiles B x A, 8 tiles C x A, 8 tiles C x B, two
//     tiles with the circulant diagonals n = 0..3 of C x C and a quarter tile with n = 4: 52 w-tiles;
//     then the three u-tiles (m itself, times the record's 1.0) and one idle tile: 56 = 8 x 7.
//     A lane's two operand addresses per tile are per-lane constants set up once (s40_a / s40_b);
//     K-step and chunk parity are immediate offsets.
//   * Stages.  As in k_sweep_split the per-(sample, pixel) arithmetic is shared through LDS: wave
//     (group g, role r) computes the raw profile (stage R) and the weights (stage W) of its group for
//     the steps = r (mod 4), every wave reads (w, u) of both groups for its MFMAs (stage C).  One
//     loop iteration is now ONE 8-step chunk (12 KiB of records) and ONE block barrier; stage W runs
//     one iteration ahead of stage C, stage R 20 raw steps ahead (a ring of 128 slots per sample).
//     The multi-DLA form gathers profile values in place of stage R.
//
//   * Epilogue.  The 32 Cholesky factorisations of a block run in registers, one sample per 16-lane
//     DPP row, four per wave, in one round (factor_rows16, sweep_kernels.hpp); LDS only transposes
//     the accumulators into rows.
//
// Results are bit-identical to the pre-expanded kernels': the same products, the same MFMA sequence
// per column, the same operations in the epilogue.  (S40_EXP_*: ablation and A/B switches of
// diagnostic builds, tools/ab_build.sh; NO* results are wrong by construction, S40_EXP_PAIRED and
// S40_EXP_LDSEPI select this round's earlier epilogues.)
#pragma once
#include <type_traits>

#include "multi_kernels.hpp"
#include "sweep_split_kernel.hpp"

#ifndef S40_EPI_PW
#define S40_EPI_PW 4  // columns per panel of factor_paired (-DS40_EXP_PAIRED)
#endif

namespace gpdla {
// s_waitcnt immediate for "vmcnt(n), nothing else" (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt
// and lgkmcnt at their maxima)
__host__ __device__ constexpr int vmcnt_imm(int n) { return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14); }
static_assert(vmcnt_imm(0) == 0x0F70, "the encoding glds_wait() uses");
}  // namespace gpdla

namespace gpdla {

constexpr int kS40Row = 48;                 // doubles per pixel of a record
constexpr int kS40Rec = 4 * kS40Row;        // 192 doubles = 1536 B per K-step
constexpr int kS40CH = 8;                   // K-steps per chunk = per loop iteration
constexpr int kS40Zero = 40, kS40One = 46;  // constants carried in every record row
constexpr int kS40Ring = 129;               // 128 raw-profile slots per sample + 1 pad (bank spread)
constexpr int kS40Tiles = 56, kS40TilesW = 52, kS40NTW = 7;
constexpr int kS40Lead = 20;                // raw steps primed; stage R of iteration i produces 8 i + 20 .. 8 i + 27
static_assert(kS40CH <= kRecordPoolPad, "a chunk copy may run this far past a quasar's last record");

// The two record-row columns whose product is column c of tile T (see the column map above).
__host__ __device__ constexpr int s40_a(int T, int c) {
  if (T < 8) return c;                                     // A x A, diagonal n = T
  if (T < 16) return 16 + c;                               // B x B, diagonal n = T - 8
  if (T == 16) return c < 8 ? c : 8 + c;                   // n = 8 of A (c < 8) and of B (c >= 8)
  if (T < 33) return 16 + (T - 17);                        // B x A, row 16 + r
  if (T < 41) return 32 + (T - 33);                        // C x A
  if (T < 49) return 32 + (T - 41);                        // C x B
  if (T < 51) return 32 + (c & 7);                         // C x C, diagonals 0 | 1 and 2 | 3
  if (T == 51) return c < 4 ? 32 + c : kS40Zero;           // C x C, diagonal 4
  if (T == 52) return c;                                   // u-tiles: m itself
  if (T == 53) return 16 + c;
  if (T == 54) return c < 8 ? 32 + c : kS40Zero;
  return kS40Zero;                                         // T == 55: idle
}
__host__ __device__ constexpr int s40_b(int T, int c) {
  if (T < 8) return (c + T) & 15;
  if (T < 16) return 16 + ((c + T - 8) & 15);
  if (T == 16) return c < 8 ? c + 8 : 16 + c;
  if (T < 41) return c;
  if (T < 49) return 16 + c;
  if (T == 49) return 32 + (((c & 7) + (c >> 3)) & 7);
  if (T == 50) return 32 + (((c & 7) + 2 + (c >> 3)) & 7);
  if (T == 51) return c < 4 ? 36 + c : kS40Zero;
  if (T == 54) return c < 8 ? kS40One : kS40Zero;
  if (T < 54) return kS40One;
  return kS40Zero;
}
// Where the epilogue's LDS row of a sample keeps that column: the packed lower triangle
// (idx(i, j) = i (i + 1) / 2 + j), v behind it at voff = 52 * 16; idle columns go to the 12 unused
// slots between them.
__host__ __device__ constexpr int s40_pos(int T, int c) {
  constexpr int voff = kS40TilesW * 16;
  if (T >= kS40TilesW) return voff + (T - kS40TilesW) * 16 + c;
  const int a = s40_a(T, c), b = s40_b(T, c);
  if (a >= 40 || b >= 40) return 820 + (c & 7);
  const int i = a > b ? a : b, j = a > b ? b : a;
  return i * (i + 1) / 2 + j;
}
// every pair (i, j), j <= i < 40, is the product of exactly one w-tile column
constexpr bool s40_map_is_a_bijection() {
  int seen[820] = {};
  for (int T = 0; T < kS40TilesW; ++T)
    for (int c = 0; c < 16; ++c) {
      const int a = s40_a(T, c), b = s40_b(T, c);
      if (a >= 40 || b >= 40) continue;
      ++seen[s40_pos(T, c)];
    }
  for (int e = 0; e < 820; ++e)
    if (seen[e] != 1) return false;
  return true;
}
static_assert(s40_map_is_a_bijection(), "column map of the k <= 40 slim sweep");

// ------------------------------------------------------------------------------------------
// k_build_slim40_records: record(q, t) = 4 pixels x [m[0..39] | 0 x 6 | 1 | 0]; columns >= k are zero.
// Record `steps` (the neutral trailing one) has zero M rows like k_build_records'.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_build_slim40_records(BuildRecordsArgs a) {
  const int q = a.order[blockIdx.x / a.blocks_per_quasar];
  const int bq = blockIdx.x % a.blocks_per_quasar;
  const QuasarMeta m = a.meta[q];
  const int k = a.k;
  double *out = a.records + m.rec_off * (int64_t)kS40Rec;
  const int64_t total = (int64_t)(m.steps + 1) * kS40Rec;
  for (int64_t e = (int64_t)bq * 256 + threadIdx.x; e < total; e += (int64_t)a.blocks_per_quasar * 256) {
    const int64_t row = e / kS40Row;  // = 4 step + jj
    const int col = (int)(e - row * kS40Row);
    double v = 0.0;
    if (col < k) v = a.Mi[(m.pix_off + row) * k + col];
    else if (col == kS40One) v = 1.0;
    out[e] = v;
  }
}

// LDS doubles of the loop phase (the epilogue reuses the array from the stage buffers on)
__host__ __device__ constexpr size_t sweep_split_slim_lds_doubles(bool multi) {
  return kExpTab + 2 * (size_t)kS40CH * kS40Rec      // exp table | two chunks of raw records
         + 2 * 2 * 2 * (size_t)kS40CH * 64           // (w, u): parity x group x {w, u} x step x lane
         + 2 * 4 * 2 * 16                            // per-role partial sums at the end
         + (multi ? 0 : 2 * 16 * (size_t)kS40Ring);  // raw ring per group
}

// LINES: number of Lyman lines when known at compile time (0: read at run time); single-DLA only.
// ND == 0: the single-DLA sweep (Args = SweepArgs); ND >= 1: the multi-DLA sweep of a model whose
// samples multiply ND profiles (Args = SweepMultiArgs; mode 0, the sub-DLA pass, has ND = 1).
template <int LINES, int ND, typename Args>
__global__ __launch_bounds__(512) void k_sweep_split_slim(Args a) {
  extern __shared__ double smem[];
  constexpr bool kMulti = ND > 0;
  constexpr int NDR = ND > 0 ? ND : 1;
  constexpr int CH = kS40CH, NTW = kS40NTW, TW = kS40TilesW, NT = kS40Tiles;
  const int64_t xj = blockIdx.x >> 3;
  const int bq = (int)(xj % a.blocks_per_quasar);
  int64_t q, ql = 0;
  if constexpr (kMulti) {
    ql = 8 * (xj / a.blocks_per_quasar) + (blockIdx.x & 7);
    if (ql >= a.nq_sub) return;
    q = a.q0 + ql;
  } else {
    const int64_t pos = 8 * (xj / a.blocks_per_quasar) + (blockIdx.x & 7);
    if (pos >= a.nq) return;
    q = a.order[pos];
  }
  const QuasarMeta m = a.meta[q];
  if (m.status != 0) return;
  if constexpr (kMulti) {
    if (a.alive[q] == 0) return;
  }
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int group = wave >> 2, role = wave & 3;
  const int s = lane & 15, jj = lane >> 4;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);

  double *exp_tab = smem;                                   // [64]
  double *stage = exp_tab + kExpTab;                        // [2][CH][kS40Rec]
  double *wu = stage + (size_t)2 * CH * kS40Rec;            // [2 parity][2 groups][2][CH][64]
  double *red = wu + 2 * 2 * 2 * CH * 64;                   // [2 groups][4 roles][2][16]
  double *ring = red + 2 * 4 * 2 * 16;                      // [2 groups][16][kS40Ring]   (single-DLA)

  const int64_t slot0 = (int64_t)bq * (2 * kSamplesPerWave) + group * kSamplesPerWave;
  const int64_t slot = slot0 + s;
  const bool is_sample = slot < a.S;
  const bool is_null = !is_sample;  // slot == S is the null model; slots beyond it are idle copies
  const PixelRow *pix = a.pix + m.pix_off;
  const double *rec_base = a.records + m.rec_off * (int64_t)kS40Rec;
  const int niter = (m.steps + CH - 1) / CH;  // = chunks

  // ---- producer state: Voigt stages (single-DLA) or profile-table rows (multi-DLA) ----------------
  [[maybe_unused]] int32_t sample = 0;
  [[maybe_unused]] int chain_ok = 1;
  [[maybe_unused]] const double *rows[4] = {nullptr, nullptr, nullptr, nullptr};
  [[maybe_unused]] int L = 0;
  [[maybe_unused]] double nscale64 = 0.0, cs = 0.0, inv_s = 0.0;
  [[maybe_unused]] double mult_r[LINES > 0 ? LINES : 1], ms_r[LINES > 0 ? LINES : 1];
  [[maybe_unused]] double *my_ring = nullptr;
  [[maybe_unused]] double opz = 1.0, inv_opz = 1.0;  // run-time line count: 1 + z_DLA and its reciprocal (wing_sum_runtime)
  [[maybe_unused]] const double *lam = nullptr;
  [[maybe_unused]] int n_pad = 0;
  if constexpr (kMulti) {
    const int64_t i = is_sample ? slot : 0;
    // rows of the profile table this lane multiplies (multi :342-351); an index outside [1, S]
    // (0 = never drawn, multi :116, :460-464) is never followed: the sample becomes NaN
    rows[0] = a.prof + ((ql * 2 + (a.mode == 0 ? 1 : 0)) * a.S + i) * a.stride;
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
  } else {
    L = LINES > 0 ? LINES : a.num_lines;
    sample = is_sample ? a.perm[slot] : 0;
    const double z_dla = m.min_z_dla + (m.max_z_dla - m.min_z_dla) * a.offset_samples[sample];  // :162-164
    const double nhi = a.nhi_samples[sample];
    if (LINES > 0) {
#pragma unroll
      for (int j = 0; j < LINES; ++j) mult_r[j] = g_lines.c / (g_lines.wavelength_cm[j] * (1 + z_dla)) / 1e8;  // voigt.c:278-279
    } else {
      opz = 1 + z_dla;
      inv_opz = 1.0 / opz;
    }
    if (tid < kExpTab) exp_tab[tid] = exp2((double)tid * (1.0 / kExpTab));
    lam = a.lam_pad + m.lam_off;
    n_pad = m.n_u + 6;
    nscale64 = -nhi * g_lines.inv_sqrt2pi_sigma * kInvSqrtPi * kExpScale;  // (pre-scaled exp, sweep_kernels.hpp)
    inv_s = g_lines.inv_sqrt2_sigma;
#pragma unroll
    for (int j = 0; j < (LINES > 0 ? LINES : 0); ++j) ms_r[j] = mult_r[j] * inv_s;
    cs = g_lines.c * inv_s;
    my_ring = ring + (size_t)(group * 16 + s) * kS40Ring;
  }
  const double tap0 = g_lines.taps[0], tap1 = g_lines.taps[1], tap2 = g_lines.taps[2], tap3 = g_lines.taps[3];

  static_assert((CH * kS40Rec) % 128 == 0, "a chunk is a whole number of KiB");
  const uint32_t stage_lds = __builtin_amdgcn_readfirstlane(lds_address(stage));
  auto issue_chunk = [&](int c) {  // (see glds_chunk in sweep_kernels.hpp)
    glds_chunk<CH * kS40Rec / 128, 8>(rec_base + (size_t)c * CH * kS40Rec,
                                      stage_lds + (uint32_t)(c & 1) * (uint32_t)(CH * kS40Rec * 8), wave_s, lane);
  };
  issue_chunk(0);

  // ---- stage C operands: two LDS addresses per tile, per lane, for K-step 0 of parity 0 -----------
  uint32_t pa[NTW], pb[NTW];
#pragma unroll
  for (int cc = 0; cc < NTW; ++cc) {
    const int T = wave_s * NTW + cc;
    pa[cc] = stage_lds + (uint32_t)(jj * kS40Row + s40_a(T, s)) * 8u;
    pb[cc] = stage_lds + (uint32_t)(jj * kS40Row + s40_b(T, s)) * 8u;
    // (opaque: the compiler otherwise keeps the index and adds the array's base in every K-step)
    asm volatile("" : "+v"(pa[cc]), "+v"(pb[cc]));
  }
  auto lds_at = [](uint32_t addr, int byte_off) -> double {  // (pointer arithmetic: the offset folds into the instruction)
    return ((const __attribute__((address_space(3))) double *)(uintptr_t)addr)[byte_off / 8];
  };
  // (w, u) table of a parity, as this lane sees it; the last wave's four u-tiles take u
  auto wu_at = [&](int par, int g, int which, int step) -> double * {
    return wu + ((((size_t)par * 2 + g) * 2 + which) * CH + step) * 64 + lane;
  };
  const int tail_which = wave_s == 7 ? 1 : 0;

  // ---- stage R for one raw step: this lane's padded pixel 4 t' + jj (single-DLA) -------------------
  auto raw_of = [&](double lamP) -> double {
    double total;
    bool near;
    if (LINES == 3) {
      total = wing_sum3(lamP, ms_r[0], ms_r[LINES > 1 ? 1 : 0], ms_r[LINES > 2 ? 2 : 0], cs, &near);
    } else {
      total = wing_sum_runtime(lamP * inv_opz, cs, L, &near);
    }
    if (__builtin_expect(__any(near), 0)) {
      if constexpr (LINES > 0) total = total_near<LINES>(lamP, mult_r[0], mult_r[LINES > 1 ? 1 : 0], mult_r[LINES > 2 ? 2 : 0], nullptr, L);
      else total = total_near_at(lamP, opz, L);
    }
    return exp_table_scaled(nscale64 * total, exp_tab);
  };
  auto lam_of = [&](int tr) -> double { return lam[min(4 * tr + jj, n_pad - 1)]; };
  auto pix_of = [&](int t) -> PixelRow { return pix[4 * min(t, m.steps) + jj]; };  // row `steps` is neutral
  const int p_last = 4 * m.steps + jj;  // profile rows are padded to 4 (steps + 1) entries
  auto gather = [&](int t, double (&r)[NDR]) {
    const int p = min(4 * t + jj, p_last);
#pragma unroll
    for (int j = 0; j < NDR; ++j) r[j] = rows[j][p];
  };

  // ---- stage W for one K-step t: absorption of pixel 4 t + jj, weights, partial sums --------------
  double quad_sum = 0.0, dprod = 1.0;
  int dexp = 0;
  auto weigh = [&](int t, const PixelRow &px, double absorb, double *w_out, double *u_out) {
    if (is_null) absorb = 1.0;
    const double r = fma(-absorb, px.mu, px.y);
    const double a2 = absorb * absorb;
    const double d = fma(px.omega2, a2, px.nu);
    const double inv_d = fast_rcp(d);
    *w_out = a2 * inv_d;
    if constexpr (kMulti) {  // (the operation order of k_sweep_multi_split: results stay bit-identical)
      *u_out = absorb * r * inv_d;
      if (t < m.steps) quad_sum = fma(r * r, inv_d, quad_sum);
    } else {
      const double ri = r * inv_d;
      *u_out = absorb * ri;
      if (t < m.steps) quad_sum = fma(r, ri, quad_sum);
    }
    if (t < m.steps) {  // (steps beyond the last are never consumed; keep them out of the sums)
      dprod *= d;
      dexp += __builtin_amdgcn_frexp_exp(dprod);
      dprod = __builtin_amdgcn_frexp_mant(dprod);
    }
  };
  auto absorb_ring = [&](int t) -> double {  // voigt.c:297-299 (symmetric taps), as in k_sweep
    const int p0 = 4 * t + jj;
    const double g0 = my_ring[p0 & 127], g1 = my_ring[(p0 + 1) & 127], g2 = my_ring[(p0 + 2) & 127],
                 g3 = my_ring[(p0 + 3) & 127], g4 = my_ring[(p0 + 4) & 127], g5 = my_ring[(p0 + 5) & 127],
                 g6 = my_ring[(p0 + 6) & 127];
    double absorb = fma(g6, tap0, g0 * tap0);
    double ab2 = fma(g5, tap1, g1 * tap1);
    absorb = fma(g2, tap2, absorb);
    ab2 = fma(g4, tap2, ab2);
    return fma(g3, tap3, absorb) + ab2;
  };
  auto absorb_gathered = [&](const double (&r)[NDR]) -> double {
    double v = r[0];
#pragma unroll
    for (int j = 1; j < NDR; ++j) v *= r[j];
    return v;
  };

  // ---- prime: raw steps 0 .. kS40Lead - 1 (role r: r, 4 + r, ...), then (w, u) of steps 0 .. 7 -----
  // multi-DLA: profile values of the two W stages of an iteration, one set per iteration parity: an
  // iteration of parity P consumes set P and requests set P ^ 1 for the next iteration -- no register
  // moves, and a gather has a whole iteration (8 K-steps) to land
  [[maybe_unused]] double g_buf[2][2][NDR];
  if constexpr (!kMulti) {
    __syncthreads();  // multipliers and the exp table visible
#pragma unroll
    for (int c5 = 0; c5 < kS40Lead / 4; ++c5) {
      const int tr = 4 * c5 + role;
      my_ring[(4 * tr + jj) & 127] = raw_of(lam_of(tr));
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int t = 4 * h + role;
      double w0, u0;
      weigh(t, pix_of(t), absorb_ring(t), &w0, &u0);
      *wu_at(0, group, 0, t) = w0;
      *wu_at(0, group, 1, t) = u0;
    }
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int t = 4 * h + role;
      double g0[NDR], w0, u0;
      gather(t, g0);
      gather(CH + t, g_buf[0][h]);  // for the W stage of iteration 0
      weigh(t, pix_of(t), absorb_gathered(g0), &w0, &u0);
      *wu_at(0, group, 0, t) = w0;
      *wu_at(0, group, 1, t) = u0;
    }
  }
  d4 acc0[NTW], acc1[NTW];
#pragma unroll
  for (int c = 0; c < NTW; ++c) {
    acc0[c] = d4{0.0, 0.0, 0.0, 0.0};
    acc1[c] = d4{0.0, 0.0, 0.0, 0.0};
  }

  glds_wait();  // chunk 0 landed
  __syncthreads();

  // One iteration = one chunk of 8 K-steps from parity PAR, in two halves of four: stage C for the
  // four steps, then this wave's stage W for step 8 (it + 1) + 4 h + role (into the other parity's
  // table) and stage R for raw step 8 it + kS40Lead + 4 h + role.  Their global operands are
  // requested at the top of the half.
  auto iteration = [&](auto PARC, int it) {
    constexpr int PAR = decltype(PARC)::value;
    // (see k_sweep: free here, keeps compiler waits out of the K-steps; multi-DLA: the ND profile values
    // requested in the second half of the previous iteration may still be in flight)
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(kMulti ? ND : 0));
    if (it + 1 < niter) issue_chunk(it + 1);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int t_w = CH * (it + 1) + 4 * h + role;
      [[maybe_unused]] const int t_r = CH * it + kS40Lead + 4 * h + role;
      const PixelRow px_w = pix_of(t_w);
      [[maybe_unused]] double lam_r = 0.0;
      if constexpr (kMulti) gather(t_w + CH, g_buf[PAR ^ 1][h]);  // for the W stage of the NEXT iteration
      else lam_r = lam_of(t_r);
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        const int tt = 4 * h + t4;
        if (CH * it + tt < m.steps) {
          constexpr int kStepBytes = kS40Rec * 8;
          const int off = (PAR * CH + tt) * kStepBytes;
          const double w0 = *wu_at(PAR, 0, 0, tt), w1 = *wu_at(PAR, 1, 0, tt);
          const double t0 = *wu_at(PAR, 0, tail_which, tt), t1 = *wu_at(PAR, 1, tail_which, tt);
          double opa[NTW], opb[NTW], bop[NTW];
#pragma unroll
          for (int cc = 0; cc < NTW; ++cc) {
            opa[cc] = lds_at(pa[cc], off);
#ifndef S40_EXP_NOMUL
            opb[cc] = lds_at(pb[cc], off);
#endif
          }
          __builtin_amdgcn_sched_barrier(0);  // all 18 reads requested before the first product waits
#pragma unroll
          for (int cc = 0; cc < NTW; ++cc) {
#ifdef S40_EXP_NOMUL
            bop[cc] = opa[cc];
#else
            bop[cc] = opa[cc] * opb[cc];
#endif
          }
#ifdef S40_EXP_NOMFMA
#pragma unroll
          for (int cc = 0; cc < NTW; ++cc) {
            asm volatile("" ::"v"(bop[cc]));
            if (cc < 1) acc0[cc][0] += (cc < 3 ? w0 : t0) * bop[cc] + w1 * t1;
          }
#else
#pragma unroll
          for (int cc = 0; cc < NTW; ++cc) {
            acc0[cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(cc < 3 ? w0 : t0, bop[cc], acc0[cc], 0, 0, 0);
            acc1[cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(cc < 3 ? w1 : t1, bop[cc], acc1[cc], 0, 0, 0);
          }
#endif
        }
      }
#ifndef S40_EXP_NOWR
      double w1, u1;
      if constexpr (kMulti) {
        weigh(t_w, px_w, absorb_gathered(g_buf[PAR][h]), &w1, &u1);
      } else {
        weigh(t_w, px_w, absorb_ring(t_w), &w1, &u1);
      }
      *wu_at(PAR ^ 1, group, 0, 4 * h + role) = w1;
      *wu_at(PAR ^ 1, group, 1, 4 * h + role) = u1;
      if constexpr (!kMulti) my_ring[(4 * t_r + jj) & 127] = raw_of(lam_r);
#else
      asm volatile("" ::"v"(px_w.y), "v"(lam_r));
#endif
    }
    // the chunk copy has landed.  Multi-DLA: the ND gathers of the second half are younger than the
    // copy and stay in flight (the wait for that half's pixel row has already drained everything older)
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(kMulti ? ND : 0));
    asm volatile("" ::: "memory");
#ifndef S40_EXP_NOBAR
    __syncthreads();
#endif
  };
  for (int it = 0; it < niter; it += 2) {
    iteration(std::integral_constant<int, 0>{}, it);
    if (it + 1 < niter) iteration(std::integral_constant<int, 1>{}, it + 1);
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
  __syncthreads();  // red is read; the epilogue may now overwrite everything behind the exp table

  // ---- epilogue: every wave spills result registers 2p, 2p + 1 of its 7 tiles, for both groups, to
  // the groups' LDS rows (row rho = 2 jj + h holds sample jj + 4 (2p + h)) through the column map
  using ES = EpilogueShape<TW, 4>;
  constexpr int ncols = ES::stride(logical_tiles(NT));
  constexpr int voff = TW * 16;
  int at[NTW];
#pragma unroll
  for (int cc = 0; cc < NTW; ++cc) at[cc] = s40_pos(wave_s * NTW + cc, s);
  double *Eg = stage;
  double *e0 = Eg + (size_t)(jj * 2) * ncols, *e1 = e0 + (size_t)ES::SPP * ncols;
#if !defined(S40_EXP_LDSEPI) && !defined(S40_EXP_PAIRED) && !defined(S40_EXP_NOEPI)
  // The accumulators pass through LDS in two rounds of 16 samples (32 x 861 doubles do not fit) only
  // to be transposed into rows: after round p the waves of roles 2p, 2p + 1 take four of its samples
  // each -- one per 16-lane row, all k + 1 rows of a sample in registers (load_rows16) -- and once
  // both rounds are through, all eight waves factor their four samples at once, in registers
  // (factor_rows16).  The waves keep to samples of their own group: its scalar sums are in their lanes.
  Rows16<40> R;
  const int round_mine = role >> 1;
  const int rho = 4 * (role & 1) + jj;  // the group's LDS row this 16-lane row takes
  {
    const double *erow = Eg + (size_t)(group * ES::SPP + rho) * ncols;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      __syncthreads();  // p = 0: the loop's buffers are dead; p = 1: round 0's rows are in registers
#pragma unroll
      for (int cc = 0; cc < NTW; ++cc) {
        e0[at[cc]] = acc0[cc][2 * p];
        e0[ncols + at[cc]] = acc0[cc][2 * p + 1];
        e1[at[cc]] = acc1[cc][2 * p];
        e1[ncols + at[cc]] = acc1[cc][2 * p + 1];
      }
      __syncthreads();
      if (round_mine == p) load_rows16<40>(R, erow, s, a.k, voff);  // wave-uniform
    }
  }
  {
    const int p = round_mine;
    const int sigma = (rho >> 1) + 4 * (2 * p + (rho & 1));  // Mat<double>::sample_of(jj, reg)
    const double q_s = __shfl(quad_sum, sigma), ld_s = __shfl(logd_sum, sigma);
    const double ll = factor_rows16<40>(R, a.k, q_s, ld_s, m.n_kept);
    const bool writer_lane = s == 0;
    const int64_t slot_s = slot0 + sigma;
    if constexpr (kMulti) {
      const bool ok_s = __shfl(chain_ok, sigma) != 0;
      if (writer_lane) {
        if (slot_s < a.S) {
          if (a.mode == 0) a.sample_ll_lls[q * a.S + slot_s] = ll + m.ll_bias - a.log_S;     // multi :376-378
          else a.sample_ll_dla[(q * a.max_dlas + (a.mode - 1)) * a.S + slot_s] = ok_s ? ll + m.ll_bias - a.log_S : NAN;  // :359-361
        } else if (slot_s == a.S && a.mode == 1) {
          a.ll_no_dla[q] = ll + m.ll_bias;                                                    // multi :296-298
        }
      }
    } else {
      const int32_t sample_s = __shfl(sample, sigma);
      if (writer_lane) {
        if (slot_s < a.S) a.sample_ll[(int64_t)q * a.S + sample_s] = ll + m.ll_bias;
        else if (slot_s == a.S) a.ll_no_dla[q] = ll + m.ll_bias;
      }
    }
  }
#else  // the round-4 predecessors, for A/B: two passes of 16 samples, 32 lanes per sample, factored next to LDS
  __syncthreads();  // the loop's buffers are dead
#pragma unroll
  for (int cc = 0; cc < NTW; ++cc) {
    e0[at[cc]] = acc0[cc][0];
    e0[ncols + at[cc]] = acc0[cc][1];
    e1[at[cc]] = acc1[cc][0];
    e1[ncols + at[cc]] = acc1[cc][1];
  }
  // ONE copy of the factorisation's straight-line code (4 to 6 thousand instructions).  The first
  // pass's spill is peeled off above, so only the second pass's result registers (56 of the 112) are
  // live while the first pass factors; with the 48 row entries per lane factor_paired keeps that
  // still fits the register file -- nothing of the epilogue may go to scratch: 100 bytes per lane
  // and block would already double the kernel's HBM traffic
#ifdef S40_EXP_UNROLLP
#pragma unroll
#else
#pragma nounroll
#endif
  for (int p = 0; p < 2; ++p) {
    if (p == 1) {
      __syncthreads();  // first pass factored
#pragma unroll
      for (int cc = 0; cc < NTW; ++cc) {
        e0[at[cc]] = acc0[cc][2];
        e0[ncols + at[cc]] = acc0[cc][3];
        e1[at[cc]] = acc1[cc][2];
        e1[ncols + at[cc]] = acc1[cc][3];
      }
    }
    __syncthreads();
    const int rho = 2 * role + (lane >> 5);
    const int sigma = (rho >> 1) + 4 * (2 * p + (rho & 1));  // Mat<double>::sample_of(jj, reg)
    const double q_s = __shfl(quad_sum, sigma), ld_s = __shfl(logd_sum, sigma);
#ifdef S40_EXP_NOEPI
    const double ll = q_s + ld_s + Eg[(size_t)(group * ES::SPP + rho) * ncols + (lane & 31)];
#elif defined(S40_EXP_LDSEPI)
    const double ll = factor_lds<2, 32>(Eg + (size_t)(group * ES::SPP + rho) * ncols, lane & 31, a.k, voff, q_s, ld_s,
                                        m.n_kept);
#else
    const double ll = factor_paired<40, S40_EPI_PW>(Eg + (size_t)(group * ES::SPP + rho) * ncols, lane & 31, a.k, voff, q_s, ld_s,
                                        m.n_kept);
#endif
    const int64_t slot_s = slot0 + sigma;
    if constexpr (kMulti) {
      const bool ok_s = __shfl(chain_ok, sigma) != 0;
      if ((lane & 31) == 0) {
        if (slot_s < a.S) {
          if (a.mode == 0) a.sample_ll_lls[q * a.S + slot_s] = ll + m.ll_bias - a.log_S;     // multi :376-378
          else a.sample_ll_dla[(q * a.max_dlas + (a.mode - 1)) * a.S + slot_s] = ok_s ? ll + m.ll_bias - a.log_S : NAN;  // :359-361
        } else if (slot_s == a.S && a.mode == 1) {
          a.ll_no_dla[q] = ll + m.ll_bias;                                                    // multi :296-298
        }
      }
    } else {
      const int32_t sample_s = __shfl(sample, sigma);
      if ((lane & 31) == 0) {
        if (slot_s < a.S) a.sample_ll[(int64_t)q * a.S + sample_s] = ll + m.ll_bias;
        else if (slot_s == a.S) a.ll_no_dla[q] = ll + m.ll_bias;
      }
    }
  }
#endif
}

}  // namespace gpdla
