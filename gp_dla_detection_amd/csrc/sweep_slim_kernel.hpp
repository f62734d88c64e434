// sweep_slim_kernel.hpp -- the fp64 sweep for k <= 20, three Lyman lines (the production case), on
// SLIM step records: the B-operand tiles vech(m m') are formed inside the sweep.
//
// k_sweep (sweep_kernels.hpp) streams, per K-step of 4 pixels, a pre-expanded record of 14 MFMA
// B-operand tiles: 7680 B of HBM per step, 2.9 MB per quasar, 2.9 GB per 1000 quasars -- 18 times
// the algorithmic traffic, 59 GB for one DR12Q shard.  Only 4 x (20 + 5) of those 960 doubles are
// information: the 4 interpolated M rows, the 4 pixel rows and the wavelengths.  Here a step record
// is those alone (896 B, 0.34 MB per quasar) and the 13 vech tiles are produced on the fly:
//
//   * Column map.  The MFMA does not care which (i, j) a tile column holds, only the epilogue does,
//     so the 210 = 208 + 2 entries of the lower triangle are dealt to (tile, column) such that a
//     lane can form its 13 products from ONE address register and immediate offsets: with c = its
//     column (0..15) and the row of M stored twice in a row in LDS,
//        tile n = 0..7 :  m[c] * m[(c + n) mod 16]       the n-th circulant diagonal of the 16 x 16 block
//        tile 8        :  m[c] * m[c + 8]  (c < 8);  for c >= 8 eight of the ten pairs (16+a, 16+b)
//        tile 9 + r    :  m[c] * m[16 + r]               r = 0..3
//     and the last two pairs, (19, 18) and (19, 19), stay on the VALU as in k_sweep (vech columns
//     208, 209).  Per tile: one ds_read_b64 (immediate offset 8 n), one v_mul_f64, one ds_write_b64.
//   * Who does it.  Wave w of the block expands K-step w of the NEXT chunk (8 steps per chunk, 8
//     waves) into the tile buffer the block will read after the next barrier, spread over K-steps
//     0..6 of the current chunk: 13 multiplies per wave and chunk, 1.6 per K-step, against 93 VALU
//     instructions a K-step already has.  Its 4 M rows arrive by a private 1-KiB LDS-DMA whose
//     per-lane source addresses lay each 16-double row down twice (that is what makes (c + n) mod 16
//     an immediate offset); nobody else reads that landing zone, so it needs no barrier.
//   * The rest of a record (pixel rows, m columns 16..19, the two VALU columns, wavelengths, and m
//     columns 0..15 in lane order, which IS the u tile) is copied whole by the block's chunk DMA,
//     double-buffered as before.  One barrier per chunk, as before.
//
// LDS (exactly the CU's 160 KiB): ring 33 792 | two parities of [8 x 13 tiles of 512 B | 8 raw records
// = 7 KiB] | 8 landing zones of 1152 B.  The 64-entry exp table lives in the ring's 128 pad slots.
// Every per-parity address is (per-lane constant) + parity x 60 416 B: four adds per chunk, none per
// K-step.
// Results are bit-identical to k_sweep's: the same products, the same MFMA sequence per column.
#pragma once
#include <type_traits>

#include "sweep_kernels.hpp"

namespace gpdla {

constexpr int kSlimExtras = 12;                      // doubles per pixel: y mu omega2 nu | m16..19 | p208 p209 | lam | pad
constexpr int kSlimRec = 4 * kSlimExtras + 64;       // 112 doubles = 896 B per K-step: extras, then m[0..15] of 4 pixels
constexpr int kSlimCH = 8;                           // K-steps per chunk = waves per block
constexpr int kSlimTilesW = 13;
constexpr int kSlimStepTiles = kSlimTilesW * 64;     // doubles of expanded tiles per K-step
constexpr int kSlimRingD = kSweepWaves * kSamplesPerWave * kRing2;   // 4224
constexpr int kSlimTileBuf = kSlimCH * kSlimStepTiles;               // 6656
constexpr int kSlimRawBuf = kSlimCH * kSlimRec;                      // 896
constexpr int kSlimLand = 4 * 32 + 4 * 4;                            // per wave: 4 doubled rows + 4 x m16..19
constexpr int kSlimBlock = kSlimTileBuf + kSlimRawBuf;               // one parity: tiles, then raw records (7552 doubles)
constexpr int kSlimLdsDoubles = kSlimRingD + 2 * kSlimBlock + kSweepWaves * kSlimLand;
static_assert((kSlimBlock * 8) % 512 == 0, "both parities reachable with ds_read2st64 offsets");
static_assert(kSlimLdsDoubles * 8 == 160 * 1024, "the slim sweep uses the whole LDS of a CU");
static_assert((kSlimCH * kSlimRec) % 128 == 0, "a raw chunk is a whole number of KiB");

// (i, j), i >= j, of column `col` of tile `tile` (see the column map above)
__host__ __device__ constexpr int slim_pair_i(int tile, int col) {
  if (tile < 8) {
    const int b = (col + tile) & 15;
    return col > b ? col : b;
  }
  if (tile == 8) {
    if (col < 8) return col + 8;
    const int l = col - 8;  // (16,16) (17,16) (17,17) (18,16) (18,17) (18,18) (19,16) (19,17)
    return l < 1 ? 16 : l < 3 ? 17 : l < 6 ? 18 : 19;
  }
  return 16 + (tile - 9);
}
__host__ __device__ constexpr int slim_pair_j(int tile, int col) {
  if (tile < 8) {
    const int b = (col + tile) & 15;
    return col > b ? b : col;
  }
  if (tile == 8) {
    if (col < 8) return col;
    const int l = col - 8;
    return 16 + (l < 1 ? 0 : l < 3 ? l - 1 : l < 6 ? l - 3 : l - 6);
  }
  return col;
}
// position in the packed lower triangle (row-wise, idx(i, j) = i (i + 1) / 2 + j) the epilogue reads
__host__ __device__ constexpr int slim_pos(int tile, int col) {
  return slim_pair_i(tile, col) * (slim_pair_i(tile, col) + 1) / 2 + slim_pair_j(tile, col);
}

// ------------------------------------------------------------------------------------------
// k_build_slim_records: record(q, t) = [4 pixels x 12 extras | 4 pixels x m[0..15]], record `steps`
// neutral, as k_build_records' trailing one.  A pure gather: 112 doubles per K-step.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_build_slim_records(BuildRecordsArgs a) {
  const int q = a.order[blockIdx.x / a.blocks_per_quasar];
  const int bq = blockIdx.x % a.blocks_per_quasar;
  const QuasarMeta m = a.meta[q];
  const int k = a.k;
  const int n_pad = m.n_u + 6;
  double *out = a.records + m.rec_off * (int64_t)kSlimRec;
  const int64_t total = (int64_t)(m.steps + 1) * kSlimRec;
  for (int64_t e = (int64_t)bq * 256 + threadIdx.x; e < total; e += (int64_t)a.blocks_per_quasar * 256) {
    const int step = (int)(e / kSlimRec), r = (int)(e - (int64_t)step * kSlimRec);
    double v = 0.0;
    if (r < 4 * kSlimExtras) {
      const int jj = r / kSlimExtras, f = r - jj * kSlimExtras;
      const int64_t row = m.pix_off + 4 * (int64_t)step + jj;
      if (f < 4) {
        const PixelRow px = a.pix[row];
        v = f == 0 ? px.y : f == 1 ? px.mu : f == 2 ? px.omega2 : px.nu;
      } else if (f < 8) {
        if (kXUColumn + (f - 4) < k) v = a.Mi[row * k + kXUColumn + (f - 4)];
      } else if (f < 10) {  // vech columns 208, 209 = (19, 18), (19, 19)
        if (k == 20) v = a.Mi[row * k + 19] * a.Mi[row * k + 18 + (f - 8)];
      } else if (f == 10) {
        int P = 4 * (step + 3) + jj;
        if (P > n_pad - 1) P = n_pad - 1;
        v = a.lam_pad[m.lam_off + P];
      }
    } else {
      const int l = r - 4 * kSlimExtras, jj = l >> 4, col = l & 15;
      if (col < k) v = a.Mi[(m.pix_off + 4 * (int64_t)step + jj) * k + col];
    }
    out[e] = v;
  }
}

// exp tables in the ring's pad slots: entry j of the 2^(j/64) table sits in slot 32 of ring row j
__device__ __forceinline__ ExpState exp_ring_begin_scaled(double t, const double *ring_pad) {
  ExpState e;
  const double nf = rint(t);
  asm("v_cvt_i32_f64 %0, %1" : "=v"(e.ni) : "v"(nf));  // saturating (see exp_table_begin)
  e.tabv = ring_pad[(e.ni & (kExpTab - 1)) * kRing2];
  e.r = t - nf;
  return e;
}

// Epilogue pass of the slim sweep: factor_pass of sweep_kernels.hpp with the spill scattered
// through the column map (tile, column) -> packed-triangle position.
__device__ __forceinline__ double slim_factor_pass(const d4 (&acc)[14], const double (&xw)[kXW],
                                                   const double (&xu)[kXU], int p, double *Eg, int lane, int k,
                                                   double quad_sum, double logd_sum, int n_kept,
                                                   int *sigma_out, bool *writer) {
  using ES = EpilogueShape<13, 1>;
  constexpr int voff = (13 + 1) * 16;
  constexpr int ncols = ES::stride(16);
  const int s = lane & 15, jj = lane >> 4;
  const int half = s >> 3;
  const int sigma = Mat<double>::sample_of(jj, ES::RPP * p + half);
  const double q_s = __shfl(quad_sum, sigma + 16 * jj);
  const double ld_s = __shfl(logd_sum, sigma + 16 * jj);
  double *e = Eg + (size_t)(jj * ES::RPP) * ncols;
#pragma unroll
  for (int cc = 0; cc < 14; ++cc) {
    const int at = cc < kSlimTilesW ? slim_pos(cc, s) : voff + s;  // the u tile: v[0..15]
#pragma unroll
    for (int h = 0; h < ES::RPP; ++h) e[h * ncols + at] = acc[cc][ES::RPP * p + h];
  }
  {
    const int r = Mat<double>::reg_of(s);
    if (jj == 0 && r / ES::RPP == p) {
      double *es = Eg + (size_t)(Mat<double>::jj_of(s) * ES::RPP + r % ES::RPP) * ncols;
#pragma unroll
      for (int x = 0; x < kXW; ++x) es[kXWColumn + x] = xw[x];
#pragma unroll
      for (int x = 0; x < kXU; ++x) es[voff + kXUColumn + x] = xu[x];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  *sigma_out = sigma;
  *writer = (s & (ES::LPS - 1)) == 0;
  return factor_rows<ES::ROWS, ES::LPS, 20>(e + half * ncols, s & (ES::LPS - 1), k, voff, q_s, ld_s, n_kept);
}

// LINES: 3 (set_parameters.m:63, the production value: the three-line wing tier wing_sum3), or 0: the
// line count is a.num_lines, read at run time (voigt.c:16, 266 default to all 31).  The 160 KiB of LDS
// are spoken for, so the run-time form keeps no per-sample table of line multipliers: the wing tier
// takes x_j = (lambda / (1 + z_DLA)) kms_j - c / (sqrt2 sigma) with kms_j from constant memory (scalar
// loads; wing_sum_runtime), and the rare near tier forms the reference's own multiplier (voigt.c:278-279) on the spot.
template <int LINES>
__global__ __launch_bounds__(512) void k_sweep_slim(SweepArgs a) {
  static_assert(LINES == 3 || LINES == 0, "three lines at compile time, or a run-time count");
  extern __shared__ double smem[];
  constexpr int WAVES = kSweepWaves, CH = kSlimCH;
  const int64_t xj = blockIdx.x >> 3;
  const int64_t pos = 8 * (xj / a.blocks_per_quasar) + (blockIdx.x & 7);
  const int bq = (int)(xj % a.blocks_per_quasar);
  if (pos >= a.nq) return;
  const int64_t q = a.order[pos];  // quasars dealt to the XCDs in order of decreasing length (k_sweep)
  const QuasarMeta m = a.meta[q];
  if (m.status != 0) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int s = lane & 15, jj = lane >> 4;

  double *ring = smem;                                   // [8][16][33]; pad slot 32 of row j: 2^(j/64)
  double *blocks = ring + kSlimRingD;                    // [2 parities]{[8 steps][13 tiles][64], [8 steps][112]}
  double *land = blocks + 2 * kSlimBlock + wave * kSlimLand;  // this wave's landing zone: [4][32] rows twice, [4][4] m16..19
  const double *exp_pad = ring + 32;

  const int64_t slot0 = (int64_t)bq * (WAVES * kSamplesPerWave) + wave * kSamplesPerWave;
  const int64_t slot = slot0 + s;
  const bool is_sample = slot < a.S;
  const bool is_null = !is_sample;  // slot == S is the null model; slots beyond it are idle copies
  const int32_t sample = is_sample ? a.perm[slot] : 0;
  const double z_dla = m.min_z_dla + (m.max_z_dla - m.min_z_dla) * a.offset_samples[sample];  // process_qsos.m:162-164
  const double nhi = a.nhi_samples[sample];
  [[maybe_unused]] double mult_r[3];
  [[maybe_unused]] const double opz = 1 + z_dla, inv_opz = 1.0 / opz;
  [[maybe_unused]] const int L = LINES > 0 ? LINES : a.num_lines;
  if constexpr (LINES == 3) {
#pragma unroll
    for (int j = 0; j < 3; ++j) mult_r[j] = g_lines.c / (g_lines.wavelength_cm[j] * (1 + z_dla)) / 1e8;  // voigt.c:278-279
  }
  if (tid < kExpTab) ring[tid * kRing2 + 32] = exp2((double)tid * (1.0 / kExpTab));
  double *my_ring = ring + (size_t)(wave * kSamplesPerWave + s) * kRing2 + jj;
  const double *lam = a.lam_pad + m.lam_off;
  const int n_pad = m.n_u + 6;
  const double nscale64 = -nhi * g_lines.inv_sqrt2pi_sigma * kInvSqrtPi * kExpScale;
  const double *rec_base = a.records + m.rec_off * (int64_t)kSlimRec;
  const int nchunks = (m.steps + CH - 1) / CH;

  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t raw_lds = __builtin_amdgcn_readfirstlane(lds_address(blocks + kSlimTileBuf));
  auto issue_chunk = [&](int c) {  // the block's copy of chunk c's raw records (7 KiB)
    glds_chunk<CH * kSlimRec / 128, WAVES>(rec_base + (size_t)c * CH * kSlimRec,
                                           raw_lds + (uint32_t)(c & 1) * (uint32_t)(kSlimBlock * 8), wave_s, lane);
  };
  // This wave's private copy of K-step `wave` of chunk c: its 4 M rows, each laid down twice
  // (lane 16 jj + p fetches doubles 2 (p & 7), 2 (p & 7) + 1 of pixel jj's row), and m[16..19] of
  // the 4 pixels (lanes 0..7).
  const double *land_src = rec_base + (size_t)wave_s * kSlimRec;
  const int src_rows = 4 * kSlimExtras + 16 * jj + 2 * (s & 7);
  const int src_x = kSlimExtras * (lane >> 1) + 4 + 2 * (lane & 1);
  auto issue_private = [&](int c) {
    const double *rec = land_src + (size_t)c * CH * kSlimRec;
    glds16(rec + src_rows, land);
    if (lane < 8) glds16(rec + src_x, land + 4 * 32);
  };
  // Expansion of this wave's K-step of the chunk that goes to tile buffer P, tiles [t0, t1): the
  // operands are requested by expand_load (early in a K-step) and multiplied and stored by
  // expand_store (late, long after they landed)
  const double *row = land + 32 * jj + s;                  // m[c] at +0, m[(c + n) & 15] at +n
  const double *bc = land + 4 * 32 + 4 * jj;               // m[16 + r] at +r
  // tile 8: lanes c < 8 multiply m[c] m[c + 8]; lanes c >= 8 the pairs (16 + a, 16 + b)
  const double *a8 = s < 8 ? row : bc + (slim_pair_i(8, s) - 16);
  const double *b8 = s < 8 ? row + 8 : bc + (slim_pair_j(8, s) - 16);
  struct Operands {
    double mc, o[2], p8;
  };
  auto expand_load = [&](int t0, int t1, Operands &x) {
    x.mc = row[0];
#pragma unroll
    for (int t = t0; t < t1; ++t) {
      if (t == 8) {
        x.o[t - t0] = a8[0];
        x.p8 = b8[0];
      } else {
        x.o[t - t0] = t == 0 ? x.mc : t < 8 ? row[t] : bc[t - 9];
      }
    }
  };
  auto expand_store = [&](double *dst, int t0, int t1, const Operands &x) {
#pragma unroll
    for (int t = t0; t < t1; ++t) dst[t * 64] = (t == 8 ? x.p8 : x.mc) * x.o[t - t0];
  };
  // per-lane LDS bases of parity 0: this wave's expansion target, the tile fragments, this lane's
  // pixel block of a raw record, the u tile of a raw record
  double *const xd0 = blocks + (size_t)wave_s * kSlimStepTiles + lane;
  const double *const tb0 = blocks + lane;
  const double *const mb0 = blocks + kSlimTileBuf + kSlimExtras * jj;
  const double *const ub0 = blocks + kSlimTileBuf + 4 * kSlimExtras + lane;
  // an address the compiler must keep in a register instead of re-deriving it in every K-step
  auto pinned = [](const double *p) {
    uint32_t v = lds_address(p);
    asm volatile("" : "+v"(v));
    return (double *)(__attribute__((address_space(3))) double *)(uintptr_t)v;
  };

  issue_private(0);
  issue_chunk(0);

  const double c_light = g_lines.c, inv_s = g_lines.inv_sqrt2_sigma;
  [[maybe_unused]] double ms_r[3];
  if constexpr (LINES == 3) {
#pragma unroll
    for (int j = 0; j < 3; ++j) ms_r[j] = mult_r[j] * inv_s;
  }
  const double cs = c_light * inv_s;
  // sqrt(pi) Sum_j lead_j Re w_j at one padded pixel: voigt.c:282-289
  auto optical_sum = [&](double lamP) -> double {
    if constexpr (LINES == 3) {
      bool near;
      double total = wing_sum3(lamP, ms_r[0], ms_r[1], ms_r[2], cs, &near);
      if (__builtin_expect(__any(near), 0)) total = total_near<3>(lamP, mult_r[0], mult_r[1], mult_r[2], nullptr, 3);
      return total;
    } else {
      bool near;
      double total = wing_sum_runtime(lamP * inv_opz, cs, L, &near);
      if (__builtin_expect(__any(near), 0)) total = total_near_at(lamP, opz, L);
      return total;
    }
  };

  __syncthreads();  // the exp table visible
  // prime the ring with padded pixels 0..11 (the raw profile runs three K-steps ahead)
  for (int c3 = 0; c3 < 3; ++c3) {
    const double lam0 = lam[min(4 * c3 + jj, n_pad - 1)];
    const double tot = optical_sum(lam0);
    const ExpState es0 = exp_ring_begin_scaled(nscale64 * tot, exp_pad);
    const double v = exp_table_end_scaled(es0);
    my_ring[4 * c3] = v;
    my_ring[4 * c3 + 16] = v;
  }

  d4 acc[14];
#pragma unroll
  for (int c = 0; c < 14; ++c) acc[c] = d4{0, 0, 0, 0};
  double quad_sum = 0.0, dprod = 1.0;
  double xw[kXW] = {0.0, 0.0}, xu[kXU] = {0.0, 0.0, 0.0, 0.0};
  int dexp = 0;
  const double tap0 = g_lines.taps[0], tap1 = g_lines.taps[1], tap2 = g_lines.taps[2], tap3 = g_lines.taps[3];

  glds_wait();  // this wave's rows of chunk 0 (and its share of the raw chunk) landed
  {
    const double mc = row[0];
    for (int t = 0; t < kSlimTilesW; ++t)
      xd0[t * 64] = (t == 8 ? b8[0] : mc) * (t == 8 ? a8[0] : t == 0 ? mc : t < 8 ? row[t] : bc[t - 9]);
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the landing zone has been read ...
  if (nchunks > 1) issue_private(1);   // ... and may be refilled
  glds_wait();
  __syncthreads();

  // One chunk of 8 K-steps per iteration, from parity c & 1.  While it runs, this wave expands its
  // K-step of chunk c + 1 into the other parity: two tiles in each of K-steps 0..5, one in K-step 6
  // (operands requested before the MFMA burst, multiplied and stored behind it; unconditionally:
  // after the last chunk the products of stale rows land in a buffer nobody reads), and in K-step
  // 7 refills its landing zone for chunk c + 2 -- every read of it has been consumed by a multiply
  // by then, and the 1-KiB copy lands during that K-step's burst, before the chunk's closing
  // barrier.  Measured against the alternatives on one box (tools/ab.sh, ms per launch): this
  // 149.6; three tiles per K-step over five K-steps 150.1; products stored before the burst 152.0;
  // without the explicit lgkmcnt(0) at the top of a K-step 153.1; pre-expanded records (k_sweep) 149.5.
  for (int c = 0; c < nchunks; ++c) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // (nothing of ours is in flight: see k_sweep)
    if (c + 1 < nchunks) issue_chunk(c + 1);
    const int par = (c & 1) * kSlimBlock;
    const double *tbuf = pinned(tb0 + par);
    const double *mine0 = pinned(mb0 + par);
    const double *ubuf = pinned(ub0 + par);
    double *xdst = pinned(xd0 + (kSlimBlock - par));
    double lam_next = 0.0;
#pragma unroll
    for (int tt = 0; tt < CH; ++tt) {
      const int rn = c * CH + tt;
      constexpr int kXS = 7;  // K-steps that carry expansion work
      constexpr int kT0[7] = {0, 2, 4, 6, 8, 10, 12}, kT1[7] = {2, 4, 6, 8, 10, 12, 13};
      if (tt == kXS && c + 2 < nchunks) issue_private(c + 2);
      if (rn < m.steps) {
        const double *tl = tbuf + (size_t)tt * kSlimStepTiles;
        const double *mine = mine0 + (size_t)tt * kSlimRec;
        const int slot_p = (4 * tt) & 15;
        const int slot_w = (slot_p + 12) & 15;
        if (tt > 0) __builtin_amdgcn_s_waitcnt(0xC07F);  // lam_next (requested before the last burst) is here
        const double lamP = tt == 0 ? mine[10] : lam_next;
        const double *g = my_ring + slot_p;
        const double g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3], g4 = g[4], g5 = g[5], g6 = g[6];
        const double2 p01 = *reinterpret_cast<const double2 *>(mine);
        const double2 p23 = *reinterpret_cast<const double2 *>(mine + 2);
        const double py = p01.x, pmu = p01.y, pom = p23.x, pnu = p23.y;
        __builtin_amdgcn_sched_barrier(0);
        // (2) instrument broadening for pixel 4 rn + jj: voigt.c:297-299 (symmetric taps).  In front of (1)
        // since round 5: the three-line wing tier takes this pixel's d along and returns 1/d from the same
        // v_rcp_f64 as its own three quotients (wing_sum3_rcp4): 151.06 -> 150.21 ms on one box
        // (profiles/r05_ab_rcp4.txt), and the two spilled registers are gone.
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
        if constexpr (LINES == 3) {
          bool near;
          total = wing_sum3_rcp4(lamP, ms_r[0], ms_r[1], ms_r[2], cs, &near, d, &inv_d);
#ifndef SLIM_EXP_NONEAR  // (ablation, results wrong by construction: what the accurate tier costs)
          if (__builtin_expect(__any(near), 0)) total = total_near<3>(lamP, mult_r[0], mult_r[1], mult_r[2], nullptr, 3);
#endif
        } else {
          total = optical_sum(lamP);
          inv_d = fast_rcp(d);
        }
        const ExpState es = exp_ring_begin_scaled(nscale64 * total, exp_pad);
        __builtin_amdgcn_sched_barrier(0);
        double bop[14];
#pragma unroll
        for (int cc = 0; cc < kSlimTilesW; ++cc) bop[cc] = tl[cc * 64];
        bop[13] = ubuf[(size_t)tt * kSlimRec];  // m[0..15] of the 4 pixels in lane order: the u tile
        __builtin_amdgcn_sched_barrier(0);
        const double raw = exp_table_end_scaled(es);
        my_ring[slot_w] = raw;
        my_ring[slot_w + 16] = raw;
        // (3) weights: process_qsos.m:192-198 folded into log_mvnpdf_low_rank.m:11-15
        const double r = fma(-absorb, pmu, py);
        const double w = a2 * inv_d;
        const double ri = r * inv_d;
        const double u = absorb * ri;
        quad_sum = fma(r, ri, quad_sum);
        dprod *= d;
        if (tt & 1) {
          dexp += __builtin_amdgcn_frexp_exp(dprod);
          dprod = __builtin_amdgcn_frexp_mant(dprod);
        }
        Operands x;
        if (tt < kXS) expand_load(kT0[tt < kXS ? tt : 0], kT1[tt < kXS ? tt : 0], x);
        if (tt + 1 < CH) lam_next = mine[kSlimRec + 10];
        __builtin_amdgcn_sched_barrier(0);
        // (4) rank-4 update of [B | v] on the matrix cores
#pragma unroll
        for (int cc = 0; cc < 14; ++cc)
          acc[cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(cc < kSlimTilesW ? w : u, bop[cc], acc[cc], 0, 0, 0);
        {  // vech columns 208, 209 and m columns 16..19 of this lane's pixel: 6 FMAs
          const double2 xp = *reinterpret_cast<const double2 *>(mine + 8);
          const double2 u01 = *reinterpret_cast<const double2 *>(mine + 4);
          const double2 u23 = *reinterpret_cast<const double2 *>(mine + 6);
          xw[0] = fma(w, xp.x, xw[0]);
          xw[1] = fma(w, xp.y, xw[1]);
          xu[0] = fma(u, u01.x, xu[0]);
          xu[1] = fma(u, u01.y, xu[1]);
          xu[2] = fma(u, u23.x, xu[2]);
          xu[3] = fma(u, u23.y, xu[3]);
        }
        if (tt < kXS) expand_store(xdst, kT0[tt < kXS ? tt : 0], kT1[tt < kXS ? tt : 0], x);
      }
    }
    glds_wait();      // the prefetched raw chunk and this wave's next rows have landed ...
    __syncthreads();  // ... everyone's tiles of the next chunk are written; this chunk's buffers are free
  }

  double logd_sum = log(dprod) + (double)dexp * 0.6931471805599453;
  quad_sum += __shfl_xor(quad_sum, 16);
  quad_sum += __shfl_xor(quad_sum, 32);
  logd_sum += __shfl_xor(logd_sum, 16);
  logd_sum += __shfl_xor(logd_sum, 32);
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

  using ES = EpilogueShape<13, 1>;
  double *Eg = smem + (size_t)wave * ES::SPP * ES::stride(16);
#pragma unroll
  for (int p = 0; p < ES::PASSES; ++p) {
    int sigma;
    bool writer;
    const double ll = slim_factor_pass(acc, xw, xu, p, Eg, lane, a.k, quad_sum, logd_sum, m.n_kept, &sigma, &writer);
    const int64_t slot_s = slot0 + sigma;
    const int32_t sample_s = __shfl(sample, sigma + 16 * jj);
    if (writer) {
      if (slot_s < a.S) a.sample_ll[(int64_t)q * a.S + sample_s] = ll + m.ll_bias;
      else if (slot_s == a.S) a.ll_no_dla[q] = ll + m.ll_bias;
    }
  }
}

}  // namespace gpdla
