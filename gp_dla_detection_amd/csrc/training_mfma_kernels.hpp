// training_mfma_kernels.hpp -- the training objective (objective.m:12-75 over spectrum_loss.m:14-76)
// on the fp64 matrix cores, value and gradient, deterministic (no atomics).
//
// All training quasars share the rest-frame pixel grid and M, so the per-quasar Woodbury pieces of
// spectrum_loss.m are three dense contractions over the whole training set -- the sweep's
// [W | U] . [P | M] with "samples" replaced by quasars:
//
//   d = nu + omega2 (1 - exp(-tau0 (1+z)^beta) + c0)^2,   w = 1/d,   u = y/d          (:22-32)
//   B_q = I + Sum_p w_qp m_p m_p',   t_q = Sum_p u_qp m_p      contraction over pixels  (:40)
//   L_q = chol(B_q),  z_q = B_q^-1 t_q,  T_q = B_q^-1 + z_q z_q'   per quasar, k x k    (:42-46)
//   -log p_q = 1/2 (Sum y^2 w - t_q'z_q + Sum log d + 2 Sum log L_jj + n log 2 pi)      (:48-52)
//   dM[p,:] = m_p' (Sum_q w_qp T_q) - Sum_q u_qp z_q'          contraction over quasars (:55-56)
//   core_qp = (K^-1 y)_p^2 - (K^-1)_pp = u^2 - 2 u w (m_p'z_q) + w^2 (m_p'T_q m_p) - w  (:59)
//   dlog_omega[p] = -Sum_q an_qp core_qp, dlog_c0 / tau0 / beta = -Sum_qp core_qp da_qp  (:62-74)
//
// using K^-1 M = D^-1 M B^-1, M'K^-1 y = z, y'K^-1 y = Sum y^2 w - t'z (identities of the Woodbury
// form; the as-written k x n matrix C of :44 is never formed).  m'T m is the dot product of
// vech(m m') with vech(T) (off-diagonals doubled), i.e. the third contraction -- over the
// k(k+1)/2 + k columns -- and every contraction runs on v_mfma_f64_16x16x4_f64.
//
// Kernels (k <= 20: 14 w-tiles + 2 u-tiles of 16 columns; k <= 40: four such tile groups):
//   k_train_records   [vech(m m') | m] from M in the two B-operand tilings; omega2 = exp(2 log omega)
//   k_train_build     first contraction, rows = quasars, steps over pixels, with its A operand made on
//                     the fly: w, u from (flux, log(1+z), noise) in registers, never stored; the
//                     per-quasar sums of log d (as a running product), y^2 w and the pixel count ride
//                     along; split along the pixel axis into partial sums
//   k_train_factor    per quasar: partials added in split order, Cholesky (64 / KMAX quasars per wave,
//                     lanes along the rows), B^-1, z, -log p; T_q and z_q in both operand tilings
//   k_train_core      m'z and m'Tm by MFMA, then the element-wise gradient terms and their sums; its
//                     result registers hold w and u in the A-operand lane order of the dM contraction,
//                     which it writes out as whole rows (wB, uB)
//   k_train_contract  rows x steps MFMA contraction for dM (rows = pixels, steps over quasars), split
//                     along the step axis into partial sums that the next kernel adds in a fixed order
//   k_train_finish    ordered sums of all partials into f and g
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sweep_kernels.hpp"

namespace gpdla {

// Rank classes.  A contraction step carries 16-tile GROUPS of B tiles (one group per wave: 16
// accumulator tiles = 128 registers): k <= 20 has one group -- 14 vech tiles (224 columns >= 210) and
// 2 projection tiles; k <= 40 has four -- 52 vech tiles (832 >= 820), 3 projection tiles, 9 of
// padding.  Columns of [vech | v]: vech at 0 .. 16 W, v at 16 W + c.  The core contraction walks the
// same columns four at a time: KsW steps of vech, KsU steps of v (padded to an even count so that a
// quasar group's operands are a whole number of KiB).
template <int KMAX> struct TrK;
template <> struct TrK<20> {
  static constexpr int W = 14, U = 2, Groups = 1, KsW = 53, KsU = 5;
};
template <> struct TrK<40> {
  static constexpr int W = 52, U = 3, Groups = 4, KsW = 205, KsU = 11;
};
template <int KMAX> struct TrC : TrK<KMAX> {
  static constexpr int Tiles = 16 * TrK<KMAX>::Groups;          // B tiles per contraction step
  static constexpr int Cols = 16 * Tiles;                       // columns of a contraction's output row
  static constexpr int Ks = TrK<KMAX>::KsW + TrK<KMAX>::KsU;    // column steps of the core contraction
  static_assert(Ks % 2 == 0, "a quasar group's operands are a whole number of KiB");
  static_assert(16 * TrK<KMAX>::W >= KMAX * (KMAX + 1) / 2 && 16 * TrK<KMAX>::U >= KMAX, "columns fit");
  static_assert(4 * TrK<KMAX>::KsW >= KMAX * (KMAX + 1) / 2 && 4 * TrK<KMAX>::KsU >= KMAX, "column steps fit");
};
constexpr int kTrGroupD = 16 * 64;  // doubles of one tile group of one step

struct TrainDims {
  int64_t nq, G;      // quasars, pixels
  int32_t k;
  int64_t NQ16, PG;   // row groups of 16: quasars, pixels
  int64_t T, TQ;      // contraction steps of 4: pixels (4 PG), quasars (4 NQ16)
  int64_t ld;         // row stride of flux / log(1+z) / noise: 16 PG (rows padded with missing pixels)
  int32_t H, H2, GS;  // splits: B build over pixels, dM over quasars, core over quasar groups
};

__host__ __device__ inline void vech_ij(int c, int *i, int *j) {
  int ii = (int)((sqrt(8.0 * c + 1.0) - 1.0) * 0.5);
  while ((ii + 1) * (ii + 2) / 2 <= c) ++ii;
  while (ii * (ii + 1) / 2 > c) --ii;
  *i = ii;
  *j = c - ii * (ii + 1) / 2;
}

// ------------------------------------------------------------------------------------------
// The scalars of the parameter vector x = [M (G k) | log omega (G) | log c0, log tau0, log beta]
// (objective.m:29-32) are exponentiated where they are used: no kernel argument changes between
// evaluations, and no separate kernel stands in front of the first contraction.
// ------------------------------------------------------------------------------------------
struct TrainScal {
  double c_0, tau_0, beta;
};
__device__ __forceinline__ TrainScal train_scal(const double *x, int64_t G, int k) {
  const double *s = x + G * (k + 1);
  return TrainScal{exp(s[0]), exp(s[1]), exp(s[2])};
}
__device__ __forceinline__ double train_omega2(const double *x, int64_t G, int k, int64_t p) {
  return exp(2 * x[G * k + p]);  // objective.m:29
}

// The mean-flux model's objective (multi_dlas/objective_lyseries.m over spectrum_loss_lyseries.m)
// differs from objective.m / spectrum_loss.m in ONE line: the optical depth of a pixel sums
// num_forest_lines Lyman lines, tau_l (lambda_1 (1+z) / lambda_l)^beta with tau_l = tau_0 lambda_l f_l /
// (lambda_1 f_1), each counted only where its redshift does not exceed the quasar's
// (spectrum_loss_lyseries.m:22-39).  Lines are sorted by decreasing wavelength, so the active lines
// of a pixel are a prefix 1..n, n = nl[q][p] (k_train_lines, the as-written comparison), and
//   optical depth = (1+z)^beta * T[n],  T[n] = tau_0 Sum_{l <= n} coef_l exp(beta logr_l)
// -- a table of nfl + 1 numbers per evaluation (T[1] = tau_0: the plain objective is nfl = 1).
constexpr int kTrMaxLines = 31;
struct TrainLines {
  int32_t nfl;             // 0 / 1: spectrum_loss.m; > 1: spectrum_loss_lyseries.m
  double coef[kTrMaxLines];  // lambda_l f_l / (lambda_1 f_1)
  double logr[kTrMaxLines];  // log(lambda_1 / lambda_l)
};
// one thread fills T[0 .. nfl] (LDS); the caller's next barrier publishes it
__device__ __forceinline__ void train_line_table(const TrainLines &L, double tau_0, double beta, double *T) {
  double acc = 0.0;
  T[0] = 0.0;
  for (int l = 0; l < L.nfl; ++l) {
    acc += tau_0 * L.coef[l] * exp(beta * L.logr[l]);
    T[l + 1] = acc;
  }
}

// k_train_lines: nl[q][p] = number of active lines of pixel p of quasar q, by the comparison of
// spectrum_loss_lyseries.m:28-31 as written (zqso_1pz = the quasar's last lya_1pz,
// objective_lyseries.m:46); *not_prefix is raised if the active lines are not 1..n somewhere.
struct TrainLinesArgs {
  int64_t nq, G, ld;
  int32_t nfl;
  double wl[kTrMaxLines];
  const double *lya_1pz;  // [nq][ld]
  uint8_t *nl;            // [nq][ld]
  int32_t *not_prefix;
};
__global__ void k_train_lines(TrainLinesArgs a) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.nq * a.ld) return;
  const int64_t q = e / a.ld, p = e - q * a.ld;
  int n = 1;
  if (p < a.G) {
    const double zqso_1pz = a.lya_1pz[q * a.ld + a.G - 1], lya = a.lya_1pz[e];
    bool open = true;
    for (int l = 1; l < a.nfl; ++l) {
      const double lyman_1pz = a.wl[0] * lya / a.wl[l];
      const bool on = lyman_1pz <= zqso_1pz;
      if (on && !open) *a.not_prefix = 1;
      if (on && open) n = l + 1;
      open = open && on;
    }
  }
  a.nl[e] = (uint8_t)n;
}

// (log(1+z) is data: it is taken once, when the training set is uploaded, so that the power of
// spectrum_loss.m:22 costs one exp per evaluation instead of a pow)
// (tau_0: the pixel's line scale T[n] -- tau_0 itself for the plain objective; beta64 = beta * 64 / ln 2:
// both exponentials go through the sweep's 64-entry table (exp_table_scaled, relative error < 2e-16),
// 11 instructions each where a series in full costs 20, and the power needs no reciprocal)
__device__ __forceinline__ void train_element(double y, double logz1, double nu, double om, double c_0,
                                              double tau_0, double beta64, const double *exp_tab, double *w,
                                              double *u, double *d_out) {
  const double od = tau_0 * exp_table_scaled(beta64 * logz1, exp_tab);  // spectrum_loss.m:22: tau0 (1+z)^beta
  const double sf = 1 - exp_table_scaled(-kExpScale * od, exp_tab) + c_0;  // :23, :26
  const double d = nu + om * (sf * sf);         // :27, :29
  *w = fast_rcp(d);                             // :31
  *u = *w * y;                                // :32
  *d_out = d;
}

// ------------------------------------------------------------------------------------------
// k_train_records: from M (G x k column-major)
// recM: [group][T + pad][16 tiles][jj][col]   B[pixel][column] of [vech(m m') | m]; step t = 4 c + e of
//       chunk c contracts the pixels 16 c + 4 jj + e, jj = 0..3 (k_train_build: lane (jj, s) loads the
//       four CONSECUTIVE pixels 16 c + 4 jj .. + 3 of quasar s and feeds them to the chunk's steps
//       e = 0..3 -- a sum over pixels does not care in which order they are taken)
// recP: [PG][Ks column steps][jj][col]                     B[column 4ks+jj][pixel 16pt+col], vech then m
// ------------------------------------------------------------------------------------------
struct TrainRecordsArgs {
  TrainDims d;
  const double *M;
  double *recM, *recP;
  int64_t group_stride;  // doubles between the tile groups of recM: (T + chunk padding) * 16 * 64
  int32_t *not_pd;       // cleared here, raised by k_train_factor
  double *omega2;        // [16 PG] exp(2 log omega) (objective.m:29), zero behind the last pixel
};

__device__ __forceinline__ double train_col_value(const double *M, int64_t G, int k, int64_t p, int kind, int idx) {
  // kind 0: vech column idx of m_p m_p'; kind 1: m_p[idx]
  if (p >= G) return 0.0;
  if (kind == 0) {
    if (idx >= k * (k + 1) / 2) return 0.0;
    int i, j;
    vech_ij(idx, &i, &j);
    return M[p + (int64_t)i * G] * M[p + (int64_t)j * G];
  }
  return idx < k ? M[p + (int64_t)idx * G] : 0.0;
}

template <int KMAX>
__global__ void k_train_records(TrainRecordsArgs a) {
  using K = TrC<KMAX>;
  const TrainDims &D = a.d;
  const int64_t nM = D.T * K::Tiles * 64, nP = D.PG * K::Ks * 64;
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.not_pd = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nM + nP; e += (int64_t)gridDim.x * blockDim.x) {
    if (e < 16 * D.PG) a.omega2[e] = e < D.G ? train_omega2(a.M, D.G, D.k, e) : 0.0;
    if (e < nM) {
      const int l = (int)(e & 63), c = (int)((e >> 6) % K::Tiles);
      const int64_t t = (e >> 6) / K::Tiles;
      const int jj = l >> 4, col = l & 15;
      const int64_t px = 16 * (t >> 2) + 4 * jj + (t & 3);
      const double v = c < K::W ? train_col_value(a.M, D.G, D.k, px, 0, 16 * c + col)
                                : train_col_value(a.M, D.G, D.k, px, 1, 16 * (c - K::W) + col);
      a.recM[(c >> 4) * a.group_stride + (t * 16 + (c & 15)) * 64 + l] = v;
    } else {
      const int64_t e2 = e - nM;
      const int l = (int)(e2 & 63), ks = (int)((e2 >> 6) % K::Ks);
      const int64_t pt = (e2 >> 6) / K::Ks;
      const int jj = l >> 4, col = l & 15;
      a.recP[e2] = ks < K::KsW ? train_col_value(a.M, D.G, D.k, 16 * pt + col, 0, 4 * ks + jj)
                               : train_col_value(a.M, D.G, D.k, 16 * pt + col, 1, 4 * (ks - K::KsW) + jj);
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_train_contract: out[R][h][16 rows][cols] = Sum over this split's steps of
// [a_w x W tiles | a_u x U tiles] (rows x 4) . Brec (4 x cols).  A wave owns 16 rows and ONE tile
// group (16 accumulator tiles); a block is 4 waves = 4 row groups that walk the SAME steps (split
// h) of the SAME tile group, so that group's B records are staged once per block: chunks of 4
// steps, double-buffered in LDS by the sweep's asynchronous global->LDS copy, one barrier per
// chunk.  64 KiB of LDS per block: two blocks share a CU, so a SIMD always has a wave of the other
// block to run while one waits at its barrier.  The A operands (512 contiguous bytes per wave and
// step) come straight from global memory, one chunk ahead.
// ------------------------------------------------------------------------------------------
struct TrainContractArgs {
  const double *Aw, *Au;   // [R][steps][64]
  const double *Brec;      // [groups][steps + pad][16][64]
  int64_t R, steps;
  int32_t nsplit;
  int32_t groups, w_tiles, cols;  // tile groups; tiles (over all groups) that take a_w; output row length
  int64_t group_stride;    // doubles between the tile groups of Brec
  double *out;             // [R][nsplit][16][cols]
};
constexpr int kTrChunk = 4;                             // steps per staged chunk
constexpr int kTrCWaves = 4;                            // row groups (waves) per block
constexpr size_t kTrContractLds = 2 * kTrChunk * kTrGroupD * sizeof(double);  // 64 KiB
constexpr int kTrBuildMaxChunks = 64;  // chunks of one split of k_train_build (its omega2 table: 8 KiB)
constexpr size_t kTrBuildLds = kTrContractLds + (16 * kTrBuildMaxChunks + kTrMaxLines + 1 + 64) * sizeof(double);  // + the line table + the exp table

// NW: tiles of the block's group that take a_w (compile-time: the A operand of every MFMA is then
// a fixed register, not a select)
// NT: tiles of the group that exist at all -- the last group of the k <= 40 class holds 4 w-tiles and 3
// u-tiles, its other 9 tiles are padding of zeros: no MFMA, no LDS read, no store for them (14 % of a
// k <= 40 contraction's matrix work; the padded columns of the partial sums are zeroed once, when the
// workspace is made, and never written).
template <int NW>
constexpr int train_group_tiles() {
  static_assert(TrK<40>::W == 52 && TrK<40>::U == 3 && TrK<20>::W + TrK<20>::U == 16, "tile counts of the two rank classes");
#ifdef TR_EXP_ALLTILES  // (A/B: the rounds-3/4 form, MFMAs on the padding tiles too)
  return 16;
#else
  return NW == 4 ? 4 + TrK<40>::U : 16;
#endif
}
template <int NW>
__device__ __forceinline__ void train_contract_body(const TrainContractArgs &a, double *smem, int tg) {
  constexpr int NT = train_group_tiles<NW>();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t bx = blockIdx.x % (gridDim.x / a.groups);  // (the tile group is the SLOWEST block index: train_tile_group)
  const int64_t rb = bx / a.nsplit;
  const int h = (int)(bx % a.nsplit);
  const int64_t r = rb * kTrCWaves + wave;
  const bool active = r < a.R;
  constexpr int nw = NW;
  const int64_t t0 = (a.steps * h) / a.nsplit, t1 = (a.steps * (h + 1)) / a.nsplit;  // balanced split
  const int nchunks = (int)((t1 - t0 + kTrChunk - 1) / kTrChunk);
  d4 acc[NT];
#pragma unroll
  for (int c = 0; c < NT; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
  const double *aw = a.Aw + ((active ? r : 0) * a.steps) * 64 + lane, *au = a.Au + ((active ? r : 0) * a.steps) * 64 + lane;
  const double *brec = a.Brec + tg * a.group_stride;
  // whole chunks, copied as per-wave spans (glds_chunk, sweep_kernels.hpp); a chunk that runs past
  // t1 reads the following steps' records or the kTrChunk records of padding behind the group
  static_assert((kTrChunk * kTrGroupD) % 128 == 0, "a chunk is a whole number of KiB");
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t smem_lds = __builtin_amdgcn_readfirstlane(lds_address(smem));
  auto issue_chunk = [&](int c) {
    glds_chunk<kTrChunk * kTrGroupD / 128, kTrCWaves>(brec + (t0 + (int64_t)c * kTrChunk) * kTrGroupD,
                                                      smem_lds + (uint32_t)(c & 1) * (uint32_t)(kTrChunk * kTrGroupD * 8), wave_s, lane);
  };
  double wn[kTrChunk], un[kTrChunk];
  auto load_a = [&](int c) {
#pragma unroll
    for (int tt = 0; tt < kTrChunk; ++tt) {
      const int64_t t = min(t0 + (int64_t)c * kTrChunk + tt, a.steps - 1);  // clamped: unused beyond t1
      wn[tt] = aw[t * 64];
      un[tt] = au[t * 64];
    }
  };
  if (nchunks > 0) {
    load_a(0);
    issue_chunk(0);
  }
  for (int c = 0; c < nchunks; ++c) {
    double wc[kTrChunk], uc[kTrChunk];
#pragma unroll
    for (int tt = 0; tt < kTrChunk; ++tt) {
      wc[tt] = wn[tt];
      uc[tt] = un[tt];
    }
    glds_wait();      // chunk c landed (this wave's part) ...
#ifndef TR_EXP_NOBAR
    __syncthreads();  // ... and everyone's; the other buffer's readers are done
#endif
    if (c + 1 < nchunks) {
      load_a(c + 1);
      issue_chunk(c + 1);
    }
    const double *buf = smem + (size_t)(c & 1) * kTrChunk * kTrGroupD + lane;
    const int csteps = (int)min((int64_t)kTrChunk, t1 - (t0 + (int64_t)c * kTrChunk));
#pragma unroll
    for (int tt = 0; tt < kTrChunk; ++tt) {
      if (tt < csteps) {
        double b[NT];
#pragma unroll
#ifdef TR_EXP_NOLDS
        for (int cc = 0; cc < NT; ++cc) b[cc] = wc[tt] + cc;
#else
        for (int cc = 0; cc < NT; ++cc) b[cc] = buf[(size_t)(tt * 16 + cc) * 64];
#endif
#pragma unroll
        for (int cc = 0; cc < NT; ++cc)
          acc[cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(cc < nw ? wc[tt] : uc[tt], b[cc], acc[cc], 0, 0, 0);
      }
    }
  }
  glds_wait();  // nothing is in flight here (the last chunk issued no copy); says so to tools/check_vmem_hazard.py
  if (!active) return;
#ifdef TR_EXP_NOSTORE
  if (acc[0][0] == acc[0][0]) return;
#endif
  // result register rr of tile c: row (lane >> 4) + 4 rr, column 16 (16 tg + c) + (lane & 15)
  double *o = a.out + ((r * a.nsplit + h) * 16) * (int64_t)a.cols + 256 * tg;
  const int jj = lane >> 4, s = lane & 15;
#pragma unroll
  for (int c = 0; c < NT; ++c)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) o[(int64_t)(jj + 4 * rr) * a.cols + 16 * c + s] = acc[c][rr];
}

// The tile group of a block is its SLOWEST index (blocks [g n, (g + 1) n) of the grid work on group g).
// Blocks are dealt round-robin over the 8 XCDs, so with the group as the fastest index (rounds 3-4)
// the blocks of group 3 -- 7 tiles instead of 16 since round 5 -- all sat on XCDs 3 and 7 and the
// other six XCDs ran only full blocks: skipping the padding tiles saved nothing (measured: 314 -> 313 us).
// Slowest, the short blocks are the grid's tail.
__device__ __forceinline__ int train_tile_group(int groups) { return (int)(blockIdx.x / (gridDim.x / groups)); }

__global__ __launch_bounds__(kTrCWaves * 64) void k_train_contract(TrainContractArgs a) {
  extern __shared__ double smem[];
  const int tg = train_tile_group(a.groups);
  const int nw = max(0, min(16, a.w_tiles - 16 * tg));  // block-uniform: 14 (k <= 20); 16, 16, 16, 4 (k <= 40)
  if (nw == 16) train_contract_body<16>(a, smem, tg);
  else if (nw == 14) train_contract_body<14>(a, smem, tg);
  else train_contract_body<4>(a, smem, tg);
}

// ------------------------------------------------------------------------------------------
// k_train_build: the first contraction with its A operand made on the fly -- [vech(B_q - I) | t_q]
// = Sum_p [w_qp x W tiles | u_qp x U tiles] . recM.  Same block shape as k_train_contract (4 waves =
// 4 quasar groups walking the same chunks of the same tile group; the records staged once per block,
// double-buffered by the asynchronous copy), but lane (jj, s) loads flux / log(1+z) / noise of the
// four consecutive pixels 16 c + 4 jj .. + 3 of quasar 16 r + s (32 contiguous bytes per array,
// one chunk ahead), turns them into w and u in registers, and step e of the chunk takes element e:
// no w / u array is written or read for this contraction.  The per-quasar sums of spectrum_loss.m
// :48-52 ride along: Sum y^2 w, the pixel count, and Sum log d as log of the running product of
// the d (mantissa and exponent kept apart: one multiply per pixel instead of one log).
// part1: [16 NQ16][H][3] = (Sum log d, Sum y^2 w, count) of the split's pixels
// ------------------------------------------------------------------------------------------
struct TrainBuildArgs {
  TrainDims d;
  const double *flux, *log_lya_1pz, *noise;  // [nq][ld], NaN flux = missing pixel (objective.m:42)
  const double *omega2;                      // [16 PG]
  const double *x;                           // the parameter vector
  const uint8_t *nl;                         // [nq][ld] active lines per pixel (lines.nfl > 1 only)
  TrainLines lines;
  const double *Brec;                        // recM
  int32_t groups, w_tiles, cols;
  int64_t group_stride;
  double *out;    // partB [NQ16][H][16][cols]
  double *part1;
};

// LY: the Lyman-series objective (lines.nfl > 1), a compile-time switch so that the plain objective pays nothing for it
template <int NW, bool LY>
__device__ __forceinline__ void train_build_body(const TrainBuildArgs &a, double *smem, int tg) {
  constexpr int NT = train_group_tiles<NW>();  // (the last group of the k <= 40 class: 7 tiles, see train_contract_body)
  const TrainDims &D = a.d;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int jj = lane >> 4, s = lane & 15;
  const int64_t bx = blockIdx.x % (gridDim.x / a.groups);  // (the tile group is the slowest block index: train_tile_group)
  const int64_t rb = bx / D.H;
  const int h = (int)(bx % D.H);
  const int64_t r = rb * kTrCWaves + wave;
  const int64_t q = r * 16 + s;
  const bool active = r < D.NQ16, qreal = q < D.nq;
  const int64_t c0 = (D.PG * h) / D.H, c1 = (D.PG * (h + 1)) / D.H;  // chunks of 16 pixels, balanced split
  const int nchunks = (int)(c1 - c0);
  const TrainScal sc = train_scal(a.x, D.G, D.k);
  d4 acc[NT];
#pragma unroll
  for (int c = 0; c < NT; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
  const int64_t row = (qreal ? q : 0) * D.ld + 4 * jj + 16 * c0;
  const double2 *pf = reinterpret_cast<const double2 *>(a.flux + row), *pz = reinterpret_cast<const double2 *>(a.log_lya_1pz + row),
                *pn = reinterpret_cast<const double2 *>(a.noise + row);
  const double *brec = a.Brec + tg * a.group_stride;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t smem_lds = __builtin_amdgcn_readfirstlane(lds_address(smem));
  auto issue_chunk = [&](int c) {
    glds_chunk<kTrChunk * kTrGroupD / 128, kTrCWaves>(brec + (c0 + c) * kTrChunk * kTrGroupD,
                                                      smem_lds + (uint32_t)(c & 1) * (uint32_t)(kTrChunk * kTrGroupD * 8), wave_s, lane);
  };
  // omega2 of the split's pixels: once per block, behind the two record buffers; then the line table
  double *om_s = smem + 2 * kTrChunk * kTrGroupD;
  for (int e = threadIdx.x; e < 16 * nchunks; e += kTrCWaves * 64) om_s[e] = a.omega2[16 * c0 + e];
  constexpr bool lyseries = LY;
  double *T_s = om_s + 16 * kTrBuildMaxChunks;
  double *exp_tab = T_s + kTrMaxLines + 1;  // 2^(j/64)
  if (threadIdx.x < kExpTab) exp_tab[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / kExpTab));
  const double beta64 = sc.beta * kExpScale;
  if (lyseries && threadIdx.x == 0) train_line_table(a.lines, sc.tau_0, sc.beta, T_s);
  const uint32_t *pl = reinterpret_cast<const uint32_t *>(a.nl + (lyseries ? row : 0));
  // The raw elements are NOT double-buffered (the registers are needed for two waves per SIMD):
  // the first two pixels of the next chunk are requested when this chunk's second element has been
  // used, the last two after its fourth -- two to three K-steps ahead of their use.
  double2 f01, z01, n01, f23, z23, n23;
  uint32_t nl4 = 0x01010101u;  // the four pixels' active-line counts
  auto load01 = [&](int c) {
    f01 = pf[(int64_t)c * 8];
    z01 = pz[(int64_t)c * 8];
    n01 = pn[(int64_t)c * 8];
    if (lyseries) nl4 = pl[(int64_t)c * 4];
  };
  auto load23 = [&](int c) {
    f23 = pf[(int64_t)c * 8 + 1];
    z23 = pz[(int64_t)c * 8 + 1];
    n23 = pn[(int64_t)c * 8 + 1];
  };
  if (nchunks > 0) {
    issue_chunk(0);
    load01(0);
    load23(0);
  }
  double mant = 1.0, yy = 0.0, cnt = 0.0;
  int esum = 0;
  auto element = [&](double y, double lz, double nv, double om, uint32_t n_lines, double &w, double &u) {
    const bool ok = qreal && !isnan(y);  // rows behind the last quasar and padded / missing pixels
    double d;
    train_element(y, lz, nv, om, sc.c_0, lyseries ? T_s[n_lines] : sc.tau_0, beta64, exp_tab, &w, &u, &d);
    w = ok ? w : 0.0;
    u = ok ? u : 0.0;
    yy += ok ? y * u : 0.0;
    cnt += ok ? 1.0 : 0.0;
    int ex;
    const double m = frexp(ok ? (d > 0.0 ? d : NAN) : 1.0, &ex);  // a non-positive variance poisons the sum (reported as not positive definite)
    mant *= m;
    esum += ex;
  };
  for (int c = 0; c < nchunks; ++c) {
#ifdef TR_EXP_BUILD_EARLYCOPY  // (A/B: rounds 3-4, the next chunk's copy issued at the top of the chunk)
    // chunk c landed (this wave's part): its copy is older than the (at most six) raw loads in flight
    __builtin_amdgcn_s_waitcnt(0x0F76);  // vmcnt(6)
    asm volatile("" ::: "memory");
    __syncthreads();  // ... and everyone's; the other buffer's readers are done
    if (c + 1 < nchunks) issue_chunk(c + 1);
#else
    // chunk c landed (this wave's part): its copy was issued in the MIDDLE of the previous chunk, behind
    // that chunk's requests for this chunk's first two pixels and in front of those for the last two
    // (three loads): only those may still be in flight
    __builtin_amdgcn_s_waitcnt(0x0F73);  // vmcnt(3)
    asm volatile("" ::: "memory");
    __syncthreads();  // ... and everyone's; the other buffer's readers are done
#endif
    const double *buf = smem + (size_t)(c & 1) * kTrChunk * kTrGroupD + lane;
    const double *omc = om_s + 16 * c + 4 * jj;
    const bool more = c + 1 < nchunks;
    const uint32_t nl_cur = nl4;  // (the next chunk's counts arrive with its first two pixels)
    // element e is turned into (w, u) right in front of step e's MFMAs (the scheduling barriers keep
    // the four elements' temporaries from being live together)
#pragma unroll
    for (int e = 0; e < kTrChunk; ++e) {
      double w, u;
      if (e == 0) element(f01.x, z01.x, n01.x, omc[0], nl_cur & 0xFF, w, u);
      if (e == 1) element(f01.y, z01.y, n01.y, omc[1], (nl_cur >> 8) & 0xFF, w, u);
      if (e == 2) {
        element(f23.x, z23.x, n23.x, omc[2], (nl_cur >> 16) & 0xFF, w, u);
#ifndef TR_EXP_BUILD_EARLYCOPY
        // The next chunk's record copy is issued HERE: behind the compiler's waits for this chunk's raw
        // values.  The compiler counts only its own loads, so with the inline-assembly copy issued at
        // the top of the chunk (rounds 3-4) its vmcnt(2) / vmcnt(0) in front of element 0 also waited
        // for the eight copy instructions just issued: every wave sat out the latency of its prefetch
        // once per chunk.  Two K-steps (32 MFMAs of this wave) remain to cover the copy.  k_train_build
        // 281 -> 275 us at k = 40, 88 -> 85 at k = 20 (profiles/r05_ab_training_k40.txt).  (Forming both
        // elements of a half together, so that the next chunk's raw loads go out two K-steps earlier,
        // was measured as well: no gain, 281 us.)
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue_chunk(c + 1);
#endif
      }
      if (e == 3) element(f23.y, z23.y, n23.y, omc[3], nl_cur >> 24, w, u);
      __builtin_amdgcn_sched_barrier(0);
      if (e == 1 && more) load01(c + 1);
      if (e == 3 && more) load23(c + 1);
      double b[NT];
#pragma unroll
      for (int cc = 0; cc < NT; ++cc) b[cc] = buf[(size_t)(e * 16 + cc) * 64];
#pragma unroll
      for (int cc = 0; cc < NT; ++cc)
        acc[cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(cc < NW ? w : u, b[cc], acc[cc], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      int ex;
      mant = frexp(mant, &ex);  // back into [1/2, 1)
      esum += ex;
    }
  }
  glds_wait();  // (no copy is in flight behind the last chunk; stated for tools/check_vmem_hazard.py)
  if (!active) return;
  // result register rr of tile c: row (lane >> 4) + 4 rr, column 16 (16 tg + c) + (lane & 15)
  double *o = a.out + ((r * D.H + h) * 16) * (int64_t)a.cols + 256 * tg;
#pragma unroll
  for (int c = 0; c < NT; ++c)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) o[(int64_t)(jj + 4 * rr) * a.cols + 16 * c + s] = acc[c][rr];
  if (tg == 0) {
    double logd = log(mant) + (double)esum * 0.693147180559945309;
    for (int sh = 16; sh <= 32; sh <<= 1) {  // the four lanes of a quasar, in a fixed order
      logd += __shfl_xor(logd, sh);
      yy += __shfl_xor(yy, sh);
      cnt += __shfl_xor(cnt, sh);
    }
    if (jj == 0) {
      double *o3 = a.part1 + (q * D.H + h) * 3;
      o3[0] = logd;
      o3[1] = yy;
      o3[2] = cnt;
    }
  }
}

template <bool LY>
__global__ __launch_bounds__(kTrCWaves * 64, 2) void k_train_build(TrainBuildArgs a) {
  extern __shared__ double smem[];
  const int tg = train_tile_group(a.groups);
  const int nw = max(0, min(16, a.w_tiles - 16 * tg));  // block-uniform: 14 (k <= 20); 16, 16, 16, 4 (k <= 40)
  if (nw == 16) train_build_body<16, LY>(a, smem, tg);
  else if (nw == 14) train_build_body<14, LY>(a, smem, tg);
  else train_build_body<4, LY>(a, smem, tg);
}

// ------------------------------------------------------------------------------------------
// k_train_factor: 64 / KMAX quasars per wave, lanes along the rows (padded quasars write zero operands).
// recD: [group][TQ + pad][16 tiles][jj = quasar % 4][col]  B[quasar][column] of [vech(T_q) | z_q]   (dM)
// recE: [NQ16][Ks][jj = column % 4][s = quasar % 16]       A[quasar][column] of [vech2(T_q) | z_q]  (core)
// ------------------------------------------------------------------------------------------
struct TrainFactorArgs {
  TrainDims d;
  const double *partB;   // [NQ16][H][16][Cols]
  const double *part1;   // [16 NQ16][H][3]
  double *recD, *recE, *nlogp;
  int32_t *not_pd;
  int64_t group_stride;  // doubles between the tile groups of recD: (TQ + chunk padding) * 16 * 64
};

// Geometry of k_train_factor: a wave factors QW quasars side by side (sub-group sg of KMAX lanes,
// lane i of it owns row i), a block is 4 waves.
// L and L^-1 are lower triangular and are stored packed, rows padded to an even length so that every
// row starts on a 16-byte boundary (two entries per LDS read): rows 2m and 2m+1 take 2m + 2 doubles
// each.  Half the LDS of a dense KMAX x KMAX array: two blocks per CU for k <= 40 too.
__host__ __device__ constexpr int tri_off(int r) { return 2 * (r / 2) * (r / 2 + 1) + (r & 1) * (2 * (r / 2) + 2); }
__host__ __device__ constexpr int tri_len(int r) { return 2 * (r / 2) + 2; }  // padded length of row r (>= r + 1)
template <int KMAX> struct TrF {
  static constexpr int QW = 64 / KMAX;  // 3 (k <= 20), 1 (k <= 40)
  static constexpr int Waves = 4;
  static constexpr int FQ = QW * Waves;  // quasars per block: a whole number of quasar steps of recD
  static_assert(FQ % 4 == 0, "a block writes whole quasar steps of recD");
  // doubles of L / L^-1 per quasar; the recE rows (4 Ks) are assembled in the same storage afterwards
  static constexpr int LSize = tri_off(KMAX) > 4 * TrC<KMAX>::Ks ? tri_off(KMAX) : 4 * TrC<KMAX>::Ks;
};

// Orders the LDS traffic of ONE wave: lanes exchange data through LDS without a block barrier (a
// wave's LDS instructions execute in issue order; this keeps the compiler from moving them).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int KMAX>
__global__ __launch_bounds__(256, 2) void k_train_factor(TrainFactorArgs a) {
  // Lane i of a sub-group owns row i of B / L in registers (static indices: every loop over KMAX
  // is unrolled; no private array is indexed at run time -- that would live in scratch memory).
  // Ranks below KMAX are padded with identity rows (B = I there: L = I, log L_jj = 0, z = 0), so
  // the code has no rank-dependent branch.  Column j of the right-looking Cholesky travels through
  // LDS once: every lane publishes a_ij, reads the pivot a_jj and the a_cj (c > j) it needs, and
  // takes ONE reciprocal square root -- l_ij = a_ij / sqrt(a_jj), row_c -= (a_ij / a_jj) a_cj.
  // L, then L^-1 (column by column), then B^-1 = L^-T L^-1 and z = B^-1 t stay within the wave.
  // The two operand tilings of [T_q | z_q] are assembled in LDS (in place: vech(B^-1) overwrites
  // the summed partials, the recE rows the storage of L^-1) and leave the block as contiguous runs.
  using K = TrC<KMAX>;
  using F = TrF<KMAX>;
  constexpr int FQ = F::FQ, QW = F::QW;
  __shared__ __attribute__((aligned(16))) double s_L[FQ][F::LSize];
  __shared__ __attribute__((aligned(16))) double s_S[FQ][K::Cols];
  __shared__ __attribute__((aligned(16))) double s_t[FQ][KMAX], s_z[FQ][KMAX], s_ld[FQ][KMAX], s_col[F::Waves][64];
  __shared__ double s_sc[FQ][4];
  __shared__ int s_good[FQ];
  __shared__ uint8_t s_vi[K::W * 16], s_vj[K::W * 16];
  static_assert(4 * K::KsW <= 16 * K::W && 4 * K::KsU <= 16 * K::U, "recE columns are a prefix of recD's");
  const TrainDims &D = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, k = D.k;
  const int64_t q0 = (int64_t)blockIdx.x * FQ, nq_pad = D.NQ16 * 16;
  const int nb = k * (k + 1) / 2;
  for (int c = tid; c < K::W * 16; c += 256) {
    int i = 0, j = 0;
    if (c < nb) vech_ij(c, &i, &j);
    s_vi[c] = (uint8_t)i;
    s_vj[c] = (uint8_t)j;
  }
  // [vech(B - I) | t] = Sum_h partial, in split order; threads along the columns (coalesced)
  for (int e = tid; e < FQ * K::Cols; e += 256) {
    const int ql = e / K::Cols, c = e % K::Cols;
    const int64_t q = q0 + ql;
    double v = 0.0;
    if (q < D.nq) {
      const double *pb = a.partB + (((q >> 4) * D.H) * 16 + (q & 15)) * (int64_t)K::Cols + c;
      for (int h0 = 0; h0 < D.H; h0 += 4) {  // four loads in flight, added in split order
        double pv[4];
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) pv[hh] = h0 + hh < D.H ? pb[(int64_t)(h0 + hh) * 16 * K::Cols] : 0.0;
#pragma unroll
        for (int hh = 0; hh < 4; ++hh)
          if (h0 + hh < D.H) v += pv[hh];
      }
    }
    s_S[ql][c] = v;
  }
  for (int e = tid; e < FQ * 3; e += 256) {
    const int ql = e / 3, which = e % 3;
    const int64_t q = q0 + ql;
    double v = 0.0;
    if (q < D.nq)
      for (int b = 0; b < D.H; ++b) v += a.part1[(q * D.H + b) * 3 + which];
    s_sc[ql][which] = v;
  }
  __syncthreads();

  // lanes past the last whole sub-group shadow sub-group 0 (same reads, no writes)
  const int sg_raw = lane / KMAX;
  const bool valid = sg_raw < QW;
  const int sg = valid ? sg_raw : 0, i = lane - sg_raw * KMAX;
  const int ql = wave * QW + sg;
  const int64_t q = q0 + ql;
  const bool real = q < D.nq;
  double *sL = s_L[ql], *sS = s_S[ql], *st = s_t[ql], *sz = s_z[ql];
  double *colp = s_col[wave] + sg * KMAX;  // this sub-group's column exchange
  double row[KMAX];  // row i of B (lower triangle), then of L
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    double v = (i < k && j <= i) ? sS[i * (i + 1) / 2 + j] : 0.0;
    if (j == i) v += 1.0;
    row[j] = v;
  }
  const double tl = i < k ? sS[K::W * 16 + i] : 0.0;
  bool pd = true;
  double dinv[KMAX], my_l = 1.0;
  // Cholesky B = L L' (spectrum_loss.m:42), right-looking
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    wave_lds_sync();  // the previous column's readers are done
    if (valid) colp[i] = row[j];
    wave_lds_sync();
    double acj[KMAX];
#pragma unroll
    for (int c = j; c < KMAX; ++c) acj[c] = colp[c];
    const double djj = acj[j];
    pd = pd && (djj > 0.0);
    const double inv = rsqrt_nr(djj);  // 1 / l_jj
    dinv[j] = inv;
    const double lij = row[j] * inv;   // meaningful for i >= j; l_jj = a_jj / sqrt(a_jj) in lane j
    const double lij2 = lij * inv;     // a_ij / a_jj
    row[j] = lij;
    if (i == j) my_l = lij;
#pragma unroll
    for (int c = j + 1; c < KMAX; ++c)
      if (i >= c) row[c] = fma(-lij2, acj[c], row[c]);
  }
  if (valid) {
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
      if (j < tri_len(i)) sL[tri_off(i) + j] = j <= i ? row[j] : 0.0;  // (the pad entry of an even row: 0)
    st[i] = tl;
    s_ld[ql][i] = log(my_l);
  }
  wave_lds_sync();
  // column i of L^-1: L x = e_i (x_r = 0 for r < i)
  double x[KMAX];
#pragma unroll
  for (int r = 0; r < KMAX; ++r) {
    double s0 = r == i ? 1.0 : 0.0, s1 = 0.0;  // two chains: the dot product is latency-, not throughput-bound
#pragma unroll
    for (int mm = 0; mm + 1 < r; mm += 2) {  // broadcast reads, two entries at a time
      const double2 lp = *reinterpret_cast<const double2 *>(sL + tri_off(r) + mm);
      s0 = fma(-lp.x, x[mm], s0);
      s1 = fma(-lp.y, x[mm + 1], s1);
    }
    if (r & 1) s0 = fma(-sL[tri_off(r) + r - 1], x[r - 1], s0);
    x[r] = r >= i ? (s0 + s1) * dinv[r] : 0.0;
  }
  wave_lds_sync();  // every lane has read L: its storage now takes L^-1
  if (valid) {
#pragma unroll
    for (int r = 0; r < KMAX; ++r)
      if (i < tri_len(r)) sL[tri_off(r) + i] = x[r];  // (x_r = 0 for r < i: the pad entry is 0)
  }
  wave_lds_sync();
  // B^-1 = L^-T L^-1: entry (i, c) = Sum_{r >= max(i, c)} Linv[r][i] Linv[r][c]; z = B^-1 t on the way
  double zacc = 0.0;
#pragma unroll
  for (int c0 = 0; c0 < KMAX; c0 += 4) {
    double v[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = c0; r < KMAX; ++r) {  // Linv[r][c] = 0 for c > r: rows c0, c0 + 1 end at column c0 + 1
      const double2 l0 = *reinterpret_cast<const double2 *>(sL + tri_off(r) + c0);
      v[0] = fma(x[r], l0.x, v[0]);
      v[1] = fma(x[r], l0.y, v[1]);
      if (r >= c0 + 2) {  // compile-time
        const double2 l1 = *reinterpret_cast<const double2 *>(sL + tri_off(r) + c0 + 2);
        v[2] = fma(x[r], l1.x, v[2]);
        v[3] = fma(x[r], l1.y, v[3]);
      }
    }
    const double2 t0 = *reinterpret_cast<const double2 *>(st + c0), t1 = *reinterpret_cast<const double2 *>(st + c0 + 2);
    zacc = fma(v[0], t0.x, zacc);
    zacc = fma(v[1], t0.y, zacc);
    zacc = fma(v[2], t1.x, zacc);
    zacc = fma(v[3], t1.y, zacc);
    if (valid && i < k) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
        if (c0 + cc <= i) sS[i * (i + 1) / 2 + c0 + cc] = v[cc];  // vech(B^-1) over the dead partial sums
    }
  }
  if (valid) sz[i] = zacc;
  wave_lds_sync();
  if (valid && i == 0) {
    double tz = 0.0, ld = 0.0;
    for (int c = 0; c < KMAX; ++c) {
      tz = fma(st[c], sz[c], tz);
      ld += s_ld[ql][c];
    }
    const double log_2pi = 1.83787706640934534;  // spectrum_loss.m:17
    const double v = 0.5 * ((s_sc[ql][1] - tz) + s_sc[ql][0] + 2 * ld + s_sc[ql][2] * log_2pi);  // :48-52
    const bool good = real && pd && v == v;  // (NaN: a pivot, or a variance d, that was not positive)
    if (real && !good) *a.not_pd = 1;
    s_good[ql] = good ? 1 : 0;
    if (q < nq_pad) a.nlogp[q] = good ? v : 0.0;  // (nlogp is allocated for the padded quasar count)
  }
  __syncthreads();
  // T = B^-1 + z z' in the two operand tilings (zero for padded / failed quasars): recD rows in
  // place in s_S, recE rows (off-diagonals doubled: m'T m = Sum_{i>=j} (2 - delta_ij) T_ij m_i m_j)
  // over the storage of L^-1
  for (int e = tid; e < FQ * K::Cols; e += 256) {
    const int qe = e / K::Cols, c = e % K::Cols;
    const bool good = s_good[qe] != 0;
    double v = 0.0;
    if (c < K::W * 16) {
      const int vi = s_vi[c], vj = s_vj[c];
      if (good && c < nb) v = s_S[qe][c] + s_z[qe][vi] * s_z[qe][vj];
      s_S[qe][c] = v;
      if (c < 4 * K::KsW) s_L[qe][c] = vi != vj ? 2.0 * v : v;
    } else {
      const int cz = c - K::W * 16;
      if (good && cz < k) v = s_z[qe][cz];
      s_S[qe][c] = v;
      if (cz < 4 * K::KsU) s_L[qe][4 * K::KsW + cz] = v;
    }
  }
  __syncthreads();
  // recD: [group][tq][16 tiles][jj = quasar % 4][col]: the block's FQ quasars are FQ / 4 whole quasar steps
  for (int e = tid; e < (FQ / 4) * K::Tiles * 64; e += 256) {
    const int tql = e / (K::Tiles * 64), r = e % (K::Tiles * 64);
    const int c = r >> 6, jj = (r >> 4) & 3, col = r & 15;
    const int64_t tq = (q0 >> 2) + tql;
    if (tq < D.TQ)
      a.recD[(c >> 4) * a.group_stride + (tq * 16 + (c & 15)) * 64 + (r & 63)] = s_S[4 * tql + jj][16 * c + col];
  }
  // recE: [g][Ks][jj = column % 4][s = quasar % 16]
  for (int e = tid; e < K::Ks * 4 * FQ; e += 256) {
    const int qe = e % FQ, kj = e / FQ;  // kj = 4 ks + jj
    const int64_t qq = q0 + qe;
    if (qq < nq_pad) a.recE[(qq >> 4) * K::Ks * 64 + (int64_t)kj * 16 + (qq & 15)] = s_L[qe][kj];
  }
}

// ------------------------------------------------------------------------------------------
// k_train_factor16 (k <= 40): the per-quasar algebra of k_train_factor in registers.
// k_train_factor keeps a quasar on 40 lanes and reads L, L^-1 and the column exchange of the
// Cholesky as broadcast LDS reads -- ~1250 of them per quasar, the pattern that bounds the old sweep
// epilogues (48 cycles per instruction with eight waves per CU, tools/dpp_f64_probe.hip).  Here a
// quasar sits on the 16 lanes of ONE DPP row, four quasars per wave, rows dealt as in Rows16
// (sweep_kernels.hpp: lane l owns rows l < 9, 9 + l and 25 + l; row 40 is t), and every cross-lane
// operand is a register of the same row, broadcast by v_fmac_f64_dpp:
//   Cholesky      left-looking, the pivot row's entries from their owner (as factor_rows16); the
//                 augmented row t comes out as y = L^-1 t, its running diagonal as -y'y = -t'z
//   L^-1          lane(a) solves L x = e_a for its rows a (the COLUMNS a of L^-1 = rows of U = L^-T):
//                 row r of L is broadcast from its owner, x stays in the lane; one slot at a time, so
//                 that L (72 entries per lane) and the columns built so far (40 + 31 + 15) fit
//   z = L^-T y    z_a = Sum_r U_a[r] y_r, y broadcast from the lane that owns row 40
//   B^-1 = U U'   entry (a, c) = Sum_r U_a[r] U_c[r]: row c of U broadcast from its owner; the entry
//                 goes straight to vech(B^-1) in s_S
// Ranks below 40 are padded with identity rows (read from the zero columns behind t), so the code has
// no rank-dependent branch.  The operation order differs from k_train_factor's (left- instead of
// right-looking, other summation orders): results agree to rounding, not bit for bit.
// ------------------------------------------------------------------------------------------
template <int KMAX>
__device__ __forceinline__ void train_factor16(double *sS, double *sz, int l, int k, double *tz_out,
                                               double *two_ld_out, bool *pd_out) {
  using RR = Rows16<KMAX>;
  constexpr int RA = RR::RA, RB = RA + 16;           // first rows of the second and third slot: 9, 25
  constexpr int voff = TrK<KMAX>::W * 16;            // t behind the vech columns
  constexpr int zoff = voff + 16 * TrK<KMAX>::U;     // the pad of zeros behind t (k_train_factor16 lays it down)
  static_assert(KMAX == 40 && RA == 9, "three slots: rows 0..8, 9..24, 25..40");
  const int ia = l, ib = RA + l, ic = RB + l;        // ic == KMAX: the row of t
  const bool fa = l < RA && ia < k, fb = ib < k, fc = ic < k, ft = ic == KMAX;
  const int roa = fa ? ia * (ia + 1) / 2 : zoff, rob = fb ? ib * (ib + 1) / 2 : zoff;
  const int roc = fc ? ic * (ic + 1) / 2 : ft ? voff : zoff;
  Rows16<KMAX> R;
  R.da = (fa ? sS[roa + ia] : 0.0) + 1.0;            // B = I + Sum (spectrum_loss.m:34-42); identity rows: 1
  R.db = (fb ? sS[rob + ib] : 0.0) + 1.0;
  R.dc = ft ? 0.0 : (fc ? sS[roc + ic] : 0.0) + 1.0;
#pragma unroll
  for (int m = 0; m < RR::NA; ++m) R.a[m] = sS[roa + m];
#pragma unroll
  for (int m = 0; m < RR::NB; ++m) R.b[m] = sS[rob + m];
#pragma unroll
  for (int m = 0; m < RR::NC; ++m) R.c[m] = sS[roc + m];  // (columns >= the row index: unused)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();  // every lane has its rows: vech(B^-1) may overwrite s_S from here on

  // ---- Cholesky (spectrum_loss.m:42), all KMAX columns; 1 / L_jj kept by the owner of row j
  double lprod = 1.0, inva = 1.0, invb = 1.0, invc = 1.0;
  int lexp = 0;
  bool pd = true;
  static_for<KMAX>([&](auto J_) __attribute__((always_inline)) {
    constexpr int j = decltype(J_)::value;
    constexpr int slot = j < RA ? 0 : j < RB ? 1 : 2;
    constexpr int ol = j - (slot == 0 ? 0 : slot == 1 ? RA : RB);
    constexpr bool act_a = j < RR::NA, act_b = j < RR::NB;
    static_for<j>([&](auto M_) __attribute__((always_inline)) {
      constexpr int mm = decltype(M_)::value;
      constexpr bool fresh = mm == j - 1;
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
    const double dj = mov_bcast<ol, true>(dsel);
    pd = pd && (dj > 0.0);
    const double inv = rsqrt_nr(dj);  // 1 / L_jj
    lprod *= dj;                      // 2 Sum log L_jj = log Prod d_j
    if constexpr ((j & 3) == 3) {
      lexp += __builtin_amdgcn_frexp_exp(lprod);
      lprod = __builtin_amdgcn_frexp_mant(lprod);
    }
    if (l == ol) {  // (a select under a constant lane mask)
      if constexpr (slot == 0) inva = inv;
      else if constexpr (slot == 1) invb = inv;
      else invc = inv;
    }
    if constexpr (act_a) {
      R.a[j] *= inv;
      R.da = fma(-R.a[j], R.a[j], R.da);
    }
    if constexpr (act_b) {
      R.b[j] *= inv;
      R.db = fma(-R.b[j], R.b[j], R.db);
    }
    R.c[j] *= inv;
    R.dc = fma(-R.c[j], R.c[j], R.dc);
  });
  *tz_out = -mov_bcast<15, true>(R.dc);  // row 40 is the third row of lane 15: -dd ended as y'y = t'z
  *two_ld_out = log(lprod) + (double)lexp * 0.6931471805599453;
  *pd_out = pd;

  // ---- columns of L^-1 (rows of U = L^-T), one slot at a time; z = U y on the way
  // row r of L: register [m] of its owner's slot array; 1 / L_rr: the owner's inv of that slot
  double xa[KMAX], za = 0.0;
  static_for<KMAX>([&](auto R_) __attribute__((always_inline)) {
    constexpr int r = decltype(R_)::value;
    constexpr int slot = r < RA ? 0 : r < RB ? 1 : 2;
    constexpr int ol = r - (slot == 0 ? 0 : slot == 1 ? RA : RB);
    double s = slot == 0 && l == ol ? 1.0 : 0.0;  // delta_ra - Sum_m L_rm x_m (fmac_bcast subtracts the product)
    static_for<r>([&](auto M_) __attribute__((always_inline)) {
      constexpr int mm = decltype(M_)::value;
      double lr;
      if constexpr (slot == 0) lr = R.a[mm];
      else if constexpr (slot == 1) lr = R.b[mm];
      else lr = R.c[mm];
      fmac_bcast<ol>(s, lr, xa[mm]);
    });
    const double invr = mov_bcast<ol, true>(slot == 0 ? inva : slot == 1 ? invb : invc);
    xa[r] = s * invr;
    fmac_bcast<15>(za, R.c[r], xa[r]);  // za -= y_r x_a[r]
  });
  double xb[KMAX - RA], zb = 0.0;  // x_b[r - RA], r >= RA
  static_for<KMAX - RA>([&](auto R_) __attribute__((always_inline)) {
    constexpr int r = RA + decltype(R_)::value;
    constexpr int slot = r < RB ? 1 : 2;
    constexpr int ol = r - (slot == 1 ? RA : RB);
    double s = slot == 1 && l == ol ? 1.0 : 0.0;
    static_for<r - RA>([&](auto M_) __attribute__((always_inline)) {
      constexpr int mm = RA + decltype(M_)::value;
      double lr;
      if constexpr (slot == 1) lr = R.b[mm];
      else lr = R.c[mm];
      fmac_bcast<ol>(s, lr, xb[mm - RA]);
    });
    const double invr = mov_bcast<ol, true>(slot == 1 ? invb : invc);
    xb[r - RA] = s * invr;
    fmac_bcast<15>(zb, R.c[r], xb[r - RA]);
  });
  double xc[KMAX - RB], zc = 0.0;  // x_c[r - RB], r >= RB
  static_for<KMAX - RB>([&](auto R_) __attribute__((always_inline)) {
    constexpr int r = RB + decltype(R_)::value;
    constexpr int ol = r - RB;
    double s = l == ol ? 1.0 : 0.0;
    static_for<r - RB>([&](auto M_) __attribute__((always_inline)) {
      constexpr int mm = RB + decltype(M_)::value;
      fmac_bcast<ol>(s, R.c[mm], xc[mm - RB]);
    });
    const double invr = mov_bcast<ol, true>(invc);
    xc[r - RB] = s * invr;
    fmac_bcast<15>(zc, R.c[r], xc[r - RB]);
  });
  // z (fmac_bcast accumulated -Sum): rows of t's lane and laneless slots are not stored
  if (l < RA) sz[ia] = -za;
  sz[ib] = -zb;
  if (ic < KMAX) sz[ic] = -zc;

  // ---- B^-1 = U U': entry (a, c), c <= a, = Sum_{r >= a} U_a[r] U_c[r]; row c of U from its owner
  static_for<KMAX>([&](auto C_) __attribute__((always_inline)) {
    constexpr int c = decltype(C_)::value;
    constexpr int slot = c < RA ? 0 : c < RB ? 1 : 2;
    constexpr int ol = c - (slot == 0 ? 0 : slot == 1 ? RA : RB);
    constexpr int r0a = c, r0b = c > RA ? c : RA, r0c = c > RB ? c : RB;  // first r with both factors possibly nonzero
    double acca = 0.0, accb = 0.0, accc = 0.0;
    static_for<KMAX - r0c>([&](auto R_) __attribute__((always_inline)) {  // the rows every slot sums over
      constexpr int r = r0c + decltype(R_)::value;
      double uc;
      if constexpr (slot == 0) uc = xa[r];
      else if constexpr (slot == 1) uc = xb[r - RA];
      else uc = xc[r - RB];
      if constexpr (slot == 0) fmac_bcast<ol>(acca, uc, xa[r]);
      if constexpr (slot <= 1) fmac_bcast<ol>(accb, uc, xb[r - RA]);
      fmac_bcast<ol>(accc, uc, xc[r - RB]);
    });
    if constexpr (slot <= 1) {  // rows r0b .. r0c - 1: the third slot's columns have not started there
      static_for<r0c - r0b>([&](auto R_) __attribute__((always_inline)) {
        constexpr int r = r0b + decltype(R_)::value;
        double uc;
        if constexpr (slot == 0) uc = xa[r];
        else uc = xb[r - RA];
        if constexpr (slot == 0) fmac_bcast<ol>(acca, uc, xa[r]);
        fmac_bcast<ol>(accb, uc, xb[r - RA]);
      });
    }
    if constexpr (slot == 0) {  // rows r0a .. r0b - 1: the first slot's columns only
      static_for<r0b - r0a>([&](auto R_) __attribute__((always_inline)) {
        constexpr int r = r0a + decltype(R_)::value;
        fmac_bcast<ol>(acca, xa[r], xa[r]);
      });
    }
    int cc = c;  // (opaque: or all compares are formed up front)
    asm volatile("" : "+s"(cc));
    if constexpr (slot == 0)
      if (l < RA && ia >= cc) sS[ia * (ia + 1) / 2 + c] = -acca;
    if constexpr (slot <= 1)
      if (ib >= cc) sS[ib * (ib + 1) / 2 + c] = -accb;
    if (ic >= cc && ic < KMAX) sS[ic * (ic + 1) / 2 + c] = -accc;
  });
}

constexpr int kTrF16Threads = 64;  // ONE wave per block: it loads, factors and stores its four quasars
template <int KMAX>
__global__ __launch_bounds__(kTrF16Threads) void k_train_factor16(TrainFactorArgs a) {
  // The shell of k_train_factor (partial sums in, the two operand tilings out) around train_factor16.
  // A block is one wave and four quasars, one per 16-lane row.  LDS is the summed row of a quasar --
  // its used columns and a pad of zeros -- and nothing else (k_train_factor keeps L / L^-1 there
  // too, stages the recE rows over them and idles three of its four waves while one factors): five
  // blocks per CU, 5120 quasars in flight on the chip.
  using K = TrC<KMAX>;
  constexpr int FQ = 4, NT = kTrF16Threads;
  constexpr int Used = K::W * 16 + 16 * K::U;  // columns that can be non-zero: vech, then t (880)
  constexpr int Row = Used + 48;               // ... and zeros behind them (train_factor16 reads identity rows there)
  static_assert(TrF<KMAX>::FQ == FQ, "the grid and the recD steps of k_train_factor");
  static_assert(Row - Used >= KMAX, "identity rows read KMAX zeros behind the used columns");
  __shared__ __attribute__((aligned(16))) double s_S[FQ][Row];
  __shared__ __attribute__((aligned(16))) double s_z[FQ][KMAX];
  __shared__ double s_sc[FQ][4];
  __shared__ int s_good[FQ];
  __shared__ uint8_t s_vi[K::W * 16], s_vj[K::W * 16];
  static_assert(4 * K::KsW <= 16 * K::W && 4 * K::KsU <= 16 * K::U, "recE columns are a prefix of recD's");
  const TrainDims &D = a.d;
  const int tid = threadIdx.x, k = D.k;
  const int64_t q0 = (int64_t)blockIdx.x * FQ, nq_pad = D.NQ16 * 16;
  const int nb = k * (k + 1) / 2;
  {  // (i, j) of vech column c: one square root per lane, then 64 columns further at a time
    int i = 0, j = 0;
    vech_ij(tid, &i, &j);
    for (int c = tid; c < K::W * 16; c += NT) {
      s_vi[c] = (uint8_t)(c < nb ? i : 0);
      s_vj[c] = (uint8_t)(c < nb ? j : 0);
      j += NT;
      while (j > i) {
        j -= i + 1;
        ++i;
      }
    }
  }
  // [vech(B - I) | t] = Sum_h partial, in split order; lanes along the columns (coalesced), seven
  // columns of a lane at a time so that 28 loads are in flight (the loop is latency-bound); only the
  // columns that can be non-zero are read, the pad behind them is laid down here
  static_assert(Used <= 14 * NT, "two halves of seven columns per lane");
#pragma unroll 1
  for (int ql = 0; ql < FQ; ++ql) {
    const int64_t q = q0 + ql;
    const bool onq = q < D.nq;
    const double *base = a.partB + (((onq ? q >> 4 : 0) * D.H) * 16 + (onq ? q & 15 : 0)) * (int64_t)K::Cols + tid;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      double v[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      for (int h0 = 0; h0 < D.H; h0 += 4) {
        double pv[7][4];
#pragma unroll
        for (int u = 0; u < 7; ++u)
#pragma unroll
          for (int hh = 0; hh < 4; ++hh) {
#ifdef TF16_EXP_NOLOAD  // ablation: no partial sums read (results wrong by construction)
            pv[u][hh] = 1e-6 * (u + hh);
#else
            const bool on = onq && tid + NT * (7 * half + u) < Used && h0 + hh < D.H;
            pv[u][hh] = on ? base[NT * (7 * half + u) + (int64_t)(h0 + hh) * 16 * K::Cols] : 0.0;
#endif
          }
#pragma unroll
        for (int u = 0; u < 7; ++u)
#pragma unroll
          for (int hh = 0; hh < 4; ++hh)
            if (h0 + hh < D.H) v[u] += pv[u][hh];  // (added in split order)
      }
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        const int c = tid + NT * (7 * half + u);
        if (c < Row) s_S[ql][c] = c < Used ? v[u] : 0.0;
      }
    }
    if (14 * NT < Row && tid + 14 * NT < Row) s_S[ql][tid + 14 * NT] = 0.0;
  }
  for (int e = tid; e < FQ * 3; e += NT) {
    const int ql = e / 3, which = e % 3;
    const int64_t q = q0 + ql;
    double v = 0.0;
    if (q < D.nq)
      for (int b = 0; b < D.H; ++b) v += a.part1[(q * D.H + b) * 3 + which];
    s_sc[ql][which] = v;
  }
  __syncthreads();
  {
    const int ql = tid >> 4, l = tid & 15;
    const int64_t q = q0 + ql;
    const bool real = q < D.nq;
    double tz, two_ld;
    bool pd;
#ifdef TF16_EXP_NOFACTOR  // ablation: no factorisation (results wrong by construction)
    tz = s_S[ql][l];
    two_ld = 0.0;
    pd = true;
    if (l < 8) s_z[ql][l] = tz;
#else
    train_factor16<KMAX>(s_S[ql], s_z[ql], l, k, &tz, &two_ld, &pd);
#endif
    if (l == 0) {
      const double log_2pi = 1.83787706640934534;  // spectrum_loss.m:17
      const double v = 0.5 * ((s_sc[ql][1] - tz) + s_sc[ql][0] + two_ld + s_sc[ql][2] * log_2pi);  // :48-52
      const bool good = real && pd && v == v;  // (NaN: a pivot, or a variance d, that was not positive)
      if (real && !good) *a.not_pd = 1;
      s_good[ql] = good ? 1 : 0;
      if (q < nq_pad) a.nlogp[q] = good ? v : 0.0;  // (nlogp is allocated for the padded quasar count)
    }
  }
  __syncthreads();
  // T = B^-1 + z z' (zero for padded / failed quasars) in place in s_S: the recD rows
  // (one wave does the work four did in k_train_factor: the loops of the shell are unrolled so that
  // their LDS reads and stores are in flight together)
#pragma unroll 5
  for (int e = tid; e < FQ * Used; e += NT) {
    const int qe = e / Used, c = e % Used;
    const bool good = s_good[qe] != 0;
    double v = 0.0;
    if (c < K::W * 16) {
      if (good && c < nb) v = s_S[qe][c] + s_z[qe][s_vi[c]] * s_z[qe][s_vj[c]];
    } else {
      const int cz = c - K::W * 16;
      if (good && cz < k) v = s_z[qe][cz];
    }
    s_S[qe][c] = v;
  }
  __syncthreads();
  // recD: [group][tq][16 tiles][jj = quasar % 4][col]: the block's four quasars are one quasar step
  // (the tiles behind the used columns are padding: recD is zeroed when it is allocated and nobody
  // ever writes anything but zeros there)
  static_assert(Used % 16 == 0, "whole tiles of used columns");
#pragma unroll 5
  for (int e = tid; e < (Used / 16) * 64; e += NT) {
    const int c = e >> 6, jj = (e >> 4) & 3, col = e & 15;
    const int64_t tq = q0 >> 2;
    if (tq < D.TQ) a.recD[(c >> 4) * a.group_stride + (tq * 16 + (c & 15)) * 64 + (e & 63)] = s_S[jj][16 * c + col];
  }
  // recE: [g][Ks][jj = column % 4][s = quasar % 16], off-diagonals doubled (m'T m = Sum_{i>=j} (2 - delta_ij)
  // T_ij m_i m_j): its columns are a prefix of the vech columns, then of z
#pragma unroll 6
  for (int e = tid; e < K::Ks * 4 * FQ; e += NT) {
    const int qe = e % FQ, kj = e / FQ;  // kj = 4 ks + jj
    const int64_t qq = q0 + qe;
    double v;
    if (kj < 4 * K::KsW) {
      v = s_S[qe][kj];
      if (s_vi[kj] != s_vj[kj]) v *= 2.0;
    } else {
      v = s_S[qe][K::W * 16 + (kj - 4 * K::KsW)];
    }
    if (qq < nq_pad) a.recE[(qq >> 4) * K::Ks * 64 + (int64_t)kj * 16 + (qq & 15)] = v;
  }
}

// ------------------------------------------------------------------------------------------
// k_train_core: one wave per (pixel group pt, split gs of the quasar groups).
// X_qp = m_p' T_q m_p (KsW column steps), Y_qp = m_p' z_q (KsU column steps) by MFMA, then
// core_qp and the sums over the wave's quasars: partcol[pt][gs][16] = Sum an core per pixel,
// partsc[pt][gs][3] = Sum core da for (c0, tau0, beta).
// ------------------------------------------------------------------------------------------
struct TrainCoreArgs {
  TrainDims d;
  const double *recP, *recE;
  const double *flux, *log_lya_1pz, *noise;
  const double *x;     // the parameter vector
  const uint8_t *nl;   // [nq][ld] active lines per pixel (lines.nfl > 1 only)
  TrainLines lines;
  double *wB, *uB;     // [PG][TQ][jj = quasar % 4][s = pixel % 16]: A operand of the dM contraction
  double *partcol, *partsc;
};
constexpr size_t kTrCoreLds = 2 * TrC<20>::Ks * 64 * sizeof(double);  // k <= 20: two quasar groups' A operands

// The element-wise gradient terms of one (16 quasars x 16 pixels) tile from X = m'Tm and Y = m'z
// (result register rr: quasar 16 g + jj + 4 rr, pixel p), accumulated into the wave's sums.
// Register rr of lane (jj, s) is quasar 4 (4 g + rr) + jj, pixel s: the A-operand lane order of
// quasar step 4 g + rr of the dM contraction, so w and u leave as whole 512-byte rows of wB / uB
// (`ob`: this lane's offset in the pixel group's rows; zeros for missing and padded elements).
struct TrainCoreRaw {
  double ye[4], lz[4], nv[4];
  uint32_t nl[4];
};
// the tile's data, requested BEFORE the tile's MFMAs so that they arrive behind them (rows are
// padded to 16 PG pixels with missing ones: no bound on p)
template <bool LY>
__device__ __forceinline__ void train_core_load(const TrainCoreArgs &a, int64_t g, int64_t p, int jj, bool active,
                                                TrainCoreRaw &r) {
  const TrainDims &D = a.d;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {  // unconditional loads (clamped row); train_core_tile drops rows behind the last quasar
    const int64_t q = min(g * 16 + jj + 4 * rr, D.nq - 1);
    const int64_t o = active ? q * D.ld + p : 0;
    r.ye[rr] = a.flux[o];
    r.lz[rr] = a.log_lya_1pz[o];
    r.nv[rr] = a.noise[o];
    r.nl[rr] = LY ? a.nl[o] : 1u;
  }
}
template <bool LY>
__device__ __forceinline__ void train_core_tile(const TrainCoreArgs &a, int64_t g, const TrainCoreRaw &raw, int jj, bool active, int64_t ob,
                                                double om, double c_0, double tau_0, double beta, const double *T_s,
                                                const double *exp_tab, const d4 &X4, const d4 &Y4, double &col,
                                                double &gc, double &gt, double &gb) {
  const double *ye = raw.ye, *lz = raw.lz, *nv = raw.nv;
  const double beta64 = beta * kExpScale;
  constexpr bool lyseries = LY;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const double y = ye[rr];
    double w = 0.0, u = 0.0;
    if (!isnan(y) && active && g * 16 + jj + 4 * rr < a.d.nq) {
      const double od = (lyseries ? T_s[raw.nl[rr]] : tau_0) * exp_table_scaled(beta64 * lz[rr], exp_tab);  // :22 (as k_train_build)
      const double ab = exp_table_scaled(-kExpScale * od, exp_tab);  // :23
      const double sf = 1 - ab + c_0;                           // :26
      const double an = om * (sf * sf);                         // :27
      w = fast_rcp(nv[rr] + an);                                // :29-31
      u = w * y;
      const double X = X4[rr], Y = Y4[rr];
      const double kiy = u - w * Y;                             // (K^-1 y)_p, :46
      const double diag = w - w * w * X + w * w * Y * Y;        // (K^-1)_pp = w - w^2 m'B^-1 m, :59
      const double core = kiy * kiy - diag;
      col = fma(an, core, col);                                 // :62
      double da = c_0 * om * sf;                                // :65
      gc = fma(core, da, gc);                                   // :66
      da = om * sf * od * ab;                                   // :69
      gt = fma(core, da, gt);                                   // :70
      da = da * lz[rr] * beta;                                  // :73
      gb = fma(core, da, gb);                                   // :74
    }
    if (active) {
      a.wB[ob + (4 * g + rr) * 64] = w;
      a.uB[ob + (4 * g + rr) * 64] = u;
    }
  }
}

__device__ __forceinline__ void train_core_store(const TrainCoreArgs &a, int64_t pt, int gs, int lane, double col,
                                                 double gc, double gt, double gb) {
  const TrainDims &D = a.d;
  const int jj = lane >> 4, s = lane & 15;
  col += __shfl_xor(col, 16);
  col += __shfl_xor(col, 32);
  if (jj == 0) a.partcol[(pt * D.GS + gs) * 16 + s] = col;
  for (int o = 32; o > 0; o >>= 1) {
    gc += __shfl_xor(gc, o);
    gt += __shfl_xor(gt, o);
    gb += __shfl_xor(gb, o);
  }
  if (lane == 0) {
    double *o3 = a.partsc + (pt * D.GS + gs) * 3;
    o3[0] = gc;
    o3[1] = gt;
    o3[2] = gb;
  }
}

// k <= 20.  A block is 4 waves = 4 pixel groups that walk the SAME quasar groups (split gs): the A
// operands of a quasar group ([vech2(T_q) | z_q] of its 16 quasars, 29 KB) are staged once per
// block, double-buffered by glds16; the B operands of the wave's pixel group stay in registers.
template <bool LY>
__global__ __launch_bounds__(256, 2) void k_train_core(TrainCoreArgs a) {
  extern __shared__ double smem[];
  using K = TrC<20>;
  const TrainDims &D = a.d;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t pblk = blockIdx.x / D.GS;
  const int gs = (int)(blockIdx.x % D.GS);
  const int64_t pt = pblk * 4 + wave;
  const bool active = pt < D.PG;
  const int64_t g0 = (D.NQ16 * gs) / D.GS, g1 = (D.NQ16 * (gs + 1)) / D.GS;  // balanced split
  const int jj = lane >> 4, s = lane & 15;
  const int64_t p = pt * 16 + s;
  const TrainScal sc = train_scal(a.x, D.G, D.k);
  const double c_0 = sc.c_0, tau_0 = sc.tau_0, beta = sc.beta;
  __shared__ double T_s[kTrMaxLines + 1], exp_tab[kExpTab];
  if (LY && threadIdx.x == 0) train_line_table(a.lines, tau_0, beta, T_s);  // (published by the first group's barrier)
  if (threadIdx.x < kExpTab) exp_tab[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / kExpTab));
  double bP[K::Ks];
#pragma unroll
  for (int ks = 0; ks < K::Ks; ++ks) bP[ks] = active ? a.recP[(pt * K::Ks + ks) * 64 + lane] : 0.0;
  const double om = (active && p < D.G) ? train_omega2(a.x, D.G, D.k, p) : 0.0;
  const int64_t ob = pt * D.TQ * 64 + lane;
  static_assert((K::Ks * 64) % 128 == 0, "a quasar group's operands are a whole number of KiB");
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t smem_lds = __builtin_amdgcn_readfirstlane(lds_address(smem));
  auto issue_group = [&](int64_t g) {  // (glds_chunk, sweep_kernels.hpp)
    glds_chunk<K::Ks * 64 / 128, 4>(a.recE + g * K::Ks * 64, smem_lds + (uint32_t)((g - g0) & 1) * (uint32_t)(K::Ks * 64 * 8),
                                    wave_s, lane);
  };
  if (g0 < g1) issue_group(g0);
  double col = 0.0, gc = 0.0, gt = 0.0, gb = 0.0;
  for (int64_t g = g0; g < g1; ++g) {
    // group g's operands landed (this wave's part).  Their copy is older than the previous tile's
    // eight stores of w and u, which may stay in flight (vector memory operations complete in
    // order): waiting for those too costs a store round trip per tile.
    if (g == g0) {
      glds_wait();
    } else {
      __builtin_amdgcn_s_waitcnt(0x0F78);  // vmcnt(8)
      asm volatile("" ::: "memory");
    }
    __syncthreads();
    if (g + 1 < g1) issue_group(g + 1);
    const double *re = smem + (size_t)((g - g0) & 1) * K::Ks * 64 + lane;
    TrainCoreRaw raw;
    train_core_load<LY>(a, g, p, jj, active, raw);
    __builtin_amdgcn_sched_barrier(0);  // (nothing that waits for these loads may move in front of the MFMAs)
    // two accumulator chains for X (registers: the kernel must stay within 256 per lane so that
    // two waves share a SIMD), one for Y
    d4 x0 = {0.0, 0.0, 0.0, 0.0}, x1 = x0, yv = x0;
#pragma unroll
    for (int ks = 0; ks < K::KsW; ks += 2) {
      x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(re[ks * 64], bP[ks], x0, 0, 0, 0);
      if (ks + 1 < K::KsW) x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(re[(ks + 1) * 64], bP[ks + 1], x1, 0, 0, 0);
    }
#pragma unroll
    for (int ks = K::KsW; ks < K::Ks; ++ks)
      yv = __builtin_amdgcn_mfma_f64_16x16x4f64(re[ks * 64], bP[ks], yv, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    const d4 xs = {x0[0] + x1[0], x0[1] + x1[1], x0[2] + x1[2], x0[3] + x1[3]};
    train_core_tile<LY>(a, g, raw, jj, active, ob, om, c_0, tau_0, beta, T_s, exp_tab, xs, yv, col, gc, gt, gb);
  }
  glds_wait();  // (no copy is in flight behind the last group; stated for tools/check_vmem_hazard.py)
  if (!active) return;
  train_core_store(a, pt, gs, lane, col, gc, gt, gb);
}

// 20 < k <= 40: 216 column steps -- too many operands for the registers (432 per lane) and 110 KB per
// quasar group, twice that per block: too many for LDS as well.  A block is kTrWidePB pixel groups and
// four consecutive splits of the quasar groups (one per wave): the pixel groups' operands -- the same
// for the four waves -- are staged through LDS in chunks of 12 column steps (6 KiB per group,
// double-buffered by the asynchronous copy, one barrier per chunk); each wave's own quasar-group
// operand comes straight from L2 into registers one chunk ahead (two register sets, the chunk loop
// unrolled by two) and is used for ALL the block's pixel groups, one accumulator chain each.
// (Round 3 first streamed BOTH operands of every MFMA from L2: 5.3 GB per evaluation through L2,
// latency-bound at 7.6 TB/s, 0.72-0.78 ms.  Rounds 3-4 had ONE pixel group per block: the quasar-group
// operands, 34.6 MB, were then re-read once per pixel group -- 2.65 GB through L2 per launch, 1.17 GB of
// it from beyond L2 (rocprofv3 FETCH_SIZE, profiles/r05_training_k40_pmc.json) against 0.29 GB of
// algorithmic traffic, 27 % of the wave cycles waiting: 0.366 ms.  Two pixel groups per block halve
// that re-read.)  The waves of a block run the same number of chunks -- a split that has one quasar
// group fewer idles through the last ones.
#ifndef TR_WIDE_CH
#define TR_WIDE_CH 12
#endif
constexpr int kTrWideCH = TR_WIDE_CH;  // column steps per chunk (TR_WIDE_CH: for A/B)
#ifndef TR_WIDE_PB
#define TR_WIDE_PB 2
#endif
constexpr int kTrWidePB = TR_WIDE_PB;  // pixel groups per block (TR_WIDE_PB=1: the round-3 form, for A/B)
template <bool LY>
__global__ __launch_bounds__(256, 2) void k_train_core_wide(TrainCoreArgs a) {
  using K = TrC<40>;
  constexpr int CH = kTrWideCH, NCH = K::Ks / CH, PB = kTrWidePB;
  static_assert(K::Ks % CH == 0 && (CH * 64) % 128 == 0 && CH % 4 == 0, "whole chunks of whole KiB");
  static_assert(K::KsW > (NCH - 1) * CH, "only the last chunk mixes X and Y steps");
  __shared__ __attribute__((aligned(16))) double sB[2][PB][CH * 64];
  __shared__ double T_s[kTrMaxLines + 1], exp_tab[kExpTab];
  if (threadIdx.x < kExpTab) exp_tab[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / kExpTab));  // (published by the first chunk's barrier)
  const TrainDims &D = a.d;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t pt0 = (blockIdx.x / (D.GS / 4)) * PB;  // (grid: ceil(PG / PB) x GS / 4 blocks; GS is a multiple of 4)
  const int gs4 = (int)(blockIdx.x % (D.GS / 4)) * 4, gs = gs4 + wave;
  const int64_t g0 = (D.NQ16 * gs) / D.GS, g1 = (D.NQ16 * (gs + 1)) / D.GS;
  int iters = 0;  // block-uniform: the longest of the four splits
  for (int w = 0; w < 4; ++w)
    iters = max(iters, (int)((D.NQ16 * (gs4 + w + 1)) / D.GS - (D.NQ16 * (gs4 + w)) / D.GS));
  const int total = iters * NCH;
  const int jj = lane >> 4, s = lane & 15;
  const TrainScal sc = train_scal(a.x, D.G, D.k);
  const double c_0 = sc.c_0, tau_0 = sc.tau_0, beta = sc.beta;
  if (LY && threadIdx.x == 0) train_line_table(a.lines, tau_0, beta, T_s);  // (published by the first chunk's barrier)
  // the block's pixel groups; one behind the last (PG not a multiple of PB) computes on the last one's
  // operands and stores nothing
  bool pvalid[PB];
  int64_t pt[PB], p[PB], ob[PB];
  double om[PB];
  const double *bsrc[PB];
#pragma unroll
  for (int b = 0; b < PB; ++b) {
    pvalid[b] = pt0 + b < D.PG;
    pt[b] = min(pt0 + b, D.PG - 1);
    p[b] = pt[b] * 16 + s;
    om[b] = p[b] < D.G ? train_omega2(a.x, D.G, D.k, p[b]) : 0.0;
    ob[b] = pt[b] * D.TQ * 64 + lane;
    bsrc[b] = a.recP + pt[b] * K::Ks * 64;
  }
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t sB_lds = __builtin_amdgcn_readfirstlane(lds_address(&sB[0][0][0]));
  auto issue = [&](int n) {  // running chunk n: chunk n % NCH of every pixel group's operand into buffer n & 1
#pragma unroll
    for (int b = 0; b < PB; ++b)
      glds_chunk<CH * 64 / 128, 4>(bsrc[b] + (size_t)(n % NCH) * CH * 64,
                                   sB_lds + (uint32_t)((n & 1) * PB + b) * (uint32_t)(CH * 64 * 8), wave_s, lane);
  };
  auto group_of = [&](int it) -> int64_t {  // (an idle iteration re-reads the split's last group; nothing of it is used)
    return min(max(min(g0 + it, g1 - 1), (int64_t)0), D.NQ16 - 1);
  };
  auto load_a = [&](int n, double (&dst)[CH]) {
    const int it = n / NCH, c = n - it * NCH;
#ifdef TR_EXP_AHOT  // ablation (results wrong by construction): every wave re-reads ONE quasar group's operand -- an L2-hot A stream
    const double *re = a.recE + (size_t)c * CH * 64 + lane + 0 * it;
#else
    const double *re = a.recE + group_of(it) * K::Ks * 64 + (size_t)c * CH * 64 + lane;
#endif
#pragma unroll
    for (int j = 0; j < CH; ++j) dst[j] = re[j * 64];
  };
  double A0[CH], A1[CH];
  if (total > 0) {
    issue(0);
    load_a(0, A0);
  }
  double col[PB], gc[PB], gt[PB], gb[PB];
  d4 xa[PB], yv[PB];  // (one chain per pixel group: PB independent chains alternate on the matrix pipe)
  TrainCoreRaw raw[PB];
#pragma unroll
  for (int b = 0; b < PB; ++b) {
    col[b] = gc[b] = gt[b] = gb[b] = 0.0;
    xa[b] = yv[b] = d4{0.0, 0.0, 0.0, 0.0};
  }
  auto phase = [&](int n, const double (&cur)[CH], double (&nxt)[CH]) {
    const int it = n / NCH, c = n - it * NCH;
    const bool live = g0 + it < g1;
    glds_wait();      // chunk n of the shared operands landed (this wave's part); so did cur
#ifndef TR_EXP_WIDE_NOBAR  // (ablation: no chunk barrier; results wrong by construction)
    __syncthreads();  // ... everyone's; the other buffer's readers are done
#endif
    if (n + 1 < total) {
      issue(n + 1);
      load_a(n + 1, nxt);
    }
    if (c == 0) {
#pragma unroll
      for (int b = 0; b < PB; ++b) xa[b] = yv[b] = d4{0.0, 0.0, 0.0, 0.0};
    }
    const double *bl = &sB[n & 1][0][0] + lane;
    if (c < NCH - 1) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
#pragma unroll
        for (int b = 0; b < PB; ++b)
          xa[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[j], bl[(b * CH + j) * 64], xa[b], 0, 0, 0);
        if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);  // (at most 4 PB LDS operands requested ahead: registers)
      }
    } else {
      // The tiles' raw data are requested HERE, in front of the group's last PB x 12 MFMAs, and live only
      // inside this branch.  Requested a chunk earlier (round-5 first form) their registers were shared
      // with the LDS operands of the main branch, and the compiler -- which must not overwrite the
      // destination of a load in flight -- put s_waitcnt vmcnt(0) into the middle of EVERY chunk's MFMAs:
      // each wave then sat out the latency of the prefetch it had just issued, 18 times per group
      // (46 % of the wave cycles parked, profiles/r05_training_k40_pmc.json).
#pragma unroll
      for (int b = 0; b < PB; ++b) train_core_load<LY>(a, group_of(it), p[b], jj, true, raw[b]);
      __builtin_amdgcn_sched_barrier(0);  // (nothing that waits for these loads may move in front of the MFMAs)
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        constexpr int ks0 = (NCH - 1) * CH;
#pragma unroll
        for (int b = 0; b < PB; ++b) {
          if (ks0 + j < K::KsW)  // compile-time
            xa[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[j], bl[(b * CH + j) * 64], xa[b], 0, 0, 0);
          else
            yv[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[j], bl[(b * CH + j) * 64], yv[b], 0, 0, 0);
        }
        if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);
      }
      if (live) {
#pragma unroll
        for (int b = 0; b < PB; ++b) {
          train_core_tile<LY>(a, g0 + it, raw[b], jj, pvalid[b], ob[b], om[b], c_0, tau_0, beta, T_s, exp_tab, xa[b], yv[b], col[b],
                              gc[b], gt[b], gb[b]);
        }
      }
    }
  };
  for (int n = 0; n < total; n += 2) {
    phase(n, A0, A1);
    if (n + 1 < total) phase(n + 1, A1, A0);
  }
  glds_wait();  // (no copy is in flight behind the last chunk; stated for tools/check_vmem_hazard.py)
#pragma unroll
  for (int b = 0; b < PB; ++b)
    if (pvalid[b]) train_core_store(a, pt[b], gs, lane, col[b], gc[b], gt[b], gb[b]);
}

// ------------------------------------------------------------------------------------------
// k_train_finish: ordered sums.  Block 0: f and the three scalar gradients (the longest block -- four
// tree reductions -- so it is dispatched FIRST and runs beside the others instead of being the
// kernel's tail); blocks 1 .. G: pixel p = block - 1 -> dM[p, :] and dlog_omega[p].  The kernel is
// latency-bound (95 % of its wave cycles are waits): the splits of a column are requested four at a
// time and added in split order (the same sum, bit for bit, as one at a time; eight at a time: no
// further gain).  k = 40: 40 -> 30 us; k = 20: 21 -> 22 (profiles/r05_ab_training_finish.txt).
// ------------------------------------------------------------------------------------------
struct TrainFinishArgs {
  TrainDims d;
  const double *M;
  const double *partD;    // [PG][H2][16][Cols]
  const double *partcol;  // [PG][GS][16]
  const double *partsc;   // [PG][GS][3]
  const double *nlogp;    // [16 NQ16]
  const int32_t *flag_in; // not-PD flag of k_train_factor ...
  double *flag_out;       // ... forwarded as a double next to f (one copy back to the host)
  const double *x;        // tau0, beta: the Kim et al. priors enter the gradient (objective.m:59-71)
  double *f, *g;          // g: [G (k+1) + 3]
};

template <int KMAX>
__global__ __launch_bounds__(256) void k_train_finish(TrainFinishArgs a) {
  using K = TrC<KMAX>;
  __shared__ double s_a[K::Cols], s_red[256];
  const TrainDims &D = a.d;
  const int tid = threadIdx.x, k = D.k;
  const int64_t G = D.G;
  if (blockIdx.x > 0) {
    const int64_t p = (int64_t)blockIdx.x - 1, pt = p >> 4;
    const int ps = (int)(p & 15);
    // (only the 16 (W + U) columns that exist: the padding tiles of the last group are never written)
    for (int e = tid; e < 16 * (K::W + K::U); e += 256) {  // A_p (vech) and C_p: sum of the quasar splits, in order
      const double *pd = a.partD + ((pt * D.H2) * 16 + ps) * (int64_t)K::Cols + e;
      double v = 0.0;
      int h = 0;
      for (; h + 4 <= D.H2; h += 4) {
        double t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = pd[(int64_t)(h + u) * 16 * K::Cols];
#pragma unroll
        for (int u = 0; u < 4; ++u) v += t[u];
      }
      for (; h < D.H2; ++h) v += pd[(int64_t)h * 16 * K::Cols];
      s_a[e] = v;
    }
    __syncthreads();
    if (tid < k) {  // dM[p, c] = Sum_e m_p[e] A_p[e, c] - C_p[c]   (:55-56)
      double acc = 0.0;
      for (int e = 0; e < k; ++e) {
        const int i = e > tid ? e : tid, j = e > tid ? tid : e;
        acc = fma(a.M[p + (int64_t)e * G], s_a[i * (i + 1) / 2 + j], acc);
      }
      a.g[p + (int64_t)tid * G] = acc - s_a[K::W * 16 + tid];
    }
    if (tid == 64) {
      double v = 0.0;
      const double *pc = a.partcol + (pt * D.GS) * 16 + ps;
      int gs = 0;
      for (; gs + 8 <= D.GS; gs += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = pc[(gs + u) * 16];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
      }
      for (; gs < D.GS; ++gs) v += pc[gs * 16];
      a.g[G * k + p] = -v;  // :62
    }
    return;
  }
  // scalars: deterministic tree over a fixed assignment
  for (int which = 0; which < 4; ++which) {
    double v = 0.0;
    if (which == 0) {
      for (int64_t q = tid; q < D.nq; q += 256) v += a.nlogp[q];
    } else {
      for (int64_t e = tid; e < D.PG * D.GS; e += 256) v += a.partsc[e * 3 + (which - 1)];
    }
    s_red[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) s_red[tid] += s_red[tid + o];
      __syncthreads();
    }
    if (tid == 0) {
      if (which == 0) *a.flag_out = *a.flag_in ? 1.0 : 0.0;
      if (which == 0) {
        *a.f = s_red[0];
      } else {
        double v = -s_red[0];  // :66, :70, :74
        const double tau_0_mu = 0.0023, tau_0_sigma = 0.0007, beta_mu = 3.65, beta_sigma = 0.21;  // objective.m:59-71
        const TrainScal sc = train_scal(a.x, G, k);
        if (which == 2) v += sc.tau_0 * (sc.tau_0 - tau_0_mu) / (tau_0_sigma * tau_0_sigma);
        if (which == 3) v += sc.beta * (sc.beta - beta_mu) / (beta_sigma * beta_sigma);
        a.g[G * (k + 1) + (which - 1)] = v;
      }
    }
    __syncthreads();
  }
}

}  // namespace gpdla
