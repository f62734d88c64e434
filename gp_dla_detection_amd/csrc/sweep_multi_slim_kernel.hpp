// sweep_multi_slim_kernel.hpp -- the multi-DLA sweep for k <= 20 (multi_dlas/
// process_qsos_multiple_dlas_meanflux.m:340-381) on the SLIM step records of k_sweep_slim: the 13
// vech(m m') tiles are formed inside the sweep.
//
// k_sweep_multi<14, 1, 8, 13, ND> streams 7680-byte pre-expanded records; this kernel is the same
// sweep -- absorption = product of ND gathered profile rows, weights, 13 + 1 MFMA tiles and 6 VALU
// columns per K-step, in-LDS factorisation -- with k_sweep_slim's record handling
// (sweep_slim_kernel.hpp): 896-byte records, the circulant column map, wave w expanding K-step w of
// the NEXT chunk into the tile buffer of the other parity from a private landing zone of doubled M
// rows, one barrier per 8 K-steps.  The profile values of a K-step are requested four K-steps before
// they are multiplied (k_sweep_multi: three; the unrolled chunk is eight steps long, so four slots
// rotate at compile time).  Results are bit-identical to k_sweep_multi's: the same products, the
// same MFMA sequence per column, the same weight arithmetic.
#pragma once
#include "multi_kernels.hpp"
#include "sweep_slim_kernel.hpp"
#include "sweep_split_slim_kernel.hpp"  // vmcnt_imm

namespace gpdla {

constexpr int kMultiSlimLdsDoubles = 2 * kSlimBlock + kSweepWaves * kSlimLand;  // 130 048 B
__host__ __device__ constexpr size_t sweep_multi_slim_lds_doubles() {
  using ES = EpilogueShape<13, 1>;
  const size_t epi = (size_t)kSweepWaves * ES::SPP * ES::stride(16);
  return epi > (size_t)kMultiSlimLdsDoubles ? epi : (size_t)kMultiSlimLdsDoubles;
}

template <int ND>
__global__ __launch_bounds__(512) void k_sweep_multi_slim(SweepMultiArgs a) {
  extern __shared__ double smem[];
  constexpr int WAVES = kSweepWaves, CH = kSlimCH, kAhead = 4;
  static_assert(CH % kAhead == 0, "gather slots rotate at compile time within a chunk");
  const int64_t xj = blockIdx.x >> 3;
  const int64_t ql = 8 * (xj / a.blocks_per_quasar) + (blockIdx.x & 7);
  const int bq = (int)(xj % a.blocks_per_quasar);
  if (ql >= a.nq_sub) return;
  const int64_t q = a.q0 + ql;
  const QuasarMeta m = a.meta[q];
  if (m.status != 0 || a.alive[q] == 0) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int s = lane & 15, jj = lane >> 4;

  double *blocks = smem;                                       // [2 parities]{[8 steps][13 tiles][64], [8 steps][112]}
  double *land = blocks + 2 * kSlimBlock + wave * kSlimLand;   // this wave's landing zone (k_sweep_slim)

  const int64_t slot0 = (int64_t)bq * (WAVES * kSamplesPerWave) + wave * kSamplesPerWave;
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
  const double *rec_base = a.records + m.rec_off * (int64_t)kSlimRec;
  const int nchunks = (m.steps + CH - 1) / CH;

  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t raw_lds = __builtin_amdgcn_readfirstlane(lds_address(blocks + kSlimTileBuf));
  auto issue_chunk = [&](int c) {  // the block's copy of chunk c's raw records (7 KiB)
    glds_chunk<CH * kSlimRec / 128, WAVES>(rec_base + (size_t)c * CH * kSlimRec,
                                           raw_lds + (uint32_t)(c & 1) * (uint32_t)(kSlimBlock * 8), wave_s, lane);
  };
  // this wave's private copy of K-step `wave` of chunk c: its 4 M rows, each laid down twice, and
  // m[16..19] of the 4 pixels (k_sweep_slim)
  const double *land_src = rec_base + (size_t)wave_s * kSlimRec;
  const int src_rows = 4 * kSlimExtras + 16 * jj + 2 * (s & 7);
  const int src_x = kSlimExtras * (lane >> 1) + 4 + 2 * (lane & 1);
  auto issue_private = [&](int c) {
    const double *rec = land_src + (size_t)c * CH * kSlimRec;
    glds16(rec + src_rows, land);
    if (lane < 8) glds16(rec + src_x, land + 4 * 32);
  };
  const double *row = land + 32 * jj + s;                  // m[c] at +0, m[(c + n) & 15] at +n
  const double *bc = land + 4 * 32 + 4 * jj;               // m[16 + r] at +r
  const double *a8 = s < 8 ? row : bc + (slim_pair_i(8, s) - 16);
  const double *b8 = s < 8 ? row + 8 : bc + (slim_pair_j(8, s) - 16);
  struct Operands {
    double mc, o[4], p8;
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
  double *const xd0 = blocks + (size_t)wave_s * kSlimStepTiles + lane;
  const double *const tb0 = blocks + lane;
  const double *const mb0 = blocks + kSlimTileBuf + kSlimExtras * jj;
  const double *const ub0 = blocks + kSlimTileBuf + 4 * kSlimExtras + lane;
  auto pinned = [](const double *p) {
    uint32_t v = lds_address(p);
    asm volatile("" : "+v"(v));
    return (double *)(__attribute__((address_space(3))) double *)(uintptr_t)v;
  };

  issue_private(0);
  issue_chunk(0);

  // profile values: requested kAhead K-steps before they are multiplied (the gather reads HBM)
  const int p_last = 4 * m.steps + jj;  // rows are padded to 4 (steps + 1) entries
  // The loads are issued as inline assembly and waited for by hand (MSLIM_EXP_CLOADS: plain C++
  // loads, for A/B).  With compiler-visible loads every K-step began with s_waitcnt vmcnt(0): the
  // K-steps are separate basic blocks (each is guarded by rn < steps) with untracked LDS-DMA
  // instructions between them, and the compiler's wait-count pass then no longer knows how many of
  // its loads are in flight -- so a gather was waited for ONE K-step after its issue, not four.
  // vmcnt counts in order: when K-step rn consumes its slot, exactly the 3 ND gathers of K-steps
  // rn - 3 .. rn - 1 (or of the priming) are younger; DMA copies in between only make the wait stricter.
  auto gather = [&](int p, double (&r)[ND]) {
#pragma unroll
    for (int j = 0; j < ND; ++j) {
#if defined(MSLIM_EXP_NOGATHER)
      r[j] = 1.0 + 1e-9 * p;  // ablation: no profile loads at all (results wrong by construction)
#elif defined(MSLIM_EXP_CLOADS)
      r[j] = rows[j][p];
#else
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r[j]) : "v"(rows[j] + p));
#endif
    }
  };
  auto gathered = [&](double (&r)[ND]) {  // the values requested kAhead K-steps ago have arrived
#if !defined(MSLIM_EXP_CLOADS) && !defined(MSLIM_EXP_NOGATHER)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kAhead - 1) * ND) : "memory");
#pragma unroll
    for (int j = 0; j < ND; ++j) asm volatile("" : "+v"(r[j]));  // (orders every later use behind the wait)
#endif
  };
  double raw[kAhead][ND];
#pragma unroll
  for (int t = 0; t < kAhead; ++t) gather(min(4 * t + jj, p_last), raw[t]);

  d4 acc[14];
#pragma unroll
  for (int c = 0; c < 14; ++c) acc[c] = d4{0, 0, 0, 0};
  double quad_sum = 0.0, dprod = 1.0;
  double xw[kXW] = {0.0, 0.0}, xu[kXU] = {0.0, 0.0, 0.0, 0.0};
  int dexp = 0;

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

  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks) issue_chunk(c + 1);
    const int par = (c & 1) * kSlimBlock;
    const double *tbuf = pinned(tb0 + par);
    const double *mine0 = pinned(mb0 + par);
    const double *ubuf = pinned(ub0 + par);
    double *xdst = pinned(xd0 + (kSlimBlock - par));
#pragma unroll
    for (int tt = 0; tt < CH; ++tt) {
      const int rn = c * CH + tt;
      // K-steps 0..3 carry the expansion of this wave's K-step of the next chunk (4 + 3 + 3 + 3 tiles); the
      // landing zone is refilled at the top of K-step 4, so that the only memory operations younger than
      // it at the chunk's end are the profile gathers of K-steps 4..7 (see the wait below)
      constexpr int kXS = 4;
      constexpr int kT0[4] = {0, 4, 7, 10}, kT1[4] = {4, 7, 10, 13};
      if (tt == kXS && c + 2 < nchunks) issue_private(c + 2);
      // Past the quasar's last K-step (only the last chunk can be short) the kernel leaves BOTH loops in
      // one jump.  Guarding every K-step by itself, or breaking out of the chunk only, computes the
      // same thing but leaves paths in the control-flow graph -- K-step 0 skipped and K-step 4 run, or a
      // short chunk followed by another chunk -- on which a gather slot is consumed with fewer than
      // (kAhead - 1) ND younger requests behind it: paths that never execute, but that the static check
      // of the hand-counted waits (tools/check_vmem_hazard.py) cannot tell from real ones.
#ifdef MSLIM_EXP_GUARD_EACH  // (A/B: the round-4 form, every K-step guarded by itself)
      if (rn < m.steps)
#else
      if (rn >= m.steps) goto k_loop_done;
#endif
      {
        const double *tl = tbuf + (size_t)tt * kSlimStepTiles;
        const double *mine = mine0 + (size_t)tt * kSlimRec;
        const double2 p01 = *reinterpret_cast<const double2 *>(mine);
        const double2 p23 = *reinterpret_cast<const double2 *>(mine + 2);
        const double py = p01.x, pmu = p01.y, pom = p23.x, pnu = p23.y;
        double bop[14];
#pragma unroll
#ifdef MSLIM_EXP_NOBOP  // ablation: no fragment reads (results wrong by construction)
        for (int cc = 0; cc < 14; ++cc) bop[cc] = 1.0 + 1e-9 * (cc + lane);
        asm volatile("" ::"v"(tl));
#else
        for (int cc = 0; cc < kSlimTilesW; ++cc) bop[cc] = tl[cc * 64];
        bop[13] = ubuf[(size_t)tt * kSlimRec];  // m[0..15] of the 4 pixels in lane order: the u tile
#endif
        // absorption of pixel 4 rn + jj: product of the gathered profiles (multi :342-351); then the
        // request for K-step rn + kAhead into the slot just consumed
        gathered(raw[tt % kAhead]);
        double absorb = raw[tt % kAhead][0];
#pragma unroll
        for (int j = 1; j < ND; ++j) absorb *= raw[tt % kAhead][j];
        if (is_null) absorb = 1.0;
        gather(min(4 * (rn + kAhead) + jj, p_last), raw[tt % kAhead]);
        // weights (the operation order of k_sweep_multi)
        const double r = fma(-absorb, pmu, py);        // multi :355
        const double a2 = absorb * absorb;
        const double d = fma(pom, a2, pnu);            // multi :357, :361
        const double inv_d = fast_rcp(d);
        const double w = a2 * inv_d;
        const double u = absorb * r * inv_d;
        quad_sum = fma(r * r, inv_d, quad_sum);
        dprod *= d;
        dexp += __builtin_amdgcn_frexp_exp(dprod);
        dprod = __builtin_amdgcn_frexp_mant(dprod);
        Operands x;
        if (tt < kXS) expand_load(kT0[tt < kXS ? tt : 0], kT1[tt < kXS ? tt : 0], x);
        __builtin_amdgcn_sched_barrier(0);
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
    // The prefetched raw chunk (issued at the top of this chunk) and this wave's next rows (issued at
    // K-step 4) must have landed -- but NOT the profile values requested in K-steps 4..7 for the next
    // chunk's first four steps: vmcnt counts in order, so "at most (8 - kXS) ND operations outstanding"
    // is exactly "everything up to the landing-zone copy has completed".  (A plain vmcnt(0) here made
    // every wave sit out an HBM gather latency once per chunk.)  In the last chunk nothing was copied.
#ifndef MSLIM_EXP_WAITALL
    __builtin_amdgcn_s_waitcnt(vmcnt_imm((CH - 4) * ND));
    asm volatile("" ::: "memory");
#else
    glds_wait();
#endif
#ifndef MSLIM_EXP_NOBAR  // (ablation: no chunk barrier; results wrong by construction)
    __syncthreads();  // ... everyone's tiles of the next chunk are written; this chunk's buffers are free
#endif
  }

k_loop_done:
  // The last four K-steps requested profile values nobody multiplies.  They were issued by inline
  // assembly, so the compiler does not know that their destination registers are still awaited and
  // hands them to the epilogue: a load landing late would overwrite whatever lives there by then.
  // Drain them before anything else reuses a register.
#if !defined(MSLIM_EXP_CLOADS) && !defined(MSLIM_EXP_NOGATHER)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int t = 0; t < kAhead; ++t) {
#pragma unroll
    for (int j = 0; j < ND; ++j) asm volatile("" : "+v"(raw[t][j]));  // (the registers stay theirs up to here)
  }
#endif
  // a short last chunk has jumped over its chunk-end barrier: the epilogue below reuses the LDS the
  // other waves' last K-steps read (all waves of a block sweep the same quasar and leave the same way)
  glds_wait();
  __syncthreads();
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
#ifdef MSLIM_EXP_NOEPI  // ablation: no factorisation (results wrong by construction)
    sigma = (lane >> 4) + 4 * (2 * p + ((lane >> 3) & 1));
    writer = (lane & 7) == 0;
    double ll = quad_sum + logd_sum + xw[0] + xu[p];
#pragma unroll
    for (int cc = 0; cc < 14; ++cc) ll += acc[cc][2 * p] + acc[cc][2 * p + 1];  // (every accumulator stays live)
#else
    const double ll = slim_factor_pass(acc, xw, xu, p, Eg, lane, a.k, quad_sum, logd_sum, m.n_kept, &sigma, &writer);
#endif
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

}  // namespace gpdla
