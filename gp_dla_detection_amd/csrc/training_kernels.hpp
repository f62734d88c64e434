// training_kernels.hpp -- the training objective of the GP quasar model on the GPU
// (SURVEY.md section 8f, row N3): spectrum_loss.m:14-76 summed over the training set as
// objective.m:41-57 does, value and gradient.
//
// Same Woodbury algebra as the inference sweep, one 256-thread block per training quasar:
//   d = nu + omega2 (1 - exp(-tau0 (1+z)^beta) + c0)^2,  B = I + M' D^-1 M = L L'
//   K^-1 y = D^-1 (y - M B^-1 M' D^-1 y)
//   K^-1 M = D^-1 M B^-1            (spectrum_loss.m:55 with C M = I - B^-1 substituted)
//   diag K^-1 = d^-1 - d^-2 m_p' B^-1 m_p                        (:59)
// and the gradients of :56-74.  Deterministic since round 3: a block is a SLOT that walks quasars
// slot, slot + S, ... in order and adds each one's gradient into the slot's own copy of g (every
// address is touched by one thread of the block, always the same one), and k_training_reduce sums
// the slots in order -- round 1 accumulated g with fp64 atomics, whose order changed the last bits
// from run to run.  Used for 20 < k <= 40 (the matrix-core path of training_mfma_kernels.hpp
// takes k <= 20) and as its cross-check (GPDLA_TRAIN_LEGACY=1).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sweep_kernels.hpp"

namespace gpdla {

struct TrainingArgs {
  int64_t nq, G, ld;      // quasars, rest-frame pixels, row stride of flux / lya_1pz / noise
  int32_t k;
  const double *flux;     // [nq][ld] (quasar-major), NaN = missing pixel (objective.m:42)
  const double *lya_1pz;  // [nq][ld]
  const double *noise;    // [nq][ld]
  const double *M;        // [G x k] column-major (the first G k entries of x)
  const double *omega2;   // [G] exp(2 log omega), objective.m:29
  double c_0, tau_0, beta;
  double *slots;          // [num_slots][G (k+1) + 4]: per slot g in the layout of x (objective.m:73), then f; zeroed by the host
  int32_t *not_pd;        // set to 1 if some B is not positive definite (chol would throw, :42)
};

// dynamic LDS: dinv[G] | y[G] | kiy[G] | L[k*k] | Binv[k*k] | t[k] | z[k] | gvec[k] | red[8] |
//              stage[2 k kTrainChunkStride]  (rows of M for a chunk of pixels; see the B build and pass 3)
constexpr int kTrainChunk = 128;        // pixels per staged chunk in the B build
constexpr int kTrainChunkStride = 129;  // odd row stride: threads read different rows at the same pixel
__host__ __device__ constexpr size_t training_lds_doubles(int64_t G, int k) {
  // stage: B build 2 k kTrainChunkStride, pass 3 k * 256; the larger of the two
  return (size_t)(3 * G + 2 * k * k + 3 * k + 8) +
         (size_t)(2 * k * kTrainChunkStride > k * 256 ? 2 * k * kTrainChunkStride : k * 256);
}
__device__ __forceinline__ void training_one_quasar(const TrainingArgs &a, const int q, double *sm, double *gs) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = a.k;
  const int64_t G = a.G;
  double *s_dinv = sm, *s_y = s_dinv + G, *s_kiy = s_y + G;
  double *s_L = s_kiy + G, *s_Binv = s_L + k * k, *s_t = s_Binv + k * k, *s_z = s_t + k,
         *s_g = s_z + k, *s_red = s_g + k, *s_stage = s_red + 8;
  const double *F = a.flux + (int64_t)q * a.ld, *Z = a.lya_1pz + (int64_t)q * a.ld,
               *V = a.noise + (int64_t)q * a.ld;
  const double log_2pi = 1.83787706640934534;  // spectrum_loss.m:17

  auto block_sum = [&](double v) -> double {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
  };

  // ---- pass 1: diagonal, its log, valid-pixel count (spectrum_loss.m:22-31) ----
  double logd = 0.0, cnt = 0.0;
  for (int64_t p = tid; p < G; p += 256) {
    const double y = F[p];
    double dinv = 0.0, yy = 0.0;
    if (!isnan(y)) {
      const double od = a.tau_0 * pow(Z[p], a.beta);   // :22
      const double sf = 1 - exp(-od) + a.c_0;           // :23, :26
      const double d = V[p] + a.omega2[p] * (sf * sf);  // :27, :29
      dinv = 1.0 / d;
      yy = y;
      logd += log(d);
      cnt += 1.0;
    }
    s_dinv[p] = dinv;
    s_y[p] = yy;
  }
  logd = block_sum(logd);
  cnt = block_sum(cnt);
  if (cnt == 0.0) return;  // quasar with no valid pixel contributes nothing
#if defined(GPDLA_TRAIN_STOP) && GPDLA_TRAIN_STOP == 1
  return;  // timing experiment (tools/flag_probe.sh): stop before this phase
#endif
  // ---- B = I + M' D^-1 M (lower triangle) and t = M' D^-1 y (:40-41) ----
  // Thread el owns one entry (i, j) of the triangle (or one entry of t).  The pixel axis is walked
  // in chunks staged through LDS -- rows of M and of D^-1 M, loaded coalesced -- instead of every
  // thread streaming two columns of M from memory on its own.
  const int nb = k * (k + 1) / 2;
  constexpr int kSlots = (GPDLA_MAX_K * (GPDLA_MAX_K + 1) / 2 + GPDLA_MAX_K + 255) / 256;  // entries per thread
  int bi[kSlots], bj[kSlots];
#pragma unroll
  for (int sl = 0; sl < kSlots; ++sl) {
    const int el = tid + 256 * sl;
    bi[sl] = bj[sl] = 0;
    if (el < nb) {
      int i = (int)((sqrt(8.0 * el + 1.0) - 1.0) * 0.5);
      while ((i + 1) * (i + 2) / 2 <= el) ++i;
      while (i * (i + 1) / 2 > el) --i;
      bi[sl] = i;
      bj[sl] = el - i * (i + 1) / 2;
    } else if (el < nb + k) {
      bi[sl] = el - nb;
    }
  }
  {
    double *s_mc = s_stage, *s_wc = s_stage + (size_t)k * kTrainChunkStride;  // M and D^-1 M rows of the chunk
    double acc[kSlots];
#pragma unroll
    for (int sl = 0; sl < kSlots; ++sl) acc[sl] = 0.0;
    for (int64_t base = 0; base < G; base += kTrainChunk) {
      const int len = (int)min((int64_t)kTrainChunk, G - base);
      __syncthreads();  // previous chunk consumed
      {  // all of this thread's loads of the chunk first (k/2 of them), then the LDS writes
        constexpr int kLoads = GPDLA_MAX_K * kTrainChunk / 256;
        double mv[kLoads];
#pragma unroll
        for (int u = 0; u < kLoads; ++u) {
          const int idx = tid + 256 * u;
          const int e = idx / kTrainChunk, pp = idx % kTrainChunk;
          mv[u] = (e < k && pp < len) ? a.M[base + pp + (int64_t)e * G] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kLoads; ++u) {
          const int idx = tid + 256 * u;
          const int e = idx / kTrainChunk, pp = idx % kTrainChunk;
          if (e < k) {
            s_mc[e * kTrainChunkStride + pp] = mv[u];
            s_wc[e * kTrainChunkStride + pp] = pp < len ? mv[u] * s_dinv[base + pp] : 0.0;
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int sl = 0; sl < kSlots; ++sl) {
        const int el = tid + 256 * sl;
        if (el < nb) {
          const double *wi = s_wc + bi[sl] * kTrainChunkStride, *mj = s_mc + bj[sl] * kTrainChunkStride;
#pragma unroll 16
          for (int pp = 0; pp < len; ++pp) acc[sl] = fma(wi[pp], mj[pp], acc[sl]);
        } else if (el < nb + k) {
          const double *wi = s_wc + bi[sl] * kTrainChunkStride;
#pragma unroll 16
          for (int pp = 0; pp < len; ++pp) acc[sl] = fma(wi[pp], s_y[base + pp], acc[sl]);
        }
      }
    }
#pragma unroll
    for (int sl = 0; sl < kSlots; ++sl) {
      const int el = tid + 256 * sl;
      if (el < nb) s_L[bi[sl] * k + bj[sl]] = bi[sl] == bj[sl] ? acc[sl] + 1.0 : acc[sl];
      else if (el < nb + k) s_t[bi[sl]] = acc[sl];
    }
  }
  __syncthreads();
#if defined(GPDLA_TRAIN_STOP) && GPDLA_TRAIN_STOP == 2
  return;  // timing experiment (tools/flag_probe.sh): stop before this phase
#endif
  // ---- Cholesky B = L L' in place (lower) (:42): right-looking, lane i of wave 0 owns row i ----
  if (wave == 0) {
    double logdiag = 0.0;
    bool pd = true;
    for (int j = 0; j < k; ++j) {
      const double djj = s_L[j * k + j];
      pd = pd && (djj > 0.0);
      const double ljj = sqrt(djj);
      logdiag += log(ljj);
      __builtin_amdgcn_wave_barrier();  // everyone has read the pivot before it is overwritten
      if (lane == j) s_L[j * k + j] = ljj;
      if (lane > j && lane < k) s_L[lane * k + j] /= ljj;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane > j && lane < k) {  // trailing update of this lane's row
        const double lij = s_L[lane * k + j];
        for (int c = j + 1; c <= lane; ++c) s_L[lane * k + c] = fma(-lij, s_L[c * k + j], s_L[lane * k + c]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
      s_red[4] = logdiag;
      s_red[5] = pd ? 0.0 : 1.0;
    }
  }
  __syncthreads();
  if (s_red[5] != 0.0) {
    if (tid == 0) *a.not_pd = 1;
    return;
  }
#if defined(GPDLA_TRAIN_STOP) && GPDLA_TRAIN_STOP == 3
  return;  // timing experiment (tools/flag_probe.sh): stop before this phase
#endif
  // ---- B^-1 column by column (threads 0..k-1), z = B^-1 t (thread k) ----
  if (tid <= k) {
    double w[GPDLA_MAX_K];
    for (int i = 0; i < k; ++i) {  // forward: L w = rhs
      double r = tid < k ? (i == tid ? 1.0 : 0.0) : s_t[i];
      for (int mm = 0; mm < i; ++mm) r = fma(-s_L[i * k + mm], w[mm], r);
      w[i] = r / s_L[i * k + i];
    }
    for (int i = k - 1; i >= 0; --i) {  // backward: L' x = w
      double r = w[i];
      for (int mm = i + 1; mm < k; ++mm) r = fma(-s_L[mm * k + i], w[mm], r);
      w[i] = r / s_L[i * k + i];
    }
    for (int i = 0; i < k; ++i) {
      if (tid < k) s_Binv[i * k + tid] = w[i];
      else s_z[i] = w[i];
    }
  }
  __syncthreads();
#if defined(GPDLA_TRAIN_STOP) && GPDLA_TRAIN_STOP == 4
  return;  // timing experiment (tools/flag_probe.sh): stop before this phase
#endif
  // ---- pass 2: K^-1 y (:46), y' K^-1 y ----
  double quad = 0.0;
  for (int64_t p = tid; p < G; p += 256) {
    double mz = 0.0;
    for (int c = 0; c < k; ++c) mz = fma(a.M[p + (int64_t)c * G], s_z[c], mz);
    const double kiy = s_dinv[p] * (s_y[p] - mz);
    s_kiy[p] = kiy;
    quad = fma(s_y[p], kiy, quad);
  }
  quad = block_sum(quad);
  // gvec = M' K^-1 y (the K_inv_y' * M of :56)
  for (int c = tid; c < k; c += 256) {
    double acc = 0.0;
    const double *Mc = a.M + (int64_t)c * G;
    for (int64_t p = 0; p < G; ++p) acc = fma(s_kiy[p], Mc[p], acc);
    s_g[c] = acc;
  }
  __syncthreads();
#if defined(GPDLA_TRAIN_STOP) && GPDLA_TRAIN_STOP == 5
  return;  // timing experiment (tools/flag_probe.sh): stop before this phase
#endif
  // ---- pass 3: gradients (:55-74) ----
  // m_p' B^-1 for this thread's pixel: the row m_p is staged in LDS (column tid of s_mrow), so the
  // k x k loop reads it with compile-time-free indices from LDS instead of a run-time indexed
  // private array (which lands in scratch memory).
  double gc = 0.0, gt = 0.0, gb = 0.0;
  double *s_mrow = s_stage;  // [k][256]
  __syncthreads();           // the B build's use of the stage region is long over; be explicit
  for (int64_t base = 0; base < G; base += 256) {
    const int64_t p = base + tid;
    const double dinv = p < G ? s_dinv[p] : 0.0;
    if (dinv == 0.0) continue;  // missing pixel (or past the end): no block-wide sync below
    {  // this pixel's row of M: all loads in flight at once, then the LDS writes
      double mv[GPDLA_MAX_K];
#pragma unroll
      for (int c = 0; c < GPDLA_MAX_K; ++c) mv[c] = c < k ? a.M[p + (int64_t)c * G] : 0.0;
#pragma unroll
      for (int c = 0; c < GPDLA_MAX_K; ++c)
        if (c < k) s_mrow[c * 256 + tid] = mv[c];
    }
    const double kiy = s_kiy[p];
    double mBm = 0.0;
    for (int c = 0; c < k; ++c) {
      double mb = 0.0;  // (m_p' B^-1)_c
#pragma unroll 4
      for (int e = 0; e < k; ++e) mb = fma(s_mrow[e * 256 + tid], s_Binv[e * k + c], mb);
      mBm = fma(mb, s_mrow[c * 256 + tid], mBm);
      const double dM = dinv * mb - kiy * s_g[c];                     // :55-56
      gs[p + (int64_t)c * G] += dM;
    }
    const double diag = dinv - dinv * dinv * mBm;                     // :59
    const double od = a.tau_0 * pow(Z[p], a.beta);
    const double ab = exp(-od);
    const double sf = 1 - ab + a.c_0;
    const double om = a.omega2[p];
    const double core = kiy * kiy - diag;
    gs[G * k + p] += -(om * (sf * sf)) * core;                        // :62
    double da = a.c_0 * om * sf;                                      // :65
    gc -= core * da;                                                  // :66
    da = om * sf * od * ab;                                           // :69
    gt -= core * da;                                                  // :70
    da = da * log(Z[p]) * a.beta;                                     // :73
    gb -= core * da;                                                  // :74
  }
  gc = block_sum(gc);
  gt = block_sum(gt);
  gb = block_sum(gb);
  if (tid == 0) {
    const double nlog_p = 0.5 * (quad + logd + 2 * s_red[4] + cnt * log_2pi);  // :48-52
    gs[G * (k + 1) + 3] += nlog_p;
    gs[G * (k + 1)] += gc;
    gs[G * (k + 1) + 1] += gt;
    gs[G * (k + 1) + 2] += gb;
  }
}

__global__ __launch_bounds__(256) void k_training_loss(TrainingArgs a) {
  extern __shared__ double sm[];
  double *gs = a.slots + (int64_t)blockIdx.x * (a.G * (a.k + 1) + 4);
  for (int64_t q = blockIdx.x; q < a.nq; q += gridDim.x) {
    training_one_quasar(a, (int)q, sm, gs);
    __syncthreads();  // the LDS arrays are reused by the next quasar
  }
}

// out[e] = Sum over the slots, in slot order, of slots[s][e], e < n (g, then f behind it)
__global__ void k_training_reduce(const double *slots, int num_slots, int64_t n, double *out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  double v = 0.0;
  for (int s = 0; s < num_slots; ++s) v += slots[(int64_t)s * n + e];
  out[e] = v;
}

// omega2 = exp(2 log omega), objective.m:29
__global__ void k_training_omega2(const double *log_omega, int64_t G, double *omega2) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < G) omega2[p] = exp(2 * log_omega[p]);
}

}  // namespace gpdla
