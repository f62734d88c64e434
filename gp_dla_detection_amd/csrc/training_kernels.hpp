// training_kernels.hpp -- the training objective of the GP quasar model on the GPU
// (SURVEY.md section 8f, row N3): spectrum_loss.m:14-76 summed over the training set as
// objective.m:41-57 does, value and gradient.
//
// Same Woodbury algebra as the inference sweep, one 256-thread block per training quasar:
//   d = nu + omega2 (1 - exp(-tau0 (1+z)^beta) + c0)^2,  B = I + M' D^-1 M = L L'
//   K^-1 y = D^-1 (y - M B^-1 M' D^-1 y)
//   K^-1 M = D^-1 M B^-1            (spectrum_loss.m:55 with C M = I - B^-1 substituted)
//   diag K^-1 = d^-1 - d^-2 m_p' B^-1 m_p                        (:59)
// and the gradients of :56-74 accumulated into g with fp64 atomics (25 k addresses, 10^4 adders
// each: the sums are order-dependent in the last bits, which an L-BFGS caller does not see).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sweep_kernels.hpp"

namespace gpdla {

struct TrainingArgs {
  int64_t nq, G;          // quasars, rest-frame pixels
  int32_t k;
  const double *flux;     // [nq][G] (quasar-major), NaN = missing pixel (objective.m:42)
  const double *lya_1pz;  // [nq][G]
  const double *noise;    // [nq][G]
  const double *M;        // [G x k] column-major (the first G k entries of x)
  const double *omega2;   // [G] exp(2 log omega), objective.m:29
  double c_0, tau_0, beta;
  double *f;              // scalar accumulator
  double *g;              // [G (k+1) + 3] accumulator, layout of x (objective.m:73)
  int32_t *not_pd;        // set to 1 if some B is not positive definite (chol would throw, :42)
};

// dynamic LDS: dinv[G] | y[G] | kiy[G] | L[k*k] | Binv[k*k] | t[k] | z[k] | gvec[k] | red[8]
__global__ __launch_bounds__(256) void k_training_loss(TrainingArgs a) {
  extern __shared__ double sm[];
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = a.k;
  const int64_t G = a.G;
  double *s_dinv = sm, *s_y = s_dinv + G, *s_kiy = s_y + G;
  double *s_L = s_kiy + G, *s_Binv = s_L + k * k, *s_t = s_Binv + k * k, *s_z = s_t + k,
         *s_g = s_z + k, *s_red = s_g + k;
  const double *F = a.flux + (int64_t)q * G, *Z = a.lya_1pz + (int64_t)q * G,
               *V = a.noise + (int64_t)q * G;
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
  // ---- B = I + M' D^-1 M (lower triangle) and t = M' D^-1 y (:40-41) ----
  const int nb = k * (k + 1) / 2;
  for (int e = tid; e < nb + k; e += 256) {
    double acc = 0.0;
    if (e < nb) {
      int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
      while ((i + 1) * (i + 2) / 2 <= e) ++i;
      while (i * (i + 1) / 2 > e) --i;
      const int j = e - i * (i + 1) / 2;
      const double *Mi = a.M + (int64_t)i * G, *Mj = a.M + (int64_t)j * G;
      for (int64_t p = 0; p < G; ++p) acc = fma(Mi[p] * s_dinv[p], Mj[p], acc);
      if (i == j) acc += 1.0;
      s_L[i * k + j] = acc;
    } else {
      const int i = e - nb;
      const double *Mi = a.M + (int64_t)i * G;
      for (int64_t p = 0; p < G; ++p) acc = fma(Mi[p], s_dinv[p] * s_y[p], acc);
      s_t[i] = acc;
    }
  }
  __syncthreads();
  // ---- Cholesky B = L L' in place (lower), thread 0 (:42) ----
  if (tid == 0) {
    double logdiag = 0.0;
    bool pd = true;
    for (int i = 0; i < k; ++i)
      for (int j = 0; j <= i; ++j) {
        double sum = s_L[i * k + j];
        for (int mm = 0; mm < j; ++mm) sum = fma(-s_L[i * k + mm], s_L[j * k + mm], sum);
        if (i == j) {
          pd = pd && (sum > 0.0);
          const double lii = sqrt(sum);
          logdiag += log(lii);
          s_L[i * k + i] = lii;
        } else {
          s_L[i * k + j] = sum / s_L[j * k + j];
        }
      }
    s_red[4] = logdiag;
    s_red[5] = pd ? 0.0 : 1.0;
  }
  __syncthreads();
  if (s_red[5] != 0.0) {
    if (tid == 0) *a.not_pd = 1;
    return;
  }
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
  // ---- pass 3: gradients (:55-74) ----
  double gc = 0.0, gt = 0.0, gb = 0.0;
  for (int64_t p = tid; p < G; p += 256) {
    const double dinv = s_dinv[p];
    if (dinv == 0.0) continue;  // missing pixel
    double mrow[GPDLA_MAX_K];
    for (int c = 0; c < k; ++c) mrow[c] = a.M[p + (int64_t)c * G];
    const double kiy = s_kiy[p];
    double mBm = 0.0;
    for (int c = 0; c < k; ++c) {
      double mb = 0.0;  // (m_p' B^-1)_c
      for (int e = 0; e < k; ++e) mb = fma(mrow[e], s_Binv[e * k + c], mb);
      mBm = fma(mb, mrow[c], mBm);
      const double dM = dinv * mb - kiy * s_g[c];                     // :55-56
      atomicAdd(a.g + p + (int64_t)c * G, dM);
    }
    const double diag = dinv - dinv * dinv * mBm;                     // :59
    const double od = a.tau_0 * pow(Z[p], a.beta);
    const double ab = exp(-od);
    const double sf = 1 - ab + a.c_0;
    const double om = a.omega2[p];
    const double core = kiy * kiy - diag;
    atomicAdd(a.g + G * k + p, -(om * (sf * sf)) * core);             // :62
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
    atomicAdd(a.f, nlog_p);
    atomicAdd(a.g + G * (k + 1), gc);
    atomicAdd(a.g + G * (k + 1) + 1, gt);
    atomicAdd(a.g + G * (k + 1) + 2, gb);
  }
}

// omega2 = exp(2 log omega), objective.m:29
__global__ void k_training_omega2(const double *log_omega, int64_t G, double *omega2) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < G) omega2[p] = exp(2 * log_omega[p]);
}

}  // namespace gpdla
