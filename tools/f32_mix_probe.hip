// f32_mix_probe.hip -- does v_mfma_f32_16x16x4_f32 (the XDL matrix core: the contraction of the
// fp32 study variant, BASELINE config 5) overlap with fp64 VALU work on its SIMD?  The fp64 MFMA
// does not (tools/fp64_mix_probe.hip).  Two shapes, at 1 and 2 waves per SIMD:
//   fine:  one MFMA : VPER v_fma_f64, interleaved instruction by instruction within a wave
//   burst: NB MFMAs back to back, then NB*VPER v_fma_f64 (the K-step of the sweep: a burst of 14
//          MFMAs behind ~110 VALU instructions) -- overlap then needs the OTHER wave of the SIMD
// Build: hipcc --offload-arch=gfx950 -O3 tools/f32_mix_probe.hip -o /tmp/f32_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool DO_MFMA, bool DO_VALU, int VPER, bool BURST>
__global__ __launch_bounds__(512) void k_mix(double *out, int iters) {
  constexpr int NB = 14;
  f4 acc[NB];
  for (int j = 0; j < NB; ++j) acc[j] = f4{0, 0, 0, 0};
  float x = threadIdx.x * 1e-3f + 1.0f, y = 2.0f - threadIdx.x * 1e-3f;
  double v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 1e-3 + j;
  const double m = 1.0000001, c = 1e-9;
  // a wave of odd index starts half an iteration late, so that two waves of a SIMD are out of phase
  for (int i = 0; i < iters; ++i) {
    if (BURST) {
      if (DO_MFMA) {
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (DO_VALU) {
#pragma unroll
        for (int r = 0; r < NB * VPER / 8; ++r) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fma(v[j], m, c);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (DO_MFMA) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (DO_VALU) {
#pragma unroll
          for (int q = 0; q < VPER; ++q) v[q & 7] = fma(v[q & 7], m, c);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  double r = 0;
  for (int j = 0; j < NB; ++j) r += acc[j][0] + acc[j][3];
  for (int j = 0; j < 8; ++j) r += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <bool M, bool V, int VPER, bool BURST>
float timeit(double *out, int threads) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k_mix<M, V, VPER, BURST><<<256, threads>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_mix<M, V, VPER, BURST><<<256, threads>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int VPER, bool BURST>
void run(double *out) {
  for (int threads : {256, 512}) {
    float a = timeit<true, false, VPER, BURST>(out, threads);
    float b = timeit<false, true, VPER, BURST>(out, threads);
    float c = timeit<true, true, VPER, BURST>(out, threads);
    printf("%s  v_fma_f64 per MFMA = %2d, %d waves/SIMD: MFMA-only %.3f ms | FMA-only %.3f ms | both %.3f ms (sum %.3f max %.3f)\n",
           BURST ? "burst" : "fine ", VPER, threads / 256, a, b, c, a + b, a > b ? a : b);
  }
}

int main() {
  double *out;
  hipMalloc(&out, 256 * 512 * 8);
  run<8, false>(out);
  run<8, true>(out);
  run<4, false>(out);
  run<4, true>(out);
  return 0;
}
