// fp64_mix_probe.hip -- can ONE wave overlap its own v_mfma_f64_16x16x4_f64 with its own fp64
// VALU work?  Every wave runs the same loop; per iteration NM MFMAs (4 independent accumulators)
// and NV independent v_fma_f64, interleaved one MFMA : NV/NM FMAs.  Compare with MFMA-only and
// FMA-only loops at 1 and 2 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_mix_probe.hip -o /tmp/fp64_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <bool DO_MFMA, bool DO_VALU, int VPER, int KIND>
__global__ __launch_bounds__(512) void k_mix(double *out, int iters) {
  d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double x = threadIdx.x * 1e-3 + 1.0, y = 2.0 - threadIdx.x * 1e-3;
  double v[12];
  float vf[24];
  unsigned vi[24];
  for (int j = 0; j < 12; ++j) v[j] = threadIdx.x * 1e-3 + j;
  for (int j = 0; j < 24; ++j) { vf[j] = threadIdx.x * 1e-3f + j; vi[j] = threadIdx.x + j; }
  const double m = 1.0000001, c = 1e-9;
  const float mf = 1.0000001f, cf = 1e-9f;
  // KIND 0: VPER v_fma_f64 (4 cycles each); 1: 2*VPER v_fma_f32 (2 cycles each); 2: 2*VPER int mads
#define VBLOCK                                                       \
  if (DO_VALU) {                                                     \
    if (KIND == 0) { _Pragma("unroll") for (int j = 0; j < VPER; ++j) v[j] = fma(v[j], m, c); } \
    else if (KIND == 1) { _Pragma("unroll") for (int j = 0; j < 2 * VPER; ++j) vf[j] = fmaf(vf[j], mf, cf); } \
    else { _Pragma("unroll") for (int j = 0; j < 2 * VPER; ++j) vi[j] = vi[j] * 1664525u + 1013904223u; } \
  }                                                                  \
  __builtin_amdgcn_sched_barrier(0);
#define MBLOCK(acc)                                                  \
  if (DO_MFMA) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0); \
  __builtin_amdgcn_sched_barrier(0);
  for (int i = 0; i < iters; ++i) {
    MBLOCK(a0) VBLOCK MBLOCK(a1) VBLOCK MBLOCK(a2) VBLOCK MBLOCK(a3) VBLOCK
  }
  double r = a0[0] + a1[1] + a2[2] + a3[3];
  for (int j = 0; j < 12; ++j) r += v[j];
  for (int j = 0; j < 24; ++j) r += vf[j] + vi[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <bool M, bool V, int VPER, int KIND>
float timeit(double *out, int threads) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k_mix<M, V, VPER, KIND><<<256, threads>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_mix<M, V, VPER, KIND><<<256, threads>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int VPER, int KIND>
void run(double *out) {
  const char *names[3] = {"v_fma_f64", "v_fma_f32 x2", "int mad x2"};
  for (int threads : {256, 512}) {
    float a = timeit<true, false, VPER, KIND>(out, threads);
    float b = timeit<false, true, VPER, KIND>(out, threads);
    float c = timeit<true, true, VPER, KIND>(out, threads);
    printf("%-13s VALU per MFMA = %2d, %d waves/SIMD: MFMA-only %.3f ms | FMA-only %.3f ms | interleaved %.3f ms (sum %.3f max %.3f)\n",
           names[KIND], VPER, threads / 256, a, b, c, a + b, a > b ? a : b);
  }
}

int main() {
  double *out;
  hipMalloc(&out, 256 * 512 * 8);
  run<4, 0>(out);
  run<12, 0>(out);
  run<12, 1>(out);
  run<12, 2>(out);
  return 0;
}
