// i8_mfma_probe.hip -- v_mfma_i32_16x16x64_i8 on gfx950: operand lane maps (checked with exact
// integer data), issue rate, and whether it overlaps with fp64 VALU work in the same wave.
// Build: hipcc --offload-arch=gfx950 -O3 tools/i8_mfma_probe.hip -o /tmp/i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));

// A: 16 x 64 (row-major int8), B: 64 x 16 (row-major int8), D: 16 x 16 int32.
// Hypothesis (as bf16 16x16x32 with 2x K): lane l holds A[row l&15][k = 16*(l>>4) + 0..15] and
// B[k = 16*(l>>4) + 0..15][col l&15]; D: col = l&15, row = 4*(l>>4) + reg.
__global__ void k_layout(const int8_t *A, const int8_t *B, int *D) {
  const int l = threadIdx.x;
  i32x4 a, b;
  int8_t *ap = reinterpret_cast<int8_t *>(&a), *bp = reinterpret_cast<int8_t *>(&b);
  for (int j = 0; j < 16; ++j) {
    ap[j] = A[(l & 15) * 64 + 16 * (l >> 4) + j];
    bp[j] = B[(16 * (l >> 4) + j) * 16 + (l & 15)];
  }
  i32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}

template <bool DO_MFMA, bool DO_VALU, int VPER>
__global__ __launch_bounds__(512) void k_mix(double *out, int iters) {
  i32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  i32x4 x = {(int)threadIdx.x, 3, 5, 7}, y = {1, (int)threadIdx.x, 2, 4};
  double v[12];
  for (int j = 0; j < 12; ++j) v[j] = threadIdx.x * 1e-3 + j;
  const double m = 1.0000001, c = 1e-9;
#define VBLOCK                                                                 \
  if (DO_VALU) {                                                               \
    _Pragma("unroll") for (int j = 0; j < VPER; ++j) v[j] = fma(v[j], m, c);   \
  }                                                                            \
  __builtin_amdgcn_sched_barrier(0);
#define MBLOCK(acc)                                                            \
  if (DO_MFMA) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, y, acc, 0, 0, 0); \
  __builtin_amdgcn_sched_barrier(0);
  for (int i = 0; i < iters; ++i) {
    MBLOCK(a0) VBLOCK MBLOCK(a1) VBLOCK MBLOCK(a2) VBLOCK MBLOCK(a3) VBLOCK
  }
  double r = a0[0] + a1[1] + a2[2] + a3[3];
  for (int j = 0; j < 12; ++j) r += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <bool M, bool V, int VPER>
float timeit(double *out, int threads) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k_mix<M, V, VPER><<<256, threads>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_mix<M, V, VPER><<<256, threads>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int VPER>
void run(double *out) {
  for (int threads : {256, 512}) {
    float a = timeit<true, false, VPER>(out, threads);
    float b = timeit<false, true, VPER>(out, threads);
    float c = timeit<true, true, VPER>(out, threads);
    printf("i8 MFMA + %d v_fma_f64 each, %d waves/SIMD: MFMA-only %.3f ms (%.1f cyc/MFMA @2.1GHz) | FMA-only %.3f | interleaved %.3f (sum %.3f max %.3f)\n",
           VPER, threads / 256, a, a * 1e-3 * 2.1e9 / (20000.0 * 4 * (threads / 256)), b, c, a + b, a > b ? a : b);
  }
}

int main() {
  std::vector<int8_t> A(16 * 64), B(64 * 16);
  std::vector<int> D(256), R(256, 0);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 64; ++k) A[i * 64 + k] = (int8_t)((i * 7 + k * 3) % 23 - 11);
  for (int k = 0; k < 64; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (int8_t)((k * 5 + j * 11) % 19 - 9 + (j == 3));
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 64; ++k) R[i * 16 + j] += (int)A[i * 64 + k] * (int)B[k * 16 + j];
  int8_t *dA, *dB; int *dD; double *out;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024); hipMalloc(&out, 256 * 512 * 8);
  hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
  k_layout<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += D[i] != R[i];
  printf("i8 16x16x64 layout check: %d mismatches of 256\n", bad);
  run<4>(out);
  run<12>(out);
  return 0;
}
