// mfma_f64_probe.hip -- measures v_mfma_f64_16x16x4_f64 issue cost on gfx950 and checks its
// operand/result lane maps with exact integer data (A = asymmetric, B = asymmetric).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k_rate(double *out, long long *cycles, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3 + 1.0, b = 2.0 - threadIdx.x * 1e-3;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

__global__ void k_layout(const double *A /*16x4 row-major*/, const double *B /*4x16 row-major*/, double *D /*16x16*/) {
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

int main() {
  double *out;
  long long *cyc;
  hipMalloc(&out, 1024 * 1024 * 8);
  hipMalloc(&cyc, 4096 * 8);
  // layout check
  std::vector<double> A(64), B(64), D(256), R(256, 0.0);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = i * 5 + k * 3 + 1;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = k * 7 + j * 2 + 1 + (j == 3 ? 11 : 0);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
  double *dA, *dB, *dD;
  hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dD, 256 * 8);
  hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
  k_layout<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += (D[i] != R[i]);
  printf("layout check: %d mismatches of 256\n", bad);
  // rate: one wave per SIMD (256 threads per block, 1 block per CU), and 2 waves per SIMD
  const int iters = 2000;
  for (int nacc : {1, 4, 16}) {
    for (int threads : {64, 256, 512}) {
      int blocks = 256;
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&]() {
        if (nacc == 1) k_rate<1><<<blocks, threads>>>(out, cyc, iters);
        else if (nacc == 4) k_rate<4><<<blocks, threads>>>(out, cyc, iters);
        else k_rate<16><<<blocks, threads>>>(out, cyc, iters);
      };
      launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      long long c0;
      hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
      double mfmas = (double)blocks * (threads / 64) * iters * nacc;
      double tflops = mfmas * 2048.0 / (ms * 1e-3) / 1e12;
      printf("nacc=%2d threads/block=%3d: %.3f ms, %.1f TFLOP/s fp64, s_memtime ticks per MFMA per wave = %.1f\n",
             nacc, threads, ms, tflops, (double)c0 / (iters * nacc));
    }
  }
  return 0;
}
