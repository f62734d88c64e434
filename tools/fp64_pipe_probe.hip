// fp64_pipe_probe.hip -- do v_mfma_f64_16x16x4_f64 and fp64 VALU (v_fma_f64) share an execution
// pipe on gfx950?  Two waves per SIMD: even waves run an MFMA-only loop, odd waves a VALU-only
// loop; compare wall time of {MFMA waves alone, VALU waves alone, both together}.  Also the same
// for fp32 VALU and for v_rcp_f64 (transcendental), and the accuracy of the v_rcp_f64 seed.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_pipe_probe.hip -o /tmp/fp64_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

// mode bit0: even waves do MFMA; bit1: odd waves do work of kind `kind`
template <int KIND>  // 0: v_fma_f64, 1: v_fma_f32, 2: v_rcp_f64, 3: int v_mad_u32
__global__ __launch_bounds__(512) void k_probe(double *out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  double r = 0.0;
  if ((wave & 1) == 0) {
    if (mode & 1) {
      d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
      double x = threadIdx.x * 1e-3 + 1.0, y = 2.0 - threadIdx.x * 1e-3;
      for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
      }
      r = a0[0] + a1[1] + a2[2] + a3[3];
    }
  } else if (mode & 2) {
    if (KIND == 0) {
      double v[16];
      for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 1e-3 + j;
      const double m = 1.0000001, c = 1e-9;
      for (int i = 0; i < iters; ++i) {  // 64 FMAs per iteration = 4 MFMAs' worth of pipe time if shared
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = fma(v[j], m, c);
      }
      for (int j = 0; j < 16; ++j) r += v[j];
    } else if (KIND == 1) {
      float v[16];
      for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 1e-3f + j;
      const float m = 1.0000001f, c = 1e-9f;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = fmaf(v[j], m, c);
      }
      for (int j = 0; j < 16; ++j) r += v[j];
    } else if (KIND == 2) {
      double v[8];
      for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 1e-3 + j + 1.5;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_rcp(v[j]);
      }
      for (int j = 0; j < 8; ++j) r += v[j];
    } else {
      unsigned v[16];
      for (int j = 0; j < 16; ++j) v[j] = threadIdx.x + j;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = v[j] * 1664525u + 1013904223u;
      }
      for (int j = 0; j < 16; ++j) r += v[j];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ void k_rcp_acc(const double *x, double *seed, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) seed[i] = __builtin_amdgcn_rcp(x[i]);
}

template <int KIND>
void run(const char *name, double *out) {
  const int iters = 20000, blocks = 256;
  float ms[4];
  for (int mode = 1; mode <= 3; ++mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_probe<KIND><<<blocks, 512>>>(out, iters, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_probe<KIND><<<blocks, 512>>>(out, iters, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms[mode], e0, e1);
  }
  printf("%-12s MFMA-only %.3f ms | %s-only %.3f ms | both %.3f ms  (sum %.3f, max %.3f)\n", name, ms[1],
         name, ms[2], ms[3], ms[1] + ms[2], fmax(ms[1], ms[2]));
}

int main() {
  double *out;
  hipMalloc(&out, 256 * 512 * 8);
  run<0>("v_fma_f64", out);
  run<1>("v_fma_f32", out);
  run<2>("v_rcp_f64", out);
  run<3>("v_mul_u32", out);
  // rcp seed accuracy
  const int n = 1 << 16;
  double *hx = new double[n], *hs = new double[n], *dx, *ds;
  for (int i = 0; i < n; ++i) hx[i] = std::exp(-10.0 + 30.0 * i / n) * (1.0 + 0.37 * ((i * 2654435761u) % 1000) / 1000.0);
  hipMalloc(&dx, n * 8);
  hipMalloc(&ds, n * 8);
  hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  k_rcp_acc<<<n / 256, 256>>>(dx, ds, n);
  hipMemcpy(hs, ds, n * 8, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int i = 0; i < n; ++i) worst = fmax(worst, fabs(hs[i] * hx[i] - 1.0));
  printf("v_rcp_f64 seed: max |x*rcp(x) - 1| = %.3e (2^%.1f)\n", worst, std::log2(worst));
  return 0;
}
