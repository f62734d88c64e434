// How fast does the chip take stores?  (k_profiles: 15.6 GB in 4.2-5.1 ms.)
// Each wave writes 16-pixel tiles of 64 profile rows the way k_profiles does -- store instruction e
// writes 128 bytes of rows 4e .. 4e+3, the rows `stride` doubles apart -- or, for comparison, the same
// bytes as one contiguous span per wave; temporal and non-temporal stores.
// hipcc --offload-arch=gfx950 -O2 tools/hbm_write_probe.hip -o /tmp/hbm_write_probe && /tmp/hbm_write_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <bool ROWS, bool NT>
__global__ __launch_bounds__(256) void writer(double *out, long stride, long rows_per_table, int tiles) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long w_global = (long)blockIdx.x * 4 + wave;          // this wave's 64 rows
  const double v = 1.0 + lane * 1e-9;
  if (ROWS) {
    double *base = out + (w_global * 64) * stride + (lane & 15);
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        double *p = base + (long)(4 * e + (lane >> 4)) * stride + 16 * t;
        if (NT) __builtin_nontemporal_store(v, p); else *p = v;
      }
    }
  } else {
    double *base = out + w_global * 64 * stride + lane;       // the wave's rows as one span
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        double *p = base + ((long)t * 16 + e) * 64;
        if (NT) __builtin_nontemporal_store(v, p); else *p = v;
      }
    }
  }
}

int main() {
  const long stride = 1504, rows = 64L * 2 * 10000;  // config 4: 64 quasars x 2 kinds x 10^4 samples
  const int tiles = 94;                                // 1504 / 16
  double *buf;
  const size_t bytes = (size_t)rows * stride * 8;
  if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int blocks = (int)(rows / 64 / 4);
#define RUN(R, N, name)                                                                            \
  for (int rep = 0; rep < 3; ++rep) {                                                              \
    hipEventRecord(a);                                                                             \
    hipLaunchKernelGGL((writer<R, N>), dim3(blocks), dim3(256), 0, 0, buf, stride, rows, tiles);   \
    hipEventRecord(b);                                                                             \
    hipEventSynchronize(b);                                                                        \
    float ms;                                                                                      \
    hipEventElapsedTime(&ms, a, b);                                                                \
    printf("%-52s %.2f ms  %.2f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12);                    \
  }
  RUN(true, true, "rows 12 KB apart, 128-byte pieces, non-temporal")
  RUN(true, false, "rows 12 KB apart, 128-byte pieces, temporal")
  RUN(false, true, "one contiguous span per wave, non-temporal")
  RUN(false, false, "one contiguous span per wave, temporal")
  return 0;
}
