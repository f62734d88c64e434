// Probes behind the epilogue of the 20 < k <= 40 sweeps (factor_paired):
//   1. does v_fmac_f64_dpp with row_newbcast:n compute  d += (-s0[lane n of the 16-lane row]) * s1  bit for bit
//      like fma(-s0[16 row + n], s1, d) ?  (and v_mov_b64_dpp: the plain broadcast)
//   2. issue cost of that instruction against plain v_fmac_f64 (four independent chains, one wave)
//   3. dependent-issue cost of v_fma_f64 (ONE chain, one wave) -- what a single dot-product chain pays
//   4. cost of a broadcast ds_read2_b64 (all lanes of a half wave read the same 16 bytes) with one
//      wave and with eight waves of a block issuing it: the LDS return path, not the banks, is the limit
// hipcc --offload-arch=gfx950 -O2 tools/dpp_f64_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int N>
__global__ void semantics(double *out, const double *d, const double *s0, const double *s1) {
  double t = d[threadIdx.x], p = s0[threadIdx.x], r = s1[threadIdx.x];
  asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(p), "v"(r), "n"(N));
  out[N * 64 + threadIdx.x] = t;
  double bc;
  asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(bc) : "v"(p), "n"(N));
  out[(16 + N) * 64 + threadIdx.x] = bc;
}

template <int OP>
__global__ void rate(double *out, unsigned long long *cycles, double a, double b) {
  double v0 = a + threadIdx.x * 1e-9, v1 = b, v2 = a * 0.5, v3 = b * 0.25, p = 1e-3 * threadIdx.x, r = 1.0 + 1e-9 * threadIdx.x;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < 256; ++it) {
    if (OP == 0) { REP16(asm volatile("v_fmac_f64 %0, %4, %5\n v_fmac_f64 %1, %4, %5\n v_fmac_f64 %2, %4, %5\n v_fmac_f64 %3, %4, %5" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(p), "v"(r));) }
    if (OP == 1) { REP16(asm volatile("v_fmac_f64_dpp %0, -%4, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, -%4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, -%4, %5 row_newbcast:9 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, -%4, %5 row_newbcast:13 row_mask:0xf bank_mask:0xf" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(p), "v"(r));) }
    if (OP == 2) { REP16(asm volatile("v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0" : "+v"(v0) : "v"(p), "v"(r));) }
    if (OP == 3) { REP16(asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, -%1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, -%1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, -%1, %2 row_newbcast:13 row_mask:0xf bank_mask:0xf" : "+v"(v0) : "v"(p), "v"(r));) }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

// every half wave reads the same 16 bytes (offset by half): what the pivot-row reads of a factorisation do
__global__ void lds_broadcast(double *out, unsigned long long *cycles) {
  __shared__ double buf[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = 1.0 + i * 1e-6;
  __syncthreads();
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)buf + ((threadIdx.x >> 5) & 15) * 512;
  double acc = 0.0;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < 256; ++it) {
    double a0, a1, b0, b1, c0, c1, d0, d1;
    asm volatile(
        "ds_read2_b64 %0, %8 offset0:0 offset1:1\n ds_read2_b64 %1, %8 offset0:2 offset1:3\n"
        "ds_read2_b64 %2, %8 offset0:4 offset1:5\n ds_read2_b64 %3, %8 offset0:6 offset1:7\n"
        "ds_read2_b64 %4, %8 offset0:8 offset1:9\n ds_read2_b64 %5, %8 offset0:10 offset1:11\n"
        "ds_read2_b64 %6, %8 offset0:12 offset1:13\n ds_read2_b64 %7, %8 offset0:14 offset1:15\n s_waitcnt lgkmcnt(0)"
        : "=v"(*(double2 *)&a0), "=v"(*(double2 *)&b0), "=v"(*(double2 *)&c0), "=v"(*(double2 *)&d0),
          "=v"(*(double2 *)&a1), "=v"(*(double2 *)&b1), "=v"(*(double2 *)&c1), "=v"(*(double2 *)&d1)
        : "v"(addr));
    acc += a0 + b0 + c0 + d0 + a1 + b1 + c1 + d1;
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
  double *out, *din;
  unsigned long long *cyc, h;
  hipMalloc(&out, 1 << 20);
  hipMalloc(&din, 3 * 64 * 8);
  hipMalloc(&cyc, 8);
  double hd[3 * 64], ho[32 * 64];
  for (int i = 0; i < 64; ++i) {
    hd[i] = 0.3 + 0.01 * i;
    hd[64 + i] = 1.0 / 3.0 + 0.1 * i;
    hd[128 + i] = 2.0 / 7.0 - 0.01 * i;
  }
  hipMemcpy(din, hd, sizeof hd, hipMemcpyHostToDevice);
#define SEM(N) hipLaunchKernelGGL(semantics<N>, dim3(1), dim3(64), 0, 0, out, din, din + 64, din + 128);
  SEM(0) SEM(1) SEM(2) SEM(3) SEM(4) SEM(5) SEM(6) SEM(7) SEM(8) SEM(9) SEM(10) SEM(11) SEM(12) SEM(13) SEM(14) SEM(15)
  hipMemcpy(ho, out, sizeof ho, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int n = 0; n < 16; ++n)
    for (int l = 0; l < 64; ++l) {
      const double want = fma(-hd[64 + (l & ~15) + n], hd[128 + l], hd[l]);
      if (memcmp(&want, &ho[n * 64 + l], 8)) ++bad;
    }
  printf("v_fmac_f64_dpp -s0 row_newbcast:n == fma(-s0[16 row + n], s1, d): %s (%d of 1024 lanes differ)\n", bad ? "NO" : "bit for bit", bad);
  bad = 0;
  for (int n = 0; n < 16; ++n)
    for (int l = 0; l < 64; ++l)
      if (memcmp(&hd[64 + (l & ~15) + n], &ho[(16 + n) * 64 + l], 8)) ++bad;
  printf("v_mov_b64_dpp row_newbcast:n == s0[16 row + n]: %s (%d of 1024 lanes differ)\n", bad ? "NO" : "yes", bad);
  const char *names[] = {"v_fmac_f64 x4 chains", "v_fmac_f64_dpp x4 chains", "v_fma_f64 one chain", "v_fmac_f64_dpp one chain"};
#define RUN(OP)                                                                 \
  hipLaunchKernelGGL(rate<OP>, dim3(1), dim3(64), 0, 0, out, cyc, 1.25, 0.75);  \
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                 \
  printf("%-28s %.2f cycles/instr (one wave)\n", names[OP], (double)h / (256.0 * 64.0));
  RUN(0) RUN(1) RUN(2) RUN(3)
  for (int waves : {1, 2, 4, 8}) {
    hipLaunchKernelGGL(lds_broadcast, dim3(1), dim3(64 * waves), 0, 0, out, cyc);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("broadcast ds_read2_b64, %d wave(s) per CU: %.2f cycles/instr per wave, %.1f bytes/cycle delivered per CU\n", waves,
           (double)h / (256.0 * 8.0), 256.0 * 8.0 * 1024.0 * waves / (double)h);
  }
  return 0;
}
