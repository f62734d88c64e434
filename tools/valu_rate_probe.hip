// Issue cost (cycles per wave64 instruction, one wave per SIMD) of the fp64 VALU instructions the
// sweep kernel's K-step uses.  hipcc --offload-arch=gfx950 -O2 tools/valu_rate_probe.hip -o /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void probe(double *out, unsigned long long *cycles, double a, double b) {
  double v0 = a + threadIdx.x * 1e-9, v1 = b, v2 = a * 0.5, v3 = b * 0.25;
  int i0 = (int)threadIdx.x, i1 = 3;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < 256; ++it) {
    // four independent chains so that latency does not limit issue
    if (OP == 0) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %2, %2, %1, %1\n v_fma_f64 %3, %3, %1, %1\n v_fma_f64 %4, %4, %1, %1" : "+v"(v0) : "v"(v1), "v"(v2), "v"(v3), "v"(a));) }
    if (OP == 1) { REP16(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));) }
    if (OP == 2) { REP16(asm volatile("v_ldexp_f64 %0, %0, %4\n v_ldexp_f64 %1, %1, %4\n v_ldexp_f64 %2, %2, %4\n v_ldexp_f64 %3, %3, %4" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(i1));) }
    if (OP == 3) { REP16(asm volatile("v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1\n v_rndne_f64 %2, %2\n v_rndne_f64 %3, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));) }
    if (OP == 4) { REP16(asm volatile("v_cvt_i32_f64 %0, %1\n v_cvt_i32_f64 %0, %2\n v_cvt_i32_f64 %0, %3\n v_cvt_i32_f64 %0, %4" : "+v"(i0) : "v"(v0), "v"(v1), "v"(v2), "v"(v3));) }
    if (OP == 5) { REP16(asm volatile("v_frexp_mant_f64 %0, %0\n v_frexp_mant_f64 %1, %1\n v_frexp_mant_f64 %2, %2\n v_frexp_mant_f64 %3, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));) }
    if (OP == 6) { REP16(asm volatile("v_frexp_exp_i32_f64 %0, %1\n v_frexp_exp_i32_f64 %0, %2\n v_frexp_exp_i32_f64 %0, %3\n v_frexp_exp_i32_f64 %0, %4" : "+v"(i0) : "v"(v0), "v"(v1), "v"(v2), "v"(v3));) }
    if (OP == 7) { REP16(asm volatile("v_min_f64 %0, %0, %4\n v_min_f64 %1, %1, %4\n v_min_f64 %2, %2, %4\n v_min_f64 %3, %3, %4" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a));) }
    if (OP == 8) { REP16(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a));) }
    if (OP == 9) { REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a));) }
    if (OP == 10) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i0) : "v"(i1) : "vcc");) }
    if (OP == 11) { REP16(asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));) }
    if (OP == 12) { REP16(asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %2\n v_cmp_lt_f64 vcc, %2, %3\n v_cmp_lt_f64 vcc, %3, %0" :: "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "vcc");) }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + i0;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
  double *out;
  unsigned long long *cyc, h;
  hipMalloc(&out, 1 << 20);
  hipMalloc(&cyc, 8);
  const char *names[] = {"v_fma_f64", "v_rcp_f64", "v_ldexp_f64", "v_rndne_f64", "v_cvt_i32_f64", "v_frexp_mant_f64",
                         "v_frexp_exp_i32_f64", "v_min_f64", "v_mul_f64", "v_add_f64", "v_cndmask_b32", "v_rsq_f64", "v_cmp_lt_f64"};
#define RUN(OP)                                                                  \
  hipLaunchKernelGGL(probe<OP>, dim3(1), dim3(64), 0, 0, out, cyc, 1.25, 0.75);  \
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                  \
  printf("%-22s %.2f cycles/instr (one wave)\n", names[OP], (double)h / (256.0 * 64.0));
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12)
  return 0;
}
