// Accuracy of v_rcp_f64 / v_rsq_f64 seeds and of one / two Newton steps on them (relative error
// against the correctly rounded result, over 2^22 arguments spread across [1, 2) and a few binades).
//   hipcc --offload-arch=gfx950 -O2 tools/rcp_accuracy_probe.hip -o /tmp/rcp_probe && /tmp/rcp_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

__global__ void probe(double *err) {  // err[0..2]: rcp seed, 1 step, 2 steps; err[3..5]: rsq likewise
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
  double worst[6] = {0, 0, 0, 0, 0, 0};
  for (int i = tid; i < (1 << 22); i += n) {
    const double a = ldexp(1.0 + (double)i / (double)(1 << 22) + 1e-9 * (i % 7), (i % 9) - 4);
    const double exact = 1.0 / a;  // IEEE division (full software sequence)
    double r = __builtin_amdgcn_rcp(a);
    worst[0] = fmax(worst[0], fabs(r - exact) / exact);
    double e = fma(-a, r, 1.0);
    r = fma(r, e, r);
    worst[1] = fmax(worst[1], fabs(r - exact) / exact);
    e = fma(-a, r, 1.0);
    r = fma(r, e, r);
    worst[2] = fmax(worst[2], fabs(r - exact) / exact);
    const double ex2 = 1.0 / sqrt(a);
    double y = __builtin_amdgcn_rsq(a);
    worst[3] = fmax(worst[3], fabs(y - ex2) / ex2);
    const double h = 0.5 * a;
    double er = fma(-h * y, y, 0.5);
    y = fma(y, er, y);
    worst[4] = fmax(worst[4], fabs(y - ex2) / ex2);
    er = fma(-h * y, y, 0.5);
    y = fma(y, er, y);
    worst[5] = fmax(worst[5], fabs(y - ex2) / ex2);
  }
  for (int k = 0; k < 6; ++k) {
    double v = worst[k];
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)&err[k], (unsigned long long)__double_as_longlong(v));
  }
}

int main() {
  double *d, h[6];
  hipMalloc(&d, 48);
  hipMemset(d, 0, 48);
  hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, d);
  hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
  printf("v_rcp_f64: seed %.3e  one Newton step %.3e  two %.3e   (2^-52 = 2.2e-16)\n", h[0], h[1], h[2]);
  printf("v_rsq_f64: seed %.3e  one Newton step %.3e  two %.3e\n", h[3], h[4], h[5]);
  return 0;
}
