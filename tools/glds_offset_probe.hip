// Does the immediate offset of global_load_lds_dwordx4 move the LDS destination as well as the
// global source?  One wave copies with `offset:1024` and M0 = LDS base; the LDS image is then
// written back and inspected on the host.
//   hipcc --offload-arch=gfx950 -O2 tools/glds_offset_probe.hip -o /tmp/glds_probe && /tmp/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void probe(const double *src, double *out) {
  __shared__ double lds[1024];  // 8 KiB
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.0;
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
  const double *g = src + 2 * threadIdx.x;  // 16 bytes per lane
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %0, off\n\t"
               "global_load_lds_dwordx4 %0, off offset:1024\n\t"
               "global_load_lds_dwordx4 %0, off offset:3072\n\t"
               "s_waitcnt vmcnt(0)"
               :
               : "v"(g), "s"(base)
               : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}

int main() {
  std::vector<double> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = i;
  double *src, *out;
  hipMalloc(&src, 8192);
  hipMalloc(&out, 8192);
  hipMemcpy(src, h.data(), 8192, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out);
  std::vector<double> r(1024);
  hipMemcpy(r.data(), out, 8192, hipMemcpyDeviceToHost);
  // expected if the offset applies to both sides: LDS KiB-block b holds source block b for b = 0, 1, 3
  for (int b = 0; b < 8; ++b) printf("LDS block %d: first %.0f last %.0f\n", b, r[128 * b], r[128 * b + 127]);
  bool both = r[128] == 128.0 && r[255] == 255.0 && r[384] == 384.0 && r[256] == -1.0;
  printf("immediate offset applies to the LDS address too: %s\n", both ? "yes" : "NO");
  return both ? 0 : 1;
}
