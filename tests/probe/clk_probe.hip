// Probe: shader clock vs 100 MHz wall clock inside short kernels, and the issue cost of dependent instruction chains
// (v_fma_f32, v_pk_fma_f32, v_cmp -> s_and).  hipcc --offload-arch=gfx950 -O3 tests/probe/clk_probe.hip -o /tmp/clk_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k_chain(float* out, unsigned long long* t, int n, int mode) {
  float a = out[threadIdx.x], b = 1.0001f;
  f2 pa = {a, a}, pb = {b, b};
  unsigned long long acc = 0;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  if (mode == 0)
    for (int i = 0; i < n; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(b));
  else if (mode == 1)
    for (int i = 0; i < n; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pa) : "v"(pb));
  else if (mode == 2)
    for (int i = 0; i < n; ++i) {
      unsigned long long m;
      asm volatile("v_cmp_gt_f32 %0, %1, %2\n\ts_and_b64 %0, %0, exec" : "=s"(m) : "v"(a), "v"(b));
      acc += m;
    }
  else
    for (int i = 0; i < n; ++i) {  // 4 independent fma chains
      asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(pa.x) : "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(pa.y) : "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(pb.x) : "v"(b));
    }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  out[threadIdx.x] = a + pa.x + pa.y + pb.x + (float)acc;
  if (threadIdx.x == 0) {
    t[0] = c1 - c0;
    t[1] = w1 - w0;
  }
}
int main() {
  float* d;
  unsigned long long *t, h[2];
  hipMalloc(&d, 4096);
  hipMemset(d, 0, 4096);
  hipMalloc(&t, 16);
  const char* names[] = {"v_fma_f32 dependent", "v_pk_fma_f32 dependent", "v_cmp + s_and", "4 independent v_fma"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode)
      for (int n : {1000, 20000}) {
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d, t, n, mode);
        hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        const double ns = h[1] * 10.0;
        printf("%-26s n=%6d: %8llu clk, %8.0f ns -> %.2f GHz, %.2f clk/iter, %.2f ns/iter\n", names[mode], n, h[0], ns,
               h[0] / ns, (double)h[0] / n, ns / n);
      }
  return 0;
}
