// Probe: what a SIMD of this chip retires per clock when W wavefronts share it — the numbers k_graph_build's bound is
// made of.  One workgroup of 4 W wavefronts (W per SIMD), every wave runs n rounds of eight INDEPENDENT instructions of
// one kind; reported: shader clocks per instruction per SIMD (all waves of a SIMD together).
//   hipcc --offload-arch=gfx950 -O3 tests/probe/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_probe(float* out, unsigned long long* t, int n, int mode) {
  __shared__ __attribute__((aligned(16))) float lds[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = (float)i;
  float b = 1.0001f;
  f2 p[8];
  float a[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = out[threadIdx.x] + i;
    p[i] = f2{a[i], a[i] + 1};
  }
  const f2 pb = {b, b};
  unsigned long long acc = 0;
  unsigned cacc = 0;
  __syncthreads();
  const unsigned long long c0 = clock64();
  if (mode == 0) {
    for (int i = 0; i < n; ++i)
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[k]) : "v"(pb));
  } else if (mode == 1) {
    for (int i = 0; i < n; ++i)
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[k]) : "v"(b));
  } else if (mode == 2) {
    for (int i = 0; i < n; ++i)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        unsigned long long m;
        asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a[k]), "v"(b));
        acc |= m;
      }
  } else if (mode == 3) {  // v_cmp -> carry-in of v_addc (the column-word shift of k_graph_build)
    for (int i = 0; i < n; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        unsigned long long m, junk;
        asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a[k]), "v"(b));
        asm volatile("v_addc_co_u32 %0, %1, %0, %0, %2" : "+v"(cacc), "=s"(junk) : "s"(m));
      }
  } else if (mode == 4) {  // 16-byte broadcast reads from LDS (all lanes one address), eight in flight
    for (int i = 0; i < n; ++i) {
      f4 v[8];
      const int base = (i & 7) * 128;
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *(const f4*)(lds + base + 4 * k);
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += v[k].x;  // (8 VALU per 8 reads; subtract mode 1's share)
    }
  } else if (mode == 5) {  // v_pk_add with neg modifiers + v_pk_mul: the loop's other packed forms
    for (int i = 0; i < n; ++i)
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[k]) : "v"(pb));
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k + 1]) : "v"(pb));
      }
  } else if (mode == 6) {  // v_min3 / v_max3
    for (int i = 0; i < n; ++i)
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(a[k + 1]), "v"(b));
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k + 1]) : "v"(a[k]), "v"(b));
      }
  }
  const unsigned long long c1 = clock64();
  float r = (float)acc + (float)cacc;
  for (int i = 0; i < 8; ++i) r += a[i] + p[i].x + p[i].y;
  out[threadIdx.x] = r;
  if (threadIdx.x == 0) t[0] = c1 - c0;
}
int main() {
  float* d;
  unsigned long long *t, h;
  hipMalloc(&d, 1 << 16);
  hipMemset(d, 0, 1 << 16);
  hipMalloc(&t, 16);
  const char* names[] = {"v_pk_fma_f32", "v_fma_f32", "v_cmp_gt_f32 -> sgpr", "v_cmp + v_addc(sgpr carry)", "ds_read_b128 broadcast (+1 v_add each)",
                         "v_pk_add(neg) / v_pk_mul", "v_min3 / v_max3"};
  const int n = 4000;
  for (int mode = 0; mode < 7; ++mode)
    for (int W : {1, 2, 4, 8}) {
      hipLaunchKernelGGL(k_probe, dim3(1), dim3(256 * W), 0, 0, d, t, n, mode);
      hipLaunchKernelGGL(k_probe, dim3(1), dim3(256 * W), 0, 0, d, t, n, mode);
      hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
      const double per = (double)h / ((double)n * 8.0 * W);  // clocks per instruction per SIMD (wave 0's span)
      printf("%-40s W=%d waves/SIMD: %9llu clk -> %.2f clk per instruction per SIMD\n", names[mode], W, h, per);
    }
  return 0;
}
