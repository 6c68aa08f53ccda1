// Host cost of the HIP calls the whole-path driver makes (run on the GPU box):  hipcc --offload-arch=gfx950 -O2 launch_cost.hip -o /tmp/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { void* p[64]; int v[16]; };   // ~576 bytes of kernel arguments, as Clouds2
__global__ void k_empty_small(int* x) { if (x && threadIdx.x == 9999) *x = 1; }
__global__ void k_empty_big(Big b) { if (b.p[0] && threadIdx.x == 9999) *(int*)b.p[0] = 1; }
__global__ void k_spin(long long clocks) { const long long t0 = clock64(); while (clock64() - t0 < clocks) {} }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s, s2; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming);
  hipEvent_t et; hipEventCreate(&et);
  Big b{}; 
  for (int rep = 0; rep < 3; ++rep) {
    const int N = 200;
    // the queue kept busy by one long kernel so that the launches' cost is the host's alone
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 4000000LL);
    double t0 = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty_small, dim3(64), dim3(256), 0, s, (int*)nullptr);
    double t1 = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty_big, dim3(64), dim3(256), 0, s, b);
    double t2 = now();
    for (int i = 0; i < N; ++i) hipEventRecord(e, s);
    double t3 = now();
    for (int i = 0; i < N; ++i) hipStreamWaitEvent(s2, e, 0);
    double t4 = now();
    for (int i = 0; i < N; ++i) hipEventRecord(et, s);
    double t5 = now();
    hipStreamSynchronize(s); hipStreamSynchronize(s2);
    // drained queue: launch + wait = the round trip of one dependent hand-over
    double t6 = now();
    for (int i = 0; i < 50; ++i) { hipLaunchKernelGGL(k_empty_small, dim3(1), dim3(64), 0, s, (int*)nullptr); hipStreamSynchronize(s); }
    double t7 = now();
    // the empty kernels' own execution, back to back on the device
    hipEventRecord(et, s);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty_big, dim3(64), dim3(256), 0, s, b);
    hipEvent_t et2; hipEventCreate(&et2); hipEventRecord(et2, s); hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, et, et2);
    printf("rep %d: launch(small args) %.2f us  launch(576 B args) %.2f us  eventRecord(no timing) %.2f us  streamWaitEvent %.2f us  eventRecord(timing) %.2f us | launch+sync %.2f us | %d empty kernels back to back: %.2f us each\n",
           rep, (t1 - t0) / N, (t2 - t1) / N, (t3 - t2) / N, (t4 - t3) / N, (t5 - t4) / N, (t7 - t6) / 50, N, 1e3 * ms / N);
  }
  return 0;
}
