// Probe: what back-to-back launches of one stream cost by themselves on this part, by launch shape: an (almost) empty kernel and one
// that writes a GB, 256 workgroups x 512 threads, with and without 160 KB of dynamic LDS.  (Round 5: a C2 launch of the power kernel
// spends 12-24 us between the end of one kernel and the start of the next: tools/launch_timeline.py.)
// Build: hipcc -O3 --offload-arch=gfx950 -o launch_gap_probe launch_gap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) empty_kernel(float *p) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0 && p == nullptr) lds[0] = 1.f;
}
__global__ void __launch_bounds__(512) write_kernel(float4 *p, size_t n4) {   // n4 float4 per workgroup, coalesced
  extern __shared__ float lds[];
  float4 *q = p + (size_t)blockIdx.x * n4;
  for (size_t i = threadIdx.x; i < n4; i += 512) q[i] = make_float4(1.f, 2.f, 3.f, (float)i);
  if (p == nullptr) lds[0] = 1.f;
}
template <class F> float per_launch_us(F launch, int reps) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1000.f / reps;
}
int main() {
  float4 *buf;
  const size_t total = (size_t)1 << 30;   // 1 GB
  (void)hipMalloc(&buf, total);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(empty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(write_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int lds : {0, 160 * 1024}) {
    printf("LDS %6d B: empty kernel %.2f us per launch;", lds, per_launch_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), lds, 0, (float *)buf); }, 200));
    for (size_t mb : {16, 64, 256, 1024}) {
      const size_t n4 = mb * 1024 * 1024 / 16 / 256;
      const float us = per_launch_us([&] { hipLaunchKernelGGL(write_kernel, dim3(256), dim3(512), lds, 0, buf, n4); }, 50);
      printf("  write %4zu MB %.1f us (%.2f TB/s)", mb, us, mb * 1.048576 / us);
    }
    printf("\n");
  }
  return 0;
}
