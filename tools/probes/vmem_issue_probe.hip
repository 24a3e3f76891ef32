// Probe: what holds a wave where it ISSUES a run of global loads -- the number of instructions or the bytes?  (Round 5: istft2048_pipe_kernel's
// 33 requests of 8 bytes per lane hold every wave ~8 000 cycles per tile.)  256 workgroups x 8 waves, each wave requests the same 8448 bytes per
// lane-row set of a [1025][938] complex array (16 lanes a 128-byte row piece, as the kernel does) as 33 x 8 B or as 17 x 16 B per lane, then
// does ~7 000 cycles of arithmetic, tile after tile; s_memtime around the issue and around the first use.
// Build: hipcc -O3 --offload-arch=gfx950 -o vmem_issue_probe vmem_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int kFrames = 938, kRows = 1025, kTiles = 58;
template <int WIDE>
__global__ void __launch_bounds__(512) k(const float2 *z, float *sink, unsigned long long *cyc) {
  const int tid = threadIdx.x, wave = tid >> 6;
  const size_t clip = (size_t)blockIdx.x * kRows * kFrames;
  float acc = 0.f;
  unsigned long long issue = 0, wait = 0;
  for (int t = 0; t < kTiles; ++t) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float2 v[34];
    if constexpr (WIDE) {   // 8 lanes a 128-byte row piece (two frames per lane), 8 rows an instruction
      const int sf = tid & 7, srow = tid >> 3;
#pragma unroll
      for (int i = 0; i < 17; ++i) {
        const int row = srow + 64 * i;
        float4 q = make_float4(0, 0, 0, 0);
        if (row < kRows) q = *reinterpret_cast<const float4 *>(z + clip + (size_t)row * kFrames + 16 * t + 2 * sf);
        v[2 * i] = make_float2(q.x, q.y);
        v[2 * i + 1] = make_float2(q.z, q.w);
      }
    } else {                // 16 lanes a row piece, 4 rows an instruction
      const int sf = tid & 15, srow = tid >> 4;
#pragma unroll
      for (int i = 0; i < 33; ++i) {
        const int row = srow + 32 * i;
        v[i] = make_float2(0, 0);
        if (row < kRows) v[i] = z[clip + (size_t)row * kFrames + 16 * t + sf];
      }
      v[33] = make_float2(0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    // arithmetic that does not touch v: ~7000 cycles
    float a = acc + (float)t;
#pragma unroll 1
    for (int r = 0; r < 400; ++r) { a = __builtin_fmaf(a, 1.0001f, 0.5f); a = __builtin_fmaf(a, 0.9999f, -0.5f); a = __builtin_fmaf(a, 1.0002f, 0.25f); a = __builtin_fmaf(a, 0.9998f, -0.25f); }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 34; ++i) a += v[i].x + v[i].y;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    acc = a;
    issue += t1 - t0;
    wait += t3 - t2;
  }
  sink[blockIdx.x * 512 + tid] = acc;
  if ((tid & 63) == 0) { cyc[(blockIdx.x * 8 + wave) * 2] = issue; cyc[(blockIdx.x * 8 + wave) * 2 + 1] = wait; }
}
template <int WIDE> void run(const float2 *z, float *sink, unsigned long long *cyc, const char *name) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<WIDE>, dim3(256), dim3(512), 0, 0, z, sink, cyc);
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<WIDE>, dim3(256), dim3(512), 0, 0, z, sink, cyc);
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(256 * 8 * 2);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double is = 0, wt = 0;
  for (size_t i = 0; i < h.size(); i += 2) { is += h[i]; wt += h[i + 1]; }
  printf("%-34s %.3f ms per launch (%.2f TB/s); per tile and wave: issue %.0f ticks, wait at first use %.0f ticks\n", name, ms / 10,
         256.0 * kTiles * 16 * kRows * 8 / (ms / 10 * 1e-3) / 1e12, is / (h.size() / 2) / kTiles, wt / (h.size() / 2) / kTiles);
}
int main() {
  float2 *z; float *sink; unsigned long long *cyc;
  const size_t n = (size_t)256 * kRows * kFrames;
  (void)hipMalloc(&z, n * 8); (void)hipMemset(z, 0, n * 8);
  (void)hipMalloc(&sink, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 2 * 8);
  run<0>(z, sink, cyc, "33 requests of 8 bytes per lane:");
  run<1>(z, sink, cyc, "17 requests of 16 bytes per lane:");
  run<0>(z, sink, cyc, "33 requests of 8 bytes per lane:");
  run<1>(z, sink, cyc, "17 requests of 16 bytes per lane:");
  return 0;
}
