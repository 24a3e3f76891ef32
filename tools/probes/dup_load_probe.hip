// Probe (round 5, a lead for the next): does the vector-memory path charge a load instruction by the bytes its LANES ask for or by the
// bytes it FETCHES?  In the 32-lane frame pipeline the two halves of a wave hold consecutive frames: half 1's samples are half 0's,
// 2048 bytes further on -- the same instruction index asks for two different 256-byte pieces, and every sample is requested by four
// frames.  With half 1's registers rotated by 8 (register j <-> point j - 8 of ITS frame; the shift theorem turns that into a factor
// i^k1 on the first transform's outputs, i.e. into the twiddle table) 24 of the 32 instructions would ask for the SAME piece in both
// halves.  Here: 256 workgroups x 8 waves walk 58 tiles of 16 frames (hop 512) of a 480 000-sample clip as the power kernel does,
// 32 loads of 8 bytes per lane and tile, ~6 000 cycles of arithmetic between them;
//   mode 0: as the kernel (distinct pieces per half)   mode 1: rotated (24 shared pieces, 8 distinct)   mode 2: half 1 masked off (what one half alone costs)
// Build: hipcc -O3 --offload-arch=gfx950 -o dup_load_probe dup_load_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int kN = 480000, kTiles = 58;
template <int MODE>
__global__ void __launch_bounds__(512) k(const float *x, float *sink, unsigned long long *cyc) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l = lane & 31, h = lane >> 5;
  const float *clip = x + (size_t)blockIdx.x * kN;
  float acc = 0.f;
  unsigned long long issue = 0, wait = 0;
  for (int t = 0; t < kTiles; ++t) {
    const float2 *f0 = reinterpret_cast<const float2 *>(clip + (size_t)(16 * t + 2 * wave) * 512) + l;   // frame 2 wave of the tile, this lane's first point
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float2 v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      int idx;
      if (MODE == 1) idx = (h == 0 || j >= 8) ? 32 * j : 32 * (j + 32);          // half 1, register j: point j - 8 of frame + 1 = absolute piece j (j >= 8), piece j + 32 - ... (j < 8: the frame's last quarter)
      else idx = 32 * j + 256 * h;                                               // half 1: the next frame, 256 complex points on
      if (MODE == 2 && h == 1) { v[j] = make_float2(0.f, 0.f); continue; }
      v[j] = f0[idx];
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float a = acc + (float)t;
#pragma unroll 1
    for (int r = 0; r < 350; ++r) { a = __builtin_fmaf(a, 1.0001f, 0.5f); a = __builtin_fmaf(a, 0.9999f, -0.5f); a = __builtin_fmaf(a, 1.0002f, 0.25f); a = __builtin_fmaf(a, 0.9998f, -0.25f); }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int j = 0; j < 32; ++j) a += v[j].x + v[j].y;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    acc = a;
    issue += t1 - t0;
    wait += t3 - t2;
  }
  sink[blockIdx.x * 512 + tid] = acc;
  if (lane == 0) { cyc[(blockIdx.x * 8 + wave) * 2] = issue; cyc[(blockIdx.x * 8 + wave) * 2 + 1] = wait; }
}
template <int MODE> void run(const float *x, float *sink, unsigned long long *cyc, const char *name) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, x, sink, cyc);
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, x, sink, cyc);
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> hc(256 * 8 * 2);
  (void)hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost);
  double is = 0, wt = 0;
  for (size_t i = 0; i < hc.size(); i += 2) { is += hc[i]; wt += hc[i + 1]; }
  printf("%-52s %.4f ms per launch; per tile and wave: issue %.0f ticks (100 MHz: x ~21 for cycles), wait at first use %.0f ticks\n", name, ms / 20,
         is / (hc.size() / 2) / kTiles, wt / (hc.size() / 2) / kTiles);
}
int main() {
  float *x, *sink; unsigned long long *cyc;
  const size_t n = (size_t)256 * kN + 65536;
  (void)hipMalloc(&x, n * 4); (void)hipMemset(x, 0, n * 4);
  (void)hipMalloc(&sink, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 2 * 8);
  run<0>(x, sink, cyc, "as the kernel: a frame per half, distinct pieces");
  run<1>(x, sink, cyc, "half 1 rotated by 8: 24 of 32 pieces shared");
  run<2>(x, sink, cyc, "half 0 alone (half 1 masked off)");
  run<0>(x, sink, cyc, "as the kernel, again");
  return 0;
}
