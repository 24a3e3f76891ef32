// Probe: cost of the spectrogram's scattered row-run stores by themselves (diagnostic, not part of the library).
// out[clip][1025][pitch] floats; a "tile" = RUN frames (RUN*4 bytes per row) x 1024 rows, written by one workgroup
// of 16 waves with global_store_dwordx4 (RUN/4 lanes per row).  Tiles are dealt to workgroups either round robin
// (mode 0: neighbours on different XCDs), XCD-contiguous (mode 2: 32 neighbouring tiles per XCD at a time) or as
// one contiguous range per workgroup (mode 1).
//   ./store_shape_probe <run frames: 16|32|64|128> <pitch floats> <mode 0|2> [shift floats]
// Build: hipcc -O3 --offload-arch=gfx950 -o store_shape_probe store_shape_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int RUN>
__global__ void __launch_bounds__(1024) k(float *out, long pitch_f, long total_tiles, int mode, int shift) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nb = gridDim.x;
  long vb = blockIdx.x;
  if (mode == 2) {
    const long q = nb / 8, r = nb % 8, xcd = vb % 8, idx = vb / 8;
    vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const long tpc = pitch_f / RUN;
  constexpr int LPR = RUN / 4;             // lanes per row (16 bytes each)
  constexpr int RPI = 64 / LPR;            // rows per instruction
  // mode 1: every workgroup walks its own contiguous range of tiles (its own clips), one after the other
  const long per = (total_tiles + nb - 1) / nb;
  const long g0 = mode == 1 ? vb * per : vb, g1 = mode == 1 ? (g0 + per < total_tiles ? g0 + per : total_tiles) : total_tiles;
  const long gs = mode == 1 ? 1 : nb;
  for (long g = g0; g < g1; g += gs) {
    float *base = out + (g / tpc) * 1025 * pitch_f + RUN * (g % tpc) + shift;
    for (int i = 0; i < 64 / RPI; ++i) {   // 64 rows per wave
      const int row = 64 * wave + RPI * i + lane / LPR;
      float *p = base + row * pitch_f + 4 * (lane % LPR);
      const float v = (float)(g + row);
      using f4 = __attribute__((ext_vector_type(4))) float;
      const f4 d = {v, v, v, v};
      asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(d) : "memory");
    }
  }
}
int main(int argc, char **argv) {
  const int run = argc > 1 ? atoi(argv[1]) : 16;
  const long frames = argc > 2 ? atol(argv[2]) : 938;
  const int mode = argc > 3 ? atoi(argv[3]) : 0;
  const int shift = argc > 4 ? atoi(argv[4]) : 0;
  const int clips = 256;
  const long total_tiles = clips * (frames / run);
  float *out;
  if (hipMalloc(&out, (size_t)clips * 1025 * frames * 4 + 65536) != hipSuccess) return 1;
  (void)hipMemset(out, 0, (size_t)clips * 1025 * frames * 4 + 65536);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  auto launch = [&]() {
    if (run == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(1024), 0, 0, out, frames, total_tiles, mode, shift);
    else if (run == 32) hipLaunchKernelGGL(k<32>, dim3(256), dim3(1024), 0, 0, out, frames, total_tiles, mode, shift);
    else if (run == 64) hipLaunchKernelGGL(k<64>, dim3(256), dim3(1024), 0, 0, out, frames, total_tiles, mode, shift);
    else hipLaunchKernelGGL(k<128>, dim3(256), dim3(1024), 0, 0, out, frames, total_tiles, mode, shift);
  };
  for (int i = 0; i < 3; ++i) launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)total_tiles * 1024 * run * 4;
  printf("run %3d frames (%3d B) pitch %ld mode %d shift %d: %.3f ms per launch, %.2f TB/s (%.1f MB)\n", run, run * 4, frames,
         mode, shift, ms / reps, bytes / (ms / reps) / 1e9, bytes / 1e6);
  return 0;
}
