// Probe: cost of the spectrogram's scattered row-run stores by themselves (diagnostic, not part of the library).
// out[clip][1025][pitch] floats; a "tile" = RUN frames (RUN*4 bytes per row) x 1024 rows, written by one workgroup
// of 16 waves with global_store_dwordx4 (RUN/4 lanes per row).  Tiles are dealt to workgroups either round robin
// (mode 0: neighbours on different XCDs), XCD-contiguous (mode 2: 32 neighbouring tiles per XCD at a time) or as
// one contiguous range per workgroup (mode 1).
// Mode 3: ranges of R consecutive tiles, the ranges dealt like mode 2 (what a frame-ring pipeline would write).
// align = 1 (or 64) / 128: every row run is moved down to its 64- / 128-byte boundary (the "skewed" block stores such a pipeline
// could issue: one whole 64-byte block per row and tile instead of a run straddling two).
//   ./store_shape_probe <run frames: 16|32|64|128> <pitch floats> <mode 0|1|2|3> [shift floats] [align 0|1] [R]
// Build: hipcc -O3 --offload-arch=gfx950 -o store_shape_probe store_shape_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int RUN>
__global__ void __launch_bounds__(1024) k(float *out, long pitch_f, long total_tiles, int mode, int shift, int align, int R) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nb = gridDim.x;
  long vb = blockIdx.x;
  if (mode == 2 || mode == 3) {
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
  const long nranges = (total_tiles + R - 1) / R;
  const long n_outer = mode == 3 ? nranges : g1;
  for (long go = g0; go < n_outer; go += gs)
  for (int t = 0; t < (mode == 3 ? R : 1); ++t) {
    const long g = mode == 3 ? go * R + t : go;
    if (g >= total_tiles) break;
    float *base = out + (g / tpc) * 1025 * pitch_f + RUN * (g % tpc) + shift;
    for (int i = 0; i < 64 / RPI; ++i) {   // 64 rows per wave
      const int row = 64 * wave + RPI * i + lane / LPR;
      float *p = base + row * pitch_f;
      if (align) p = (float *)((unsigned long)p & ~(unsigned long)((align == 1 ? 64 : align) - 1));
      p += 4 * (lane % LPR);
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
  const int align = argc > 5 ? atoi(argv[5]) : 0;
  const int R = argc > 6 ? atoi(argv[6]) : 4;
  const int clips = 256;
  const long total_tiles = clips * (frames / run);
  float *out;
  if (hipMalloc(&out, (size_t)clips * 1025 * frames * 4 + 65536) != hipSuccess) return 1;
  (void)hipMemset(out, 0, (size_t)clips * 1025 * frames * 4 + 65536);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  auto launch = [&]() {
    if (run == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(1024), 0, 0, out, frames, total_tiles, mode, shift, align, R);
    else if (run == 32) hipLaunchKernelGGL(k<32>, dim3(256), dim3(1024), 0, 0, out, frames, total_tiles, mode, shift, align, R);
    else if (run == 64) hipLaunchKernelGGL(k<64>, dim3(256), dim3(1024), 0, 0, out, frames, total_tiles, mode, shift, align, R);
    else hipLaunchKernelGGL(k<128>, dim3(256), dim3(1024), 0, 0, out, frames, total_tiles, mode, shift, align, R);
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
  printf("run %3d frames (%3d B) pitch %ld mode %d shift %d align %d R %d: %.3f ms per launch, %.2f TB/s (%.1f MB)\n", run, run * 4,
         frames, mode, shift, align, R, ms / reps, bytes / (ms / reps) / 1e9, bytes / 1e6);
  return 0;
}
