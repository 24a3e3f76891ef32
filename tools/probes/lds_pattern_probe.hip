// Probe: LDS cost of the access patterns of the 32-lane frame pipeline (stft_fast_p32.hpp), per wave-instruction and CU,
// with 8 and 16 waves per CU all issuing the same pattern back to back (throughput) and with one wave (latency-ish).
//   T0 transposition write   lane l (of 32; two frames per wave in columns 2w, 2w+1): cell l + 33 j      ds_write_b32
//   T1 transposition read    cell i + 33 l                                                             ds_read_b32
//   T2 result write          row l + 32 s                                                              ds_write_b32
//   T3 flush read            rows {0-3,16-19} + 4 h, 4 floats per lane                                  ds_read_b32 x4 (or read2)
//   T4 table read            lane-contiguous float2                                                    ds_read_b64
//   T5 plain                 lane-contiguous float                                                     ds_read_b32 / T6 ds_write_b32
// Build: hipcc -O3 --offload-arch=gfx950 -o lds_pattern_probe lds_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int kStride = 17, kRows = 1056;
template <int T>
__global__ void __launch_bounds__(1024) k(unsigned long long *cyc, int reps, float *sink) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 31, h = lane >> 5;
  const int w8 = wave & 7, g = wave >> 3;
  float *tile = lds + g * kRows * kStride;
  const int col = 2 * w8 + h;
  for (int i = threadIdx.x; i < 2 * kRows * kStride; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  float acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = (float)(lane + i);
  const int own = l * kStride + col, rd = 33 * l * kStride + col;
  const int fr = (128 * w8 + ((l >> 2) & 3) + 16 * (l >> 4) + 4 * h) * kStride + 4 * (lane & 3);
  float2 *tab = reinterpret_cast<float2 *>(lds + 2 * kRows * kStride);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    if constexpr (T == 0) {
#pragma unroll
      for (int j = 0; j < 32; ++j) tile[own + 33 * kStride * j] = acc[j];
    }
    if constexpr (T == 1) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] += tile[rd + kStride * i];
    }
    if constexpr (T == 2) {
#pragma unroll
      for (int s = 0; s < 32; ++s) tile[own + 32 * kStride * s] = acc[s];
    }
    if constexpr (T == 3) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const float *src = tile + fr + (32 * (p >> 1) + 8 * (p & 1)) * kStride;
        acc[4 * p] += src[0]; acc[4 * p + 1] += src[1]; acc[4 * p + 2] += src[2]; acc[4 * p + 3] += src[3];
      }
    }
    if constexpr (T == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float2 w = tab[l + 32 * i]; acc[2 * i] += w.x; acc[2 * i + 1] += w.y; }
    }
    if constexpr (T == 5) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] += lds[lane + 64 * i + 64 * 32 * (wave & 3)];
    }
    if constexpr (T == 6) {
#pragma unroll
      for (int i = 0; i < 32; ++i) lds[lane + 64 * i + 64 * 32 * (wave & 3)] = acc[i];
    }
    if constexpr (T == 7) {   // transposition as in the kernel: 32 writes, 32 reads, 32 writes, 32 reads (in place)
#pragma unroll
      for (int j = 0; j < 32; ++j) tile[own + 33 * kStride * j] = acc[j];
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = tile[rd + kStride * i];
    }
    if constexpr (T == 8) {   // round 5: the transposition with the strides swapped -- lane l writes register j to cell 33 l + j (a PAIR
      // of registers per ds_write2_b32: cells 68 bytes apart), lane k1 reads register l' from cell 33 l' + k1.  Both sides conflict free.
      const unsigned wb = (unsigned)((33 * l * kStride + col + g * kRows * kStride) * 4);
#pragma unroll
      for (int j = 0; j < 32; j += 2)
        asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(wb + (unsigned)(j >= 14 ? 14 * kStride * 4 : 0) + (unsigned)(j >= 28 ? 14 * kStride * 4 : 0)), "v"(acc[j]), "v"(acc[j + 1]),
                     "n"((j % 14) * kStride), "n"((j % 14 + 1) * kStride) : "memory");
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = tile[own + 33 * kStride * i];
    }
    if constexpr (T == 9) {   // the write half of T8 alone
      const unsigned wb = (unsigned)((33 * l * kStride + col + g * kRows * kStride) * 4);
#pragma unroll
      for (int j = 0; j < 32; j += 2)
        asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(wb + (unsigned)(j >= 14 ? 14 * kStride * 4 : 0) + (unsigned)(j >= 28 ? 14 * kStride * 4 : 0)), "v"(acc[j]), "v"(acc[j + 1]),
                     "n"((j % 14) * kStride), "n"((j % 14 + 1) * kStride) : "memory");
    }
    if constexpr (T == 10) {   // T7 with the reads as ds_read2_b32 (half the instructions, same LDS-array cycles)
#pragma unroll
      for (int j = 0; j < 32; ++j) tile[own + 33 * kStride * j] = acc[j];
      const unsigned rb = (unsigned)((rd + g * kRows * kStride) * 4);
      float2 t[16];
#pragma unroll
      for (int i = 0; i < 32; i += 2)
        asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(t[i / 2]) : "v"(rb + (unsigned)(i >= 14 ? 14 * kStride * 4 : 0) + (unsigned)(i >= 28 ? 14 * kStride * 4 : 0)),
                     "n"((i % 14) * kStride), "n"((i % 14 + 1) * kStride) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "+v"(t[8]), "+v"(t[9]),
                   "+v"(t[10]), "+v"(t[11]), "+v"(t[12]), "+v"(t[13]), "+v"(t[14]), "+v"(t[15])::"memory");
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc[2 * i] = t[i].x; acc[2 * i + 1] = t[i].y; }
    }
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float a = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) a += acc[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int T>
void run(const char *name, int ops_per_rep) {
  unsigned long long *cyc;
  float *sink;
  (void)hipMalloc(&cyc, 256 * 8);
  (void)hipMalloc(&sink, 256 * 1024 * 4);
  const int reps = 400;
  const int ldsb = 2 * kRows * kStride * 4 + 16384;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<T>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  printf("%-34s", name);
  for (int threads : {64, 256, 512, 1024}) {
    hipLaunchKernelGGL(k<T>, dim3(256), dim3(threads), ldsb, 0, cyc, reps, sink);
    hipLaunchKernelGGL(k<T>, dim3(256), dim3(threads), ldsb, 0, cyc, reps, sink);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : h) mean += c;
    mean /= 256;
    const int waves = threads / 64;
    printf("  %2dw: %6.2f cyc per wave-instr per CU (%.0f per wave)", waves, mean / ((double)reps * ops_per_rep * waves), mean / ((double)reps * ops_per_rep));
  }
  printf("\n");
}
int main() {
  run<0>("transposition write (b32)", 32);
  run<1>("transposition read (b32)", 32);
  run<2>("result write (b32)", 32);
  run<3>("flush read (4 floats per lane)", 32);
  run<4>("table read (b64, lane-contiguous)", 16);
  run<5>("plain read b32", 32);
  run<6>("plain write b32", 32);
  run<7>("transposition write + read", 64);
  run<8>("  strides swapped, ds_write2_b32", 64);
  run<9>("  its write half alone (write2)", 32);
  run<10>("  T7 with ds_read2_b32 reads", 64);
  return 0;
}
