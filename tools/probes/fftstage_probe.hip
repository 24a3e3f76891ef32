// Probe: cycles per vector instruction of the COMPILED radix-16 stage code of the fused STFT kernels
// (fft_device.hpp's fft16_pass1/2 + 15 register twiddles, with and without the in-wave transposes), alone on the
// SIMDs: no LDS, no memory, no scalar work besides the loop counter.  Instruction counts per iteration come from
// the disassembly (tools/probes/count_loop.py) and are passed on the command line.
// Build: hipcc -O3 -ffp-contract=fast -I../../soundml_amd/csrc --offload-arch=gfx950 -o fftstage_probe fftstage_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fft_device.hpp"
using namespace smx::fftdev;

template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_mov(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, 0xF, BANK_MASK, false));
}
#define SMX_SWAP8(OP, V, A0, B0, A1, B1, A2, B2, A3, B3, A4, B4, A5, B5, A6, B6, A7, B7)                              \
  asm("s_nop 1\n\t" OP " %0, %1\n\t" OP " %2, %3\n\t" OP " %4, %5\n\t" OP " %6, %7\n\t" OP " %8, %9\n\t" OP            \
      " %10, %11\n\t" OP " %12, %13\n\t" OP " %14, %15"                                                             \
      : "+v"((V)[A0]), "+v"((V)[B0]), "+v"((V)[A1]), "+v"((V)[B1]), "+v"((V)[A2]), "+v"((V)[B2]), "+v"((V)[A3]),     \
        "+v"((V)[B3]), "+v"((V)[A4]), "+v"((V)[B4]), "+v"((V)[A5]), "+v"((V)[B5]), "+v"((V)[A6]), "+v"((V)[B6]),     \
        "+v"((V)[A7]), "+v"((V)[B7]))
__device__ __forceinline__ void transpose16(float (&v)[16]) {
  SMX_SWAP8("v_permlane32_swap_b32", v, 0, 8, 1, 9, 2, 10, 3, 11, 4, 12, 5, 13, 6, 14, 7, 15);
  SMX_SWAP8("v_permlane16_swap_b32", v, 0, 4, 1, 5, 2, 6, 3, 7, 8, 12, 9, 13, 10, 14, 11, 15);
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (!(k & 2)) {
      const float A = v[k], B = v[k | 2];
      v[k | 2] = dpp_mov<0x128, 0x3>(B, A);
      v[k] = dpp_mov<0x128, 0xC>(A, B);
    }
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (!(k & 1)) {
      const float A = v[k], B = v[k | 1];
      v[k | 1] = dpp_mov<0x104, 0x5>(B, A);
      v[k] = dpp_mov<0x114, 0xA>(A, B);
    }
}

template <int T>
__global__ void __launch_bounds__(1024) k(unsigned long long *cyc, int reps, float2 *io, const float2 *twg) {
  c32 v[16];
  float2 tw[16];
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float2 a = io[tid * 16 + j];
    v[j] = {a.x, a.y};
    tw[j] = twg[(threadIdx.x & 63) * 16 + j];
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    asm volatile("; LOOP_BEGIN");
    fft16_pass1(v);
    fft16_pass2(v);
#pragma unroll
    for (int q = 1; q < 16; ++q) v[q] = cmul(v[q], c32{tw[q].x, tw[q].y});
    if constexpr (T == 1) {
      float re[16], im[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) { re[q] = v[q].x; im[q] = v[q].y; }
      transpose16(re);
      transpose16(im);
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = {re[q], im[q]};
    }
    asm volatile("; LOOP_END");
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int j = 0; j < 16; ++j) io[tid * 16 + j] = make_float2(v[j].x, v[j].y);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int T>
void run(const char *name, int vec_per_iter) {
  unsigned long long *cyc;
  float2 *io, *tw;
  (void)hipMalloc(&cyc, 256 * 8);
  (void)hipMalloc(&io, 256 * 1024 * 16 * 8);
  (void)hipMalloc(&tw, 64 * 16 * 8);
  (void)hipMemset(io, 0, 256 * 1024 * 16 * 8);
  (void)hipMemset(tw, 0, 64 * 16 * 8);
  const int reps = 2000;
  printf("%-40s (%d vector instr / iteration)", name, vec_per_iter);
  for (int threads : {256, 512, 768, 1024}) {
    hipLaunchKernelGGL(k<T>, dim3(256), dim3(threads), 0, 0, cyc, reps, io, tw);
    hipLaunchKernelGGL(k<T>, dim3(256), dim3(threads), 0, 0, cyc, reps, io, tw);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : h) mean += c;
    mean /= 256;
    const double per_iter = mean / reps;
    printf("  %dw: %6.0f cyc/iter %5.2f/instr/wave %5.2f/SIMD", threads / 256, per_iter, per_iter / vec_per_iter, per_iter / vec_per_iter / (threads / 256));
  }
  printf("\n");
}
int main(int argc, char **argv) {
  run<0>("radix-16 + 15 twiddles", argc > 1 ? atoi(argv[1]) : 228);
  run<1>("radix-16 + 15 twiddles + 2 transposes", argc > 2 ? atoi(argv[2]) : 356);
  return 0;
}
