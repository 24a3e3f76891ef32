// Probe: issue cost (cycles per wave-instruction per SIMD) of the cross-lane / data-movement
// instructions the FFT kernel depends on, at the kernel's occupancy (1024 threads = 4 waves/SIMD)
// and with 1 wave/SIMD.  Each test runs R x 64 instructions on 16 independent registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_mov(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, 0xF, BANK, false));
}
template <int T>
__global__ void __launch_bounds__(1024) k(float *out, unsigned long long *cyc, int reps, float s) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
  const int lane = threadIdx.x & 63;
  const bool odd = lane & 1;
  int addr = ((lane ^ 5) & 63) * 4;
  __shared__ float lds[4096];
  lds[threadIdx.x] = v[0];
  __syncthreads();
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (T == 0) v[i] = fmaf(v[i], s, 1.0f);                                   // v_fma (independent x16)
        if (T == 1) v[i] = dpp_mov<0x128, 0xF>(v[i], v[(i + 1) & 15]);            // v_mov_dpp row_ror:8
        if (T == 2) v[i] = dpp_mov<0x4E, 0xF>(v[i], v[(i + 1) & 15]);             // v_mov_dpp quad_perm
        if (T == 3) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[(i + 8) & 15]));
        if (T == 4) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[(i + 8) & 15]));
        if (T == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));   // v_cndmask
        if (T == 6) v[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v[i])));
        if (T == 7) v[i] = v[i] + v[(i + 3) & 15];                                // v_add
        if (T == 8) { v[i] += lds[(threadIdx.x + 64 * i + r) & 4095]; }           // ds_read_b32 (+ add)
        if (T == 9) { lds[(threadIdx.x + 64 * i + r) & 4095] = v[i]; v[i] += 1.0f; }  // ds_write_b32 (+ add)
        if (T == 11) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(s));            // v_fma, not packable
        if (T == 12) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*reinterpret_cast<float2*>(&v[i & 14])) : "v"(make_float2(s, s)));
        if (T == 10) v[i] = fmaf(v[i], s, dpp_mov<0x4E, 0xF>(0.f, v[i]));         // dpp mov + fma (quad radix step)
      }
    }
  }
  __syncthreads();   // every wave of the workgroup has finished: throughput, not the oldest wave's latency
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int T>
void run(const char *name, int threads) {
  float *out; unsigned long long *cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
  const int reps = 200;
  hipLaunchKernelGGL(k<T>, dim3(256), dim3(threads), 0, 0, out, cyc, reps, 1.0001f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto c : h) mean += c; mean /= 256;
  const double instr_per_wave = reps * 64.0;
  const int waves_per_simd = threads / 256;
  printf("%-28s %4d thr: %.2f cyc/instr/wave, %.2f cyc per wave-instr per SIMD\n", name, threads, mean / instr_per_wave,
         mean / instr_per_wave / waves_per_simd);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int threads : {256, 1024}) {
    run<0>("v_fma_f32 (compiler)", threads); run<11>("v_fma_f32 (asm)", threads); run<12>("v_pk_fma_f32 (asm)", threads); run<7>("v_add_f32", threads); run<5>("v_cndmask_b32", threads);
    run<1>("v_mov_dpp row_ror:8", threads); run<2>("v_mov_dpp quad_perm", threads); run<10>("dpp mov + fma", threads);
    run<3>("v_permlane32_swap", threads); run<4>("v_permlane16_swap", threads);
    run<6>("ds_bpermute_b32", threads); run<8>("ds_read_b32", threads); run<9>("ds_write_b32", threads);
  }
  return 0;
}
