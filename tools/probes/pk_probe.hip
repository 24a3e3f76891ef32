// Probe: SIMD throughput of scalar vs packed f32 VALU forms with all-VGPR operands (4 waves/SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int T>
__global__ void __launch_bounds__(1024) k(float *out, unsigned long long *cyc, int reps) {
  f2 v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i};
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        f2 &a = v[i]; const f2 b = v[(i + 5) & 15]; const f2 c = v[(i + 10) & 15];
        if (T == 0) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a.x) : "v"(b.x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a.y) : "v"(b.y)); }
        if (T == 1) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(b)); }
        if (T == 2) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a.x) : "v"(b.x), "v"(c.x)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a.y) : "v"(b.y), "v"(c.y)); }
        if (T == 3) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c)); }
        if (T == 4) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a.x) : "v"(b.x)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a.y) : "v"(b.y)); }
        if (T == 5) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b)); }
        if (T == 6) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a.x) : "v"(b.x), "v"(c.x)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a.y) : "v"(b.y), "v"(c.y)); }
        if (T == 7) { asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a.x) : "v"(b.x)); asm volatile("v_add_f32 %0, %1, %0" : "+v"(a.y) : "v"(b.y)); }
        if (T == 8) { asm volatile("v_mov_b32 %0, %1" : "+v"(a.x) : "v"(b.x)); asm volatile("v_mov_b32 %0, %1" : "+v"(a.y) : "v"(b.y)); }
        if (T == 9) { asm volatile("v_cndmask_b32 %0, %0, %1, s[10:11]" : "+v"(a.x) : "v"(b.x)); asm volatile("v_cndmask_b32 %0, %0, %1, s[10:11]" : "+v"(a.y) : "v"(b.y)); }
        if (T == 10) { asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(a) : "v"(b)); }
      }
    }
  }
  __syncthreads();
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += v[i].x + v[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int T>
void run(const char *name, int per_iter) {
  float *out; unsigned long long *cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
  const int reps = 200;
  hipLaunchKernelGGL(k<T>, dim3(256), dim3(1024), 0, 0, out, cyc, reps);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto c : h) mean += c; mean /= 256;
  printf("%-40s %.2f cycles per complex element-op per SIMD (%d instr each)\n", name, mean / (reps * 64.0) / 4, per_iter);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("2 x v_add_f32 (VGPR,VGPR)", 2); run<1>("v_pk_add_f32", 1); run<10>("v_pk_add_f32 op_sel/neg (rot by i)", 1);
  run<7>("v_sub + v_add", 2);
  run<2>("2 x v_fma_f32 (3 VGPR)", 2); run<6>("2 x v_fmac_f32", 2); run<3>("v_pk_fma_f32 (3 VGPR)", 1);
  run<4>("2 x v_mul_f32", 2); run<5>("v_pk_mul_f32", 1);
  run<8>("2 x v_mov_b32", 2); run<9>("2 x v_cndmask_b32 (sgpr mask)", 2);
  return 0;
}
