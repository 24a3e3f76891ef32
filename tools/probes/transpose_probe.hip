// Probe: 16x16 (lane>>2, register) transpose inside a wave with permlane32/16_swap + DPP row ops.
// Build: hipcc --offload-arch=gfx950 -O3 transpose_probe.hip -o transpose_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_mov(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, 0xF, BANK, false));
}
// inline asm: this hipcc drops the second result of __builtin_amdgcn_permlane{32,16}_swap.
// "s_nop 1" covers the VALU-write -> v_permlane read hazard (2 wait states) inside the statement.
__device__ __forceinline__ void swap32(float &a, float &b) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap16(float &a, float &b) {
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void transpose16(float (&v)[16]) {
#pragma unroll
  for (int k = 0; k < 16; ++k) if (!(k & 8)) swap32(v[k], v[k | 8]);
#pragma unroll
  for (int k = 0; k < 16; ++k) if (!(k & 4)) swap16(v[k], v[k | 4]);
#pragma unroll
  for (int k = 0; k < 16; ++k) if (!(k & 2)) { const float A = v[k], B = v[k | 2]; v[k | 2] = dpp_mov<0x128, 0x3>(B, A); v[k] = dpp_mov<0x128, 0xC>(A, B); }
#pragma unroll
  for (int k = 0; k < 16; ++k) if (!(k & 1)) { const float A = v[k], B = v[k | 1]; v[k | 1] = dpp_mov<0x104, 0x5>(B, A); v[k] = dpp_mov<0x114, 0xA>(A, B); }
}
__global__ void k(const float* in, float* out) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = in[i * 64 + threadIdx.x];
  transpose16(v);
#pragma unroll
  for (int i = 0; i < 16; ++i) out[i * 64 + threadIdx.x] = v[i];
}
int main() {
  float h[1024], o[1024], *di, *dout;
  for (int k = 0; k < 16; ++k) for (int l = 0; l < 64; ++l) h[k * 64 + l] = (float)(l * 16 + k);
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(h));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j) {
    const int i = l >> 2, a = l & 3;
    const float want = (float)((j * 4 + a) * 16 + i);   // lane (j, a), register i
    if (o[j * 64 + l] != want) { if (bad < 8) printf("lane %d reg %d got %g want %g\n", l, j, o[j * 64 + l], want); ++bad; }
  }
  printf("transpose probe: %d mismatches\n", bad);
  return bad != 0;
}
