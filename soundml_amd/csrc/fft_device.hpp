// Shared device-side FFT building blocks (registers only): complex helpers and the
// 4- / 16-point forward DFTs used by the STFT and FIR kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace smx {
namespace fftdev {

struct c32 {
  float x, y;
};
__device__ __forceinline__ c32 operator+(c32 a, c32 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c32 operator-(c32 a, c32 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c32 cmul(c32 a, c32 w) {
  return {a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x};
}
__device__ __forceinline__ c32 mul_neg_i(c32 a) { return {a.y, -a.x}; }

// 4-point forward DFT in place: (a,b,c,d) -> (X0,X1,X2,X3)
__device__ __forceinline__ void fft4(c32 &a, c32 &b, c32 &c, c32 &d) {
  const c32 t0 = a + c, t1 = a - c, t2 = b + d, t3 = mul_neg_i(b - d);
  a = t0 + t2;
  b = t1 + t3;
  c = t0 - t2;
  d = t1 - t3;
}

// 16-point forward DFT, natural order in and out, fully in registers (4 x 4), in two passes
// (pass 1: four 4-point DFTs + inner twiddles; pass 2: four 4-point DFTs + index transpose).
__device__ __forceinline__ void fft16_pass1(c32 (&v)[16]) {
  constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f;
  constexpr float h = 0.70710678118654752f;
#pragma unroll
  for (int n0 = 0; n0 < 4; ++n0) fft4(v[n0], v[4 + n0], v[8 + n0], v[12 + n0]);
  // u[n0][k0] sits at v[4 k0 + n0]; multiply by W16^(n0 k0)
  v[4 * 1 + 1] = cmul(v[4 * 1 + 1], c32{c1, -s1});   // W^1
  v[4 * 1 + 2] = cmul(v[4 * 1 + 2], c32{h, -h});     // W^2
  v[4 * 1 + 3] = cmul(v[4 * 1 + 3], c32{s1, -c1});   // W^3
  v[4 * 2 + 1] = cmul(v[4 * 2 + 1], c32{h, -h});     // W^2
  v[4 * 2 + 2] = mul_neg_i(v[4 * 2 + 2]);            // W^4
  v[4 * 2 + 3] = cmul(v[4 * 2 + 3], c32{-h, -h});    // W^6
  v[4 * 3 + 1] = cmul(v[4 * 3 + 1], c32{s1, -c1});   // W^3
  v[4 * 3 + 2] = cmul(v[4 * 3 + 2], c32{-h, -h});    // W^6
  v[4 * 3 + 3] = cmul(v[4 * 3 + 3], c32{-c1, s1});   // W^9
}
__device__ __forceinline__ void fft16_pass2(c32 (&v)[16]) {
#pragma unroll
  for (int k0 = 0; k0 < 4; ++k0) fft4(v[4 * k0], v[4 * k0 + 1], v[4 * k0 + 2], v[4 * k0 + 3]);
  // X[k0 + 4 k1] sits at v[4 k0 + k1]: transpose the 4x4 index
  c32 t[16];
#pragma unroll
  for (int k0 = 0; k0 < 4; ++k0)
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) t[k0 + 4 * k1] = v[4 * k0 + k1];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = t[i];
}
__device__ __forceinline__ void fft16(c32 (&v)[16]) {
  fft16_pass1(v);
  fft16_pass2(v);
}

#ifdef SMX_STAMPS
// Diagnostic build only (make STAMPS=1): per-phase cycle sums of every wave, read back with
// smx_debug_read_stamps().  Never compiled into the shipped library; no output depends on it.
constexpr int kStampSlots = 24;
__device__ unsigned long long g_stamp_sums[4096 * 16 * kStampSlots];
#define SMX_STAMP(i)                                                                      \
  do {                                                                                    \
    unsigned long long t__;                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    stamp_sum[i] += t__ - stamp_prev;                                                     \
    stamp_prev = t__;                                                                     \
  } while (0)
#else
#define SMX_STAMP(i) do { } while (0)
#endif


}  // namespace fftdev
}  // namespace smx
