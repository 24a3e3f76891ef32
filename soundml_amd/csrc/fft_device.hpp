// Shared device-side FFT building blocks (registers only): complex helpers and the
// 4- / 16-point forward DFTs used by the STFT and FIR kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace smx {
namespace fftdev {

// complex value of scalar type S in registers; c32 is what the float32 kernels use, c64 the float64 interior
template <typename S>
struct cpx {
  S x, y;
};
using c32 = cpx<float>;
using c64 = cpx<double>;
template <typename S> struct vec2_of;
template <> struct vec2_of<float> { using type = float2; };
template <> struct vec2_of<double> { using type = double2; };

template <typename S> __device__ __forceinline__ cpx<S> operator+(cpx<S> a, cpx<S> b) { return {a.x + b.x, a.y + b.y}; }
template <typename S> __device__ __forceinline__ cpx<S> operator-(cpx<S> a, cpx<S> b) { return {a.x - b.x, a.y - b.y}; }
// (two products and two fused multiply-adds, written out: the translation units that include this header are built with
// -ffp-contract=off, so the compiler fuses nothing on its own and every instantiation rounds the same way)
template <typename S> __device__ __forceinline__ cpx<S> cmul(cpx<S> a, cpx<S> w) {
  return {__builtin_fma(a.x, w.x, -(a.y * w.y)), __builtin_fma(a.x, w.y, a.y * w.x)};
}
template <> __device__ __forceinline__ cpx<float> cmul<float>(cpx<float> a, cpx<float> w) {
  return {__builtin_fmaf(a.x, w.x, -(a.y * w.y)), __builtin_fmaf(a.x, w.y, a.y * w.x)};
}
template <typename S> __device__ __forceinline__ cpx<S> mul_neg_i(cpx<S> a) { return {a.y, -a.x}; }

// 4-point forward DFT in place: (a,b,c,d) -> (X0,X1,X2,X3)
template <typename S>
__device__ __forceinline__ void fft4(cpx<S> &a, cpx<S> &b, cpx<S> &c, cpx<S> &d) {
  const cpx<S> t0 = a + c, t1 = a - c, t2 = b + d, t3 = mul_neg_i(b - d);
  a = t0 + t2;
  b = t1 + t3;
  c = t0 - t2;
  d = t1 - t3;
}

// 16-point forward DFT, natural order in and out, fully in registers (4 x 4), in two passes
// (pass 1: four 4-point DFTs + inner twiddles; pass 2: four 4-point DFTs + index transpose).
static_assert((float)0.92387953251128674 == 0.92387953251128674f && (float)0.38268343236508977 == 0.38268343236508977f &&
                  (float)0.70710678118654752 == 0.70710678118654752f,
              "the float32 constants are the rounded float64 ones");
template <typename S>
__device__ __forceinline__ void fft16_pass1(cpx<S> (&v)[16]) {
  constexpr S c1 = (S)0.92387953251128674, s1 = (S)0.38268343236508977;
  constexpr S h = (S)0.70710678118654752;
  using C = cpx<S>;
#pragma unroll
  for (int n0 = 0; n0 < 4; ++n0) fft4(v[n0], v[4 + n0], v[8 + n0], v[12 + n0]);
  // u[n0][k0] sits at v[4 k0 + n0]; multiply by W16^(n0 k0)
  v[4 * 1 + 1] = cmul(v[4 * 1 + 1], C{c1, -s1});   // W^1
  v[4 * 1 + 2] = cmul(v[4 * 1 + 2], C{h, -h});     // W^2
  v[4 * 1 + 3] = cmul(v[4 * 1 + 3], C{s1, -c1});   // W^3
  v[4 * 2 + 1] = cmul(v[4 * 2 + 1], C{h, -h});     // W^2
  v[4 * 2 + 2] = mul_neg_i(v[4 * 2 + 2]);          // W^4
  v[4 * 2 + 3] = cmul(v[4 * 2 + 3], C{-h, -h});    // W^6
  v[4 * 3 + 1] = cmul(v[4 * 3 + 1], C{s1, -c1});   // W^3
  v[4 * 3 + 2] = cmul(v[4 * 3 + 2], C{-h, -h});    // W^6
  v[4 * 3 + 3] = cmul(v[4 * 3 + 3], C{-c1, s1});   // W^9
}
template <typename S>
__device__ __forceinline__ void fft16_pass2(cpx<S> (&v)[16]) {
#pragma unroll
  for (int k0 = 0; k0 < 4; ++k0) fft4(v[4 * k0], v[4 * k0 + 1], v[4 * k0 + 2], v[4 * k0 + 3]);
  // X[k0 + 4 k1] sits at v[4 k0 + k1]: transpose the 4x4 index
  cpx<S> t[16];
#pragma unroll
  for (int k0 = 0; k0 < 4; ++k0)
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) t[k0 + 4 * k1] = v[4 * k0 + k1];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = t[i];
}
template <typename S>
__device__ __forceinline__ void fft16(cpx<S> (&v)[16]) {
  fft16_pass1(v);
  fft16_pass2(v);
}

// ---- Stockham (autosort) passes through LDS: the FFT of the FIR kernel and of the fast synthesis kernel ----
// In-place passes of radix 16 / 4 / 2; N/16 threads own one transform, every thread keeps 16 complex points in
// registers, so a pass is: read 16 (lane-contiguous, conflict free), sync, twiddle + register DFT, write 16 to
// the autosort positions.  LDS addresses are XOR-swizzled (a ^ ((a >> 5) & 31)): reads conflict free, writes at
// most 2-way (tools/sim_fir_fft.py).  Twiddles: one table read (exp(-2 pi i j / N), j < N/2) per pass and radix
// group, powers by binary multiplication (depth <= 4).  The scalar type S (float / double) comes from the registers.
__device__ __forceinline__ int swz(int a) { return a ^ ((a >> 5) & 31); }

// powers w^1 .. w^(R-1) of a unit twiddle by binary multiplication (depth <= 4)
template <int R, typename S>
__device__ __forceinline__ void twiddle_powers(cpx<S> w1, cpx<S> (&w)[16]) {
  w[1] = w1;
  if constexpr (R >= 4) {
    w[2] = cmul(w1, w1);
    w[3] = cmul(w[2], w1);
  }
  if constexpr (R >= 16) {
    w[4] = cmul(w[2], w[2]);
    w[5] = cmul(w[4], w1);
    w[6] = cmul(w[4], w[2]);
    w[7] = cmul(w[4], w[3]);
    w[8] = cmul(w[4], w[4]);
#pragma unroll
    for (int j = 9; j < 16; ++j) w[j] = cmul(w[8], w[j - 8]);
  }
}

// One Stockham pass of radix R with sub-transform size NS (see tools/sim_fir_fft.py).
// r[i*R + j] is element j of radix group i (G = 16/R groups per thread, t_i = tid + T*i).
// READ: registers come from LDS (false for the very first pass: they hold the loaded samples);
// WRITE: results go back to LDS at the autosort positions (false for the last pass of a transform,
// whose outputs stay in registers with natural index out_index<R,NS>(tid, i, j)).
// WAVE: the transform belongs to one wave (N = 1024: T = 64 threads) with a private buffer; DS operations of a wave
// execute in order, so the workgroup barriers reduce to compiler fences.
template <bool WAVE>
__device__ __forceinline__ void stockham_sync() {
  if constexpr (WAVE) asm volatile("" ::: "memory");
  else __syncthreads();
}

// `wpre`: the pass's table twiddles already in registers (fft_passes<.., PRE>: every pass's table read is issued before
// the first pass, so its global-memory latency is paid once per transform instead of once per pass); nullptr = read here.
template <int N, int R, int NS, bool READ, bool WRITE, bool WAVE = false, typename S = float>
__device__ __forceinline__ void stockham_pass(cpx<S> (&r)[16], typename vec2_of<S>::type *z, int tid,
                                              const typename vec2_of<S>::type *tw,
                                              const typename vec2_of<S>::type *wpre = nullptr) {
  using V = typename vec2_of<S>::type;
  constexpr int T = N / 16, G = 16 / R;
  if constexpr (READ) {
    stockham_sync<WAVE>();   // the previous pass's writes are visible
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const V v = z[swz(tid + T * (i + G * j))];
        r[i * R + j] = {v.x, v.y};
      }
    stockham_sync<WAVE>();   // everyone holds its points: the buffer may be overwritten
  }
#pragma unroll
  for (int i = 0; i < G; ++i) {
    if constexpr (NS > 1) {
      const int k = (tid + T * i) % NS;
      const V w1 = wpre ? wpre[i] : tw[k * (N / (NS * R))];
      cpx<S> w[16];
      twiddle_powers<R>(cpx<S>{w1.x, w1.y}, w);
#pragma unroll
      for (int j = 1; j < R; ++j) r[i * R + j] = cmul(r[i * R + j], w[j]);
    }
    if constexpr (R == 16) {
      fft16(r);
    } else if constexpr (R == 4) {
      fft4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
    } else {
      const cpx<S> u = r[2 * i], v = r[2 * i + 1];
      r[2 * i] = u + v;
      r[2 * i + 1] = u - v;
    }
  }
  if constexpr (WRITE) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int t = tid + T * i, k = t % NS;
      const int j0 = (t / NS) * NS * R + k;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        V o;
        o.x = r[i * R + j].x;
        o.y = r[i * R + j].y;
        z[swz(j0 + j * NS)] = o;
      }
    }
  }
}

template <int R, int NS, int T>
__device__ __forceinline__ int out_index(int tid, int i, int j) {
  const int t = tid + T * i, k = t % NS;
  return (t / NS) * NS * R + k + j * NS;
}

// forward FFT of the 16 points per thread; FIRST: registers already hold element tid + T*m in r[m].
// On return r[i*R + j] (last radix R, its NS) holds natural-order element out_index<R, NS, T>(tid, i, j).
#ifndef SMX_FFT_PRE_DEFAULT
#define SMX_FFT_PRE_DEFAULT false
#endif
template <int LOG2N, bool FIRST, bool WAVE = false, typename S = float, bool PRE = SMX_FFT_PRE_DEFAULT>
__device__ __forceinline__ void fft_passes(cpx<S> (&r)[16], typename vec2_of<S>::type *z, int tid,
                                           const typename vec2_of<S>::type *tw) {
  constexpr int N = 1 << LOG2N;
  static_assert(LOG2N >= 8 && LOG2N <= 14, "transform sizes 256 .. 16384");
  if constexpr (PRE) {   // same passes, same arithmetic; the table reads of passes 2.. are issued up front
    using V = typename vec2_of<S>::type;
    constexpr int T = N / 16;
    constexpr int R3 = LOG2N == 8 ? 1 : (LOG2N == 9 ? 2 : (LOG2N == 10 || LOG2N == 11) ? 4 : 16);       // third pass
    constexpr int R4 = (LOG2N == 11 || LOG2N == 13) ? 2 : (LOG2N == 14 ? 4 : 1);                          // fourth pass
    constexpr int NS4 = LOG2N == 11 ? 1024 : 4096;
    V w2[1], w3[R3 > 1 ? 16 / R3 : 1], w4[R4 > 1 ? 16 / R4 : 1];
    w2[0] = tw[(tid % 16) * (N / 256)];
    if constexpr (R3 > 1) {
#pragma unroll
      for (int i = 0; i < 16 / R3; ++i) w3[i] = tw[((tid + T * i) % 256) * (N / (256 * R3))];
    }
    stockham_pass<N, 16, 1, !FIRST, true, WAVE>(r, z, tid, tw);
    if constexpr (LOG2N == 8) {
      stockham_pass<N, 16, 16, true, false, WAVE>(r, z, tid, tw, w2);
    } else {
      stockham_pass<N, 16, 16, true, true, WAVE>(r, z, tid, tw, w2);
      if constexpr (R4 > 1) {   // the last pass's (up to 8) values one pass ahead only: 16 registers less across the first two passes
#pragma unroll
        for (int i = 0; i < 16 / R4; ++i) w4[i] = tw[((tid + T * i) % NS4) * (N / (NS4 * R4))];
      }
      stockham_pass<N, R3, 256, true, (R4 > 1), WAVE>(r, z, tid, tw, w3);
      if constexpr (R4 > 1) stockham_pass<N, R4, NS4, true, false, WAVE>(r, z, tid, tw, w4);
    }
    return;
  }
  stockham_pass<N, 16, 1, !FIRST, true, WAVE>(r, z, tid, tw);
  if constexpr (LOG2N == 8) {
    stockham_pass<N, 16, 16, true, false, WAVE>(r, z, tid, tw);
  } else {
    stockham_pass<N, 16, 16, true, true, WAVE>(r, z, tid, tw);
    if constexpr (LOG2N == 9) {
      stockham_pass<N, 2, 256, true, false, WAVE>(r, z, tid, tw);
    } else if constexpr (LOG2N == 10) {
      stockham_pass<N, 4, 256, true, false, WAVE>(r, z, tid, tw);
    } else if constexpr (LOG2N == 11) {
      stockham_pass<N, 4, 256, true, true, WAVE>(r, z, tid, tw);
      stockham_pass<N, 2, 1024, true, false, WAVE>(r, z, tid, tw);
    } else if constexpr (LOG2N == 12) {
      stockham_pass<N, 16, 256, true, false, WAVE>(r, z, tid, tw);
    } else if constexpr (LOG2N == 13) {
      stockham_pass<N, 16, 256, true, true, WAVE>(r, z, tid, tw);
      stockham_pass<N, 2, 4096, true, false, WAVE>(r, z, tid, tw);
    } else {
      stockham_pass<N, 16, 256, true, true, WAVE>(r, z, tid, tw);
      stockham_pass<N, 4, 4096, true, false, WAVE>(r, z, tid, tw);
    }
  }
}

template <int LOG2N>
struct LastPass {   // radix / sub-size of the final pass of fft_passes<LOG2N>
  static constexpr int R = (LOG2N == 10 || LOG2N == 14) ? 4 : ((LOG2N == 12 || LOG2N == 8) ? 16 : 2);
  static constexpr int NS = (1 << LOG2N) / R;
};


// ---- mixed-radix Stockham passes (radix 2 / 3 / 4 / 5 / 7) for lengths 2^a 3^b 5^c 7^d: one wave owns a transform ------------------
// Butterfly j of a pass with sub-transform length ns reads j + t L / R from `src`, multiplies by exp(-2 pi i t (j mod ns) / (ns R))
// taken from ONE table tw_l[j] = exp(-2 pi i j / L), and writes (j - j mod ns) R + j mod ns + t ns to `dst` (autosort: natural
// order in, natural order out).  Used by stft_mixed_power16_kernel (stft_generic.hip) and istft_mixed_frames_kernel (istft.hip).
template <int R, typename S>
__device__ __forceinline__ void dft_small(cpx<S> (&v)[7]) {
  using C = cpx<S>;
  if constexpr (R == 2) {
    const C a = v[0], b = v[1];
    v[0] = a + b;
    v[1] = a - b;
  } else if constexpr (R == 4) {
    fft4(v[0], v[1], v[2], v[3]);
  } else if constexpr (R == 3) {
    constexpr S s = (S)0.86602540378443865, h = (S)0.5;
    const C t1 = v[1] + v[2], d = v[1] - v[2];
    const C t2 = {v[0].x - h * t1.x, v[0].y - h * t1.y};
    const C r = {s * d.y, -s * d.x};            // -i s d
    v[0] = v[0] + t1;
    v[1] = t2 + r;
    v[2] = t2 - r;
  } else if constexpr (R == 7) {
    // x_j +- x_(7-j) once, then three cosine and three sine combinations (forward kernel exp(-2 pi i j k / 7))
    constexpr S c1 = (S)0.62348980185873353, c2 = (S)-0.22252093395631440, c3 = (S)-0.90096886790241913;   // cos(2 pi k / 7)
    constexpr S s1 = (S)0.78183148246802981, s2 = (S)0.97492791218182361, s3 = (S)0.43388373911755812;    // sin(2 pi k / 7)
    const C a1 = v[1] + v[6], a2 = v[2] + v[5], a3 = v[3] + v[4], b1 = v[1] - v[6], b2 = v[2] - v[5], b3 = v[3] - v[4];
    const C x0 = v[0];
    const C e1 = {x0.x + c1 * a1.x + c2 * a2.x + c3 * a3.x, x0.y + c1 * a1.y + c2 * a2.y + c3 * a3.y};
    const C e2 = {x0.x + c2 * a1.x + c3 * a2.x + c1 * a3.x, x0.y + c2 * a1.y + c3 * a2.y + c1 * a3.y};
    const C e3 = {x0.x + c3 * a1.x + c1 * a2.x + c2 * a3.x, x0.y + c3 * a1.y + c1 * a2.y + c2 * a3.y};
    const C d1 = {s1 * b1.x + s2 * b2.x + s3 * b3.x, s1 * b1.y + s2 * b2.y + s3 * b3.y};
    const C d2 = {s2 * b1.x - s3 * b2.x - s1 * b3.x, s2 * b1.y - s3 * b2.y - s1 * b3.y};
    const C d3 = {s3 * b1.x - s1 * b2.x + s2 * b3.x, s3 * b1.y - s1 * b2.y + s2 * b3.y};
    v[0] = x0 + a1 + a2 + a3;
    v[1] = {e1.x + d1.y, e1.y - d1.x};            // e - i d
    v[6] = {e1.x - d1.y, e1.y + d1.x};
    v[2] = {e2.x + d2.y, e2.y - d2.x};
    v[5] = {e2.x - d2.y, e2.y + d2.x};
    v[3] = {e3.x + d3.y, e3.y - d3.x};
    v[4] = {e3.x - d3.y, e3.y + d3.x};
  } else {
    constexpr S c1 = (S)0.30901699437494742, c2 = (S)-0.80901699437494742;   // cos(2 pi / 5), cos(4 pi / 5)
    constexpr S s1 = (S)0.95105651629515357, s2 = (S)0.58778525229247313;    // sin(2 pi / 5), sin(4 pi / 5)
    const C a1 = v[1] + v[4], a2 = v[2] + v[3], b1 = v[1] - v[4], b2 = v[2] - v[3];
    const C x0 = v[0];
    const C e1 = {x0.x + c1 * a1.x + c2 * a2.x, x0.y + c1 * a1.y + c2 * a2.y};
    const C e2 = {x0.x + c2 * a1.x + c1 * a2.x, x0.y + c2 * a1.y + c1 * a2.y};
    const C d1 = {s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y};
    const C d2 = {s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y};
    v[0] = x0 + a1 + a2;
    v[1] = {e1.x + d1.y, e1.y - d1.x};            // e1 - i d1
    v[4] = {e1.x - d1.y, e1.y + d1.x};
    v[2] = {e2.x + d2.y, e2.y - d2.x};
    v[3] = {e2.x - d2.y, e2.y + d2.x};
  }
}

template <int R, typename S>
__device__ __forceinline__ void mixed_pass(const typename vec2_of<S>::type *src, typename vec2_of<S>::type *dst, int L, int ns, int lane,
                                           const typename vec2_of<S>::type *tw_l) {
  using V = typename vec2_of<S>::type;
  const int nb = L / R, stride = L / (ns * R);
  for (int j = lane; j < nb; j += 64) {
    const int k = j % ns;
    cpx<S> v[7];
#pragma unroll
    for (int t = 0; t < R; ++t) {
      const V u = src[j + t * nb];
      v[t] = {u.x, u.y};
    }
    if (ns > 1) {
      const int step = k * stride;              // t * step < R * L / R = L: no wrap
#pragma unroll
      for (int t = 1; t < R; ++t) {
        const V w = tw_l[t * step];
        v[t] = cmul(v[t], cpx<S>{w.x, w.y});
      }
    }
    dft_small<R, S>(v);
    const int o = (j - k) * R + k;
#pragma unroll
    for (int t = 0; t < R; ++t) {
      V q;
      q.x = v[t].x;
      q.y = v[t].y;
      dst[o + t * ns] = q;
    }
  }
}

// every pass of a plan (radices in radix[0 .. npass)), ping-pong between two buffers of L values; returns the buffer that
// holds the transform.  `lane` is the lane of the owning wave; DS operations of a wave complete in order, so the passes need
// only compiler fences between them.  S = float or double (the float64 interior).
template <typename S>
__device__ __forceinline__ typename vec2_of<S>::type *mixed_transform(typename vec2_of<S>::type *a, typename vec2_of<S>::type *b, int L, int npass,
                                                                     unsigned long long radices /* 4 bits per pass */, int lane,
                                                                     const typename vec2_of<S>::type *tw_l) {
  using V = typename vec2_of<S>::type;
  V *src = a, *dst = b;
  int ns = 1;
  for (int p = 0; p < npass; ++p) {
    const int r = (int)((radices >> (4 * p)) & 15ull);   // uniform; packed, so that the plan is scalars and not an indexed array in memory
    if (r == 4) mixed_pass<4, S>(src, dst, L, ns, lane, tw_l);
    else if (r == 2) mixed_pass<2, S>(src, dst, L, ns, lane, tw_l);
    else if (r == 5) mixed_pass<5, S>(src, dst, L, ns, lane, tw_l);
    else if (r == 7) mixed_pass<7, S>(src, dst, L, ns, lane, tw_l);
    else mixed_pass<3, S>(src, dst, L, ns, lane, tw_l);
    ns *= r;
    asm volatile("" ::: "memory");
    V *sw = src;
    src = dst;
    dst = sw;
  }
  return src;
}

#ifdef SMX_STAMPS
// Diagnostic build only (make STAMPS=1): per-phase cycle sums of every wave, read back with
// smx_debug_read_stamps().  Never compiled into the shipped library; no output depends on it.
constexpr int kStampSlots = 24;
__device__ unsigned long long g_stamp_sums[4096 * 16 * kStampSlots];
#ifdef SMX_STAMPS_COARSE
// only the whole-loop clock check (slots 20 / 21: shader cycles and 100 MHz reference ticks around the tile loop):
// the in-kernel clock without the per-phase stamps' own cost (MI355X_MICROARCH.md 'DVFS give-back' item 6)
#define SMX_STAMP(i) do { (void)stamp_prev; } while (0)
#else
#define SMX_STAMP(i)                                                                      \
  do {                                                                                    \
    unsigned long long t__;                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    stamp_sum[i] += t__ - stamp_prev;                                                     \
    stamp_prev = t__;                                                                     \
  } while (0)
#endif
#else
#define SMX_STAMP(i) do { } while (0)
#endif


}  // namespace fftdev
}  // namespace smx
