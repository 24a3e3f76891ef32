// STFT kernels for every geometry the fused fft-2048 kernels of stft_fast.hip do not serve: any fft_size, hop,
// alignment and pad mode, float32 or float64 audio, float32 or float64 interior (reference semantics for every
// Stft.Config, stft.ml:356-364 + :670-674).  In the order the dispatcher (launch_stft_generic, at the end) prefers them:
//
//   stft_stockham_power16_kernel     fft 512 .. 8192, power output (and, MEL, the fused mel / projection tail): FT
//                                    frames per workgroup on the Stockham passes of fft_device.hpp, real form
//                                    (one half-size complex transform + post-pass), NO stage: each frame's column
//                                    returns into its own work buffer and leaves through columns_out; float32, or
//                                    double for the float64 interior;
//   stft_stockham_complex16_kernel   the same for complex output (X[M] packed with X[0], XOR-placed columns);
//   stft_bluestein_power16_kernel    even sizes up to 1024 that are not powers of two (fft 400 ...): chirp-z of
//                                    length N/2 on the same passes, same column tail;
//   stft_stockham_real_kernel        the staged form ([bins][FT + 1] stage in LDS) for what is left of fft 512 ..
//   stft_stockham_kernel             16384 (full-size complex form for fft 256 and from 4096 on);
//   stft_bluestein_real_kernel /     chirp-z, half length for even sizes, full length for odd ones, up to fft 8192;
//   stft_bluestein_kernel
//   stft_generic_kernel              everything else: radix-2 passes in LDS (power of two) or a direct DFT against an
//                                    N-entry float64-built table, float32 or float64, FT frames staged and written
//                                    frames-fastest.
#include <cstdlib>

#include "fft_device.hpp"
#include "smx_internal.hpp"

namespace smx {
namespace {

template <typename T> struct Vec2;
template <> struct Vec2<float> { using type = float2; };
template <> struct Vec2<double> { using type = double2; };

struct GenericArgs {
  const void *x;
  int64_t n, x_stride, lead;
  int64_t fft, hop, left;
  int pad;
  double pad_value;
  int64_t p0, count;
  int mode;
  double power;
  void *out;
  int64_t out_stride, out_offset, bins;
  const void *window;
  const void *twiddle;
  int log2n;   // >= 0 for the power-of-two kernel
  int ft;      // frames per workgroup
  int direct;  // Stockham kernel: results go straight from registers to memory (no room for an LDS stage)
};

// two neighbouring samples as one access; only element alignment is promised (frames start anywhere)
template <typename T> struct Pair;
template <> struct Pair<float> { typedef float type __attribute__((ext_vector_type(2), aligned(4))); };
template <> struct Pair<double> { typedef double type __attribute__((ext_vector_type(2), aligned(8))); };

template <typename Tin>
__device__ inline double fetch_sample(const Tin *x, int64_t n, int64_t s, int pad, double pad_value) {
  if (s >= 0 && s < n) return (double)x[s];
  if (pad == SMX_PAD_REFLECT) {  // stft.ml:300-305
    if (n == 1) return (double)x[0];
    const int64_t period = 2 * (n - 1);
    int64_t m = s % period;
    if (m < 0) m += period;
    return (double)x[m < n ? m : period - m];
  }
  if (pad == SMX_PAD_EDGE) return (double)x[s < 0 ? 0 : n - 1];
  return (double)(Tin)pad_value;   // the padded signal is built in the input dtype (stft.ml:318-338), then widened
}

// |z|^p with the reference's rounding order (stft.ml:670-674): the spectrum is
// rounded to the storage component type first, the magnitude is taken in the
// real dtype, then squared / kept / raised.
// kept out of line: the general power is rare and long, and the callers below are unrolled 16 times
__device__ __noinline__ float general_power_f32(float p2, float half_power) { return powf(p2, half_power); }

template <typename Tacc, typename Tout>
__device__ inline Tout magnitude_pow(Tacc re, Tacc im, double power) {
  if constexpr (sizeof(Tacc) == 4) {
    const float p2 = re * re + im * im;
    if (power == 2.0) return (Tout)p2;
    if (power == 1.0) return (Tout)sqrtf(p2);
    return (Tout)general_power_f32(p2, (float)(0.5 * power));
  } else {
    const Tout r = (Tout)re, i = (Tout)im;
    const Tout m = (Tout)sqrt((double)r * (double)r + (double)i * (double)i);
    if (power == 2.0) return m * m;
    if (power == 1.0) return m;
    return (Tout)pow((double)m, power);
  }
}

template <typename Tacc>
__device__ inline unsigned bitrev(unsigned v, int bits) {
  return bits == 0 ? 0u : (__brev(v) >> (32 - bits));
}

// LDS layout: work[N] complex<Tacc> | stage[bins][FT+1] of Tout or complex<Tout>
template <typename Tin, typename Tacc, typename Tout, bool POW2>
__global__ void __launch_bounds__(256) stft_generic_kernel(GenericArgs a) {
  using C = typename Vec2<Tacc>::type;
  using CO = typename Vec2<Tout>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C *work = reinterpret_cast<C *>(smem);
  const int64_t N = a.fft;
  const int ft = a.ft;
  const int64_t tiles = (a.count + ft - 1) / ft;
  const int64_t clip = blockIdx.x / tiles;
  const int64_t tile = blockIdx.x % tiles;
  const Tin *x = reinterpret_cast<const Tin *>(a.x) + clip * a.x_stride;
  const Tacc *window = reinterpret_cast<const Tacc *>(a.window);
  const C *tw = reinterpret_cast<const C *>(a.twiddle);
  const int64_t bins = a.bins;
  const int tid = threadIdx.x;
  // staging region after the work buffer
  const size_t work_bytes = POW2 ? (size_t)N * sizeof(C) : (((size_t)N * sizeof(Tacc) + 15) / 16 * 16);
  unsigned char *stage = smem + work_bytes;
  const int sstride = ft + 1;

  const int64_t f0 = tile * ft;
  const int nf = (int)((a.count - f0) < ft ? (a.count - f0) : ft);
  for (int f = 0; f < nf; ++f) {
    const int64_t p = a.p0 + f0 + f;
    const int64_t s0 = p * a.hop - a.left;
    if constexpr (POW2) {
      for (int64_t i = tid; i < N; i += blockDim.x) {
        const Tacc v = (Tacc)(fetch_sample<Tin>(x, a.n, s0 + i, a.pad, a.pad_value)) * window[i];
        C z;
        z.x = v;
        z.y = (Tacc)0;
        work[bitrev<Tacc>((unsigned)i, a.log2n)] = z;
      }
      for (int64_t half = 1; half < N; half <<= 1) {
        __syncthreads();
        const int64_t tstep = (N >> 1) / half;
        for (int64_t b = tid; b < (N >> 1); b += blockDim.x) {
          const int64_t j = b & (half - 1);
          const int64_t i0 = ((b - j) << 1) + j;
          const int64_t i1 = i0 + half;
          const C w = tw[j * tstep];
          const C u = work[i0], v = work[i1];
          C t;
          t.x = w.x * v.x - w.y * v.y;
          t.y = w.x * v.y + w.y * v.x;
          C lo, hi;
          lo.x = u.x + t.x; lo.y = u.y + t.y;
          hi.x = u.x - t.x; hi.y = u.y - t.y;
          work[i0] = lo;
          work[i1] = hi;
        }
      }
      __syncthreads();
      for (int64_t k = tid; k < bins; k += blockDim.x) {
        const C z = work[k];
        if (a.mode == OUT_COMPLEX) {
          CO o;
          o.x = (Tout)z.x;
          o.y = (Tout)z.y;
          reinterpret_cast<CO *>(stage)[k * sstride + f] = o;
        } else {
          reinterpret_cast<Tout *>(stage)[k * sstride + f] = magnitude_pow<Tacc, Tout>(z.x, z.y, a.power);
        }
      }
      __syncthreads();
    } else {
      Tacc *xw = reinterpret_cast<Tacc *>(smem);
      for (int64_t i = tid; i < N; i += blockDim.x)
        xw[i] = (Tacc)(fetch_sample<Tin>(x, a.n, s0 + i, a.pad, a.pad_value)) * window[i];
      __syncthreads();
      for (int64_t k = tid; k < bins; k += blockDim.x) {
        Tacc re = 0, im = 0;
        int64_t idx = 0;
        for (int64_t i = 0; i < N; ++i) {
          const C w = tw[idx];
          re += xw[i] * w.x;
          im += xw[i] * w.y;
          idx += k;
          if (idx >= N) idx -= N;
        }
        if (a.mode == OUT_COMPLEX) {
          CO o;
          o.x = (Tout)re;
          o.y = (Tout)im;
          reinterpret_cast<CO *>(stage)[k * sstride + f] = o;
        } else {
          reinterpret_cast<Tout *>(stage)[k * sstride + f] = magnitude_pow<Tacc, Tout>(re, im, a.power);
        }
      }
      __syncthreads();
    }
  }
  // flush: frames fastest
  const int total = (int)bins * nf;       // bins <= 8193 on this path, nf <= 16
  const int64_t obase = clip * bins * a.out_stride + a.out_offset + f0;
  const int shift = 31 - __builtin_clz((unsigned)a.ft);   // ft is a power of two: only a ragged last tile divides
  for (int e = tid; e < total; e += blockDim.x) {
    int64_t k;
    int f;
    if (nf == a.ft) {
      k = e >> shift;
      f = e & (a.ft - 1);
    } else {
      k = e / nf;
      f = e - (int)k * nf;
    }
    if (a.mode == OUT_COMPLEX)
      reinterpret_cast<CO *>(a.out)[obase + k * a.out_stride + f] =
          reinterpret_cast<const CO *>(stage)[k * sstride + f];
    else
      reinterpret_cast<Tout *>(a.out)[obase + k * a.out_stride + f] =
          reinterpret_cast<const Tout *>(stage)[k * sstride + f];
  }
}

constexpr size_t kLdsLimit = 160 * 1024;

// stage [bins][ft + 1] -> out[clip][bins][frames], frames fastest: runs of nf elements per bin row.  ft is a power
// of two, so a full tile splits the element index with a shift; only a clip's last, ragged tile divides.
template <typename Tout>
__device__ __forceinline__ void flush_stage(const GenericArgs &a, const unsigned char *stage, int64_t clip, int64_t f0, int nf) {
  using CO = typename Vec2<Tout>::type;
  const int ft = a.ft, sstride = ft + 1;
  const int total = (int)a.bins * nf;
  const int64_t obase = clip * a.bins * a.out_stride + a.out_offset + f0;
  const int shift = 31 - __builtin_clz((unsigned)ft);
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    int k, f;
    if (nf == ft) {
      k = e >> shift;
      f = e & (ft - 1);
    } else {
      k = e / nf;
      f = e - k * nf;
    }
    if (a.mode == OUT_COMPLEX)
      reinterpret_cast<CO *>(a.out)[obase + (int64_t)k * a.out_stride + f] = reinterpret_cast<const CO *>(stage)[k * sstride + f];
    else
      reinterpret_cast<Tout *>(a.out)[obase + (int64_t)k * a.out_stride + f] = reinterpret_cast<const Tout *>(stage)[k * sstride + f];
  }
}

// ---- power-of-two sizes 1024 .. 16384, float32 interior: Stockham passes of fft_device.hpp ---------------------
// N/16 threads own one frame (16 points each in registers, 3-4 LDS round trips instead of log2 N radix-2 passes);
// a workgroup of >= 256 threads transforms G = 256 / (N/16) frames at a time (one for N >= 4096) and FT frames in
// all, staged in LDS and written frames-fastest like the kernels above.  The frame is transformed as a complex
// signal with zero imaginary part; bins above N/2 are dropped.
template <int LOG2N, typename Tin>
__global__ void __launch_bounds__((1 << LOG2N) / 16 < 256 ? 256 : (1 << LOG2N) / 16) stft_stockham_kernel(GenericArgs a) {
  using namespace fftdev;
  constexpr int N = 1 << LOG2N, T = N / 16, G = T < 256 ? 256 / T : 1;
  constexpr int RL = LastPass<LOG2N>::R, NSL = LastPass<LOG2N>::NS, GL = 16 / RL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2 *work = reinterpret_cast<float2 *>(smem);                       // G buffers of N complex
  unsigned char *stage = smem + (size_t)G * N * sizeof(float2);
  const int ft = a.ft, sstride = ft + 1;
  const int64_t tiles = (a.count + ft - 1) / ft;
  const int64_t clip = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const Tin *x = reinterpret_cast<const Tin *>(a.x) + clip * a.x_stride;
  const float *window = reinterpret_cast<const float *>(a.window);
  const float2 *tw = reinterpret_cast<const float2 *>(a.twiddle);         // exp(-2 pi i j / N), j < N
  const int64_t bins = a.bins;
  const int tid = threadIdx.x % T, grp = threadIdx.x / T;
  const int64_t f0 = tile * ft;
  const int nf = (int)((a.count - f0) < ft ? (a.count - f0) : ft);
  for (int fb = 0; fb < nf; fb += G) {
    const int f = fb + grp;
    const bool have = f < nf;                          // uniform per group of T threads (T >= 64: per wave)
    c32 r[16];
    if (have) {
      const int64_t s0 = (a.p0 + f0 + f) * a.hop - a.left;
      if (s0 >= 0 && s0 + N <= a.n) {   // the frame lies inside the signal (uniform per group): plain loads
        const Tin *xs = x + s0;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          const int i = tid + T * m;
          r[m] = {(float)xs[i] * window[i], 0.0f};
        }
      } else {
#pragma unroll 1
        for (int m = 0; m < 16; ++m) {
          const int i = tid + T * m;
          const float v = (float)fetch_sample<Tin>(x, a.n, s0 + i, a.pad, a.pad_value) * window[i];
#pragma unroll
          for (int mm = 0; mm < 16; ++mm)
            if (mm == m) r[mm] = {v, 0.0f};
        }
      }
    } else {
#pragma unroll
      for (int m = 0; m < 16; ++m) r[m] = {0.0f, 0.0f};
    }
    // every thread takes part in the barriers of the passes; groups without a frame transform zeros
    fft_passes<LOG2N, true, (T <= 64)>(r, work + (size_t)grp * N, tid, tw);   // groups of at most 64 threads are wave-private
    if (have) {
#pragma unroll
      for (int i = 0; i < GL; ++i)
#pragma unroll
        for (int j = 0; j < RL; ++j) {
          const int k = out_index<RL, NSL, T>(tid, i, j);
          if (k < bins) {
            const c32 z = r[i * RL + j];
            if (a.direct) {
              const int64_t o = clip * bins * a.out_stride + a.out_offset + f0 + f + (int64_t)k * a.out_stride;
              if (a.mode == OUT_COMPLEX) reinterpret_cast<float2 *>(a.out)[o] = make_float2(z.x, z.y);
              else reinterpret_cast<float *>(a.out)[o] = magnitude_pow<float, float>(z.x, z.y, a.power);
            } else if (a.mode == OUT_COMPLEX) {
              reinterpret_cast<float2 *>(stage)[k * sstride + f] = make_float2(z.x, z.y);
            } else {
              reinterpret_cast<float *>(stage)[k * sstride + f] = magnitude_pow<float, float>(z.x, z.y, a.power);
            }
          }
        }
    }
    __syncthreads();   // the next round's first pass writes the work buffers again
  }
  if (a.direct) return;
  flush_stage<float>(a, stage, clip, f0, nf);
}

template <int LOG2N>
bool launch_stockham(const StftJob &job, GenericArgs a) {
  constexpr int N = 1 << LOG2N, T = N / 16, G = T < 256 ? 256 / T : 1, THREADS = T < 256 ? 256 : T;
  const size_t elem_out = (job.mode == OUT_COMPLEX ? 2 : 1) * sizeof(float);
  const size_t work = (size_t)G * N * sizeof(float2);
  auto stage_bytes = [&](int ft) { return (size_t)a.bins * (size_t)(ft + 1) * elem_out + 16; };
  int ft = 16;
  while (ft > G && work + stage_bytes(ft) > kLdsLimit) ft >>= 1;
  a.direct = work + stage_bytes(ft) > kLdsLimit ? 1 : 0;   // fft 16384: the work buffer alone is 128 KB
  if (a.direct) ft = G;
  if (work > kLdsLimit) return false;
  a.ft = ft;
  const int64_t blocks = a.lead * ((a.count + ft - 1) / ft);
  if (blocks > 2147483647LL) throw Failure("stft: too many frame tiles for one launch");
  const size_t lds = a.direct ? work : work + stage_bytes(ft);
  auto kernel = stft_stockham_kernel<LOG2N, float>;
  SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(THREADS), lds, job.stream, a);
  SMX_HIP_CHECK(hipGetLastError());
  return true;
}

// ---- power-of-two sizes 512 .. 16384, float32 interior: the frame as ONE half-size complex transform ----------
// z[i] = x[2i] w[2i] + i x[2i+1] w[2i+1] (M = N/2 points, the window pre-halved), Z = FFT_M(z), then
//   X[k] = (Z[k] + conj Z[M-k]) - i W_N^k (Z[k] - conj Z[M-k]),  X[M] = Re Z[0] - Im Z[0]
// -- half the passes' work of the kernel above for one more LDS round trip (the partner Z[M-k] lives in another
// thread).  M/16 threads own a frame; transforms of at most 64 threads are wave-private (no workgroup barriers).
// S = double: the reference's float64 interior (float32 or float64 audio, results rounded once to Tout) on the same
// passes, fft 512 .. 4096 (16 complex doubles per thread need the 256-register budget of a 256-thread workgroup).
template <int LOG2N, typename Tin, typename S, typename Tout>
__global__ void __launch_bounds__((1 << LOG2N) / 32 < 256 ? 256 : (1 << LOG2N) / 32)
    stft_stockham_real_kernel(GenericArgs a, const typename fftdev::vec2_of<S>::type *tw_m,
                              const typename fftdev::vec2_of<S>::type *tw_n) {
  using namespace fftdev;
  using V = typename vec2_of<S>::type;
  using CO = typename Vec2<Tout>::type;
  constexpr int N = 1 << LOG2N, LOG2M = LOG2N - 1, M = N / 2, T = M / 16, G = T < 256 ? 256 / T : 1;
  constexpr bool WAVE = T <= 64;
  constexpr int RL = LastPass<LOG2M>::R, NSL = LastPass<LOG2M>::NS, GL = 16 / RL;
  // float32: the window table is pre-halved; float64: the config's own window, halved here (exact either way)
  constexpr S kHalf = sizeof(S) == 8 ? (S)0.5 : (S)1.0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  V *work = reinterpret_cast<V *>(smem);                                 // G buffers of M complex
  unsigned char *stage = smem + (size_t)G * M * sizeof(V);
  const int ft = a.ft, sstride = ft + 1;
  const int64_t tiles = (a.count + ft - 1) / ft;
  const int64_t clip = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const Tin *x = reinterpret_cast<const Tin *>(a.x) + clip * a.x_stride;
  const S *window = reinterpret_cast<const S *>(a.window);
  const int64_t bins = a.bins;
  const int tid = threadIdx.x % T, grp = threadIdx.x / T;
  V *z = work + (size_t)grp * M;
  const int64_t f0 = tile * ft;
  const int nf = (int)((a.count - f0) < ft ? (a.count - f0) : ft);
  auto emit = [&](int f, int k, S re, S im) {
    if (a.direct) {
      const int64_t o = clip * bins * a.out_stride + a.out_offset + f0 + f + (int64_t)k * a.out_stride;
      if (a.mode == OUT_COMPLEX) {
        CO c;
        c.x = (Tout)re;
        c.y = (Tout)im;
        reinterpret_cast<CO *>(a.out)[o] = c;
      } else {
        reinterpret_cast<Tout *>(a.out)[o] = magnitude_pow<S, Tout>(re, im, a.power);
      }
    } else if (a.mode == OUT_COMPLEX) {
      CO c;
      c.x = (Tout)re;
      c.y = (Tout)im;
      reinterpret_cast<CO *>(stage)[k * sstride + f] = c;
    } else {
      reinterpret_cast<Tout *>(stage)[k * sstride + f] = magnitude_pow<S, Tout>(re, im, a.power);
    }
  };
  for (int fb = 0; fb < nf; fb += G) {
    const int f = fb + grp;
    const bool have = f < nf;                          // uniform per group of T threads
    cpx<S> r[16];
    if (have) {
      const int64_t s0 = (a.p0 + f0 + f) * a.hop - a.left;
      if (s0 >= 0 && s0 + N <= a.n) {   // the frame lies inside the signal (uniform per group): plain loads
        const Tin *xs = x + s0;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          const int i = 2 * (tid + T * m);
          // one 8-byte access for the two samples (element-aligned only) and one for their window values
        const auto xv = *reinterpret_cast<const typename Pair<Tin>::type *>(xs + i);
        const V wv = reinterpret_cast<const V *>(window)[tid + T * m];
        r[m] = {(S)xv.x * wv.x * kHalf, (S)xv.y * wv.y * kHalf};
        }
      } else {
#pragma unroll 1
        for (int m = 0; m < 16; ++m) {
          const int i = 2 * (tid + T * m);
          const S v0 = (S)fetch_sample<Tin>(x, a.n, s0 + i, a.pad, a.pad_value) * window[i] * kHalf;
          const S v1 = (S)fetch_sample<Tin>(x, a.n, s0 + i + 1, a.pad, a.pad_value) * window[i + 1] * kHalf;
#pragma unroll
          for (int mm = 0; mm < 16; ++mm)
            if (mm == m) r[mm] = {v0, v1};
        }
      }
    } else {
#pragma unroll
      for (int m = 0; m < 16; ++m) r[m] = {(S)0, (S)0};
    }
    fft_passes<LOG2M, true, WAVE>(r, z, tid, tw_m);
    // the last pass left its results in registers (its reads of z are behind a sync): Z in natural order
#pragma unroll
    for (int i = 0; i < GL; ++i)
#pragma unroll
      for (int j = 0; j < RL; ++j) {
        V o;
        o.x = r[i * RL + j].x;
        o.y = r[i * RL + j].y;
        z[swz(out_index<RL, NSL, T>(tid, i, j))] = o;
      }
    stockham_sync<WAVE>();
    if (have) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int k = tid + T * m;
        const V zk = z[swz(k)], zm = z[swz((M - k) & (M - 1))];
        const S er = zk.x + zm.x, ei = zk.y - zm.y;            // Z[k] + conj Z[M-k]
        const S dr = zk.x - zm.x, di = zk.y + zm.y;            // Z[k] - conj Z[M-k]
        const V w = tw_n[k];                                   // exp(-2 pi i k / N)
        // -i w d = -i (w.x + i w.y)(dr + i di) = (w.x di + w.y dr) - i (w.x dr - w.y di)
        emit(f, k, er + (w.x * di + w.y * dr), ei - (w.x * dr - w.y * di));
      }
      if (tid == 0) {
        const V z0 = z[swz(0)];
        emit(f, M, (S)2 * (z0.x - z0.y), (S)0);
      }
    }
    __syncthreads();   // the next round's first pass writes the work buffers again
  }
  if (a.direct) return;
  flush_stage<Tout>(a, stage, clip, f0, nf);
}

template <int LOG2N, typename Tin, typename S, typename Tout>
bool launch_stockham_real(const StftJob &job, GenericArgs a, const StftTables &t) {
  using V = typename fftdev::vec2_of<S>::type;
  constexpr int N = 1 << LOG2N, M = N / 2, T = M / 16, G = T < 256 ? 256 / T : 1, THREADS = T < 256 ? 256 : T;
  constexpr bool wide = sizeof(S) == 8;
  static_assert(!wide || THREADS == 256, "the float64 form needs the register budget of a 256-thread workgroup");
  const void *tw_m = wide ? (const void *)t.fast_w_m_f64 : (const void *)t.fast_w_m;
  const void *tw_n = wide ? (const void *)t.twiddle_f64 : (const void *)t.fast_w_n;
  const void *window = wide ? (const void *)t.window_f64 : (const void *)t.fast_window;
  if (!window || !tw_m || !tw_n) return false;
  const size_t elem_out = (job.mode == OUT_COMPLEX ? 2 : 1) * sizeof(Tout);
  const size_t work = (size_t)G * M * sizeof(V);
  auto stage_bytes = [&](int ft) { return (size_t)a.bins * (size_t)(ft + 1) * elem_out + 16; };
  int ft = 16;
  while (ft > G && work + stage_bytes(ft) > kLdsLimit) ft >>= 1;
  a.direct = work + stage_bytes(ft) > kLdsLimit ? 1 : 0;
  if (a.direct) ft = G;
  if (work > kLdsLimit) return false;
  a.ft = ft;
  a.window = window;
  const int64_t blocks = a.lead * ((a.count + ft - 1) / ft);
  if (blocks > 2147483647LL) throw Failure("stft: too many frame tiles for one launch");
  const size_t lds = a.direct ? work : work + stage_bytes(ft);
  auto kernel = stft_stockham_real_kernel<LOG2N, Tin, S, Tout>;
  SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(THREADS), lds, job.stream, a, (const V *)tw_m, (const V *)tw_n);
  SMX_HIP_CHECK(hipGetLastError());
  return true;
}

// the float64-interior form for one (audio, output) dtype pair: fft 512 .. 4096
template <typename Tin, typename Tout>
bool launch_stockham_wide(const StftJob &job, const GenericArgs &a, const StftTables &t, int64_t fft) {
  switch (fft) {
    case 512: return launch_stockham_real<9, Tin, double, Tout>(job, a, t);
    case 1024: return launch_stockham_real<10, Tin, double, Tout>(job, a, t);
    case 2048: return launch_stockham_real<11, Tin, double, Tout>(job, a, t);
    case 4096: return launch_stockham_real<12, Tin, double, Tout>(job, a, t);
    default: return false;
  }
}

// ---- fft 512 / 1024, float32, power output: 16 frames per workgroup with NO separate stage ----------------------
// The kernel above keeps a [bins][17] stage next to its work buffers, which leaves two 4-wave workgroups per CU at
// fft 1024.  Here a workgroup owns 16 frames at once (16 x M/16 threads), and after the post-pass each frame's
// |X|^p column goes back into that frame's own, now dead, work buffer (M + 1 floats in the room of M complex),
// rotated by 2 f floats so that the flush -- 16 frames of one bin per 16 lanes -- reads 32 distinct banks per
// half-wave.  LDS is the work buffers alone: 32 KB (fft 512, five workgroups per CU) / 64 KB (fft 1024, two
// 8-wave workgroups).  Transforms are wave-private (at most 32 threads each): one workgroup barrier in all.
// MEL: the 16 power columns never leave the chip: W (banded, zero-padded float32 image of the float64 filterbank) x
// columns on v_mfma_f32_16x16x4_f32, one 16-row tile of W per wave at a time over the tile's own band of bins, and
// the [n_mels; 16 frames] block goes out as 64-byte row runs (Soundml.mel_spectrogram for fft 512 / 1024).
struct MelTail {
  const float *w;             // [n_mels_pad / 16][k_pad / 4][64]: the weights in MFMA A-operand order
  const int *band_lo, *band_hi;   // per 16-row tile: first / one-past-last bin any of its rows touches
  int n_mels, k_pad;
  float *out;                 // [lead; n_mels; count]
};

// The common tail of the stage-free kernels: FT power columns lie in LDS, frame f at cols + f * BUF + 2 f (BUF floats
// per frame buffer, `bins` values each).  MEL: banded filterbank x columns on the fp32 MFMA, [n_mels; 16] out;
// otherwise the columns leave as 16-byte stores of 4 frames of one bin.
// the (lo, hi) bin range of the first two row tiles a wave will take, fetched at kernel entry so that the mel tail does
// not start with a memory round trip (measured: that wait, per workgroup, was most of the tail)
struct TileBands {
  int lo[2], hi[2];
};
template <bool MEL>
__device__ __forceinline__ TileBands load_tile_bands(const MelTail &mt) {
  TileBands tb{{0, 0}, {0, 0}};
  if constexpr (MEL) {
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6, ntiles = (mt.n_mels + 15) >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int t = wave + i * nwaves;
      if (i == 0 && wave >= ntiles && ntiles < nwaves) t = ntiles - 1 - ((wave - ntiles) % ntiles);   // a helper's tile
      if (16 * t < mt.n_mels) {
        tb.lo[i] = mt.band_lo[t];
        tb.hi[i] = mt.band_hi[t];
      }
    }
  }
  return tb;
}

template <int BUF, int FT, bool MEL, typename Tout = float>
__device__ __forceinline__ void columns_out(const GenericArgs &a, const MelTail &mt, const Tout *cols, int bins, int nf,
                                            int64_t clip, int64_t f0, const TileBands &tb = TileBands{}, float *partials = nullptr,
                                            int pstride = 256 /* floats between the helper waves' partial tiles */) {
  if constexpr (MEL) {
    static_assert(!MEL || (FT == 16 && sizeof(Tout) == 4), "the MFMA tile is 16 float32 frames wide");
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int n = lane & 15, kk = lane >> 4;           // B[k = kk][n = frame], A[m = n][k = kk], D[4 kk + i][n]
    const float *colb = cols + n * (BUF) + 2 * n + kk;
    // One 16-row tile of the weights over the bin range [k_begin, k_end) (16-aligned), into acc.  Four k-steps (16
    // bins) per trip, two trips' operands in flight, two accumulator chains; the weights are in operand order (one
    // 256-byte run per load); trips may run past the band: W is zero there.
    auto tile_product = [&](int r0, int k_begin, int k_end) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const float *wt = mt.w + (int64_t)(r0 >> 4) * (mt.k_pad / 4) * 64 + lane;   // this tile's operands, this lane's slot
      int k0 = k_begin;
      for (; k0 + 32 <= k_end; k0 += 32) {
        float av[8], bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          av[j] = wt[(k0 / 4 + j) * 64];
          bv[j] = k0 + 4 * j + kk < bins ? colb[k0 + 4 * j] : 0.0f;   // past the Nyquist bin the column is not defined
        }
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j + 1], bv[j + 1], acc1, 0, 0, 0);
        }
      }
      for (; k0 < k_end; k0 += 16) {
        float av[4], bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          av[j] = wt[(k0 / 4 + j) * 64];
          bv[j] = k0 + 4 * j + kk < bins ? colb[k0 + 4 * j] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j + 1], bv[j + 1], acc1, 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += acc1[i];
      return acc;
    };
    auto tile_range = [&](int lo, int hi, int &k_begin, int &k_end) {
      k_begin = lo & ~15;                                    // 16-aligned trips never straddle the end of a padded row
      k_end = k_begin + (hi - k_begin + 15) / 16 * 16;
      k_end = k_end < mt.k_pad ? k_end : mt.k_pad;          // k_pad is a multiple of 32
    };
    auto store_tile = [&](int r0, const f32x4 &acc) {
      if (n < nf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = r0 + 4 * kk + i;
          if (row < mt.n_mels) mt.out[(clip * mt.n_mels + row) * a.count + f0 + n] = acc[i];
        }
      }
    };
    const int ntiles = (mt.n_mels + 15) >> 4;
    if (ntiles == 0) return;
    if (ntiles < nwaves && partials) {
      // Fewer tiles than waves (80 mels on 8 waves; one tile of dense weights): the idle waves take shares of the
      // widest -- for a mel bank the highest -- tiles' bin ranges.  Helper j serves tile ntiles - 1 - (j mod ntiles);
      // a tile's range is cut into 1 + (its helpers) equal 16-aligned pieces, piece 0 is the owner's; helpers leave
      // their 16 x 16 partial in LDS and the owner adds them in piece order: the same sum on every run.
      const int helpers = nwaves - ntiles;
      int tile, piece, pieces;
      if (wave < ntiles) {
        tile = wave;
        piece = 0;
        const int back = ntiles - 1 - tile;                  // helper indices back, back + ntiles, ... serve this tile
        pieces = 1 + (back < helpers ? (helpers - 1 - back) / ntiles + 1 : 0);
      } else {
        const int j = wave - ntiles;
        tile = ntiles - 1 - (j % ntiles);
        piece = 1 + j / ntiles;
        const int back = j % ntiles;
        pieces = 1 + (helpers - 1 - back) / ntiles + 1;
      }
      const int r0 = 16 * tile;
      int lo = tb.lo[0], hi = tb.hi[0];                      // owner's or helper's tile: fetched at kernel entry
      lo = __builtin_amdgcn_readfirstlane(lo);
      hi = __builtin_amdgcn_readfirstlane(hi);
      int k_begin, k_end;
      tile_range(lo, hi, k_begin, k_end);
      const int trips = (k_end - k_begin) / 16, per = (trips + pieces - 1) / pieces;
      int kb = k_begin + 16 * per * piece, ke = kb + 16 * per;
      kb = kb < k_end ? kb : k_end;
      ke = ke < k_end ? ke : k_end;
      f32x4 acc = tile_product(r0, kb, ke);
      if (piece > 0) {
        float *slot = partials + (wave - ntiles) * pstride;
#pragma unroll
        for (int i = 0; i < 4; ++i) slot[i * 64 + lane] = acc[i];
      }
      __syncthreads();
      if (piece == 0) {
        const int back = ntiles - 1 - tile;
        for (int j = back; j < helpers; j += ntiles) {       // piece order: helper back is piece 1, back + ntiles piece 2, ...
          const float *slot = partials + j * pstride;
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] += slot[i * 64 + lane];
        }
        store_tile(r0, acc);
      }
      return;
    }
    int ti = 0;
    for (int r0 = 16 * wave; r0 < mt.n_mels; r0 += 16 * nwaves, ++ti) {
      // bins any of the tile's 16 rows touches (host-built per tile; the first two came in at kernel entry)
      int lo, hi;
      if (ti == 0) { lo = tb.lo[0]; hi = tb.hi[0]; }
      else if (ti == 1) { lo = tb.lo[1]; hi = tb.hi[1]; }
      else { lo = mt.band_lo[r0 >> 4]; hi = mt.band_hi[r0 >> 4]; }
      lo = __builtin_amdgcn_readfirstlane(lo);
      hi = __builtin_amdgcn_readfirstlane(hi);
      int k_begin, k_end;
      tile_range(lo, hi, k_begin, k_end);
      store_tile(r0, tile_product(r0, k_begin, k_end));
    }
    return;
  }
  const int total = bins * nf;
  const int64_t obase = clip * a.bins * a.out_stride + a.out_offset + f0;
  Tout *out = reinterpret_cast<Tout *>(a.out);
  if (nf == FT) {   // a lane takes 16 bytes of one bin's row (4 float or 2 double frames): conflict-free LDS reads, one store
    constexpr int PER = 16 / sizeof(Tout), QF = FT / PER;
    for (int e = threadIdx.x; e < bins * QF; e += blockDim.x) {
      const int k = e / QF, g = PER * (e % QF);
      const Tout *src = cols + g * (BUF) + 2 * g + k;
      Tout *dst = out + obase + (int64_t)k * a.out_stride + g;
      if constexpr (sizeof(Tout) == 4) {
        using f32x4 = __attribute__((ext_vector_type(4))) float;
        const f32x4 v = {src[0], src[BUF + 2], src[2 * (BUF + 2)], src[3 * (BUF + 2)]};
        asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(dst), "v"(v) : "memory");
      } else {
        using f64x2 = __attribute__((ext_vector_type(2))) double;
        const f64x2 v = {src[0], src[BUF + 2]};
        asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(dst), "v"(v) : "memory");
      }
    }
    return;
  }
  for (int e = threadIdx.x; e < total; e += blockDim.x) {   // a clip's ragged last tile
    const int k = e / nf, g = e - k * nf;
    out[obase + (int64_t)k * a.out_stride + g] = cols[g * (BUF) + 2 * g + k];
  }
}

// S = double: the float64 interior for float32 audio (window, transform and |.|^p in float64, one rounding into the
// float32 column), fewer frames per workgroup since a frame's buffer is twice as large.
template <int LOG2N, typename Tin, bool MEL, int FT, typename S = float, typename Tout = float>
__global__ void __launch_bounds__(FT * ((1 << LOG2N) / 32)) stft_stockham_power16_kernel(GenericArgs a, const typename fftdev::vec2_of<S>::type *tw_m,
                                                                                 const typename fftdev::vec2_of<S>::type *tw_n, MelTail mt) {
  using namespace fftdev;
  using V = typename vec2_of<S>::type;
  constexpr S kHalf = sizeof(S) == 8 ? (S)0.5 : (S)1.0;   // float32: the window table is pre-halved
  constexpr int N = 1 << LOG2N, LOG2M = LOG2N - 1, M = N / 2, T = M / 16;
  constexpr bool WAVE = T <= 64;   // wave-private transforms (fft 512 / 1024 / 2048); fft 4096 synchronises its 128 threads
  static_assert(FT == 16 || FT == 8 || FT == 4, "16 frames per workgroup, 8 at fft 4096, 4 at fft 8192");
  constexpr int RL = LastPass<LOG2M>::R, NSL = LastPass<LOG2M>::NS, GL = 16 / RL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  V *work = reinterpret_cast<V *>(smem);                                 // FT buffers of M complex
  const int64_t tiles = (a.count + FT - 1) / FT;
  const int64_t clip = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const Tin *x = reinterpret_cast<const Tin *>(a.x) + clip * a.x_stride;
  const S *window = reinterpret_cast<const S *>(a.window);
  const int tid = threadIdx.x % T, f = threadIdx.x / T;
  V *z = work + (size_t)f * M;
  const int64_t f0 = tile * FT;
  const int nf = (int)((a.count - f0) < FT ? (a.count - f0) : FT);
  const bool have = f < nf;                            // uniform per group of T threads
  const TileBands tb = load_tile_bands<MEL>(mt);
  cpx<S> r[16];
  if (have) {
    const int64_t s0 = (a.p0 + f0 + f) * a.hop - a.left;
    if (s0 >= 0 && s0 + N <= a.n) {
      const Tin *xs = x + s0;
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int i = 2 * (tid + T * m);
        // one 8-byte access for the two samples (element-aligned only) and one for their window values
        const auto xv = *reinterpret_cast<const typename Pair<Tin>::type *>(xs + i);
        const V wv = reinterpret_cast<const V *>(window)[tid + T * m];
        r[m] = {(S)xv.x * wv.x * kHalf, (S)xv.y * wv.y * kHalf};
      }
    } else {
#pragma unroll 1
      for (int m = 0; m < 16; ++m) {
        const int i = 2 * (tid + T * m);
        const S v0 = (S)fetch_sample<Tin>(x, a.n, s0 + i, a.pad, a.pad_value) * window[i] * kHalf;
        const S v1 = (S)fetch_sample<Tin>(x, a.n, s0 + i + 1, a.pad, a.pad_value) * window[i + 1] * kHalf;
#pragma unroll
        for (int mm = 0; mm < 16; ++mm)
          if (mm == m) r[mm] = {v0, v1};
      }
    }
  } else {
#pragma unroll
    for (int m = 0; m < 16; ++m) r[m] = {(S)0, (S)0};
  }
  fft_passes<LOG2M, true, WAVE>(r, z, tid, tw_m);
#pragma unroll
  for (int i = 0; i < GL; ++i)
#pragma unroll
    for (int j = 0; j < RL; ++j) {
      V o;
      o.x = r[i * RL + j].x;
      o.y = r[i * RL + j].y;
      z[swz(out_index<RL, NSL, T>(tid, i, j))] = o;
    }
  stockham_sync<WAVE>();
  Tout val[16], nyq = (Tout)0;
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int k = tid + T * m;
    const V zk = z[swz(k)], zm = z[swz((M - k) & (M - 1))];
    const S er = zk.x + zm.x, ei = zk.y - zm.y;
    const S dr = zk.x - zm.x, di = zk.y + zm.y;
    const V w = tw_n[k];
    val[m] = magnitude_pow<S, Tout>(er + (w.x * di + w.y * dr), ei - (w.x * dr - w.y * di), a.power);
  }
  if (tid == 0) {
    const V z0 = z[0];
    nyq = magnitude_pow<S, Tout>((S)2 * (z0.x - z0.y), (S)0, a.power);
  }
  stockham_sync<WAVE>();   // every read of this frame's Z is done (the frame's threads share a wave, or a barrier): reuse its buffer
  Tout *col = reinterpret_cast<Tout *>(z) + 2 * f;
#pragma unroll
  for (int m = 0; m < 16; ++m) col[tid + T * m] = val[m];
  if (tid == 0) col[M] = nyq;
  __syncthreads();
  columns_out<(int)(M * sizeof(V) / sizeof(Tout)), FT, MEL, Tout>(a, mt, reinterpret_cast<const Tout *>(work), M + 1, nf, clip, f0, tb,
                                                                   MEL ? reinterpret_cast<float *>(smem + (size_t)FT * M * sizeof(V)) : nullptr);
}

template <int LOG2N, int FT = 16>
bool launch_stockham_power16(const StftJob &job, GenericArgs a, const StftTables &t, const MelTail *mel = nullptr) {
  constexpr int M = (1 << LOG2N) / 2, THREADS = FT * (M / 16);   // FT frames x M/16 threads
  if (!t.fast_window || !t.fast_w_m || !t.fast_w_n) return false;
  a.window = t.fast_window;
  const int64_t blocks = a.lead * ((a.count + FT - 1) / FT);
  if (blocks > 2147483647LL) throw Failure("stft: too many frame tiles for one launch");
  const size_t lds = (size_t)FT * M * sizeof(float2);
  if constexpr (FT == 16) {
    if (mel) {
      const size_t lds_mel = lds + (size_t)(THREADS / 64) * 1024;   // the helper waves' partial tiles (columns_out)
      auto kernel = stft_stockham_power16_kernel<LOG2N, float, true, 16>;
      SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mel));
      SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(THREADS), lds_mel, job.stream, a, (const float2 *)t.fast_w_m,
                         (const float2 *)t.fast_w_n, *mel);
      SMX_HIP_CHECK(hipGetLastError());
      return true;
    }
  }
  if (mel) return false;
  auto kernel = stft_stockham_power16_kernel<LOG2N, float, false, FT>;
  SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(THREADS), lds, job.stream, a, (const float2 *)t.fast_w_m,
                     (const float2 *)t.fast_w_n, MelTail{});
  SMX_HIP_CHECK(hipGetLastError());
  return true;
}

// ---- fft 512 / 1024, float32, complex output: the same stage-free scheme for Stft.transform ---------------------
// A frame's spectrum is M + 1 complex values, one more than its work buffer holds -- but X[0] and X[M] are both real,
// so X[M] rides in the imaginary slot of X[0].  There is no spare room to rotate the columns; position k of frame f
// sits at k ^ f instead (f < 16 stays inside k's aligned group of 16), which leaves the flush -- a lane takes two
// frames of one bin, 16 bytes -- with 2-way bank conflicts at worst.
template <int LOG2N, typename Tin, int FT = 16, typename S = float, typename Tout = float>
__global__ void __launch_bounds__(FT * ((1 << LOG2N) / 32)) stft_stockham_complex16_kernel(GenericArgs a, const typename fftdev::vec2_of<S>::type *tw_m,
                                                                                           const typename fftdev::vec2_of<S>::type *tw_n) {
  using namespace fftdev;
  using V = typename vec2_of<S>::type;
  using CO = typename Vec2<Tout>::type;
  constexpr int N = 1 << LOG2N, LOG2M = LOG2N - 1, M = N / 2, T = M / 16;
  constexpr bool WAVE = T <= 64;
  constexpr S kHalf = sizeof(S) == 8 ? (S)0.5 : (S)1.0;   // float32: the window table is pre-halved
  static_assert(FT == 16 || FT == 8 || FT == 4, "frames per workgroup");
  constexpr int RL = LastPass<LOG2M>::R, NSL = LastPass<LOG2M>::NS, GL = 16 / RL;
  constexpr int CSTRIDE = (int)(M * sizeof(V) / sizeof(CO));    // output elements per frame buffer
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  V *work = reinterpret_cast<V *>(smem);
  const int64_t tiles = (a.count + FT - 1) / FT;
  const int64_t clip = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const Tin *x = reinterpret_cast<const Tin *>(a.x) + clip * a.x_stride;
  const S *window = reinterpret_cast<const S *>(a.window);
  const int tid = threadIdx.x % T, f = threadIdx.x / T;
  V *z = work + (size_t)f * M;
  const int64_t f0 = tile * FT;
  const int nf = (int)((a.count - f0) < FT ? (a.count - f0) : FT);
  const bool have = f < nf;
  cpx<S> r[16];
  if (have) {
    const int64_t s0 = (a.p0 + f0 + f) * a.hop - a.left;
    if (s0 >= 0 && s0 + N <= a.n) {
      const Tin *xs = x + s0;
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int i = 2 * (tid + T * m);
        // one 8-byte access for the two samples (element-aligned only) and one for their window values
        const auto xv = *reinterpret_cast<const typename Pair<Tin>::type *>(xs + i);
        const V wv = reinterpret_cast<const V *>(window)[tid + T * m];
        r[m] = {(S)xv.x * wv.x * kHalf, (S)xv.y * wv.y * kHalf};
      }
    } else {
#pragma unroll 1
      for (int m = 0; m < 16; ++m) {
        const int i = 2 * (tid + T * m);
        const S v0 = (S)fetch_sample<Tin>(x, a.n, s0 + i, a.pad, a.pad_value) * window[i] * kHalf;
        const S v1 = (S)fetch_sample<Tin>(x, a.n, s0 + i + 1, a.pad, a.pad_value) * window[i + 1] * kHalf;
#pragma unroll
        for (int mm = 0; mm < 16; ++mm)
          if (mm == m) r[mm] = {v0, v1};
      }
    }
  } else {
#pragma unroll
    for (int m = 0; m < 16; ++m) r[m] = {(S)0, (S)0};
  }
  fft_passes<LOG2M, true, WAVE>(r, z, tid, tw_m);
#pragma unroll
  for (int i = 0; i < GL; ++i)
#pragma unroll
    for (int j = 0; j < RL; ++j) {
      V o;
      o.x = r[i * RL + j].x;
      o.y = r[i * RL + j].y;
      z[swz(out_index<RL, NSL, T>(tid, i, j))] = o;
    }
  stockham_sync<WAVE>();
  CO val[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int k = tid + T * m;
    const V zk = z[swz(k)], zm = z[swz((M - k) & (M - 1))];
    const S er = zk.x + zm.x, ei = zk.y - zm.y;
    const S dr = zk.x - zm.x, di = zk.y + zm.y;
    const V w = tw_n[k];
    val[m].x = (Tout)(er + (w.x * di + w.y * dr));
    val[m].y = (Tout)(ei - (w.x * dr - w.y * di));
    if (k == 0) {                                       // (X[0], X[M]): both real
      val[m].x = (Tout)((S)2 * (zk.x + zk.y));
      val[m].y = (Tout)((S)2 * (zk.x - zk.y));
    }
  }
  stockham_sync<WAVE>();   // every read of this frame's Z is done: reuse its buffer
  CO *col = reinterpret_cast<CO *>(z);
#pragma unroll
  for (int m = 0; m < 16; ++m) col[(tid + T * m) ^ f] = val[m];
  __syncthreads();
  const CO *cols = reinterpret_cast<const CO *>(work);
  CO *out = reinterpret_cast<CO *>(a.out);
  const int64_t obase = clip * a.bins * a.out_stride + a.out_offset + f0;
  auto fetch = [&](int k, int g) {   // X[k] of frame g; bins 0 and M unpack (X[0], X[M])
    const int kk = k == M ? 0 : k;
    CO c = cols[g * CSTRIDE + (kk ^ g)];
    if (k == 0) c.y = (Tout)0;
    if (k == M) {
      c.x = c.y;
      c.y = (Tout)0;
    }
    return c;
  };
  if (nf == FT) {   // a lane takes 16 bytes of one bin's row: two complex64 frames or one complex128
    constexpr int PER = 16 / sizeof(CO), QF = FT / PER;
    for (int e = threadIdx.x; e < (M + 1) * QF; e += blockDim.x) {
      const int k = e / QF, g = PER * (e % QF);
      CO *dst = out + obase + (int64_t)k * a.out_stride + g;
      if constexpr (sizeof(CO) == 8) {
        using f32x4 = __attribute__((ext_vector_type(4))) float;
        const CO c0 = fetch(k, g), c1 = fetch(k, g + 1);
        const f32x4 v = {(float)c0.x, (float)c0.y, (float)c1.x, (float)c1.y};
        asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(dst), "v"(v) : "memory");
      } else {
        *dst = fetch(k, g);
      }
    }
    return;
  }
  for (int e = threadIdx.x; e < (M + 1) * nf; e += blockDim.x) {   // a clip's ragged last tile
    const int k = e / nf, g = e - k * nf;
    out[obase + (int64_t)k * a.out_stride + g] = fetch(k, g);
  }
}

template <int LOG2N, int FT = 16>
bool launch_stockham_complex16(const StftJob &job, GenericArgs a, const StftTables &t) {
  constexpr int M = (1 << LOG2N) / 2, THREADS = FT * (M / 16);
  if (!t.fast_window || !t.fast_w_m || !t.fast_w_n) return false;
  a.window = t.fast_window;
  const int64_t blocks = a.lead * ((a.count + FT - 1) / FT);
  if (blocks > 2147483647LL) throw Failure("stft: too many frame tiles for one launch");
  const size_t lds = (size_t)FT * M * sizeof(float2);
  auto kernel = stft_stockham_complex16_kernel<LOG2N, float, FT>;
  SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(THREADS), lds, job.stream, a, (const float2 *)t.fast_w_m, (const float2 *)t.fast_w_n);
  SMX_HIP_CHECK(hipGetLastError());
  return true;
}

// the float64 interior, complex output: float32 audio -> complex64, float64 audio -> complex128
template <int LOG2N, int FT, typename Tio>
bool launch_stockham_complex16_wide(const StftJob &job, GenericArgs a, const StftTables &t) {
  constexpr int M = (1 << LOG2N) / 2, THREADS = FT * (M / 16);
  static_assert(THREADS <= 512, "16 complex doubles per thread need the 256-register budget");
  if (!t.window_f64 || !t.fast_w_m_f64 || !t.twiddle_f64) return false;
  a.window = t.window_f64;
  const int64_t blocks = a.lead * ((a.count + FT - 1) / FT);
  if (blocks > 2147483647LL) throw Failure("stft: too many frame tiles for one launch");
  const size_t lds = (size_t)FT * M * sizeof(double2);
  auto kernel = stft_stockham_complex16_kernel<LOG2N, Tio, FT, double, Tio>;
  SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(THREADS), lds, job.stream, a, (const double2 *)t.fast_w_m_f64,
                     (const double2 *)t.twiddle_f64);
  SMX_HIP_CHECK(hipGetLastError());
  return true;
}

template <typename Tio>
bool launch_stockham_complex16_wide_any(const StftJob &job, const GenericArgs &a, const StftTables &t, int64_t fft) {
  switch (fft) {
    case 512: return launch_stockham_complex16_wide<9, 16, Tio>(job, a, t);
    case 1024: return launch_stockham_complex16_wide<10, 16, Tio>(job, a, t);
    case 2048: return launch_stockham_complex16_wide<11, 8, Tio>(job, a, t);
    case 4096: return launch_stockham_complex16_wide<12, 4, Tio>(job, a, t);
    default: return false;
  }
}

// the float64 interior, power output: the stage-free kernel on doubles (FT frames of M double2), float32 audio with
// float32 columns or float64 audio with float64 ones
template <int LOG2N, int FT, typename Tio>
bool launch_stockham_power16_wide(const StftJob &job, GenericArgs a, const StftTables &t) {
  constexpr int M = (1 << LOG2N) / 2, THREADS = FT * (M / 16);
  static_assert(THREADS <= 512, "16 complex doubles per thread need the 256-register budget");
  if (!t.window_f64 || !t.fast_w_m_f64 || !t.twiddle_f64) return false;
  a.window = t.window_f64;
  const int64_t blocks = a.lead * ((a.count + FT - 1) / FT);
  if (blocks > 2147483647LL) throw Failure("stft: too many frame tiles for one launch");
  const size_t lds = (size_t)FT * M * sizeof(double2);
  auto kernel = stft_stockham_power16_kernel<LOG2N, Tio, false, FT, double, Tio>;
  SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(THREADS), lds, job.stream, a, (const double2 *)t.fast_w_m_f64,
                     (const double2 *)t.twiddle_f64, MelTail{});
  SMX_HIP_CHECK(hipGetLastError());
  return true;
}

#include "stft_wide_p64.hpp"   // fft 2048, float32 audio: the persistent-workgroup form of the same arithmetic

template <typename Tio>
bool launch_stockham_power16_wide_any(const StftJob &job, const GenericArgs &a, const StftTables &t, int64_t fft) {
  if constexpr (sizeof(Tio) == 4) {
    if (fft == 2048 && wide64::launch(job, a, t)) return true;
  }
  switch (fft) {
    case 512: return launch_stockham_power16_wide<9, 16, Tio>(job, a, t);
    case 1024: return launch_stockham_power16_wide<10, 16, Tio>(job, a, t);
    case 2048: return launch_stockham_power16_wide<11, 8, Tio>(job, a, t);
    case 4096: return launch_stockham_power16_wide<12, 4, Tio>(job, a, t);
    default: return false;
  }
}

// ---- any other size up to 8192, float32 interior: chirp-z (Bluestein) on the same Stockham passes ------------
// X[k] = c_k sum_n (x[n] w[n] c_n) conj(c)_(k-n),  c_n = exp(-i pi n^2 / N): one circular convolution of length
// M = 2^LOG2M >= 2 N - 1, i.e. FFT_M -> multiply by the filter's spectrum -> inverse FFT_M (as conj(FFT(conj .))),
// exactly the structure of the FIR kernel.  O(M log M) per frame instead of the direct DFT's O(N^2).
struct BluArgs {
  const float2 *chirp, *post, *filter, *tw;
};

template <int LOG2M, typename Tin>
__global__ void __launch_bounds__((1 << LOG2M) / 16 < 256 ? 256 : (1 << LOG2M) / 16) stft_bluestein_kernel(GenericArgs a, BluArgs b) {
  using namespace fftdev;
  constexpr int M = 1 << LOG2M, T = M / 16, G = T < 256 ? 256 / T : 1;
  constexpr int RL = LastPass<LOG2M>::R, NSL = LastPass<LOG2M>::NS, GL = 16 / RL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2 *work = reinterpret_cast<float2 *>(smem);
  unsigned char *stage = smem + (size_t)G * M * sizeof(float2);
  const int ft = a.ft, sstride = ft + 1;
  const int N = (int)a.fft;
  const int64_t tiles = (a.count + ft - 1) / ft;
  const int64_t clip = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const Tin *x = reinterpret_cast<const Tin *>(a.x) + clip * a.x_stride;
  const int64_t bins = a.bins;
  const int tid = threadIdx.x % T, grp = threadIdx.x / T;
  float2 *z = work + (size_t)grp * M;
  const int64_t f0 = tile * ft;
  const int nf = (int)((a.count - f0) < ft ? (a.count - f0) : ft);
  for (int fb = 0; fb < nf; fb += G) {
    const int f = fb + grp;
    const bool have = f < nf;
    c32 r[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) r[m] = {0.0f, 0.0f};
    if (have) {
      const int64_t s0 = (a.p0 + f0 + f) * a.hop - a.left;
      const bool inside = s0 >= 0 && s0 + N <= a.n;   // uniform per group: plain loads
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int i = tid + T * m;
        if (i < N) {
          const float v = inside ? (float)x[s0 + i] : (float)fetch_sample<Tin>(x, a.n, s0 + i, a.pad, a.pad_value);
          const float2 c = b.chirp[i];          // window folded in
          r[m] = {v * c.x, v * c.y};
        }
      }
    }
    fft_passes<LOG2M, true, false, float, true>(r, z, tid, b.tw);
    // product with the filter's spectrum (1/M folded in), conjugated for the inverse transform
#pragma unroll
    for (int i = 0; i < GL; ++i)
#pragma unroll
      for (int j = 0; j < RL; ++j) {
        const int idx = out_index<RL, NSL, T>(tid, i, j);
        const float2 h = b.filter[idx];
        const c32 y = cmul(r[i * RL + j], c32{h.x, h.y});
        z[swz(idx)] = make_float2(y.x, -y.y);
      }
    fft_passes<LOG2M, false, false, float, true>(r, z, tid, b.tw);
    if (have) {
#pragma unroll
      for (int i = 0; i < GL; ++i)
#pragma unroll
        for (int j = 0; j < RL; ++j) {
          const int k = out_index<RL, NSL, T>(tid, i, j);
          if (k < bins) {
            const float2 c = b.post[k];
            const c32 v = cmul(c32{r[i * RL + j].x, -r[i * RL + j].y}, c32{c.x, c.y});   // conj of the transform, times c_k
            if (a.direct) {
              const int64_t o = clip * bins * a.out_stride + a.out_offset + f0 + f + (int64_t)k * a.out_stride;
              if (a.mode == OUT_COMPLEX) reinterpret_cast<float2 *>(a.out)[o] = make_float2(v.x, v.y);
              else reinterpret_cast<float *>(a.out)[o] = magnitude_pow<float, float>(v.x, v.y, a.power);
            } else if (a.mode == OUT_COMPLEX) {
              reinterpret_cast<float2 *>(stage)[k * sstride + f] = make_float2(v.x, v.y);
            } else {
              reinterpret_cast<float *>(stage)[k * sstride + f] = magnitude_pow<float, float>(v.x, v.y, a.power);
            }
          }
        }
    }
    __syncthreads();
  }
  if (a.direct) return;
  flush_stage<float>(a, stage, clip, f0, nf);
}

// Even N: z[i] = (x[2i], x[2i+1]) . half window, Z = DFT_L(z) (L = N/2) by chirp-z of length M >= 2 L - 1, then the
// real-input post-pass X[k] = (Z[k] + conj Z[L-k]) - i W_N^k (Z[k] - conj Z[L-k]), X[L] = 2 (Re Z[0] - Im Z[0]):
// a quarter of the convolution work of the kernel above (fft 400: two 512-point transforms instead of two 1024s).
struct Blu2Args {
  const float2 *chirp, *filter, *tw, *tw_n;
};

template <int LOG2M, typename Tin>
__global__ void __launch_bounds__((1 << LOG2M) / 16 < 256 ? 256 : (1 << LOG2M) / 16) stft_bluestein_real_kernel(GenericArgs a, Blu2Args b) {
  using namespace fftdev;
  constexpr int M = 1 << LOG2M, T = M / 16, G = T < 256 ? 256 / T : 1;
  constexpr bool WAVE = T <= 64;
  constexpr int RL = LastPass<LOG2M>::R, NSL = LastPass<LOG2M>::NS, GL = 16 / RL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2 *work = reinterpret_cast<float2 *>(smem);
  unsigned char *stage = smem + (size_t)G * M * sizeof(float2);
  const int ft = a.ft, sstride = ft + 1;
  const int N = (int)a.fft, L = N / 2;
  const int64_t tiles = (a.count + ft - 1) / ft;
  const int64_t clip = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const Tin *x = reinterpret_cast<const Tin *>(a.x) + clip * a.x_stride;
  const float *window = reinterpret_cast<const float *>(a.window);       // 0.5 * analysis window
  const int64_t bins = a.bins;
  const int tid = threadIdx.x % T, grp = threadIdx.x / T;
  float2 *z = work + (size_t)grp * M;
  const int64_t f0 = tile * ft;
  const int nf = (int)((a.count - f0) < ft ? (a.count - f0) : ft);
  auto emit = [&](int f, int k, float re, float im) {
    if (a.direct) {
      const int64_t o = clip * bins * a.out_stride + a.out_offset + f0 + f + (int64_t)k * a.out_stride;
      if (a.mode == OUT_COMPLEX) reinterpret_cast<float2 *>(a.out)[o] = make_float2(re, im);
      else reinterpret_cast<float *>(a.out)[o] = magnitude_pow<float, float>(re, im, a.power);
    } else if (a.mode == OUT_COMPLEX) {
      reinterpret_cast<float2 *>(stage)[k * sstride + f] = make_float2(re, im);
    } else {
      reinterpret_cast<float *>(stage)[k * sstride + f] = magnitude_pow<float, float>(re, im, a.power);
    }
  };
  for (int fb = 0; fb < nf; fb += G) {
    const int f = fb + grp;
    const bool have = f < nf;
    c32 r[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) r[m] = {0.0f, 0.0f};
    if (have) {
      const int64_t s0 = (a.p0 + f0 + f) * a.hop - a.left;
      const bool inside = s0 >= 0 && s0 + N <= a.n;   // uniform per group: plain loads
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int i = tid + T * m;
        if (i < L) {
          const float v0 = inside ? (float)x[s0 + 2 * i] : (float)fetch_sample<Tin>(x, a.n, s0 + 2 * i, a.pad, a.pad_value);
          const float v1 = inside ? (float)x[s0 + 2 * i + 1] : (float)fetch_sample<Tin>(x, a.n, s0 + 2 * i + 1, a.pad, a.pad_value);
          const float2 c = b.chirp[i];
          r[m] = cmul(c32{v0 * window[2 * i], v1 * window[2 * i + 1]}, c32{c.x, c.y});
        }
      }
    }
    fft_passes<LOG2M, true, WAVE, float, true>(r, z, tid, b.tw);
    // product with the filter's spectrum (1/M folded in), conjugated for the inverse transform
#pragma unroll
    for (int i = 0; i < GL; ++i)
#pragma unroll
      for (int j = 0; j < RL; ++j) {
        const int idx = out_index<RL, NSL, T>(tid, i, j);
        const float2 h = b.filter[idx];
        const c32 y = cmul(r[i * RL + j], c32{h.x, h.y});
        z[swz(idx)] = make_float2(y.x, -y.y);
      }
    fft_passes<LOG2M, false, WAVE, float, true>(r, z, tid, b.tw);
    // Z[k] = c_k conj(transform)[k], k < L, into the (now free) buffer in natural order
#pragma unroll
    for (int i = 0; i < GL; ++i)
#pragma unroll
      for (int j = 0; j < RL; ++j) {
        const int k = out_index<RL, NSL, T>(tid, i, j);
        if (k < L) {
          const float2 c = b.chirp[k];
          const c32 v = cmul(c32{r[i * RL + j].x, -r[i * RL + j].y}, c32{c.x, c.y});
          z[swz(k)] = make_float2(v.x, v.y);
        }
      }
    stockham_sync<WAVE>();
    if (have) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int k = tid + T * m;
        if (k < L) {
          const float2 zk = z[swz(k)], zm = z[swz(k == 0 ? 0 : L - k)];
          const float er = zk.x + zm.x, ei = zk.y - zm.y;
          const float dr = zk.x - zm.x, di = zk.y + zm.y;
          const float2 w = b.tw_n[k];                              // exp(-2 pi i k / N)
          emit(f, k, er + (w.x * di + w.y * dr), ei - (w.x * dr - w.y * di));
        }
      }
      if (tid == 0) {
        const float2 z0 = z[0];
        emit(f, L, 2.0f * (z0.x - z0.y), 0.0f);
      }
    }
    __syncthreads();
  }
  if (a.direct) return;
  flush_stage<float>(a, stage, clip, f0, nf);
}

// The same transform as a stage-free 16-frame kernel (power / mel output, M <= 1024, i.e. even sizes up to 1024 --
// whisper's fft 400 among them): every frame's L + 1 values go back into its own work buffer and leave through
// columns_out, with MEL straight through the fp32 MFMA.
template <int LOG2M, typename Tin, bool MEL>
__global__ void __launch_bounds__(1 << LOG2M) stft_bluestein_power16_kernel(GenericArgs a, Blu2Args b, MelTail mt) {
  using namespace fftdev;
  constexpr int M = 1 << LOG2M, T = M / 16, FT = 16;
  static_assert(T <= 64, "wave-private transforms");
  constexpr int RL = LastPass<LOG2M>::R, NSL = LastPass<LOG2M>::NS, GL = 16 / RL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2 *work = reinterpret_cast<float2 *>(smem);
  const int N = (int)a.fft, L = N / 2;
  const int64_t tiles = (a.count + FT - 1) / FT;
  const int64_t clip = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const Tin *x = reinterpret_cast<const Tin *>(a.x) + clip * a.x_stride;
  const float *window = reinterpret_cast<const float *>(a.window);       // 0.5 * analysis window
  const int tid = threadIdx.x % T, f = threadIdx.x / T;
  float2 *z = work + (size_t)f * M;
  const int64_t f0 = tile * FT;
  const int nf = (int)((a.count - f0) < FT ? (a.count - f0) : FT);
  const bool have = f < nf;
  const TileBands tb = load_tile_bands<MEL>(mt);
  c32 r[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) r[m] = {0.0f, 0.0f};
  if (have) {
    const int64_t s0 = (a.p0 + f0 + f) * a.hop - a.left;
    const bool inside = s0 >= 0 && s0 + N <= a.n;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int i = tid + T * m;
      if (i < L) {
        const float v0 = inside ? (float)x[s0 + 2 * i] : (float)fetch_sample<Tin>(x, a.n, s0 + 2 * i, a.pad, a.pad_value);
        const float v1 = inside ? (float)x[s0 + 2 * i + 1] : (float)fetch_sample<Tin>(x, a.n, s0 + 2 * i + 1, a.pad, a.pad_value);
        const float2 c = b.chirp[i];
        r[m] = cmul(c32{v0 * window[2 * i], v1 * window[2 * i + 1]}, c32{c.x, c.y});
      }
    }
  }
  fft_passes<LOG2M, true, true, float, true>(r, z, tid, b.tw);
#pragma unroll
  for (int i = 0; i < GL; ++i)
#pragma unroll
    for (int j = 0; j < RL; ++j) {
      const int idx = out_index<RL, NSL, T>(tid, i, j);
      const float2 h = b.filter[idx];
      const c32 y = cmul(r[i * RL + j], c32{h.x, h.y});
      z[swz(idx)] = make_float2(y.x, -y.y);
    }
  fft_passes<LOG2M, false, true, float, true>(r, z, tid, b.tw);
#pragma unroll
  for (int i = 0; i < GL; ++i)
#pragma unroll
    for (int j = 0; j < RL; ++j) {
      const int k = out_index<RL, NSL, T>(tid, i, j);
      if (k < L) {
        const float2 c = b.chirp[k];
        const c32 v = cmul(c32{r[i * RL + j].x, -r[i * RL + j].y}, c32{c.x, c.y});
        z[swz(k)] = make_float2(v.x, v.y);
      }
    }
  stockham_sync<true>();
  float val[8], nyq = 0.0f;                      // L <= M / 2: a thread owns at most 8 bins below L
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int k = tid + T * m;
    val[m] = 0.0f;
    if (k < L) {
      const float2 zk = z[swz(k)], zm = z[swz(k == 0 ? 0 : L - k)];
      const float er = zk.x + zm.x, ei = zk.y - zm.y;
      const float dr = zk.x - zm.x, di = zk.y + zm.y;
      const float2 w = b.tw_n[k];
      val[m] = magnitude_pow<float, float>(er + (w.x * di + w.y * dr), ei - (w.x * dr - w.y * di), a.power);
    }
  }
  if (tid == 0) {
    const float2 z0 = z[0];
    nyq = magnitude_pow<float, float>(2.0f * (z0.x - z0.y), 0.0f, a.power);
  }
  stockham_sync<true>();
  float *col = reinterpret_cast<float *>(z) + 2 * f;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int k = tid + T * m;
    if (k < L) col[k] = val[m];
  }
  if (tid == 0) col[L] = nyq;
  __syncthreads();
  columns_out<2 * M, FT, MEL>(a, mt, reinterpret_cast<const float *>(work), L + 1, nf, clip, f0, tb,
                              MEL ? reinterpret_cast<float *>(smem + (size_t)FT * M * sizeof(float2)) : nullptr);
}

template <int LOG2M>
bool launch_bluestein_power16(const StftJob &job, GenericArgs a, const StftTables &t, const MelTail *mel) {
  constexpr int M = 1 << LOG2M;
  a.window = t.blu2_window;
  const Blu2Args b{t.blu2_chirp, t.blu2_filter, t.blu2_tw, (const float2 *)t.twiddle_f32};
  const int64_t blocks = a.lead * ((a.count + 15) / 16);
  if (blocks > 2147483647LL) throw Failure("stft: too many frame tiles for one launch");
  const size_t lds = (size_t)16 * M * sizeof(float2);
  if (mel) {
    const size_t lds_mel = lds + (size_t)(M / 64) * 1024;   // the helper waves' partial tiles (columns_out)
    auto kernel = stft_bluestein_power16_kernel<LOG2M, float, true>;
    SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mel));
    SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(M), lds_mel, job.stream, a, b, *mel);
  } else {
    auto kernel = stft_bluestein_power16_kernel<LOG2M, float, false>;
    SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(M), lds, job.stream, a, b, MelTail{});
  }
  SMX_HIP_CHECK(hipGetLastError());
  return true;
}

// ---- even sizes whose half length L = N / 2 is 2^a 3^b 5^c (fft 400, 480, 640, 800, 960, 1000, ...): a direct mixed-radix
// transform instead of the chirp-z convolution (two power-of-two transforms of length >= 2 L - 1 and three pointwise
// products: five times the arithmetic at fft 400).  One wave owns a frame: the windowed (even, odd) sample pairs go to the
// frame's buffer in LDS, Stockham (autosort) passes of radix 4 / 2 / 5 / 3 ping-pong between its two buffers -- butterfly j of a
// pass with sub-transform length Ns reads j + t L / R, multiplies by exp(-2 pi i t (j mod Ns) / (Ns R)) from ONE table of
// exp(-2 pi i j / L), and writes (j - j mod Ns) R + j mod Ns + t Ns -- wave-private, no workgroup barrier; then the real
// post-pass and |X|^p as in the power-of-two kernel, and the columns leave through columns_out (MEL: straight through the
// float32 MFMA).  Same interface and tile shape as stft_bluestein_power16_kernel, which it replaces for these sizes.
template <typename S>
struct MixedPlan {
  int npass;
  unsigned long long radices;   // 4 bits per pass
  const typename fftdev::vec2_of<S>::type *tw_l;   // exp(-2 pi i j / L), j < L
  const typename fftdev::vec2_of<S>::type *tw_n;   // exp(-2 pi i k / N), k <= L
};

// S = double: the float64 interior (float32 audio -> float32 / complex64 output rounded once, float64 audio -> float64 / complex128):
// window, transform and |.|^p in float64, like stft_stockham_power16_kernel<.., double>.
// FULL (odd N): the frame itself is the complex signal (imaginary parts zero), a transform of N points, bins 0 .. N / 2 read off
// directly -- no half-size trick, no post-pass; the window table is the plain analysis window.
template <int LOG2LP, typename Tin, bool MEL, int FT = 16, bool CPLX = false, typename S = float, typename Tout = float, bool FULL = false>   // LP = frame buffer capacity in complex values
__global__ void __launch_bounds__(64 * FT) stft_mixed_power16_kernel(GenericArgs a, MixedPlan<S> pl, MelTail mt) {
  static_assert(!(MEL && CPLX), "the mel tail takes powers");
  static_assert(!MEL || (sizeof(S) == 4 && sizeof(Tout) == 4), "the MFMA tail is float32");
  using namespace fftdev;
  using V = typename vec2_of<S>::type;
  using CO = typename Vec2<Tout>::type;
  constexpr S kHalf = (sizeof(S) == 8 && !FULL) ? (S)0.5 : (S)1.0;   // float32: the window table is pre-halved; FULL: no halving at all
  constexpr int LP = 1 << LOG2LP;
  constexpr int BUF = (int)(2 * LP * sizeof(V) / sizeof(Tout));   // Tout elements per frame region (two buffers of LP complex values)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  V *work = reinterpret_cast<V *>(smem);
  const int N = (int)a.fft, L = FULL ? N : N / 2;        // L: length of the complex transform
  const int nb = FULL ? N / 2 + 1 : L;                   // bins the lanes form below (the half-size form adds bin L from Z[0])
  const int64_t tiles = (a.count + FT - 1) / FT;
  const int64_t clip = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const Tin *x = reinterpret_cast<const Tin *>(a.x) + clip * a.x_stride;
  const S *window = reinterpret_cast<const S *>(a.window);       // float32: 0.5 * analysis window; float64: the analysis window
  const int lane = threadIdx.x & 63, f = threadIdx.x >> 6;
  V *za = work + (size_t)f * (2 * LP), *zb = za + LP;
  const int64_t f0 = tile * FT;
  const int nf = (int)((a.count - f0) < FT ? (a.count - f0) : FT);
  const bool have = f < nf;
  const TileBands tb = load_tile_bands<MEL>(mt);
  // both twiddle tables (L values each) move to LDS once per workgroup: every butterfly of every pass reads R - 1 of them, and from
  // global memory each of those reads was a dependent trip to L1 / L2 in the middle of a pass
  V *tw_l = work + (size_t)FT * (2 * LP), *tw_n = tw_l + LP;
  for (int i = threadIdx.x; i < L; i += 64 * FT) {
    tw_l[i] = pl.tw_l[i];
    if constexpr (!FULL) tw_n[i] = pl.tw_n[i];
  }
  __syncthreads();
  Tout val[LP / 64], vim[CPLX ? LP / 64 : 1], nyq = (Tout)0;
#pragma unroll
  for (int m = 0; m < LP / 64; ++m) {
    val[m] = (Tout)0;
    if constexpr (CPLX) vim[m] = (Tout)0;
  }
  if (have) {   // wave-uniform
    const int64_t s0 = (a.p0 + f0 + f) * a.hop - a.left;
    const bool inside = s0 >= 0 && s0 + N <= a.n;
    if constexpr (FULL) {
      for (int i = lane; i < L; i += 64) {
        const S v0 = inside ? (S)x[s0 + i] : (S)fetch_sample<Tin>(x, a.n, s0 + i, a.pad, a.pad_value);
        V q;
        q.x = v0 * window[i];
        q.y = (S)0;
        za[i] = q;
      }
    } else if (inside) {   // one 8-byte access for the two samples (element-aligned only) and one for their window values
      using P = typename Pair<Tin>::type;
      for (int i = lane; i < L; i += 64) {
        const P xv = *reinterpret_cast<const P *>(x + s0 + 2 * i);
        const V wv = reinterpret_cast<const V *>(window)[i];
        V q;
        q.x = (S)xv.x * wv.x * kHalf;
        q.y = (S)xv.y * wv.y * kHalf;
        za[i] = q;
      }
    } else {
      for (int i = lane; i < L; i += 64) {
        V q;
        q.x = (S)fetch_sample<Tin>(x, a.n, s0 + 2 * i, a.pad, a.pad_value) * window[2 * i] * kHalf;
        q.y = (S)fetch_sample<Tin>(x, a.n, s0 + 2 * i + 1, a.pad, a.pad_value) * window[2 * i + 1] * kHalf;
        za[i] = q;
      }
    }
    stockham_sync<true>();
    const V *z = mixed_transform<S>(za, zb, L, pl.npass, pl.radices, lane, tw_l);   // the transform, natural order
#pragma unroll
    for (int m = 0; m < LP / 64; ++m) {
      const int k = lane + 64 * m;
      if constexpr (FULL) {
        if (k < nb) {
          const V zk = z[k];
          if constexpr (CPLX) {
            val[m] = (Tout)zk.x;
            vim[m] = (Tout)(k == 0 ? (S)0 : zk.y);                 // X[0] is real
          } else {
            val[m] = magnitude_pow<S, Tout>(zk.x, k == 0 ? (S)0 : zk.y, a.power);
          }
        }
      } else if (k < L) {
        const V zk = z[k], zm = z[k == 0 ? 0 : L - k];
        const S er = zk.x + zm.x, ei = zk.y - zm.y;
        const S dr = zk.x - zm.x, di = zk.y + zm.y;
        const V w = tw_n[k];
        const S xr = er + (w.x * di + w.y * dr), xi = ei - (w.x * dr - w.y * di);
        if constexpr (CPLX) {
          val[m] = (Tout)(k == 0 ? (S)2 * (zk.x + zk.y) : xr);     // X[0] is real
          vim[m] = (Tout)(k == 0 ? (S)0 : xi);
        } else {
          val[m] = magnitude_pow<S, Tout>(xr, xi, a.power);
        }
      }
    }
    if (!FULL && lane == 0) {
      const V z0 = z[0];
      nyq = CPLX ? (Tout)((S)2 * (z0.x - z0.y)) : magnitude_pow<S, Tout>((S)2 * (z0.x - z0.y), (S)0, a.power);   // X[L] is real
    }
    stockham_sync<true>();
  } else {   // no frame: a column of zeros (the MFMA tail reads a few values past bin L, times zero weights: keep them finite)
    V zero;
    zero.x = (S)0;
    zero.y = (S)0;
    for (int i = lane; i < LP; i += 64) za[i] = zero;
    stockham_sync<true>();
  }
  if constexpr (CPLX) {   // Stft.transform: complex columns of L + 1 values, rows leave as 16-byte pieces
    CO *ccol = reinterpret_cast<CO *>(za) + 2 * f;
#pragma unroll
    for (int m = 0; m < LP / 64; ++m) {
      const int k = lane + 64 * m;
      if (k < nb) {
        CO c;
        c.x = val[m];
        c.y = vim[m];
        ccol[k] = c;
      }
    }
    if (!FULL && lane == 0) {
      CO c;
      c.x = nyq;
      c.y = (Tout)0;
      ccol[L] = c;
    }
    __syncthreads();
    const CO *cols = reinterpret_cast<const CO *>(work);
    CO *out = reinterpret_cast<CO *>(a.out);
    const int64_t obase = clip * a.bins * a.out_stride + a.out_offset + f0;
    constexpr int CS = (int)(2 * LP * sizeof(V) / sizeof(CO));   // CO elements per frame region
    if (nf == FT) {
      constexpr int PER = 16 / sizeof(CO), QF = FT / PER;   // two complex64 frames or one complex128 per 16-byte piece
      for (int e = threadIdx.x; e < (int)a.bins * QF; e += blockDim.x) {
        const int k = e / QF, g = PER * (e % QF);
        CO *dst = out + obase + (int64_t)k * a.out_stride + g;
        if constexpr (sizeof(CO) == 8) {
          using f32x4 = __attribute__((ext_vector_type(4))) float;
          const CO c0 = cols[g * CS + 2 * g + k], c1 = cols[(g + 1) * CS + 2 * (g + 1) + k];
          const f32x4 v = {(float)c0.x, (float)c0.y, (float)c1.x, (float)c1.y};
          asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(dst), "v"(v) : "memory");
        } else {
          *dst = cols[g * CS + 2 * g + k];
        }
      }
      return;
    }
    for (int e = threadIdx.x; e < (int)a.bins * nf; e += blockDim.x) {   // a clip's ragged last tile
      const int k = e / nf, g = e - k * nf;
      out[obase + (int64_t)k * a.out_stride + g] = cols[g * CS + 2 * g + k];
    }
    return;
  }
  Tout *col = reinterpret_cast<Tout *>(za) + 2 * f;   // column f of the tile: a.bins values in the frame's own region
#pragma unroll
  for (int m = 0; m < LP / 64; ++m) {
    const int k = lane + 64 * m;
    if (k < nb) col[k] = val[m];
  }
  if (!FULL && lane == 0) col[L] = nyq;
  __syncthreads();
  // the helper waves' partial tiles (1 KB each) go into the frames' second buffers, which are free by now: no LDS beyond the frames' own
  columns_out<BUF, FT, MEL, Tout>(a, mt, reinterpret_cast<const Tout *>(work), (int)a.bins, nf, clip, f0, tb,
                                  MEL ? reinterpret_cast<float *>(work + LP) : nullptr, (int)(2 * LP * sizeof(V) / sizeof(float)));
}

template <int LOG2LP, int FT = 16, bool FULL = false>
bool launch_mixed_power16(const StftJob &job, GenericArgs a, const StftTables &t, const MixedPlan<float> &pl, const MelTail *mel) {
  constexpr int LP = 1 << LOG2LP;
  a.window = FULL ? (const void *)t.window_f32 : (const void *)t.blu2_window;
  const int64_t blocks = a.lead * ((a.count + FT - 1) / FT);
  if (blocks > 2147483647LL) throw Failure("stft: too many frame tiles for one launch");
  const size_t lds = (size_t)FT * 4 * LP * sizeof(float) + (size_t)2 * LP * sizeof(float2);   // frames + the two twiddle tables
  auto launch = [&](auto kernel, const MelTail &tail) {
    SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(64 * FT), lds, job.stream, a, pl, tail);
    SMX_HIP_CHECK(hipGetLastError());
    return true;
  };
  if (job.mode == OUT_COMPLEX) return mel ? false : launch(stft_mixed_power16_kernel<LOG2LP, float, false, FT, true, float, float, FULL>, MelTail{});
  if constexpr (FT != 16 || FULL) {   // the MFMA tail is 16 frames wide (and the odd sizes have no fused mel face)
    return mel ? false : launch(stft_mixed_power16_kernel<LOG2LP, float, false, FT, false, float, float, FULL>, MelTail{});
  } else {
    // (the helper waves' partial tiles live in the frames' second buffers: no LDS beyond lds)
    if (mel) return launch(stft_mixed_power16_kernel<LOG2LP, float, true>, *mel);
    return launch(stft_mixed_power16_kernel<LOG2LP, float, false>, MelTail{});
  }
}

// the float64 interior: Tio = float (float32 audio, float32 / complex64 out) or double (float64 audio, float64 / complex128 out)
template <int LOG2LP, int FT, typename Tio, bool FULL = false>
bool launch_mixed_power16_wide(const StftJob &job, GenericArgs a, const StftTables &t, const MixedPlan<double> &pl) {
  constexpr int LP = 1 << LOG2LP;
  a.window = t.window_f64;
  const int64_t blocks = a.lead * ((a.count + FT - 1) / FT);
  if (blocks > 2147483647LL) throw Failure("stft: too many frame tiles for one launch");
  const size_t lds = (size_t)FT * 2 * LP * sizeof(double2) + (size_t)2 * LP * sizeof(double2);   // frames + the two twiddle tables
  auto launch = [&](auto kernel) {
    SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(64 * FT), lds, job.stream, a, pl, MelTail{});
    SMX_HIP_CHECK(hipGetLastError());
  };
  if (job.mode == OUT_COMPLEX) launch(stft_mixed_power16_kernel<LOG2LP, Tio, false, FT, true, double, Tio, FULL>);
  else launch(stft_mixed_power16_kernel<LOG2LP, Tio, false, FT, false, double, Tio, FULL>);
  return true;
}

template <typename Tio>
bool launch_mixed16_wide_any(const StftJob &job, const GenericArgs &a, const StftTables &t) {
  if (t.mixed_npass <= 0 || !t.mixed_tw_f64 || !t.window_f64 || !t.twiddle_f64 || !(t.blu2_window || t.mixed_full)) return false;   // (the forward plan's sizes)
  static const bool off = env_flag("SMX_MIXED_OFF") == 1;
  if (off) return false;
  MixedPlan<double> pl{};
  pl.npass = t.mixed_npass;
  for (int i = 0; i < t.mixed_npass; ++i) pl.radices |= (unsigned long long)t.mixed_radix[i] << (4 * i);
  pl.tw_l = t.mixed_tw_f64;
  pl.tw_n = (const double2 *)t.twiddle_f64;
  if (t.mixed_full) {   // odd N: a transform of N points
    const int64_t n = a.fft;
    if (n <= 128) return launch_mixed_power16_wide<7, 16, Tio, true>(job, a, t, pl);
    if (n <= 256) return launch_mixed_power16_wide<8, 16, Tio, true>(job, a, t, pl);
    if (n <= 512) return launch_mixed_power16_wide<9, 8, Tio, true>(job, a, t, pl);
    if (n <= 1024) return launch_mixed_power16_wide<10, 4, Tio, true>(job, a, t, pl);
    return false;
  }
  const int64_t l = a.fft / 2;
  if (l <= 128) return launch_mixed_power16_wide<7, 16, Tio>(job, a, t, pl);
  if (l <= 256) return launch_mixed_power16_wide<8, 16, Tio>(job, a, t, pl);
  if (l <= 512) return launch_mixed_power16_wide<9, 8, Tio>(job, a, t, pl);
  if (l <= 1024) return launch_mixed_power16_wide<10, 4, Tio>(job, a, t, pl);
  return false;
}

// true when the size has a mixed-radix plan (StftTables::mixed_npass > 0) and the kernel took the launch
bool launch_mixed16_any(const StftJob &job, const GenericArgs &a, const StftTables &t, const MelTail *mel) {
  if (t.mixed_npass <= 0 || !t.mixed_tw || !(t.blu2_window || (t.mixed_full && t.window_f32)) || !t.twiddle_f32) return false;
  static const bool off = env_flag("SMX_MIXED_OFF") == 1;   // tests / A/B timing: chirp-z instead
  if (off) return false;
  MixedPlan<float> pl{};
  pl.npass = t.mixed_npass;
  for (int i = 0; i < t.mixed_npass; ++i) pl.radices |= (unsigned long long)t.mixed_radix[i] << (4 * i);
  pl.tw_l = t.mixed_tw;
  pl.tw_n = (const float2 *)t.twiddle_f32;
  if (t.mixed_full) {   // odd N: a transform of N points
    const int64_t n = a.fft;
    if (n <= 128) return launch_mixed_power16<7, 16, true>(job, a, t, pl, mel);
    if (n <= 256) return launch_mixed_power16<8, 16, true>(job, a, t, pl, mel);
    if (n <= 512) return launch_mixed_power16<9, 16, true>(job, a, t, pl, mel);
    if (n <= 1024) return launch_mixed_power16<10, 8, true>(job, a, t, pl, mel);
    return false;
  }
  const int64_t l = a.fft / 2;
  if (l <= 128) return launch_mixed_power16<7>(job, a, t, pl, mel);
  if (l <= 256) return launch_mixed_power16<8>(job, a, t, pl, mel);
  if (l <= 512) return launch_mixed_power16<9>(job, a, t, pl, mel);
  if (l <= 1024) return launch_mixed_power16<10, 8>(job, a, t, pl, mel);   // 128 KB of LDS: eight frames per workgroup, power only
  return false;
}

// even non-power-of-two sizes up to 1024 (chirp-z length M <= 1024): power / mel through the 16-frame kernel
bool launch_bluestein16_any(const StftJob &job, const GenericArgs &a, const StftTables &t, const MelTail *mel) {
  switch (t.blu2_log2m) {
    case 8: return launch_bluestein_power16<8>(job, a, t, mel);
    case 9: return launch_bluestein_power16<9>(job, a, t, mel);
    case 10: return launch_bluestein_power16<10>(job, a, t, mel);
    default: return false;
  }
}

template <int LOG2M>
bool launch_bluestein_real(const StftJob &job, GenericArgs a, const StftTables &t) {
  constexpr int M = 1 << LOG2M, T = M / 16, G = T < 256 ? 256 / T : 1, THREADS = T < 256 ? 256 : T;
  const size_t elem_out = (job.mode == OUT_COMPLEX ? 2 : 1) * sizeof(float);
  const size_t work = (size_t)G * M * sizeof(float2);
  auto stage_bytes = [&](int ft) { return (size_t)a.bins * (size_t)(ft + 1) * elem_out + 16; };
  int ft = 16;
  while (ft > G && work + stage_bytes(ft) > kLdsLimit) ft >>= 1;
  a.direct = work + stage_bytes(ft) > kLdsLimit ? 1 : 0;
  if (a.direct) ft = G;
  if (work > kLdsLimit) return false;
  a.ft = ft;
  a.window = t.blu2_window;
  const Blu2Args b{t.blu2_chirp, t.blu2_filter, t.blu2_tw, (const float2 *)t.twiddle_f32};
  const int64_t blocks = a.lead * ((a.count + ft - 1) / ft);
  if (blocks > 2147483647LL) throw Failure("stft: too many frame tiles for one launch");
  const size_t lds = a.direct ? work : work + stage_bytes(ft);
  auto kernel = stft_bluestein_real_kernel<LOG2M, float>;
  SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(THREADS), lds, job.stream, a, b);
  SMX_HIP_CHECK(hipGetLastError());
  return true;
}

template <int LOG2M>
bool launch_bluestein(const StftJob &job, GenericArgs a, const BluArgs &b) {
  constexpr int M = 1 << LOG2M, T = M / 16, G = T < 256 ? 256 / T : 1, THREADS = T < 256 ? 256 : T;
  const size_t elem_out = (job.mode == OUT_COMPLEX ? 2 : 1) * sizeof(float);
  const size_t work = (size_t)G * M * sizeof(float2);
  auto stage_bytes = [&](int ft) { return (size_t)a.bins * (size_t)(ft + 1) * elem_out + 16; };
  int ft = 16;
  while (ft > G && work + stage_bytes(ft) > kLdsLimit) ft >>= 1;
  a.direct = work + stage_bytes(ft) > kLdsLimit ? 1 : 0;
  if (a.direct) ft = G;
  if (work > kLdsLimit) return false;
  a.ft = ft;
  const int64_t blocks = a.lead * ((a.count + ft - 1) / ft);
  if (blocks > 2147483647LL) throw Failure("stft: too many frame tiles for one launch");
  const size_t lds = a.direct ? work : work + stage_bytes(ft);
  auto kernel = stft_bluestein_kernel<LOG2M, float>;
  SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(THREADS), lds, job.stream, a, b);
  SMX_HIP_CHECK(hipGetLastError());
  return true;
}

template <typename Tin, typename Tacc, typename Tout>
void launch_typed(const StftJob &job, GenericArgs a) {
  const int64_t N = a.fft;
  const bool pow2 = (N & (N - 1)) == 0;
  const size_t elem_out = (job.mode == OUT_COMPLEX ? 2 : 1) * sizeof(Tout);
  auto stage_bytes = [&](int ft) { return (size_t)a.bins * (size_t)(ft + 1) * elem_out + 16; };
  size_t work_pow2 = (size_t)N * 2 * sizeof(Tacc);
  size_t work_dft = ((size_t)N * sizeof(Tacc) + 15) / 16 * 16;
  bool use_pow2 = pow2 && work_pow2 + stage_bytes(1) <= kLdsLimit;
  size_t work = use_pow2 ? work_pow2 : work_dft;
  if (work + stage_bytes(1) > kLdsLimit)
    throw Failure(format("stft: an FFT of size %lld does not fit the on-chip buffers of this device path",
                         (long long)N));
  int ft = 16;
  while (ft > 1 && work + stage_bytes(ft) > kLdsLimit / 2) ft >>= 1;   // keep 2 workgroups per CU
  while (ft > 1 && work + stage_bytes(ft) > kLdsLimit) ft >>= 1;
  a.ft = ft;
  a.log2n = -1;
  if (use_pow2) {
    a.log2n = 0;
    while ((int64_t(1) << a.log2n) < N) ++a.log2n;
  }
  const size_t lds = work + stage_bytes(ft);
  const int64_t tiles = (a.count + ft - 1) / ft;
  const int64_t blocks = a.lead * tiles;
  if (blocks > 2147483647LL) throw Failure("stft: too many frame tiles for one launch");
  auto kernel = use_pow2 ? stft_generic_kernel<Tin, Tacc, Tout, true> : stft_generic_kernel<Tin, Tacc, Tout, false>;
  SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(256), lds, job.stream, a);
  SMX_HIP_CHECK(hipGetLastError());
}

}  // namespace

// Soundml.mel_spectrogram for fft 512 / 1024 (float32 audio and interior): framing, transform, |X|^p and the mel
// product in one launch, the spectrogram never written.  false = not this geometry.
bool launch_mel_spectrogram_16(const MelSpecJob &job) {
  const StftJob &sj = job.stft;
  const smx_stft_config &c = *sj.cfg;
  if (sj.in_bytes != 4 || sj.interior == SMX_INTERIOR_F64 || sj.mode == OUT_COMPLEX || fast_path_disabled()) return false;
  const StftTables &t = c.tables();
  const bool pow2_16 = c.fft_size == 512 || c.fft_size == 1024;
  const bool chirp_16 = !pow2_16 && t.blu2_log2m >= 8 && t.blu2_log2m <= 10;   // even sizes up to 1024 (fft 400 ...)
  if (!pow2_16 && !chirp_16) return false;
  if (diag_flag("SMX_MEL16_OFF") == 1) return false;
  if (sj.count <= 0 || sj.lead <= 0) return true;
  const smx_mel_config::Tables &mtab = job.mel->tables();
  GenericArgs a{};
  a.x = sj.x;
  a.n = sj.n;
  a.x_stride = sj.x_stride;
  a.lead = sj.lead;
  a.fft = c.fft_size;
  a.hop = c.hop;
  a.left = sj.left;
  a.pad = sj.pad;
  a.pad_value = sj.pad_value;
  a.p0 = sj.p0;
  a.count = sj.count;
  a.mode = (int)sj.mode;
  a.power = sj.power;
  a.bins = c.bins();
  MelTail mt{};
  mt.w = mtab.w_tile;
  mt.band_lo = mtab.tile_lo;
  mt.band_hi = mtab.tile_hi;
  mt.n_mels = (int)job.mel->n_mels;
#ifdef SMX_DIAG   // result-altering timing switch: diagnostic builds only (make DIAG=1)
  if (diag_flag("SMX_MEL16_NOTAIL") == 1) mt.n_mels = 0;   // the kernel without its MFMA tail
#endif
  mt.k_pad = (int)mtab.k_pad;
  mt.out = reinterpret_cast<float *>(job.out);
  if (chirp_16) {
    // Sizes with a mixed-radix plan take the composition power kernel + Mel.apply at EVERY batch size.  The fused launch is
    // ahead only below a few tiles per CU (one launch instead of two; above, fft 400 / hop 160, 80 mels, 256 x 30 s:
    // 1.03 + 0.22 ms against 1.50), and its MFMA tail sums in another order than Mel.apply -- a switch by batch size made a
    // clip's values depend on what it was batched with (the reference's slice law, mel_props.ml:136-155).
    // SMX_MEL16_FUSED=1 selects the fused form, again for every batch.
    static const bool force = diag_flag("SMX_MEL16_FUSED") == 1;
    if (t.mixed_npass > 0 && !force) return false;
    return launch_mixed16_any(sj, a, t, &mt) || launch_bluestein16_any(sj, a, t, &mt);
  }
  return c.fft_size == 512 ? launch_stockham_power16<9>(sj, a, t, &mt) : launch_stockham_power16<10>(sj, a, t, &mt);
}

void launch_stft_generic(const StftJob &job) {
  if (job.count <= 0 || job.lead <= 0) return;
  const smx_stft_config &c = *job.cfg;
  const StftTables &t = c.tables();
  GenericArgs a{};
  a.x = job.x;
  a.n = job.n;
  a.x_stride = job.x_stride;
  a.lead = job.lead;
  a.fft = c.fft_size;
  a.hop = c.hop;
  a.left = job.left;
  a.pad = job.pad;
  a.pad_value = job.pad_value;
  a.p0 = job.p0;
  a.count = job.count;
  a.mode = (int)job.mode;
  a.power = job.power;
  a.out = job.out;
  a.out_stride = job.out_stride;
  a.out_offset = job.out_offset;
  a.bins = c.bins();
  const bool f64_interior = job.in_bytes == 8 || job.interior == SMX_INTERIOR_F64;
  a.window = f64_interior ? (const void *)t.window_f64 : (const void *)t.window_f32;
  a.twiddle = f64_interior ? (const void *)t.twiddle_f64 : (const void *)t.twiddle_f32;
  if (job.in_bytes == 4 && !f64_interior && !fast_path_disabled()) {
    bool done = false;
    // the half-size real form wins up to 2048 (fft 1024: 343 vs 315 Mframes/s, 512: 735 vs 687); from 4096 on both
    // forms sit at one 256-thread workgroup per CU (the stage fills the LDS) and the full-size one measured faster
    const bool full_complex = diag_flag("SMX_STOCKHAM_COMPLEX") == 1;   // diagnostic: force the full-size complex form
    const bool real_form = !full_complex && c.fft_size <= 2048;
    const bool staged = diag_flag("SMX_STOCKHAM_STAGED") == 1;   // diagnostic: the staged kernel for fft 512 / 1024 power too
    const bool power16 = real_form && job.mode != OUT_COMPLEX && !staged;
    const bool complex16 = real_form && job.mode == OUT_COMPLEX && !staged;
    if (complex16 && c.fft_size == 512) done = launch_stockham_complex16<9>(job, a, t);
    if (complex16 && c.fft_size == 1024) done = launch_stockham_complex16<10>(job, a, t);
    if (complex16 && c.fft_size == 2048) done = launch_stockham_complex16<11>(job, a, t);   // where the fused kernels do not apply
    if (job.mode == OUT_COMPLEX && !staged && !full_complex && c.fft_size == 4096) done = launch_stockham_complex16<12, 8>(job, a, t);   // eight frames per workgroup (128 KB)
    if (job.mode == OUT_COMPLEX && !staged && !full_complex && c.fft_size == 8192) done = launch_stockham_complex16<13, 4>(job, a, t);   // four
    if (power16 && c.fft_size == 512) done = launch_stockham_power16<9>(job, a, t);
    if (power16 && c.fft_size == 1024) done = launch_stockham_power16<10>(job, a, t);
    if (power16 && c.fft_size == 2048) done = launch_stockham_power16<11>(job, a, t);      // where the fused kernels do not apply
    if (job.mode != OUT_COMPLEX && !staged && c.fft_size == 4096) done = launch_stockham_power16<12, 8>(job, a, t);
    if (job.mode != OUT_COMPLEX && !staged && c.fft_size == 8192) done = launch_stockham_power16<13, 4>(job, a, t);
    if (!done) switch (c.fft_size) {
      case 256: done = launch_stockham<8>(job, a); break;
      case 512: done = real_form ? launch_stockham_real<9, float, float, float>(job, a, t) : launch_stockham<9>(job, a); break;
      case 1024: done = real_form ? launch_stockham_real<10, float, float, float>(job, a, t) : launch_stockham<10>(job, a); break;
      case 2048: done = real_form ? launch_stockham_real<11, float, float, float>(job, a, t) : launch_stockham<11>(job, a); break;
      case 4096: done = real_form ? launch_stockham_real<12, float, float, float>(job, a, t) : launch_stockham<12>(job, a); break;
      case 8192: done = real_form ? launch_stockham_real<13, float, float, float>(job, a, t) : launch_stockham<13>(job, a); break;
      case 16384: done = real_form ? launch_stockham_real<14, float, float, float>(job, a, t) : launch_stockham<14>(job, a); break;
      default: break;
    }
    const bool blu_full = diag_flag("SMX_BLUESTEIN_FULL") == 1;   // diagnostic: the full-length chirp-z for even sizes too
    if (!done && !blu_full && !staged)   // N / 2 = 2^a 3^b 5^c <= 1024: direct mixed-radix transform (power or complex)
      done = launch_mixed16_any(job, a, t, nullptr);
    if (!done && t.blu2_log2m >= 8 && t.blu2_log2m <= 10 && job.mode != OUT_COMPLEX && !blu_full && !staged)
      done = launch_bluestein16_any(job, a, t, nullptr);
    if (!done && t.blu2_log2m >= 8 && !blu_full) {   // even, not a power of two: half-length chirp-z
      switch (t.blu2_log2m) {
        case 8: done = launch_bluestein_real<8>(job, a, t); break;
        case 9: done = launch_bluestein_real<9>(job, a, t); break;
        case 10: done = launch_bluestein_real<10>(job, a, t); break;
        case 11: done = launch_bluestein_real<11>(job, a, t); break;
        case 12: done = launch_bluestein_real<12>(job, a, t); break;
        case 13: done = launch_bluestein_real<13>(job, a, t); break;
        default: break;
      }
    }
    if (!done && t.blu_log2m >= 8) {   // not a power of two: chirp-z
      const BluArgs b{t.blu_chirp, t.blu_post, t.blu_filter, t.blu_tw};
      switch (t.blu_log2m) {
        case 8: done = launch_bluestein<8>(job, a, b); break;
        case 9: done = launch_bluestein<9>(job, a, b); break;
        case 10: done = launch_bluestein<10>(job, a, b); break;
        case 11: done = launch_bluestein<11>(job, a, b); break;
        case 12: done = launch_bluestein<12>(job, a, b); break;
        case 13: done = launch_bluestein<13>(job, a, b); break;
        case 14: done = launch_bluestein<14>(job, a, b); break;
        default: break;
      }
    }
    if (done) return;
  }
  if (f64_interior && job.mode == OUT_COMPLEX && !fast_path_disabled()) {   // stage-free complex kernel on doubles
    if (diag_flag("SMX_STOCKHAM_STAGED") != 1 && (job.in_bytes == 8 ? launch_stockham_complex16_wide_any<double>(job, a, t, c.fft_size)
                                                        : launch_stockham_complex16_wide_any<float>(job, a, t, c.fft_size)))
      return;
  }
  if (f64_interior && job.mode != OUT_COMPLEX && !fast_path_disabled()) {   // stage-free power kernel on doubles
    if (diag_flag("SMX_STOCKHAM_STAGED") != 1 && (job.in_bytes == 8 ? launch_stockham_power16_wide_any<double>(job, a, t, c.fft_size)
                                                        : launch_stockham_power16_wide_any<float>(job, a, t, c.fft_size)))
      return;
  }
  if (f64_interior && !fast_path_disabled()) {   // even sizes with N / 2 = 2^a 3^b 5^c: the mixed-radix kernel on doubles
    if (job.in_bytes == 8 ? launch_mixed16_wide_any<double>(job, a, t) : launch_mixed16_wide_any<float>(job, a, t)) return;
  }
  if (f64_interior && !fast_path_disabled()) {   // float64 interior on the Stockham passes (fft 512 .. 4096)
    if (job.in_bytes == 8 ? launch_stockham_wide<double, double>(job, a, t, c.fft_size)
                          : launch_stockham_wide<float, float>(job, a, t, c.fft_size))
      return;
  }
  if (job.in_bytes == 8)
    launch_typed<double, double, double>(job, a);
  else if (f64_interior)
    launch_typed<float, double, float>(job, a);
  else
    launch_typed<float, float, float>(job, a);
}

}  // namespace smx

#ifdef SMX_STAMPS
extern "C" int smx_debug_read_stamps64(unsigned long long *out, int count) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(smx::fftdev::g_stamp_sums), sizeof(unsigned long long) * (size_t)count);
}
#endif
