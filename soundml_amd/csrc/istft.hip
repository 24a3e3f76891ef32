// Least-squares synthesis, Stft.invert (stft.ml:693-939):
//   x[m] = (sum_p w[m - p hop] y_p[m - p hop]) / (sum_p w^2[m - p hop]),   y_p = irfft(Z[:, p]).
// Two kernels, any fft_size / hop / alignment, float32 or float64 interior:
//   istft_frames_kernel  one workgroup inverts FT consecutive frames of one clip: the [bins; FT] block of
//                        the spectrum is staged through LDS (the frame axis is the fast one in memory),
//                        each frame's Hermitian extension is transformed (radix-2 passes in LDS for a power
//                        of two, a direct real inverse DFT otherwise) and leaves, windowed, as fft_size
//                        contiguous values of the scratch array y[clip][frame][fft_size];
//   istft_ola_kernel     one thread per output sample gathers the <= ceil(fft/hop) frames that reach it, in
//                        the order the reference's overlap_add adds them (frame index descending,
//                        stft.ml:806-831), divides by the envelope and trims / zero-extends.
// For fft 512 .. 4096 the frames come from the Stockham passes of fft_device.hpp instead
// (istft_stockham_frames_kernel, float32; istft_stockham_frames_wide_kernel, the float64 interior), which for
// hop = N/4 or N/2 also does the overlap-add in the same launch (no scratch array), and for fft 2048 / hop 512 the
// hand-laid istft2048_kernel below does both.  Those kernels can form Griffin-Lim's S * unit(c_k - beta c_(k-1))
// while they stage the spectrum (IstftJob::mag / unit / prev).
// No atomics: the sum is deterministic.  The envelope (partial sums on both borders, one period of the
// folded squared window in between, stft.ml:836-889) is built on the host in float64, cached on the device per
// (config, frame count).
#include <cfloat>
#include <cstdlib>
#include <type_traits>

#include "fft_device.hpp"
#include "smx_internal.hpp"

namespace smx {
namespace {

template <typename T> struct Vec2;
template <> struct Vec2<float> { using type = float2; };
template <> struct Vec2<double> { using type = double2; };

struct IstftArgs {
  const void *z;        // [lead; bins; frames] complex, frames fastest
  int64_t lead, frames, bins, count;   // count: frames actually inverted (stft.ml:915-921)
  int64_t fft, hop;
  const void *window, *twiddle;
  void *y;              // [lead; count; fft]
  int log2n, ft;
  // Griffin-Lim folding (Stockham frames kernel only): optional real factors, and with `unit` the phase update
  // unit(z - beta prev) formed on the way in (see SynArgs below)
  const float *mag;
  const float2 *prev;
  float beta;
  int unit;
};

inline __device__ unsigned bitrev_n(unsigned v, int bits) { return bits == 0 ? 0u : (__brev(v) >> (32 - bits)); }

// LDS: stage[bins][FT + 1] complex<Tacc> | work[N] complex<Tacc> (power of two only)
template <typename Tz, typename Tacc, bool POW2>
__global__ void __launch_bounds__(256) istft_frames_kernel(IstftArgs a) {
  using C = typename Vec2<Tacc>::type;
  using CZ = typename Vec2<Tz>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int64_t N = a.fft, bins = a.bins;
  const int ft = a.ft, sstride = ft + 1;
  C *stage = reinterpret_cast<C *>(smem);
  C *work = stage + bins * sstride;
  const int64_t tiles = (a.count + ft - 1) / ft;
  const int64_t clip = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int64_t f0 = tile * ft;
  const int nf = (int)((a.count - f0) < ft ? (a.count - f0) : ft);
  const int tid = threadIdx.x;
  const CZ *z = reinterpret_cast<const CZ *>(a.z) + clip * bins * a.frames + f0;
  for (int64_t e = tid; e < bins * nf; e += blockDim.x) {
    const int64_t k = e / nf;
    const int f = (int)(e % nf);
    const CZ v = z[k * a.frames + f];
    C c;
    c.x = (Tacc)v.x;
    c.y = (Tacc)v.y;
    stage[k * sstride + f] = c;
  }
  __syncthreads();
  const Tacc *window = reinterpret_cast<const Tacc *>(a.window);
  const C *tw = reinterpret_cast<const C *>(a.twiddle);   // exp(-2 pi i j / N)
  const Tacc inv_n = (Tacc)1 / (Tacc)N;
  Tacc *y = reinterpret_cast<Tacc *>(a.y) + (clip * a.count + f0) * N;
  for (int f = 0; f < nf; ++f) {
    if constexpr (POW2) {
      // x[n] = Re FFT(conj Z_full)[n] / N; the imaginary parts of the DC and Nyquist bins do not take part
      for (int64_t k = tid; k < N; k += blockDim.x) {
        const int64_t kk = k < bins ? k : N - k;
        C c = stage[kk * sstride + f];
        if (k < bins) c.y = -c.y;                 // conj of Z[k]; the mirrored half is conj(conj(Z[N-k]))
        if (k == 0 || 2 * k == N) c.y = (Tacc)0;
        work[bitrev_n((unsigned)k, a.log2n)] = c;
      }
      for (int64_t half = 1; half < N; half <<= 1) {
        __syncthreads();
        const int64_t tstep = (N >> 1) / half;
        for (int64_t b = tid; b < (N >> 1); b += blockDim.x) {
          const int64_t j = b & (half - 1);
          const int64_t i0 = ((b - j) << 1) + j, i1 = i0 + half;
          const C w = tw[j * tstep];
          const C u = work[i0], v = work[i1];
          C t;
          t.x = w.x * v.x - w.y * v.y;
          t.y = w.x * v.y + w.y * v.x;
          work[i0] = C{u.x + t.x, u.y + t.y};
          work[i1] = C{u.x - t.x, u.y - t.y};
        }
      }
      __syncthreads();
      for (int64_t n = tid; n < N; n += blockDim.x) y[(int64_t)f * N + n] = work[n].x * inv_n * window[n];
      __syncthreads();
    } else {
      // direct: x[n] = (Re Z0 + [N even] (-1)^n Re Z_{N/2} + 2 sum_{0<k<N/2} (Re Z_k cos(2 pi k n / N) - Im Z_k sin(..))) / N
      for (int64_t n = tid; n < N; n += blockDim.x) {
        Tacc acc = stage[f].x;
        int64_t idx = 0;
        for (int64_t k = 1; k < bins; ++k) {
          idx += n;
          if (idx >= N) idx -= N;
          const C w = tw[idx];                      // (cos, -sin) of 2 pi k n / N
          const C c = stage[k * sstride + f];
          if (2 * k == N) acc += c.x * w.x;         // Nyquist: real, counted once
          else acc += (Tacc)2 * (c.x * w.x + c.y * w.y);
        }
        y[(int64_t)f * N + n] = acc * inv_n * window[n];
      }
    }
  }
}

// ---- powers of two 512 .. 4096, float32: the frames on the Stockham passes (real form) ------------------------
// FT frames per workgroup, M/16 threads each (M = N/2).  The [M + 1; FT] block of the spectrum is staged through two
// float planes (row pieces of FT x 8 bytes, every load of a thread issued before its first use); a frame's threads
// form the half-size spectrum Z'[k] = E + i conj(w_k) D (the inverse of the analysis post-pass, as in the fused
// kernel below), the planes then become the frames' work buffers, z = conj(FFT_M(conj Z')) on the passes of
// fft_device.hpp, and x[2n] + i x[2n+1] = z[n] times the synthesis window (1/(2M) and the signs folded in) goes
// straight to y[clip][frame][N] in 512-byte runs.  3-4 LDS round trips against log2 N barrier-separated passes.
// FUSED (hop = N / 4, FT = 16): the overlap-add happens here too, like the fft-2048 kernel below: tiles advance by 13
// frames and start 3 early, the windowed frames stay in their LDS buffers, and the workgroup completes the 13 hops
// of output no other frame reaches -- no scratch array, no second launch, the reference's summation order.
struct FusedOla {
  int64_t env_q0 = 0;   // envelope position of padded position 0 (streaming synthesis: IstftJob::env_q0)
  float *out;               // [lead; out_len]
  int64_t out_len, left, span, head, stop;
  const double *env_head, *env_period, *env_tail;
  const double *env_rperiod;   // 1 / env_period: a float64 product per sample instead of a division (same quotients after the rounding to float32)
  int tiles_per_clip;
};

template <int LOG2N, int FT, int RATIO>   // RATIO = N / hop of the fused form (4 or 2), 0 = frames only
__global__ void __launch_bounds__(FT * ((1 << LOG2N) / 32)) istft_stockham_frames_kernel(IstftArgs a, const float2 *w_m, const float2 *w_n,
                                                                                        const float2 *synth_window, FusedOla o) {
  using namespace fftdev;
  constexpr int N = 1 << LOG2N, LOG2M = LOG2N - 1, M = N / 2, T = M / 16, THREADS = FT * T;
  constexpr bool WAVE = T <= 64;
  constexpr int RL = LastPass<LOG2M>::R, NSL = LastPass<LOG2M>::NS, GL = 16 / RL;
  constexpr int STRIDE = FT + 1, LOGFT = FT == 16 ? 4 : 3;
  constexpr int PER = ((M + 1) * FT + THREADS - 1) / THREADS;     // staged elements per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *re = reinterpret_cast<float *>(smem);
  float *im = re + (M + 1) * STRIDE;
  constexpr bool FUSED = RATIO != 0;
  constexpr int LAP = FUSED ? RATIO - 1 : 0, ADV = 16 - LAP;   // frames re-inverted per tile / complete hops per tile
  static_assert(!FUSED || FT == 16, "the fused tile is 16 frames");
  static_assert(RATIO == 0 || RATIO == 4 || RATIO == 2, "hop = N / 4 or N / 2");
  const int64_t tiles = FUSED ? (int64_t)o.tiles_per_clip : (a.count + FT - 1) / FT;
  const int64_t clip = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int64_t f0 = FUSED ? ADV * tile - LAP : tile * FT;    // first frame of the tile (may be negative when fused)
  auto valid = [&](int f) { return f0 + f >= 0 && f0 + f < a.count; };
  const float2 *z = reinterpret_cast<const float2 *>(a.z) + clip * (int64_t)(M + 1) * a.frames + f0;
  auto stage_elements = [&](auto first, auto count) {   // elements [first, first + count) of this thread's PER
    constexpr int I0 = decltype(first)::value, NI = decltype(count)::value;
    float2 v[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int e = threadIdx.x + THREADS * (I0 + j);
      const int row = e >> LOGFT, f = e & (FT - 1);
      v[j] = make_float2(0.f, 0.f);
      if (row <= M && valid(f)) v[j] = z[(int64_t)row * a.frames + f];
    }
    if (a.unit) {   // uniform: v is c_k; form unit(c_k - beta c_(k-1)) as the update kernel does
      if (a.prev) {
        const float2 *pv = a.prev + clip * (int64_t)(M + 1) * a.frames + f0;
        float2 q[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int e = threadIdx.x + THREADS * (I0 + j);
          const int row = e >> LOGFT, f = e & (FT - 1);
          q[j] = make_float2(0.f, 0.f);
          if (row <= M && valid(f)) q[j] = pv[(int64_t)row * a.frames + f];
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          v[j].x -= a.beta * q[j].x;
          v[j].y -= a.beta * q[j].y;
        }
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const float m = (float)hypot((double)v[j].x, (double)v[j].y) + FLT_MIN;
        v[j].x /= m;
        v[j].y /= m;
      }
    }
    if (a.mag) {   // uniform
      const float *mg = a.mag + clip * (int64_t)(M + 1) * a.frames + f0;
      float m[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int e = threadIdx.x + THREADS * (I0 + j);
        const int row = e >> LOGFT, f = e & (FT - 1);
        m[j] = 0.f;
        if (row <= M && valid(f)) m[j] = mg[(int64_t)row * a.frames + f];
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        v[j].x *= m[j];
        v[j].y *= m[j];
      }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int e = threadIdx.x + THREADS * (I0 + j);
      const int row = e >> LOGFT, f = e & (FT - 1);
      if (row <= M) {
        re[row * STRIDE + f] = v[j].x;
        im[row * STRIDE + f] = v[j].y;
      }
    }
  };
  if (a.unit) {   // three load streams: two batches keep the registers in budget
    constexpr int H = (PER + 1) / 2;
    stage_elements(std::integral_constant<int, 0>{}, std::integral_constant<int, H>{});
    stage_elements(std::integral_constant<int, H>{}, std::integral_constant<int, PER - H>{});
  } else {
    stage_elements(std::integral_constant<int, 0>{}, std::integral_constant<int, PER>{});
  }
  __syncthreads();
  const int tid = threadIdx.x % T, f = threadIdx.x / T;
  const bool have = valid(f);                           // uniform per group of T threads
  c32 r[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int k = tid + T * m, km = M - k;
    float zr = re[k * STRIDE + f], zi = im[k * STRIDE + f];
    float pr = re[km * STRIDE + f], pi = im[km * STRIDE + f];
    if (k == 0) { zi = 0.f; pi = 0.f; }                 // the imaginary parts of the DC and Nyquist bins do not take part
    const float er = zr + pr, ei = zi - pi;             // E = Z[k] + conj Z[M-k]
    const float dr = zr - pr, di = zi + pi;             // D = Z[k] - conj Z[M-k]
    const float2 w = w_n[k];                            // exp(-2 pi i k / N)
    const float tr = er - (w.x * di - w.y * dr);        // Z' = E + i conj(w) D
    const float ti = ei + (w.x * dr + w.y * di);
    r[m] = {tr, -ti};                                   // conj(Z')
  }
  __syncthreads();   // every column is in registers: the planes become the frames' work buffers
  float2 *buf = reinterpret_cast<float2 *>(smem) + (size_t)f * M;
  if constexpr (!FUSED) {
    if (WAVE && !have) return;                          // wave-private transforms: nothing left to synchronise with
    fft_passes<LOG2M, true, WAVE>(r, buf, tid, w_m);
    if (!have) return;
    float *y = reinterpret_cast<float *>(a.y) + (clip * a.count + f0 + f) * (int64_t)N;
#pragma unroll
    for (int i = 0; i < GL; ++i)
#pragma unroll
      for (int j = 0; j < RL; ++j) {
        const int n = out_index<RL, NSL, T>(tid, i, j);
        const float2 w = synth_window[n];
        reinterpret_cast<float2 *>(y)[n] = make_float2(r[i * RL + j].x * w.x, r[i * RL + j].y * w.y);
      }
  } else {
    static_assert(!FUSED || WAVE, "fused tiles are for fft 512 .. 2048");
    if (have) {
      fft_passes<LOG2M, true, true>(r, buf, tid, w_m);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < GL; ++i)
#pragma unroll
        for (int j = 0; j < RL; ++j) {
          const int n = out_index<RL, NSL, T>(tid, i, j);
          const float2 w = synth_window[n];
          buf[n] = make_float2(r[i * RL + j].x * w.x, r[i * RL + j].y * w.y);   // samples 2n, 2n + 1 of the frame
        }
    }
    __syncthreads();
    // overlap-add of the complete hops: local position q in [LAP hop, 16 hop), frame index descending (stft.ml:806-831)
    constexpr int HOP = FUSED ? N / RATIO : N, LOGHOP = FUSED ? (RATIO == 4 ? LOG2N - 2 : LOG2N - 1) : LOG2N;
    const float *slots = reinterpret_cast<const float *>(smem);   // frame f: N floats at f * N
    float *out = o.out + clip * o.out_len;
    const int64_t q0 = (int64_t)HOP * f0;
    for (int q = LAP * HOP + (int)threadIdx.x; q < 16 * HOP; q += THREADS) {
      const int64_t Q = q0 + q;                                   // padded position
      const int64_t mo = Q - o.left;
      if (mo >= 0 && mo < o.out_len) {
        float v = 0.f;
        if (Q < o.span) {
          float acc = 0.f;
          const int fh = q >> LOGHOP;
#pragma unroll
          for (int d = 0; d <= LAP; ++d) {
            const int g = fh - d;
            if (valid(g)) acc += slots[g * N + (q - HOP * g)];
          }
          const int64_t E = Q + o.env_q0;
          if (E >= o.head && E < o.stop) {
            v = (float)((double)acc * o.env_rperiod[E & (HOP - 1)]);
          } else {   // the clip's first and last hops
            const double env = E < o.head ? o.env_head[E] : o.env_tail[E - o.stop];
            v = (float)((double)acc / env);
          }
        }
        out[mo] = v;
      }
    }
  }
}

// The float64 interior (complex128 spectra, or complex64 under the float64 interior): the same frames kernel on
// doubles, frames only -- the windowed frames go to the float64 scratch array and istft_ola_kernel adds them up.
template <int LOG2N, int FT, typename Tz>
__global__ void __launch_bounds__(FT * ((1 << LOG2N) / 32)) istft_stockham_frames_wide_kernel(IstftArgs a, const double2 *w_m, const double2 *w_n,
                                                                                             const double2 *synth_window) {
  using namespace fftdev;
  using CZ = typename Vec2<Tz>::type;
  constexpr int N = 1 << LOG2N, LOG2M = LOG2N - 1, M = N / 2, T = M / 16, THREADS = FT * T;
  constexpr bool WAVE = T <= 64;
  constexpr int RL = LastPass<LOG2M>::R, NSL = LastPass<LOG2M>::NS, GL = 16 / RL;
  constexpr int STRIDE = FT + 1, LOGFT = FT == 16 ? 4 : (FT == 8 ? 3 : (FT == 4 ? 2 : 1));
  constexpr int PER = ((M + 1) * FT + THREADS - 1) / THREADS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *re = reinterpret_cast<double *>(smem);
  double *im = re + (M + 1) * STRIDE;
  const int64_t tiles = (a.count + FT - 1) / FT;
  const int64_t clip = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int64_t f0 = tile * FT;
  const int nf = (int)((a.count - f0) < FT ? (a.count - f0) : FT);
  const CZ *z = reinterpret_cast<const CZ *>(a.z) + clip * (int64_t)(M + 1) * a.frames + f0;
  {
    CZ v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = threadIdx.x + THREADS * i;
      const int row = e >> LOGFT, f = e & (FT - 1);
      v[i].x = (Tz)0;
      v[i].y = (Tz)0;
      if (row <= M && f < nf) v[i] = z[(int64_t)row * a.frames + f];
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = threadIdx.x + THREADS * i;
      const int row = e >> LOGFT, f = e & (FT - 1);
      if (row <= M) {
        re[row * STRIDE + f] = (double)v[i].x;
        im[row * STRIDE + f] = (double)v[i].y;
      }
    }
  }
  __syncthreads();
  const int tid = threadIdx.x % T, f = threadIdx.x / T;
  const bool have = f < nf;
  c64 r[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int k = tid + T * m, km = M - k;
    double zr = re[k * STRIDE + f], zi = im[k * STRIDE + f];
    double pr = re[km * STRIDE + f], pi = im[km * STRIDE + f];
    if (k == 0) { zi = 0.0; pi = 0.0; }
    const double er = zr + pr, ei = zi - pi;
    const double dr = zr - pr, di = zi + pi;
    const double2 w = w_n[k];
    const double tr = er - (w.x * di - w.y * dr);
    const double ti = ei + (w.x * dr + w.y * di);
    r[m] = {tr, -ti};
  }
  __syncthreads();
  double2 *buf = reinterpret_cast<double2 *>(smem) + (size_t)f * M;
  if (WAVE && !have) return;
  fft_passes<LOG2M, true, WAVE>(r, buf, tid, w_m);
  if (!have) return;
  double *y = reinterpret_cast<double *>(a.y) + (clip * a.count + f0 + f) * (int64_t)N;
#pragma unroll
  for (int i = 0; i < GL; ++i)
#pragma unroll
    for (int j = 0; j < RL; ++j) {
      const int n = out_index<RL, NSL, T>(tid, i, j);
      const double2 w = synth_window[n];
      reinterpret_cast<double2 *>(y)[n] = make_double2(r[i * RL + j].x * w.x, r[i * RL + j].y * w.y);
    }
}

template <int LOG2N, int FT, typename Tz>
void launch_stockham_frames_wide(const IstftArgs &a, const StftTables &t, hipStream_t stream) {
  constexpr int M = (1 << LOG2N) / 2, THREADS = FT * (M / 16);
  static_assert(THREADS <= 512, "16 complex doubles per thread need the 256-register budget");
  const size_t planes = 2 * (size_t)(M + 1) * (FT + 1) * sizeof(double), work = (size_t)FT * M * sizeof(double2);
  const size_t lds = (planes > work ? planes : work) + 16;
  const int64_t blocks = a.lead * ((a.count + FT - 1) / FT);
  if (blocks > 2147483647LL) throw Failure("invert: too many frame tiles for one launch");
  auto kernel = istft_stockham_frames_wide_kernel<LOG2N, FT, Tz>;
  SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(THREADS), lds, stream, a, (const double2 *)t.fast_w_m_f64, (const double2 *)t.twiddle_f64,
                     (const double2 *)t.fast_synth_window_f64);
  SMX_HIP_CHECK(hipGetLastError());
}

template <typename Tz>
bool launch_stockham_frames_wide_any(const IstftArgs &a, const StftTables &t, hipStream_t stream) {
  const bool fast_off = fast_path_disabled();
  if (fast_off || diag_flag("SMX_ISTFT_RADIX2") == 1 || !t.fast_w_m_f64 || !t.twiddle_f64 || !t.fast_synth_window_f64) return false;
  switch (a.fft) {
    case 512: launch_stockham_frames_wide<9, 16, Tz>(a, t, stream); return true;
    case 1024: launch_stockham_frames_wide<10, 8, Tz>(a, t, stream); return true;
    case 2048: launch_stockham_frames_wide<11, 8, Tz>(a, t, stream); return true;
    case 4096: launch_stockham_frames_wide<12, 2, Tz>(a, t, stream); return true;   // two float64 planes of 4 frames exceed the LDS
    default: return false;
  }
}

template <int LOG2N, int FT>
void launch_stockham_frames(const IstftArgs &a, const StftTables &t, hipStream_t stream) {
  constexpr int M = (1 << LOG2N) / 2, THREADS = FT * (M / 16);
  const size_t planes = 2 * (size_t)(M + 1) * (FT + 1) * sizeof(float), work = (size_t)FT * M * sizeof(float2);
  const size_t lds = (planes > work ? planes : work) + 16;
  const int64_t blocks = a.lead * ((a.count + FT - 1) / FT);
  if (blocks > 2147483647LL) throw Failure("invert: too many frame tiles for one launch");
  auto kernel = istft_stockham_frames_kernel<LOG2N, FT, 0>;
  SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(THREADS), lds, stream, a, (const float2 *)t.fast_w_m, (const float2 *)t.fast_w_n,
                     (const float2 *)t.fast_synth_window, FusedOla{});
  SMX_HIP_CHECK(hipGetLastError());
}

// hop = N / 4 or N / 2: frames and overlap-add in one launch (fft 512 / 1024 / 2048)
template <int LOG2N, int RATIO>
bool launch_stockham_fused(const IstftArgs &a, const StftTables &t, const FusedOla &o, hipStream_t stream) {
  constexpr int M = (1 << LOG2N) / 2, THREADS = 16 * (M / 16);
  const size_t planes = 2 * (size_t)(M + 1) * 17 * sizeof(float), work = (size_t)16 * M * sizeof(float2);
  const size_t lds = (planes > work ? planes : work) + 16;
  const int64_t blocks = a.lead * (int64_t)o.tiles_per_clip;
  if (blocks > 2147483647LL) return false;
  auto kernel = istft_stockham_frames_kernel<LOG2N, 16, RATIO>;
  SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(THREADS), lds, stream, a, (const float2 *)t.fast_w_m, (const float2 *)t.fast_w_n,
                     (const float2 *)t.fast_synth_window, o);
  SMX_HIP_CHECK(hipGetLastError());
  return true;
}

// float32 spectra, float32 interior, fft 512 .. 4096: true when the Stockham frames kernel took the launch
bool launch_stockham_frames_any(const IstftArgs &a, const StftTables &t, hipStream_t stream) {
  if (!t.fast_w_m || !t.fast_w_n || !t.fast_synth_window) return false;
  const bool fast_off = fast_path_disabled();
  if (fast_off) return false;
  switch (a.fft) {
    case 512: launch_stockham_frames<9, 16>(a, t, stream); return true;
    case 1024: launch_stockham_frames<10, 16>(a, t, stream); return true;
    case 2048: launch_stockham_frames<11, 16>(a, t, stream); return true;
    case 4096: launch_stockham_frames<12, 8>(a, t, stream); return true;
    default: return false;
  }
}

struct OlaArgs {
  int64_t env_q0 = 0;   // envelope position of padded position 0 (streaming synthesis: IstftJob::env_q0)
  const void *y;        // [lead; count; fft]
  void *out;            // [lead; out_len]
  int64_t lead, count, fft, hop, left, out_len, span;
  const double *env_head, *env_period, *env_tail;   // stft.ml:863-889
  int64_t head, stop;
};

template <typename Tacc, typename Tout>
__global__ void __launch_bounds__(256) istft_ola_kernel(OlaArgs a) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t clip = blockIdx.y;
  if (m >= a.out_len) return;
  Tout *out = reinterpret_cast<Tout *>(a.out) + clip * a.out_len;
  const int64_t q = a.left + m;                      // padded position
  if (q >= a.span) {                                 // past the frames: zero extension (stft.ml:927-931)
    out[m] = (Tout)0;
    return;
  }
  const Tacc *y = reinterpret_cast<const Tacc *>(a.y) + clip * a.count * a.fft;
  int64_t p_hi = q / a.hop;
  if (p_hi > a.count - 1) p_hi = a.count - 1;
  int64_t p_lo = q - a.fft + 1 <= 0 ? 0 : (q - a.fft + 1 + a.hop - 1) / a.hop;
  Tacc acc = (Tacc)0;
  for (int64_t p = p_hi; p >= p_lo; --p) acc += y[p * a.fft + (q - p * a.hop)];
  const int64_t e = q + a.env_q0;
  const double env = e < a.head ? a.env_head[e] : (e < a.stop ? a.env_period[e % a.hop] : a.env_tail[e - a.stop]);
  out[m] = (Tout)((double)acc / env);
}


// ---- fused synthesis for fft 2048 / hop 512, float32 interior --------------------------------------------
// One workgroup = 16 waves = 16 consecutive frames f_lo .. f_lo + 15, of which it completes the 13 hops of output
// that need no other frame: padded positions [512 (f_lo + 3), 512 (f_lo + 16)).  Tiles therefore advance by 13
// frames and re-invert 3 (19 % more transforms) in exchange for no carry between workgroups, no atomics and the
// reference's summation order (frame index descending) at every position.
//   1. the [1025; 16] block of the spectrum goes through LDS (rows are 128 contiguous bytes in memory);
//   2. wave f reads column f and the mirrored column, forms the half-size spectrum
//      Z'[k] = E + i conj(w_k) D,  E = Z[k] + conj Z[M-k],  D = Z[k] - conj Z[M-k]   (the inverse of the
//      analysis post-pass; 1/2 and 1/M are in the synthesis window), M = 1024;
//   3. z = conj(FFT_M(conj Z')) by the Stockham passes of fft_device.hpp in a wave-private 8 KB buffer;
//   4. x[2n] + i x[2n+1] = z[n], windowed, stays in that buffer as the wave's 2048 samples;
//   5. every thread gathers 6-7 output positions from the <= 4 frames that reach them, divides by the
//      envelope and stores: 26 KB of contiguous output per workgroup.
constexpr int kSynM = 1024, kSynFrames = 16, kSynHops = 13;
constexpr int kSynStride = kSynFrames + 1;                       // plane row stride (floats)
constexpr size_t kSynPlane = (size_t)(kSynM + 1) * kSynStride * sizeof(float);          // 69,700
constexpr size_t kSynRegionA = 2 * kSynPlane > (size_t)kSynFrames * kSynM * 8 ? 2 * kSynPlane : (size_t)kSynFrames * kSynM * 8;
constexpr size_t kSynLds = ((kSynRegionA + 15) / 16) * 16 + (size_t)kSynM * sizeof(float2) + 512 * sizeof(double);   // planes / slots, window, envelope reciprocals

struct SynArgs {
  int64_t env_q0 = 0;   // envelope position of padded position 0 (streaming synthesis: IstftJob::env_q0)
  const float2 *z;       // [lead; 1025; frames]
  float *out;            // [lead; out_len]
  int64_t frames, count, out_len, left, span;
  int tiles_per_clip;
  const float2 *w_m, *w_n, *synth_window;
  const double *env_head, *env_period, *env_tail;
  int64_t head, stop;
  int64_t blocks, per_xcd;   // per_xcd > 0: XCD-contiguous tile order over `blocks` tiles
  const float *mag;          // optional [lead; bins; frames] factors of z (Griffin-Lim: magnitudes x unit phases)
  // Griffin-Lim's phase update folded into the staging (stft.ml:1003-1012): with `unit`, z is the rebuilt spectrum
  // c_k and what is inverted is mag * unit(c_k - beta * c_(k-1)), unit(e) = e / (|e| + min_float); prev may be null
  const float2 *prev;
  float beta;
  int unit;
};

__global__ void __launch_bounds__(1024) istft2048_kernel(SynArgs a) {
  using namespace fftdev;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *re = reinterpret_cast<float *>(smem);
  float *im = re + (kSynM + 1) * kSynStride;
  float2 *swin = reinterpret_cast<float2 *>(smem + ((kSynRegionA + 15) / 16) * 16);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroups are dealt to the 8 XCDs round robin: give each XCD one contiguous run of tiles, so that the two
  // neighbours which share every straddled 128-byte line of a spectrum row meet in the same L2
  int64_t logical = blockIdx.x;
  if (a.per_xcd > 0) {
    logical = (int64_t)(blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
    if (logical >= a.blocks) return;
  }
  const int64_t clip = logical / a.tiles_per_clip;
  const int tile = (int)(logical % a.tiles_per_clip);
  const int64_t f_lo = (int64_t)kSynHops * tile - 3;
  const float2 *z = a.z + clip * (int64_t)(kSynM + 1) * a.frames;
  swin[tid] = a.synth_window[tid];
  // the periodic part of the envelope as reciprocals: the overlap-add multiplies (one float64 product per sample where a division
  // costs ~35 float64 operations; the quotients agree on every sample of the C2 round trip)
  double *renv = reinterpret_cast<double *>(swin + kSynM);
  if (tid < 512) renv[tid] = 1.0 / a.env_period[tid];
  // 1. stage: element e = (row, frame) with the frame fastest: 16 lanes read one 128-byte row piece.  All loads of
  // the thread are issued before the first use (the optional factors in their own batch: a branch inside the load
  // loop would serialise them)
  auto stage_elements = [&](auto first, auto count) {   // elements [first, first + count) of this thread's 17
    constexpr int I0 = decltype(first)::value, NI = decltype(count)::value;
    float2 v[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int e = tid + 1024 * (I0 + j);
      const int row = e >> 4;
      const int64_t p = f_lo + (e & 15);
      v[j] = make_float2(0.f, 0.f);
      if (row <= kSynM && p >= 0 && p < a.count) v[j] = z[(int64_t)row * a.frames + p];
    }
    if (a.unit) {   // uniform: v is c_k; form unit(c_k - beta c_(k-1)) as the update kernel does
      if (a.prev) {
        const float2 *pv = a.prev + clip * (int64_t)(kSynM + 1) * a.frames;
        float2 q[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int e = tid + 1024 * (I0 + j);
          const int row = e >> 4;
          const int64_t p = f_lo + (e & 15);
          q[j] = make_float2(0.f, 0.f);
          if (row <= kSynM && p >= 0 && p < a.count) q[j] = pv[(int64_t)row * a.frames + p];
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          v[j].x -= a.beta * q[j].x;
          v[j].y -= a.beta * q[j].y;
        }
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const float m = (float)hypot((double)v[j].x, (double)v[j].y) + FLT_MIN;
        v[j].x /= m;
        v[j].y /= m;
      }
    }
    if (a.mag) {   // uniform
      const float *mg = a.mag + clip * (int64_t)(kSynM + 1) * a.frames;
      float m[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int e = tid + 1024 * (I0 + j);
        const int row = e >> 4;
        const int64_t p = f_lo + (e & 15);
        m[j] = 0.f;
        if (row <= kSynM && p >= 0 && p < a.count) m[j] = mg[(int64_t)row * a.frames + p];
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        v[j].x *= m[j];
        v[j].y *= m[j];
      }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int e = tid + 1024 * (I0 + j);
      const int row = e >> 4, f = e & 15;
      if (row <= kSynM) {
        re[row * kSynStride + f] = v[j].x;
        im[row * kSynStride + f] = v[j].y;
      }
    }
  };
  if (a.unit) {   // three load streams: two batches keep the registers in budget
    stage_elements(std::integral_constant<int, 0>{}, std::integral_constant<int, 9>{});
    stage_elements(std::integral_constant<int, 9>{}, std::integral_constant<int, 8>{});
  } else {
    stage_elements(std::integral_constant<int, 0>{}, std::integral_constant<int, 17>{});
  }
  __syncthreads();
  // 2. half-size spectrum of frame `wave`, conjugated for the conj(FFT(conj .)) inverse
  const bool have = f_lo + wave >= 0 && f_lo + wave < a.count;   // wave-uniform
  c32 r[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int k = lane + 64 * m, km = kSynM - k;
    float zr = re[k * kSynStride + wave], zi = im[k * kSynStride + wave];
    float pr = re[km * kSynStride + wave], pi = im[km * kSynStride + wave];
    if (k == 0) { zi = 0.f; pi = 0.f; }             // the imaginary parts of the DC and Nyquist bins do not take part
    const float er = zr + pr, ei = zi - pi;           // E = Z[k] + conj Z[M-k]
    const float dr = zr - pr, di = zi + pi;           // D = Z[k] - conj Z[M-k]
    const float2 w = a.w_n[k];                        // exp(-2 pi i k / N); conj(w) = (w.x, -w.y)
    // i conj(w) D = i (w.x - i w.y)(dr + i di) = i (w.x dr + w.y di) - (w.x di - w.y dr)
    const float tr = er - (w.x * di - w.y * dr);
    const float ti = ei + (w.x * dr + w.y * di);
    r[m] = {tr, -ti};                                 // conj(Z')
  }
  __syncthreads();   // every column is in registers: the planes become the waves' private buffers
  // 3. forward transform of conj(Z') in the wave's 8 KB buffer
  float2 *buf = reinterpret_cast<float2 *>(smem) + wave * kSynM;
  if (have) {
    fft_passes<10, true, true>(r, buf, lane, a.w_m);
    // 4. z[n] = conj(r) / M; samples 2n, 2n+1, windowed (the signs and 1/(2M) are in the table)
    constexpr int RL = LastPass<10>::R, NSL = LastPass<10>::NS, GL = 16 / RL;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < GL; ++i)
#pragma unroll
      for (int j = 0; j < RL; ++j) {
        const int n = out_index<RL, NSL, 64>(lane, i, j);
        const float2 w = swin[n];
        buf[n] = make_float2(r[i * RL + j].x * w.x, r[i * RL + j].y * w.y);
      }
  }
  __syncthreads();
  // 5. overlap-add of the 13 complete hops: local position q in [1536, 8192)
  const float *slots = reinterpret_cast<const float *>(smem);     // slot f = 2048 floats at f * 2048
  float *out = a.out + clip * a.out_len;
  const int64_t q0 = 512 * f_lo;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int q = 1536 + tid + 1024 * i;
    if (q < 512 * kSynFrames) {
      const int64_t Q = q0 + q;                                    // padded position
      const int64_t mo = Q - a.left;
      if (mo >= 0 && mo < a.out_len) {
        float v = 0.f;
        if (Q < a.span) {
          float acc = 0.f;
          const int fh = q >> 9;
#pragma unroll
          for (int d = 0; d < 4; ++d) {                            // frame index descending (stft.ml:806-831)
            const int f = fh - d;
            const int64_t p = f_lo + f;
            if (p >= 0 && p < a.count) acc += slots[f * 2048 + (q - 512 * f)];
          }
          const int64_t E = Q + a.env_q0;
          if (E >= a.head && E < a.stop) {
            v = (float)((double)acc * renv[E & 511]);
          } else {   // the clip's first and last hops
            const double env = E < a.head ? a.env_head[E] : a.env_tail[E - a.stop];
            v = (float)((double)acc / env);
          }
        }
        out[mo] = v;
      }
    }
  }
}

#include "istft_pipe32.hpp"   // istft2048_pipe_kernel: the same synthesis on the 32-lane frame pipeline, persistent workgroups (round 5)

// ---- even sizes whose half length L = N / 2 is 2^a 3^b 5^c, float32 (fft 400, 480, 960, 1000, 1200 ...) ------------------------
// The direct real inverse DFT of istft_frames_kernel is O(N^2) per frame (fft 400 / hop 160 on 256 x 30 s: 103 ms).  Here one wave
// owns a frame: it reads its column of the spectrum (the 16 waves of a workgroup read the 16 neighbouring frames of the same
// 128-byte lines), forms the half-size spectrum Z'[k] = E + i conj(w_k) D as the Stockham frames kernel does, runs the mixed-radix
// passes of fft_device.hpp (conj(FFT_L(conj Z')), wave-private) and writes x[2n] + i x[2n+1] = z[n] times the synthesis window
// into y[clip][frame][N]; istft_ola_kernel adds the frames up as before.
template <typename S>
struct MixedInv {
  int npass;
  unsigned long long radices;   // 4 bits per pass
  const typename fftdev::vec2_of<S>::type *tw_l;   // exp(-2 pi i j / L)
  const typename fftdev::vec2_of<S>::type *tw_n;   // exp(-2 pi i k / N)
};
// Tz: the spectrum's scalar type (complex64 / complex128); S: the interior (float, or double = the float64 interior; y is S)
// FULL (odd N): the Hermitian extension of the bins as a complex signal of N points, transformed whole; the real parts are the frame.
template <int LOG2LP, int FT, typename Tz, typename S, bool FULL = false>
__global__ void __launch_bounds__(64 * FT) istft_mixed_frames_kernel(IstftArgs a, MixedInv<S> pl) {
  using namespace fftdev;
  using V = typename vec2_of<S>::type;
  using CZ = typename Vec2<Tz>::type;
  constexpr int LP = 1 << LOG2LP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, f = threadIdx.x >> 6;
  V *za = reinterpret_cast<V *>(smem) + (size_t)f * (2 * LP), *zb = za + LP;
  const int N = (int)a.fft, L = FULL ? N : N / 2;  // L: length of the complex transform
  const int nbins = N / 2 + 1;
  const int64_t tiles = (a.count + FT - 1) / FT;
  const int64_t clip = blockIdx.x / tiles, frame = (blockIdx.x % tiles) * FT + f;
  V *tw_l = reinterpret_cast<V *>(smem) + (size_t)FT * (2 * LP), *tw_n = tw_l + LP;   // both twiddle tables in LDS, once per workgroup
  for (int i = threadIdx.x; i < L; i += 64 * FT) {
    tw_l[i] = pl.tw_l[i];
    if constexpr (!FULL) tw_n[i] = pl.tw_n[i];
  }
  __syncthreads();
  if (frame >= a.count) return;                    // wave-uniform; no workgroup barrier below
  const CZ *zin = reinterpret_cast<const CZ *>(a.z) + clip * (int64_t)nbins * a.frames + frame;
  if constexpr (FULL) {
    // conj(Z_full): bins 0 .. N/2 conjugated, the mirrored half is conj(conj Z[N - k]) = Z[k]; then x = Re FFT_N(conj Z_full) / N
    for (int k = lane; k < nbins; k += 64) {
      const CZ c = zin[(int64_t)k * a.frames];
      V q;
      q.x = (S)c.x;
      q.y = k == 0 ? (S)0 : -(S)c.y;                // the imaginary part of the DC bin does not take part
      za[k] = q;
      if (k > 0) {
        q.y = (S)c.y;
        za[N - k] = q;
      }
    }
    asm volatile("" ::: "memory");
    const V *r = mixed_transform<S>(za, zb, L, pl.npass, pl.radices, lane, tw_l);
    const S *window = reinterpret_cast<const S *>(a.window);
    const S inv_n = (S)1 / (S)N;
    S *y = reinterpret_cast<S *>(a.y) + (clip * a.count + frame) * (int64_t)N;
    for (int n = lane; n < N; n += 64) y[n] = r[n].x * inv_n * window[n];
    return;
  }
  for (int k = lane; k <= L; k += 64) {
    const CZ c = zin[(int64_t)k * a.frames];
    V q;
    q.x = (S)c.x;
    q.y = (S)c.y;
    zb[k] = q;
  }
  asm volatile("" ::: "memory");
  for (int k = lane; k < L; k += 64) {
    V zk = zb[k], zp = zb[L - k];
    if (k == 0) { zk.y = (S)0; zp.y = (S)0; }       // the imaginary parts of the DC and Nyquist bins do not take part
    const S er = zk.x + zp.x, ei = zk.y - zp.y;     // E = Z[k] + conj Z[L-k]
    const S dr = zk.x - zp.x, di = zk.y + zp.y;     // D = Z[k] - conj Z[L-k]
    const V w = tw_n[k];
    V q;
    q.x = er - (w.x * di - w.y * dr);               // Z' = E + i conj(w) D
    q.y = -(ei + (w.x * dr + w.y * di));            // stored conjugated
    za[k] = q;
  }
  asm volatile("" ::: "memory");
  const V *r = mixed_transform<S>(za, zb, L, pl.npass, pl.radices, lane, tw_l);
  const S *window = reinterpret_cast<const S *>(a.window);
  const S inv_n = (S)1 / (S)N;
  V *y = reinterpret_cast<V *>(reinterpret_cast<S *>(a.y) + (clip * a.count + frame) * (int64_t)N);
  for (int n = lane; n < L; n += 64) {
    const V v = r[n];
    V q;
    q.x = v.x * inv_n * window[2 * n];
    q.y = -v.y * inv_n * window[2 * n + 1];
    y[n] = q;
  }
}

template <int LOG2LP, int FT, typename Tz, typename S, bool FULL = false>
void launch_mixed_frames(const IstftArgs &a, const MixedInv<S> &pl, hipStream_t stream) {
  const int64_t blocks = a.lead * ((a.count + FT - 1) / FT);
  if (blocks > 2147483647LL) throw Failure("invert: too many frame tiles for one launch");
  const size_t lds = (size_t)(FT + 1) * 2 * (size_t(1) << LOG2LP) * sizeof(typename fftdev::vec2_of<S>::type);   // frames + the two twiddle tables
  auto kernel = istft_mixed_frames_kernel<LOG2LP, FT, Tz, S, FULL>;
  SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(64 * FT), lds, stream, a, pl);
  SMX_HIP_CHECK(hipGetLastError());
}

// no Griffin-Lim factors: true when the mixed-radix frames kernel took the launch.  <float, float>: complex64 spectra, float32
// interior; <float, double>: complex64 under the float64 interior; <double, double>: complex128.
template <typename Tz, typename S>
bool launch_mixed_frames_any(const IstftArgs &a, const StftTables &t, hipStream_t stream) {
  if (t.mixed_npass <= 0 || a.mag || a.unit || (a.fft % 2 != 0) != (t.mixed_full != 0)) return false;
  static const bool off = env_flag("SMX_MIXED_OFF") == 1;
  if (off) return false;
  MixedInv<S> pl{};
  pl.npass = t.mixed_npass;
  for (int i = 0; i < t.mixed_npass; ++i) pl.radices |= (unsigned long long)t.mixed_radix[i] << (4 * i);
  if constexpr (sizeof(S) == 4) {
    if (!t.mixed_tw || !t.twiddle_f32) return false;
    pl.tw_l = t.mixed_tw;
    pl.tw_n = (const float2 *)t.twiddle_f32;
  } else {
    if (!t.mixed_tw_f64 || !t.twiddle_f64) return false;
    pl.tw_l = t.mixed_tw_f64;
    pl.tw_n = (const double2 *)t.twiddle_f64;
  }
  constexpr int W = sizeof(S) == 8 ? 2 : 1;        // a double frame is twice the LDS: half the frames per workgroup
  if (t.mixed_full) {   // odd N: a transform of N points
    const int64_t n = a.fft;
    if (n <= 128) launch_mixed_frames<7, 16, Tz, S, true>(a, pl, stream);
    else if (n <= 256) launch_mixed_frames<8, 16 / W, Tz, S, true>(a, pl, stream);
    else if (n <= 512) launch_mixed_frames<9, 16 / W, Tz, S, true>(a, pl, stream);
    else if (n <= 1024) launch_mixed_frames<10, 8 / W, Tz, S, true>(a, pl, stream);
    else return false;
    return true;
  }
  const int64_t l = a.fft / 2;
  if (l < 128) launch_mixed_frames<7, 16, Tz, S>(a, pl, stream);
  else if (l < 256) launch_mixed_frames<8, 16 / W, Tz, S>(a, pl, stream);
  else if (l < 512) launch_mixed_frames<9, 16 / W, Tz, S>(a, pl, stream);
  else if (l < 1024) launch_mixed_frames<10, 8 / W, Tz, S>(a, pl, stream);
  else return false;
  return true;
}

constexpr size_t kLdsLimit = 160 * 1024;

template <typename Tz, typename Tacc>
void launch_frames(const IstftJob &job, IstftArgs a, hipStream_t stream) {
  const int64_t N = a.fft;
  const bool pow2 = (N & (N - 1)) == 0;
  auto lds_bytes = [&](int ft, bool p2) {
    return (size_t)a.bins * (size_t)(ft + 1) * 2 * sizeof(Tacc) + (p2 ? (size_t)N * 2 * sizeof(Tacc) : 0) + 16;
  };
  const bool use_pow2 = pow2 && lds_bytes(1, true) <= kLdsLimit;
  if (lds_bytes(1, use_pow2) > kLdsLimit)
    throw Failure(format("invert: an FFT of size %lld does not fit the on-chip buffers of this device path", (long long)N));
  int ft = 16;
  while (ft > 1 && lds_bytes(ft, use_pow2) > kLdsLimit / 2) ft >>= 1;
  while (ft > 1 && lds_bytes(ft, use_pow2) > kLdsLimit) ft >>= 1;
  a.ft = ft;
  a.log2n = 0;
  while ((int64_t(1) << a.log2n) < N) ++a.log2n;
  const int64_t blocks = a.lead * ((a.count + ft - 1) / ft);
  if (blocks > 2147483647LL) throw Failure("invert: too many frame tiles for one launch");
  auto kernel = use_pow2 ? istft_frames_kernel<Tz, Tacc, true> : istft_frames_kernel<Tz, Tacc, false>;
  const size_t lds = lds_bytes(ft, use_pow2);
  SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SMX_LAUNCH(kernel, dim3((unsigned)blocks), dim3(256), lds, stream, a);
  SMX_HIP_CHECK(hipGetLastError());
  (void)job;
}

}  // namespace

// the fused kernel: fft 2048, hop 512, complex64 spectrum, float32 interior
static bool istft_fused_2048(const IstftJob &job) {
  const bool fast_off = fast_path_disabled();
  const bool f64 = job.z_bytes == 16 || job.interior == SMX_INTERIOR_F64;
  return !fast_off && !f64 && job.cfg->fft_size == 2048 && job.cfg->hop == 512 && job.lead <= 0x7fffffff / 4096;
}

// the kernels that stage the spectrum themselves: the fused fft-2048 / hop-512 one and the Stockham frames kernel
// (float32, fft 512 .. 4096)
bool istft_takes_factors(const IstftJob &job) {
  const bool fast_off = fast_path_disabled();
  const bool f64 = job.z_bytes == 16 || job.interior == SMX_INTERIOR_F64;
  const int64_t n = job.cfg->fft_size;
  return istft_fused_2048(job) || (!fast_off && !f64 && (n == 512 || n == 1024 || n == 2048 || n == 4096));
}

// the frame-major form of a synthesis (Griffin-Lim's own spectra) exists on the persistent pipeline only
bool istft_frame_major_ok(const IstftJob &job) {
  return istft_fused_2048(job) && diag_flag("SMX_ISTFT_NEW2048") != 1 && env_flag("SMX_INVERT_PIPELINE") != 0 && job.frames < (int64_t(1) << 23) &&
         job.fm_pitch >= job.cfg->fft_size / 2 + 1 && job.fm_rows >= job.frames && job.fm_rows * job.fm_pitch * 8 < (int64_t(1) << 32);
}

template <typename T>
__global__ void __launch_bounds__(256) synthesis_release_kernel(const T *carry, int64_t carry_len, const T *quot, int64_t nq, int64_t drop,
                                                                int64_t release, int64_t hold, T *out, int64_t out_stride, T *carry_out) {
  const int64_t ch = blockIdx.y, total = carry_len + nq;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const T v = i < carry_len ? carry[ch * hold + i] : quot[ch * nq + (i - carry_len)];
    if (i >= release) carry_out[ch * hold + (i - release)] = v;
    else if (i >= drop) out[ch * out_stride + (i - drop)] = v;
  }
}

void launch_synthesis_release(const void *carry, int64_t carry_len, const void *quot, int64_t nq, int64_t channels, int64_t drop,
                              int64_t release, int64_t hold, void *out, int64_t out_stride, void *carry_out, int elem_bytes,
                              hipStream_t stream) {
  const int64_t total = carry_len + nq;
  if (channels <= 0 || total <= 0) return;
  if (channels > 65535) throw Failure("synthesis: more than 65535 channels in one call");
  dim3 grid((unsigned)std::min<int64_t>((total + 255) / 256, 1024), (unsigned)channels);
  if (elem_bytes == 8)
    SMX_LAUNCH(synthesis_release_kernel<double>, grid, dim3(256), 0, stream, reinterpret_cast<const double *>(carry), carry_len,
               reinterpret_cast<const double *>(quot), nq, drop, release, hold, reinterpret_cast<double *>(out), out_stride,
               reinterpret_cast<double *>(carry_out));
  else
    SMX_LAUNCH(synthesis_release_kernel<float>, grid, dim3(256), 0, stream, reinterpret_cast<const float *>(carry), carry_len,
               reinterpret_cast<const float *>(quot), nq, drop, release, hold, reinterpret_cast<float *>(out), out_stride,
               reinterpret_cast<float *>(carry_out));
  SMX_HIP_CHECK(hipGetLastError());
}

void launch_istft(const IstftJob &job) {
  if (job.fm_pitch > 0 && !istft_frame_major_ok(job)) throw Failure("invert: no kernel takes this synthesis frame-major");
  const smx_stft_config &c = *job.cfg;
  if (job.mag && !istft_takes_factors(job)) throw Failure("istft: factors are only taken by the fused kernel");
  if (job.lead <= 0 || job.out_len <= 0) return;
  const int64_t elem_out = job.z_bytes == 16 ? 8 : 4;
  if (job.count <= 0) {   // nothing reaches the output: zeros (stft.ml:922-924)
    SMX_HIP_CHECK(hipMemsetAsync(job.out, 0, (size_t)job.lead * (size_t)job.out_len * (size_t)elem_out, job.stream));
    return;
  }
  const StftTables &t = c.tables();
  const bool f64 = job.z_bytes == 16 || job.interior == SMX_INTERIOR_F64;
  const int64_t fft = c.fft_size, hop = c.hop, count = job.count;
  const int64_t span = (count - 1) * hop + fft;
  // envelope pieces: host float64 in the reference's summation order, cached on the device per (config, count)
  const EnvelopeTable env = c.envelope(job.env_count > 0 ? job.env_count : count);
  const double *d_env = env.dev;
  const int64_t head_n = env.head_n, stop = job.env_open ? (int64_t(1) << 60) : env.stop;
  const int64_t left = job.left >= 0 ? job.left : c.left_width();
  // fused path: fft 2048, hop 512, complex64 spectrum, float32 interior
  if (istft_fused_2048(job) && diag_flag("SMX_ISTFT_NEW2048") != 1) {
    SynArgs sa{};
    sa.mag = reinterpret_cast<const float *>(job.mag);
    sa.unit = job.unit ? 1 : 0;
    sa.prev = reinterpret_cast<const float2 *>(job.prev);
    sa.beta = (float)job.beta;
    sa.z = reinterpret_cast<const float2 *>(job.z);
    sa.out = reinterpret_cast<float *>(job.out);
    sa.frames = job.frames;
    sa.count = count;
    sa.out_len = job.out_len;
    sa.left = left;
    sa.env_q0 = job.env_q0;
    sa.span = span;
    // tiles cover padded positions [0, 512 * 13 * tiles): everything the output can ask for
    const int64_t need = std::max<int64_t>(span, sa.left + job.out_len);
    const int64_t tiles = (need + 512 * kSynHops - 1) / (512 * kSynHops);
    sa.tiles_per_clip = (int)tiles;
    sa.w_m = t.fast_w_m;
    sa.w_n = t.fast_w_n;
    sa.synth_window = t.fast_synth_window;
    sa.env_head = d_env;
    sa.env_period = d_env + env.head;
    sa.env_tail = d_env + env.head + env.period;
    sa.head = head_n;
    sa.stop = stop;
    // Round 5: the persistent pipeline kernel (istft_pipe32.hpp) takes every synthesis of this geometry, whatever its size
    // (Griffin-Lim's factors included), so that a position has one value however the frames reach the kernel (offline,
    // streaming chunks).
    // SMX_INVERT_PIPELINE=0: the one-tile-per-workgroup kernel of rounds 1-4 (tests, A/B timing).
    if (env_flag("SMX_INVERT_PIPELINE") != 0 && job.frames < (int64_t(1) << 23)) {
      PipeArgs pa{};
      pa.s = sa;
      const int64_t tiles16 = (need + 512 * kIpFT - 1) / (512 * kIpFT);
      pa.s.tiles_per_clip = (int)tiles16;
      pa.total_tiles = job.lead * tiles16;
      const int cu_count = device_cu_count();   // (per device, thread-safe: tables.cpp)
      pa.blocks = (int)std::min<int64_t>(pa.total_tiles, cu_count);
      pa.range_base = pa.total_tiles / pa.blocks;
      pa.range_extra = pa.total_tiles % pa.blocks;
      pa.aligned_out = (reinterpret_cast<uintptr_t>(job.out) % 8 == 0 && job.out_len % 2 == 0 && sa.left % 2 == 0) ? 1 : 0;
      pa.fm_pitch = (int)job.fm_pitch;
      pa.fm_clip = job.fm_pitch * job.fm_rows;
      auto kp = job.fm_pitch > 0 ? ((job.mag || job.unit) ? istft2048_pipe_kernel<true, true> : istft2048_pipe_kernel<false, true>)
                                 : ((job.mag || job.unit) ? istft2048_pipe_kernel<true> : istft2048_pipe_kernel<false>);
      SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kIpLds));
      SMX_LAUNCH(kp, dim3((unsigned)pa.blocks), dim3(512), kIpLds, job.stream, pa);
      SMX_HIP_CHECK(hipGetLastError());
      return;
    }
    const int64_t blocks = job.lead * tiles;
    if (blocks <= 0x7ffffff0) {
      const bool linear = diag_flag("SMX_ISTFT_LINEAR") == 1;
      sa.blocks = blocks;
      sa.per_xcd = linear ? 0 : (blocks + 7) / 8;
      const int64_t launched = sa.per_xcd > 0 ? sa.per_xcd * 8 : blocks;
      SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(istft2048_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSynLds));
      SMX_LAUNCH(istft2048_kernel, dim3((unsigned)launched), dim3(1024), kSynLds, job.stream, sa);
      SMX_HIP_CHECK(hipGetLastError());
      return;
    }
  }
  // float32, fft 512 / 1024 / 2048 advanced by a quarter or a half of the size: frames + overlap-add fused, no scratch array
  if (!f64 && job.z_bytes == 8 && (hop * 4 == fft || hop * 2 == fft) && (fft == 512 || fft == 1024 || fft == 2048) && istft_takes_factors(job) &&
      diag_flag("SMX_ISTFT_UNFUSED") != 1) {
    IstftArgs fa{};
    fa.z = job.z;
    fa.lead = job.lead;
    fa.frames = job.frames;
    fa.bins = c.bins();
    fa.count = count;
    fa.fft = fft;
    fa.hop = hop;
    fa.mag = reinterpret_cast<const float *>(job.mag);
    fa.prev = reinterpret_cast<const float2 *>(job.prev);
    fa.beta = (float)job.beta;
    fa.unit = job.unit ? 1 : 0;
    FusedOla o{};
    o.out = reinterpret_cast<float *>(job.out);
    o.out_len = job.out_len;
    o.left = left;
    o.env_q0 = job.env_q0;
    o.span = span;
    o.head = head_n;
    o.stop = stop;
    o.env_head = d_env;
    o.env_period = d_env + env.head;
    o.env_tail = d_env + env.head + env.period;
    o.env_rperiod = d_env + env.head + env.period + env.tail;
    const int64_t need = std::max<int64_t>(span, o.left + job.out_len);
    const int64_t adv = hop * 4 == fft ? 13 : 15;      // complete hops per 16-frame tile
    o.tiles_per_clip = (int)((need + hop * adv - 1) / (hop * adv));
    const bool quarter = hop * 4 == fft;
    const bool done = fft == 512 ? (quarter ? launch_stockham_fused<9, 4>(fa, t, o, job.stream) : launch_stockham_fused<9, 2>(fa, t, o, job.stream))
                    : fft == 1024 ? (quarter ? launch_stockham_fused<10, 4>(fa, t, o, job.stream) : launch_stockham_fused<10, 2>(fa, t, o, job.stream))
                                  : (quarter ? launch_stockham_fused<11, 4>(fa, t, o, job.stream) : launch_stockham_fused<11, 2>(fa, t, o, job.stream));
    if (done) return;
  }
  // clips in chunks so that the windowed frames y stay within ~1 GiB
  const size_t acc_bytes = f64 ? 8 : 4;
  const size_t per_clip = (size_t)count * (size_t)fft * acc_bytes;
  int64_t chunk = (int64_t)((size_t(1) << 30) / (per_clip ? per_clip : 1));
  if (chunk < 1) chunk = 1;
  if (chunk > job.lead) chunk = job.lead;
  if (chunk > 65535) chunk = 65535;
  void *d_y = nullptr;
  SMX_HIP_CHECK(smx::pool_malloc_async(&d_y, (size_t)chunk * per_clip, job.stream));
  const int64_t z_clip = c.bins() * job.frames * (job.z_bytes / 1);   // bytes per clip of z
  for (int64_t c0 = 0; c0 < job.lead; c0 += chunk) {
    const int64_t nclips = job.lead - c0 < chunk ? job.lead - c0 : chunk;
    IstftArgs fa{};
    fa.z = reinterpret_cast<const unsigned char *>(job.z) + c0 * z_clip;
    fa.lead = nclips;
    fa.frames = job.frames;
    fa.bins = c.bins();
    fa.count = count;
    fa.fft = fft;
    fa.hop = hop;
    fa.window = f64 ? (const void *)t.window_f64 : (const void *)t.window_f32;
    fa.twiddle = f64 ? (const void *)t.twiddle_f64 : (const void *)t.twiddle_f32;
    fa.y = d_y;
    fa.mag = job.mag ? reinterpret_cast<const float *>(job.mag) + c0 * c.bins() * job.frames : nullptr;
    fa.prev = job.prev ? reinterpret_cast<const float2 *>(job.prev) + c0 * c.bins() * job.frames : nullptr;
    fa.beta = (float)job.beta;
    fa.unit = job.unit ? 1 : 0;
    if (job.z_bytes == 16) {
      if (!launch_stockham_frames_wide_any<double>(fa, t, job.stream) && !launch_mixed_frames_any<double, double>(fa, t, job.stream))
        launch_frames<double, double>(job, fa, job.stream);
    } else if (f64) {
      if (!launch_stockham_frames_wide_any<float>(fa, t, job.stream) && !launch_mixed_frames_any<float, double>(fa, t, job.stream))
        launch_frames<float, double>(job, fa, job.stream);
    }
    else if (!launch_stockham_frames_any(fa, t, job.stream) && !launch_mixed_frames_any<float, float>(fa, t, job.stream)) launch_frames<float, float>(job, fa, job.stream);
    OlaArgs oa{};
    oa.y = d_y;
    oa.out = reinterpret_cast<unsigned char *>(job.out) + c0 * job.out_len * elem_out;
    oa.lead = nclips;
    oa.count = count;
    oa.fft = fft;
    oa.hop = hop;
    oa.left = left;
    oa.env_q0 = job.env_q0;
    oa.out_len = job.out_len;
    oa.span = span;
    oa.env_head = d_env;
    oa.env_period = d_env + env.head;
    oa.env_tail = d_env + env.head + env.period;
    oa.head = head_n;
    oa.stop = stop;
    dim3 grid((unsigned)((job.out_len + 255) / 256), (unsigned)nclips);
    if (job.z_bytes == 16) SMX_LAUNCH((istft_ola_kernel<double, double>), grid, dim3(256), 0, job.stream, oa);
    else if (f64) SMX_LAUNCH((istft_ola_kernel<double, float>), grid, dim3(256), 0, job.stream, oa);
    else SMX_LAUNCH((istft_ola_kernel<float, float>), grid, dim3(256), 0, job.stream, oa);
    SMX_HIP_CHECK(hipGetLastError());
  }
  SMX_HIP_CHECK(hipFreeAsync(d_y, job.stream));
}

}  // namespace smx

#ifdef SMX_STAMPS
extern "C" int smx_debug_read_stamps_istft(unsigned long long *out, int count) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(smx::fftdev::g_stamp_sums), sizeof(unsigned long long) * (size_t)count);
}
#endif
