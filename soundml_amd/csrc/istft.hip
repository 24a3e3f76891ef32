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
// No atomics: the sum is deterministic.  The envelope (partial sums on both borders, one period of the
// folded squared window in between, stft.ml:836-889) is built on the host in float64 and uploaded.
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

struct OlaArgs {
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
  const double env = q < a.head ? a.env_head[q] : (q < a.stop ? a.env_period[q % a.hop] : a.env_tail[q - a.stop]);
  out[m] = (Tout)((double)acc / env);
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
  hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), lds, stream, a);
  SMX_HIP_CHECK(hipGetLastError());
  (void)job;
}

}  // namespace

void launch_istft(const IstftJob &job) {
  const smx_stft_config &c = *job.cfg;
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
  // envelope pieces (host, float64, the reference's summation order)
  std::vector<double> head, period, tail;
  int64_t head_n = 0, stop = 0;
  stft_envelope(c, count, head, period, tail, head_n, stop);
  const size_t env_doubles = head.size() + period.size() + tail.size();
  double *d_env = nullptr;
  SMX_HIP_CHECK(hipMallocAsync((void **)&d_env, (env_doubles + 1) * sizeof(double), job.stream));
  std::vector<double> packed;
  packed.reserve(env_doubles + 1);
  packed.insert(packed.end(), head.begin(), head.end());
  packed.insert(packed.end(), period.begin(), period.end());
  packed.insert(packed.end(), tail.begin(), tail.end());
  packed.push_back(1.0);
  SMX_HIP_CHECK(hipMemcpyAsync(d_env, packed.data(), packed.size() * sizeof(double), hipMemcpyHostToDevice, job.stream));
  SMX_HIP_CHECK(hipStreamSynchronize(job.stream));   // `packed` is pageable host memory that dies with this call
  // clips in chunks so that the windowed frames y stay within ~1 GiB
  const size_t acc_bytes = f64 ? 8 : 4;
  const size_t per_clip = (size_t)count * (size_t)fft * acc_bytes;
  int64_t chunk = (int64_t)((size_t(1) << 30) / (per_clip ? per_clip : 1));
  if (chunk < 1) chunk = 1;
  if (chunk > job.lead) chunk = job.lead;
  if (chunk > 65535) chunk = 65535;
  void *d_y = nullptr;
  SMX_HIP_CHECK(hipMallocAsync(&d_y, (size_t)chunk * per_clip, job.stream));
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
    if (job.z_bytes == 16) launch_frames<double, double>(job, fa, job.stream);
    else if (f64) launch_frames<float, double>(job, fa, job.stream);
    else launch_frames<float, float>(job, fa, job.stream);
    OlaArgs oa{};
    oa.y = d_y;
    oa.out = reinterpret_cast<unsigned char *>(job.out) + c0 * job.out_len * elem_out;
    oa.lead = nclips;
    oa.count = count;
    oa.fft = fft;
    oa.hop = hop;
    oa.left = c.left_width();
    oa.out_len = job.out_len;
    oa.span = span;
    oa.env_head = d_env;
    oa.env_period = d_env + head.size();
    oa.env_tail = d_env + head.size() + period.size();
    oa.head = head_n;
    oa.stop = stop;
    dim3 grid((unsigned)((job.out_len + 255) / 256), (unsigned)nclips);
    if (job.z_bytes == 16) hipLaunchKernelGGL((istft_ola_kernel<double, double>), grid, dim3(256), 0, job.stream, oa);
    else if (f64) hipLaunchKernelGGL((istft_ola_kernel<double, float>), grid, dim3(256), 0, job.stream, oa);
    else hipLaunchKernelGGL((istft_ola_kernel<float, float>), grid, dim3(256), 0, job.stream, oa);
    SMX_HIP_CHECK(hipGetLastError());
  }
  SMX_HIP_CHECK(hipFreeAsync(d_y, job.stream));
  SMX_HIP_CHECK(hipFreeAsync(d_env, job.stream));
}

}  // namespace smx
