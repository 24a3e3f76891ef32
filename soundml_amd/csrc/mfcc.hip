// Log-mel / MFCC tail (Soundml.mfcc, soundml.ml:50-95; Convert.power_to_db, convert.ml:30-50):
//   db = 10 log10 (max (mel, amin)),  clamped at (max over the WHOLE tensor) - 80 dB,
//   cepstrum[k] = scale_k * sum_m db[m] * 2 cos (pi k (2 m + 1) / (2 n_mels)),  scale_0 = 1/sqrt(4 n_mels),
//   scale_k = 1/sqrt(2 n_mels), optionally times the sinusoidal lifter, rounded once to the audio dtype.
// The interior is float64 whatever the audio dtype (the reference's contract).  Two launches: a max reduction
// of the mel spectrogram (the logarithm is monotonic, so the maximum of db is db of the maximum), then one
// thread per (clip, frame) walking the mel axis with frames across lanes (coalesced).
#include <tuple>

#include "smx_internal.hpp"

namespace smx {
namespace {

// 16 bytes per lane and access, two accesses in flight per trip (one dword at a time this pass took 200 us of the C2
// tail's 440; the mel array is the library's own 256-byte aligned scratch, any other alignment takes the scalar walk)
template <typename T> struct alignas(16) MelVec { T v[16 / sizeof(T)]; };

template <typename T>
__global__ void __launch_bounds__(256) mel_max_kernel(const T *mel, int64_t total, unsigned long long *result) {
  using Vec = MelVec<T>;
  constexpr int V = 16 / sizeof(T);
  double m = 0.0;
  auto see = [&](T x) {
    const double v = (double)x;
    m = v > m ? v : m;
  };
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nthreads = (int64_t)gridDim.x * 256;
  const int64_t nvec = ((uintptr_t)mel & 15) ? 0 : total / V;
  const Vec *mv = reinterpret_cast<const Vec *>(mel);
  int64_t i = tid;
  for (; i + nthreads < nvec; i += 2 * nthreads) {
    const Vec a = mv[i], b = mv[i + nthreads];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < V; ++e) see(a.v[e]), see(b.v[e]);
  }
  if (i < nvec) {
    const Vec a = mv[i];
#pragma unroll
    for (int e = 0; e < V; ++e) see(a.v[e]);
  }
  for (int64_t j = nvec * V + tid; j < total; j += nthreads) see(mel[j]);
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(m, off);
    m = o > m ? o : m;
  }
  // one atomic per workgroup: thousands of atomics on one address cost more than the pass over the data
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) m = part[w] > m ? part[w] : m;
    // non-negative doubles order like their bit patterns
    atomicMax(result, (unsigned long long)__double_as_longlong(m));
  }
}

struct MfccArgs {
  const void *mel;          // [lead; n_mels; frames]
  void *out;                // [lead; n_mfcc; frames]
  const double *dct;        // [n_mfcc; n_mels] raw type-II rows 2 cos(pi k (2m+1) / (2 n_mels))
  const double *post;       // [n_mfcc; 2]: orthonormal scale, lifter weight (1 when absent)
  const double *dct_t;      // [n_mels; 32]: the rows transposed, zeros beyond n_mfcc (n_mfcc <= 32)
  const unsigned long long *max_bits;
  int64_t lead, frames;
  int n_mels, n_mfcc;
};

constexpr int kChunk = 32;   // cepstral coefficients accumulated per pass over the mel axis (one pass for the usual 13-20: the logarithm is the cost)

template <typename T>
__global__ void __launch_bounds__(256) mfcc_kernel(MfccArgs a) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t clip = blockIdx.y;
  if (t >= a.frames) return;
  const T *mel = reinterpret_cast<const T *>(a.mel) + clip * a.n_mels * a.frames + t;
  T *out = reinterpret_cast<T *>(a.out) + clip * a.n_mfcc * a.frames + t;
  constexpr double amin = 1e-10;                      // convert.ml:46, soundml.ml:80
  const double decade = 10.0 / log(10.0);             // convert.ml:23
  const double top = __longlong_as_double((long long)*a.max_bits);
  const double floor_db = decade * log(top > amin ? top : amin) - 80.0;   // offset is 0 for reference 1
  for (int k0 = 0; k0 < a.n_mfcc; k0 += kChunk) {
    double acc[kChunk];
#pragma unroll
    for (int i = 0; i < kChunk; ++i) acc[i] = 0.0;
    for (int m = 0; m < a.n_mels; ++m) {
      const double v = (double)mel[(int64_t)m * a.frames];
      double db = decade * log(v > amin ? v : amin);
      db = db > floor_db ? db : floor_db;
#pragma unroll
      for (int i = 0; i < kChunk; ++i)
        if (k0 + i < a.n_mfcc) acc[i] += db * a.dct[(int64_t)(k0 + i) * a.n_mels + m];
    }
#pragma unroll
    for (int i = 0; i < kChunk; ++i)
      if (k0 + i < a.n_mfcc) {
        double c = acc[i] * a.post[2 * (k0 + i)];
        c = c * a.post[2 * (k0 + i) + 1];
        out[(int64_t)(k0 + i) * a.frames] = (T)c;
      }
  }
}

// The same for float32 data, where the double-precision logarithm was the cost (mfcc_kernel<float>: 0.25 ms at C3, three quarters
// of it the library log, ~80 instructions per value).  A float64 logarithm good to 1.5e-14 absolute in 12 instructions: v = m 2^e,
// the top 7 mantissa bits pick a centre c (table of 1 / c and ln c, 128 entries, built on the host with the host's log),
// r = m / c - 1 (|r| < 2^-8), ln v = e ln 2 + ln c + (r - r^2/2 + r^3/3 - r^4/4 + r^5/5).  The decibels it feeds are float64 as
// the reference's are (soundml.ml:50-95), rounded ONCE to float32 at the end: the result differs from the library log's by a
// float32 rounding flip in ~1e-7 of the values.  The DCT rows sit in LDS (a value read by all lanes of a wave is one broadcast
// access); four mel rows are requested per trip.
constexpr int kFastMfccMaxCoeffs = 32;
// (kept out of line so that it stays a branch: inlined, the compiler evaluates it beside the table form and selects)
__device__ __noinline__ double library_log(double v) { return log(v); }
__device__ __forceinline__ double table_log(double v, const double2 *tab) {
  const long long b = __double_as_longlong(v);
  const int e = (int)((b >> 52) & 0x7ff) - 1023;
  const double2 t = tab[(int)((b >> 45) & 127)];
  const double m = __longlong_as_double((b & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);
  const double r = fma(m, t.x, -1.0);
  const double p = r * fma(r, fma(r, fma(r, fma(r, 0.2, -0.25), 1.0 / 3.0), -0.5), 1.0);
  return fma((double)e, 0.69314718055994530942, t.y + p);
}
// NC: the coefficient count rounded up to a multiple of 8: the accumulation is NC fused multiply-adds per mel row with no branch
// between them; a row's coefficients come as wave-uniform (scalar) loads from the transposed table [mel][32] (zeros beyond n_mfcc).
// A workgroup is 64 frames x the mel axis in FOUR quarters, one per wave (lane = frame: a wave reads 256 contiguous bytes of a mel
// row): four times the waves, a quarter of the dependent requests per thread; the quarters' sums meet in LDS and are added in
// the fixed order ((q0 + q1) + q2) + q3 before the one rounding to float32.
template <int NC>
__global__ void __launch_bounds__(256) mfcc_fast_kernel(MfccArgs a, const double2 *log_table) {
  __shared__ double2 ltab[128];                 // {1 / c, ln c}
  __shared__ double part[3][NC][64];
  for (int e = threadIdx.x; e < 128; e += 256) ltab[e] = log_table[e];
  __syncthreads();
  const int lane = threadIdx.x & 63, quarter = threadIdx.x >> 6;
  const int64_t t = (int64_t)blockIdx.x * 64 + lane;
  const int64_t clip = blockIdx.y;
  const int64_t tc = t < a.frames ? t : a.frames - 1;               // (the last block's idle lanes read a valid column)
  const float *mel = reinterpret_cast<const float *>(a.mel) + clip * a.n_mels * a.frames + tc;
  constexpr double amin = 1e-10;                      // convert.ml:46, soundml.ml:80
  const double decade = 10.0 / log(10.0);             // convert.ml:23
  const double top = __longlong_as_double((long long)*a.max_bits);
  const double floor_db = decade * log(top > amin ? top : amin) - 80.0;   // offset is 0 for reference 1
  double acc[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) acc[i] = 0.0;
  auto one = [&](int m, float x) {
    const double v = (double)x;
    const double u = v > amin ? v : amin;
    // (inf and NaN take the library's log: the table form reads their bits as a number)
    double lg;
    if (u < 1.0e300) lg = table_log(u, ltab);
    else lg = library_log(u);
    double db = decade * lg;
    db = db > floor_db ? db : floor_db;
    const double *row = a.dct_t + m * 32;     // wave-uniform address: scalar loads
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = fma(db, row[i], acc[i]);
  };
  const int per = (a.n_mels + 3) / 4;
  int m = __builtin_amdgcn_readfirstlane(quarter * per);
  const int m_end = __builtin_amdgcn_readfirstlane(m + per < a.n_mels ? m + per : a.n_mels);
  for (; m + 4 <= m_end; m += 4) {   // four mel rows requested per trip (sixteen: slower -- 16 x NC scalar operands do not fit)
    const float x0 = mel[(int64_t)m * a.frames], x1 = mel[(int64_t)(m + 1) * a.frames];
    const float x2 = mel[(int64_t)(m + 2) * a.frames], x3 = mel[(int64_t)(m + 3) * a.frames];
    one(m, x0);
    one(m + 1, x1);
    one(m + 2, x2);
    one(m + 3, x3);
  }
  for (; m < m_end; ++m) one(m, mel[(int64_t)m * a.frames]);
  if (quarter > 0) {
#pragma unroll
    for (int i = 0; i < NC; ++i) part[quarter - 1][i][lane] = acc[i];
  }
  __syncthreads();
  if (quarter > 0 || t >= a.frames) return;
  float *out = reinterpret_cast<float *>(a.out) + clip * a.n_mfcc * a.frames + t;
#pragma unroll
  for (int i = 0; i < NC; ++i)
    if (i < a.n_mfcc) {
      double c = ((acc[i] + part[0][i][lane]) + part[1][i][lane]) + part[2][i][lane];
      c = c * a.post[2 * i];
      c = c * a.post[2 * i + 1];
      out[(int64_t)i * a.frames] = (float)c;
    }
}

// Convert.power_to_db / amplitude_to_db (convert.ml:30-62) in the data's own dtype: |s| first for amplitudes, floor at
// amin, scale * ln, minus the reference offset; with top_db a second pass clamps under the maximum of the WHOLE tensor.
// The maximum is kept as an order-preserving integer key of the value's bits (decibels are negative as often as not).
template <typename T> struct OrderedKey;
template <> struct OrderedKey<float> {
  using U = unsigned;
  static __device__ U of(float v) { const U b = __float_as_uint(v); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
  static __device__ float back(U k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
};
template <> struct OrderedKey<double> {
  using U = unsigned long long;
  static __device__ U of(double v) {
    const U b = (U)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
  }
  static __device__ double back(U k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
  }
};

template <typename T>
__device__ __forceinline__ T db_of(T v, T amin, T scale, T offset, int magnitude) {
  if (magnitude) v = v < (T)0 ? -v : v;
  const T floored = v > amin ? v : amin;          // maximum s amin: a NaN stays a NaN, as Nx.maximum leaves it
  return (v != v) ? v : (T)(log(floored) * scale - offset);
}

// 16 bytes per lane and access; two accesses in flight per trip
template <typename T> struct alignas(16) DbVec { T v[16 / sizeof(T)]; };

// Pass 1 (only with top_db): the maximum of the WHOLE tensor.  Decibels are a non-decreasing function of the (absolute)
// value, so the maximum decibel is the decibel of the maximum value: this pass only reads and compares keys.
template <typename T>
__global__ void __launch_bounds__(256) db_max_kernel(const T *s, int64_t total, int magnitude, typename OrderedKey<T>::U *max_key) {
  using K = OrderedKey<T>;
  using Vec = DbVec<T>;
  constexpr int V = 16 / sizeof(T);
  typename K::U best = 0;
  auto see = [&](T v) {
    if (magnitude) v = v < (T)0 ? -v : v;
    const typename K::U key = K::of(v);
    best = key > best ? key : best;
  };
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nthreads = (int64_t)gridDim.x * 256;
  const int64_t head = std::min<int64_t>(total, (int64_t)((16 - ((uintptr_t)s & 15)) & 15) / (int64_t)sizeof(T));
  const int64_t nvec = (total - head) / V;
  const Vec *sv = reinterpret_cast<const Vec *>(s + head);
  int64_t i = tid;
  for (; i + nthreads < nvec; i += 2 * nthreads) {
    const Vec a = sv[i], b = sv[i + nthreads];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < V; ++e) see(a.v[e]), see(b.v[e]);
  }
  if (i < nvec) {
    const Vec a = sv[i];
#pragma unroll
    for (int e = 0; e < V; ++e) see(a.v[e]);
  }
  if (tid < head) see(s[tid]);
  if (tid < total - head - nvec * V) see(s[head + nvec * V + tid]);
  for (int off = 32; off > 0; off >>= 1) {
    const typename K::U o = __shfl_down(best, off);
    best = o > best ? o : best;
  }
  __shared__ typename K::U part[4];   // one atomic per workgroup (see mel_max_kernel)
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) best = part[w] > best ? part[w] : best;
    atomicMax(max_key, best);
  }
}

// Pass 2: decibels, clamped under (maximum - range) when max_key is given.  Every element is read and written by
// the same lane, so out may be s.
template <typename T>
__global__ void __launch_bounds__(256) to_db_kernel(const T *s, T *out, int64_t total, T amin, T scale, T offset, int magnitude,
                                                    const typename OrderedKey<T>::U *max_key, T range) {
  using Vec = DbVec<T>;
  constexpr int V = 16 / sizeof(T);
  const bool clamp = max_key != nullptr;
  // the largest decibel is the decibel of the largest value, by the same arithmetic as everywhere else
  const T top = clamp ? db_of<T>(OrderedKey<T>::back(*max_key), amin, scale, offset, 0) : (T)0;
  const T lowest = (T)(top - range);
  auto one = [&](T v) {
    const T db = db_of<T>(v, amin, scale, offset, magnitude);
    return !clamp || db > lowest ? db : (db != db ? db : lowest);
  };
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nthreads = (int64_t)gridDim.x * 256;
  const bool aligned = (((uintptr_t)s ^ (uintptr_t)out) & 15) == 0;
  const int64_t head = !aligned ? total : std::min<int64_t>(total, (int64_t)((16 - ((uintptr_t)s & 15)) & 15) / (int64_t)sizeof(T));
  const int64_t nvec = (total - head) / V;
  const Vec *sv = reinterpret_cast<const Vec *>(s + head);
  Vec *ov = reinterpret_cast<Vec *>(out + head);
  int64_t i = tid;
  for (; i + nthreads < nvec; i += 2 * nthreads) {
    Vec a = sv[i], b = sv[i + nthreads];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < V; ++e) a.v[e] = one(a.v[e]), b.v[e] = one(b.v[e]);
    ov[i] = a;
    ov[i + nthreads] = b;
  }
  if (i < nvec) {
    Vec a = sv[i];
#pragma unroll
    for (int e = 0; e < V; ++e) a.v[e] = one(a.v[e]);
    ov[i] = a;
  }
  for (int64_t j = tid; j < head; j += nthreads) out[j] = one(s[j]);
  for (int64_t j = head + nvec * V + tid; j < total; j += nthreads) out[j] = one(s[j]);
}

template <typename T>
void run_to_db(const ToDbJob &job) {
  using U = typename OrderedKey<T>::U;
  constexpr int64_t per_block = 256 * 2 * (16 / (int64_t)sizeof(T));
  const unsigned blocks = (unsigned)std::min<int64_t>((job.total + per_block - 1) / per_block, 2048);
  const double scale = job.gain / 10.0 * (10.0 / std::log(10.0));
  const double offset = scale * std::log(std::max(job.amin, job.reference));
  U *d_key = nullptr;
  if (job.has_top_db) {
    SMX_HIP_CHECK(smx::pool_malloc_async((void **)&d_key, sizeof(U), job.stream));
    SMX_HIP_CHECK(hipMemsetAsync(d_key, 0, sizeof(U), job.stream));
    SMX_LAUNCH(db_max_kernel<T>, dim3(blocks), dim3(256), 0, job.stream, (const T *)job.s, job.total, job.magnitude ? 1 : 0, d_key);
    SMX_HIP_CHECK(hipGetLastError());
  }
  SMX_LAUNCH(to_db_kernel<T>, dim3(blocks), dim3(256), 0, job.stream, (const T *)job.s, (T *)job.out, job.total, (T)job.amin,
                     (T)scale, (T)offset, job.magnitude ? 1 : 0, (const U *)d_key, (T)job.top_db);
  SMX_HIP_CHECK(hipGetLastError());
  if (d_key) SMX_HIP_CHECK(hipFreeAsync(d_key, job.stream));
}

}  // namespace

void launch_to_db(const ToDbJob &job) {
  if (job.total <= 0) return;
  if (job.elem_bytes == 8) run_to_db<double>(job);
  else run_to_db<float>(job);
}

namespace {
struct MfccKey {
  int device, n_mels, n_mfcc;
  double lifter;
  bool operator<(const MfccKey &o) const {
    return std::tie(device, n_mels, n_mfcc, lifter) < std::tie(o.device, o.n_mels, o.n_mfcc, o.lifter);
  }
};
std::mutex g_mfcc_mutex;
std::map<MfccKey, double *> g_mfcc_tables;

// [n_mfcc; n_mels] raw type-II rows, then [n_mfcc; 2] (orthonormal scale, lifter weight), device resident
const double *mfcc_tables(int n_mels, int n_mfcc, double lifter) {
  MfccKey key{0, n_mels, n_mfcc, lifter};
  SMX_HIP_CHECK(hipGetDevice(&key.device));
  std::lock_guard<std::mutex> lock(g_mfcc_mutex);
  auto it = g_mfcc_tables.find(key);
  if (it != g_mfcc_tables.end()) return it->second;
  std::vector<double> host((size_t)n_mfcc * n_mels + 2 * (size_t)n_mfcc + (n_mfcc <= 32 ? (size_t)n_mels * 32 : 0));
  const double pi = 3.14159265358979323846;
  for (int k = 0; k < n_mfcc; ++k)
    for (int m = 0; m < n_mels; ++m)
      host[(size_t)k * n_mels + m] = 2.0 * std::cos(pi * (double)k * (double)(2 * m + 1) / (double)(2 * n_mels));
  double *post = host.data() + (size_t)n_mfcc * n_mels;
  for (int k = 0; k < n_mfcc; ++k) {
    post[2 * k] = k == 0 ? 1.0 / std::sqrt(4.0 * n_mels) : 1.0 / std::sqrt(2.0 * n_mels);
    post[2 * k + 1] = lifter > 0.0 ? 1.0 + lifter / 2.0 * std::sin(pi * (double)(k + 1) / lifter) : 1.0;
  }
  if (n_mfcc <= 32) {   // the same rows transposed, [n_mels][32] with zeros beyond n_mfcc: what mfcc_fast_kernel reads with scalar loads
    double *tr = post + 2 * (size_t)n_mfcc;
    for (int m = 0; m < n_mels; ++m)
      for (int k = 0; k < 32; ++k) tr[(size_t)m * 32 + k] = k < n_mfcc ? host[(size_t)k * n_mels + m] : 0.0;
  }
  double *dev = nullptr;
  SMX_HIP_CHECK(hipMalloc((void **)&dev, host.size() * sizeof(double)));
  SMX_HIP_CHECK(hipMemcpy(dev, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice));
  g_mfcc_tables.emplace(key, dev);
  return dev;
}
}  // namespace

namespace {
// {1 / c, ln c} for c = 1 + (i + 0.5) / 128, i < 128 (table_log), per device, built once
const double2 *log_table_device() {
  static std::mutex mutex;
  static std::map<int, double2 *> tables;
  int device = 0;
  SMX_HIP_CHECK(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(mutex);
  auto it = tables.find(device);
  if (it != tables.end()) return it->second;
  std::vector<double2> host(128);
  for (int i = 0; i < 128; ++i) {
    const double c = 1.0 + ((double)i + 0.5) / 128.0;
    host[(size_t)i] = make_double2(1.0 / c, std::log(c));
  }
  double2 *dev = nullptr;
  SMX_HIP_CHECK(hipMalloc((void **)&dev, host.size() * sizeof(double2)));
  SMX_HIP_CHECK(hipMemcpy(dev, host.data(), host.size() * sizeof(double2), hipMemcpyHostToDevice));
  tables.emplace(device, dev);
  return dev;
}
}  // namespace

void launch_mfcc(const MfccJob &job) {
  if (job.lead <= 0 || job.frames <= 0) return;
  const int n_mels = job.n_mels, n_mfcc = job.n_mfcc;
  // tables: raw DCT-II rows, orthonormal scales (soundml.ml:26-33), lifter weights (soundml.ml:35-42) -- built once per
  // (device, n_mels, n_mfcc, lifter) and kept, so a call uploads nothing and never synchronises its stream
  const double *d_tab = mfcc_tables(n_mels, n_mfcc, job.lifter);
  unsigned long long *d_max = nullptr;
  SMX_HIP_CHECK(smx::pool_malloc_async((void **)&d_max, sizeof(unsigned long long), job.stream));
  SMX_HIP_CHECK(hipMemsetAsync(d_max, 0, sizeof(unsigned long long), job.stream));
  const int64_t total = job.lead * (int64_t)n_mels * job.frames;
  const unsigned blocks = (unsigned)std::min<int64_t>((total + 2047) / 2048, 2048);
  if (job.elem_bytes == 8)
    SMX_LAUNCH(mel_max_kernel<double>, dim3(blocks), dim3(256), 0, job.stream, (const double *)job.mel, total, d_max);
  else
    SMX_LAUNCH(mel_max_kernel<float>, dim3(blocks), dim3(256), 0, job.stream, (const float *)job.mel, total, d_max);
  SMX_HIP_CHECK(hipGetLastError());
  if (job.lead > 65535) throw Failure("mfcc: too many leading slices for one launch");
  MfccArgs a{};
  a.mel = job.mel;
  a.out = job.out;
  a.dct = d_tab;
  a.post = d_tab + (size_t)n_mfcc * n_mels;
  a.dct_t = a.post + 2 * (size_t)n_mfcc;
  a.max_bits = d_max;
  a.lead = job.lead;
  a.frames = job.frames;
  a.n_mels = n_mels;
  a.n_mfcc = n_mfcc;
  dim3 grid((unsigned)((job.frames + 255) / 256), (unsigned)job.lead);
  const int nc = (n_mfcc + 7) / 8 * 8;
  if (job.elem_bytes == 8) {
    SMX_LAUNCH(mfcc_kernel<double>, grid, dim3(256), 0, job.stream, a);
  } else if (n_mfcc <= kFastMfccMaxCoeffs && diag_flag("SMX_MFCC_PLAIN") != 1) {
    const dim3 grid4((unsigned)((job.frames + 63) / 64), (unsigned)job.lead);
    auto go = [&](auto kernel) { SMX_LAUNCH(kernel, grid4, dim3(256), 0, job.stream, a, log_table_device()); };
    if (nc == 8) go(mfcc_fast_kernel<8>);
    else if (nc == 16) go(mfcc_fast_kernel<16>);
    else if (nc == 24) go(mfcc_fast_kernel<24>);
    else go(mfcc_fast_kernel<32>);
  } else {
    SMX_LAUNCH(mfcc_kernel<float>, grid, dim3(256), 0, job.stream, a);
  }
  SMX_HIP_CHECK(hipGetLastError());
  SMX_HIP_CHECK(hipFreeAsync(d_max, job.stream));
}

}  // namespace smx
