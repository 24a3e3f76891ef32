// Spectral-shape features (spectral.ml:171-255) and the linear-frequency chroma projection (chroma.ml:285-317)
// over a device-resident spectrogram [lead; bins; frames] (frames fastest).
//
// Both are reductions along the bin axis with a float64 interior and one rounding to the spectrogram's dtype,
// in the reference's operation order for float64 input (normalise the frame, then weight, then reduce; s^power floored,
// then the two means); float32 input divides by the frame's sum after the weighted reduction instead of before it (one
// walk over the spectrogram less; the same value to a rounding of the float64 interior).  Frames sit across lanes, so every load of a bin row is one coalesced run of 64 frames; the
// features are HBM-bound (one to three passes over the spectrogram, the later ones mostly from L2 / MALL).
//
//   features:  a workgroup owns 64 frames; its Q = 4 waves each walk a quarter of the bins and the partial
//              sums meet in LDS in a fixed order ((q0 + q1) + (q2 + q3): deterministic).  The roll-off is a
//              running sum compared against a threshold, so it keeps the sequential order: Q = 1.
//   chroma:    one thread per frame accumulates a chunk of 12 chroma rows in float64 registers, the weights of a
//              bin (transposed table, uniform address) come through scalar loads; raw projections go to a
//              float64 scratch, the per-frame normalisation is a second small launch.
#include <cfloat>

#include "smx_internal.hpp"

namespace smx {
namespace {

struct SpectralArgs {
  const void *s;
  void *out;
  const double *freqs;      // device [bins] or null
  const void *centroid;     // device [lead; frames] or null
  double step, p, inv_p, power;
  int64_t lead, bins, frames, ftiles;
  int *invalid;
};

__device__ __forceinline__ double pow_fixed(double v, double p) {   // Nx.pow_s with the common exponents exact
  if (p == 2.0) return v * v;
  if (p == 1.0) return v;
  return pow(v, p);
}

// walks bins [k0, k1) of one frame column in order, kDepth loads in flight at a time
constexpr int kDepth = 8;
template <typename T, typename F>
__device__ __forceinline__ void for_bins(const T *col, int64_t frames, int64_t k0, int64_t k1, F &&f) {
  int64_t k = k0;
  for (; k + kDepth <= k1; k += kDepth) {
    T v[kDepth];
#pragma unroll
    for (int j = 0; j < kDepth; ++j) v[j] = col[(k + j) * frames];
#pragma unroll
    for (int j = 0; j < kDepth; ++j) f(k + j, (double)v[j]);
  }
  for (; k < k1; ++k) f(k, (double)col[k * frames]);
}

// A workgroup = four waves: the Q bin ranges of one 64-frame tile, or (Q = 1, the sequential roll-off walk) four
// neighbouring tiles, whose 256-byte row pieces are then one contiguous run per row and workgroup (0.46 -> 0.38 ms at
// C2; sixteen-wave workgroups of 4 tiles x 4 ranges measured slower for the other three features).
template <int Q> constexpr int tiles_of = Q == 1 ? 4 : 1;
template <typename T, int FEATURE, int Q>
__global__ void __launch_bounds__(64 * Q * tiles_of<Q>) spectral_kernel(SpectralArgs a) {
  constexpr int kTiles = tiles_of<Q>;
  __shared__ double red_all[kTiles][2][Q][64];
  const int fx = threadIdx.x & 63, q = (threadIdx.x >> 6) % Q, sub = threadIdx.x / (64 * Q);
  double (*red)[Q][64] = red_all[sub];
  const int64_t groups = (a.ftiles + kTiles - 1) / kTiles;
  const int64_t clip = blockIdx.x / groups, tile = (blockIdx.x % groups) * kTiles + sub;
  const int64_t t = tile * 64 + fx;                    // a tile past the last one shadows the last frame too
  const bool live = t < a.frames;
  const int64_t tc = live ? t : a.frames - 1;          // idle lanes shadow the last frame (no stores)
  const T *col = reinterpret_cast<const T *>(a.s) + clip * a.bins * a.frames + tc;
  const int64_t bq = (a.bins + Q - 1) / Q;
  const int64_t k0 = q * bq, k1 = (k0 + bq < a.bins) ? k0 + bq : a.bins;
  auto fq = [&](int64_t k) { return a.freqs ? a.freqs[k] : (double)k * a.step; };
  // partial sums of the Q waves meet in a fixed order
  auto combine = [&](double v, int slot) {
    if constexpr (Q == 1) return v;
    red[slot][q][fx] = v;
    __syncthreads();
    const double r = (red[slot][0][fx] + red[slot][1][fx]) + (red[slot][2][fx] + red[slot][3][fx]);
    __syncthreads();
    return r;
  };
  bool bad = false;
  double result = 0.0;

  if constexpr (FEATURE == SPECTRAL_FLATNESS) {
    // sum of logs as the log of a product: mantissas multiply (renormalised every 8 factors, so the product stays
    // above 2^-9), exponents add as integers; one log per wave instead of one per bin (the float64 log was what this
    // feature spent its time on: 246 M of them at C2).  Same value to a few ulp of the float64 sum.
    double prod = 1.0, sa = 0.0;
    int64_t esum = 0;
    for_bins(col, a.frames, k0, k1, [&](int64_t k, double v) {
      bad |= !(v >= 0.0);
      const double pw = pow_fixed(v, a.power);
      const double f = pw > a.p ? pw : a.p;            // a.p = amin > 0
      sa += f;
      if (f < INFINITY) {
        prod *= __builtin_amdgcn_frexp_mant(f);
        esum += __builtin_amdgcn_frexp_exp(f);
      } else {
        prod = f;                                       // an infinite (or NaN) bin: the sum of logs is that too
      }
      if (((k - k0) & 7) == 7) {                        // wave-uniform
        if (prod < INFINITY) {
          esum += __builtin_amdgcn_frexp_exp(prod);
          prod = __builtin_amdgcn_frexp_mant(prod);
        }
      }
    });
    double sl = log(prod) + (double)esum * 0.693147180559945309417232121458;
    sl = combine(sl, 0);
    sa = combine(sa, 1);
    result = exp(sl / (double)a.bins) / (sa / (double)a.bins);
  } else if constexpr (FEATURE == SPECTRAL_ROLLOFF) {
    double cum = 0.0;
    for_bins(col, a.frames, 0, a.bins, [&](int64_t, double v) {
      bad |= !(v >= 0.0);
      cum += v;
    });
    const double threshold = cum * a.p;                // a.p = roll_percent; the total is the last cumulative value
    cum = 0.0;
    double best = INFINITY;
    for_bins(col, a.frames, 0, a.bins, [&](int64_t k, double v) {
      cum += v;
      const double f = fq(k);
      best = (cum >= threshold && f < best) ? f : best;
    });
    result = best;
  } else {
    // one walk gives the frame's sum and its first moment: sum(f_k (v_k / len)) is evaluated as sum(f_k v_k) / len, the
    // same value to a rounding of the float64 interior (the reference normalises first: spectral.ml:155-163, 171-196)
    // float32 spectrograms only: with float64 input f_k v_k can overflow where f_k (v_k / len) does not, so that dtype
    // keeps the reference's order and its extra walk
    constexpr bool kFused = sizeof(T) == 4;
    double len = 0.0, m1 = 0.0;
    const bool need_c = !(FEATURE == SPECTRAL_BANDWIDTH && a.centroid);
    for_bins(col, a.frames, k0, k1, [&](int64_t k, double v) {
      bad |= !(v >= 0.0);
      len += v;
      if (kFused && need_c) m1 += fq(k) * v;
    });
    len = combine(len, 0);
    const double safe = len < DBL_MIN ? 1.0 : len;     // spectral.ml:155-163
    double c64;
    if (!need_c) {
      c64 = (double)reinterpret_cast<const T *>(a.centroid)[clip * a.frames + tc];
    } else if constexpr (kFused) {
      c64 = combine(m1, 1) / safe;
    } else {
      for_bins(col, a.frames, k0, k1, [&](int64_t k, double v) { m1 += fq(k) * (v / safe); });
      c64 = combine(m1, 1);
    }
    if constexpr (FEATURE == SPECTRAL_CENTROID) {
      result = c64;
    } else {
      double w = 0.0;
      for_bins(col, a.frames, k0, k1, [&](int64_t k, double v) {
        const double deviation = fabs(c64 - fq(k));
        w += (kFused ? v : v / safe) * pow_fixed(deviation, a.p);
      });
      w = combine(w, 0);
      if (kFused) w /= safe;
      result = a.p == 2.0 ? sqrt(w) : (a.p == 1.0 ? w : pow(w, a.inv_p));
    }
  }
  if (bad && live) atomicOr(a.invalid, 1);
  if (live && q == 0) reinterpret_cast<T *>(a.out)[clip * a.frames + t] = (T)result;
}

template <typename T, int FEATURE, int Q>
void launch_feature(const SpectralArgs &a, hipStream_t stream) {
  constexpr int kTiles = tiles_of<Q>;
  const int64_t blocks = a.lead * ((a.ftiles + kTiles - 1) / kTiles);
  if (blocks > 2147483647LL) throw Failure("spectral: too many frame tiles for one launch");
  SMX_LAUNCH((spectral_kernel<T, FEATURE, Q>), dim3((unsigned)blocks), dim3(64 * Q * kTiles), 0, stream, a);
  SMX_HIP_CHECK(hipGetLastError());
}

template <typename T>
void dispatch_feature(int feature, const SpectralArgs &a, hipStream_t stream) {
  switch (feature) {
    case SPECTRAL_CENTROID: launch_feature<T, SPECTRAL_CENTROID, 4>(a, stream); break;
    case SPECTRAL_BANDWIDTH: launch_feature<T, SPECTRAL_BANDWIDTH, 4>(a, stream); break;
    case SPECTRAL_ROLLOFF: launch_feature<T, SPECTRAL_ROLLOFF, 1>(a, stream); break;
    case SPECTRAL_FLATNESS: launch_feature<T, SPECTRAL_FLATNESS, 4>(a, stream); break;
    default: throw Failure("spectral: unknown feature");
  }
}

// ---- chroma ------------------------------------------------------------------------------------------------
constexpr int kChromaChunk = 12;   // chroma rows accumulated per pass over the bins

struct ChromaArgs {
  const void *s;            // [lead; bins; frames]
  double *raw;              // [lead; n_chroma; frames] float64 scratch
  const double *wt;         // [bins; rows_pad], rows_pad = n_chroma rounded up to a multiple of kChromaChunk
  int64_t rows_pad;
  void *out;                // [lead; n_chroma; frames]
  int64_t lead, bins, frames, ftiles;
  int n_chroma, norm;
  double norm_p, inv_p, tiny;
};

template <typename T>
__global__ void __launch_bounds__(256) chroma_project_kernel(ChromaArgs a) {
  const int64_t clip = blockIdx.x / a.ftiles, tile = blockIdx.x % a.ftiles;
  const int64_t t = tile * 256 + threadIdx.x;
  if (t >= a.frames) return;
  const int c0 = blockIdx.y * kChromaChunk;
  const T *col = reinterpret_cast<const T *>(a.s) + clip * a.bins * a.frames + t;
  const double *wt = a.wt + c0;                        // rows padded with zeros to a whole chunk: no guards below
  double acc[kChromaChunk];
#pragma unroll
  for (int i = 0; i < kChromaChunk; ++i) acc[i] = 0.0;
  int64_t k = 0;
  for (; k + kDepth <= a.bins; k += kDepth) {
    T v[kDepth];
#pragma unroll
    for (int j = 0; j < kDepth; ++j) v[j] = col[(k + j) * a.frames];
#pragma unroll
    for (int j = 0; j < kDepth; ++j) {
      const double *w = wt + (k + j) * a.rows_pad;     // uniform: scalar loads
      const double d = (double)v[j];
#pragma unroll
      for (int i = 0; i < kChromaChunk; ++i) acc[i] += w[i] * d;
    }
  }
  for (; k < a.bins; ++k) {
    const double *w = wt + k * a.rows_pad;
    const double d = (double)col[k * a.frames];
#pragma unroll
    for (int i = 0; i < kChromaChunk; ++i) acc[i] += w[i] * d;
  }
  double *raw = a.raw + (clip * a.n_chroma + c0) * a.frames + t;
#pragma unroll
  for (int i = 0; i < kChromaChunk; ++i)
    if (c0 + i < a.n_chroma) raw[(int64_t)i * a.frames] = acc[i];
}

// chroma.ml:58-88: each frame divided by its own length in the norm; lengths below `tiny` divide by one
template <typename T>
__global__ void __launch_bounds__(256) chroma_normalise_kernel(ChromaArgs a) {
  const int64_t clip = blockIdx.x / a.ftiles, tile = blockIdx.x % a.ftiles;
  const int64_t t = tile * 256 + threadIdx.x;
  if (t >= a.frames) return;
  const double *raw = a.raw + clip * a.n_chroma * a.frames + t;
  T *out = reinterpret_cast<T *>(a.out) + clip * a.n_chroma * a.frames + t;
  double length = 1.0;
  if (a.norm != SMX_CHROMA_NORM_NONE) {
    double acc = 0.0;
    for (int c = 0; c < a.n_chroma; ++c) {
      const double m = fabs(raw[(int64_t)c * a.frames]);
      if (a.norm == SMX_CHROMA_NORM_INF) acc = m > acc ? m : acc;
      else acc += pow_fixed(m, a.norm_p);
    }
    if (a.norm == SMX_CHROMA_NORM_P && a.norm_p != 1.0) acc = a.norm_p == 2.0 ? sqrt(acc) : pow(acc, a.inv_p);
    length = acc < a.tiny ? 1.0 : acc;
  }
  for (int c = 0; c < a.n_chroma; ++c) {
    const double v = raw[(int64_t)c * a.frames];
    out[(int64_t)c * a.frames] = (T)(a.norm == SMX_CHROMA_NORM_NONE ? v : v / length);
  }
}

}  // namespace

bool launch_spectral(const SpectralJob &job) {
  if (job.lead <= 0 || job.bins <= 0 || job.frames <= 0) return true;
  SpectralArgs a{};
  a.s = job.s;
  a.out = job.out;
  a.centroid = job.centroid;
  a.step = job.step;
  a.p = job.p;
  a.inv_p = 1.0 / job.p;
  a.power = job.power;
  a.lead = job.lead;
  a.bins = job.bins;
  a.frames = job.frames;
  a.ftiles = (job.frames + 63) / 64;
  // scratch: the invalid-entry flag, then the caller's frequency grid
  const size_t grid_bytes = job.freqs ? (size_t)job.bins * sizeof(double) : 0;
  unsigned char *scratch = nullptr;
  SMX_HIP_CHECK(smx::pool_malloc_async((void **)&scratch, 16 + grid_bytes, job.stream));
  a.invalid = reinterpret_cast<int *>(scratch);
  SMX_HIP_CHECK(hipMemsetAsync(scratch, 0, 16, job.stream));
  if (job.freqs) {
    a.freqs = reinterpret_cast<const double *>(scratch + 16);
    SMX_HIP_CHECK(hipMemcpyAsync(scratch + 16, job.freqs, grid_bytes, hipMemcpyHostToDevice, job.stream));
    SMX_HIP_CHECK(hipStreamSynchronize(job.stream));   // the grid is the caller's pageable memory
  }
  if (job.elem_bytes == 8) dispatch_feature<double>(job.feature, a, job.stream);
  else dispatch_feature<float>(job.feature, a, job.stream);
  int invalid = 0;
  SMX_HIP_CHECK(hipMemcpyAsync(&invalid, a.invalid, sizeof(int), hipMemcpyDeviceToHost, job.stream));
  SMX_HIP_CHECK(hipStreamSynchronize(job.stream));
  SMX_HIP_CHECK(hipFreeAsync(scratch, job.stream));
  return invalid == 0;
}

void launch_chroma(const ChromaJob &job) {
  const smx_chroma_config &c = *job.config;
  if (job.lead <= 0 || job.frames <= 0) return;
  ChromaArgs a{};
  a.s = job.s;
  a.out = job.out;
  a.wt = c.device_weights();
  a.lead = job.lead;
  a.bins = c.bins();
  a.frames = job.frames;
  a.ftiles = (job.frames + 255) / 256;
  a.n_chroma = (int)c.n_chroma;
  a.rows_pad = (c.n_chroma + kChromaChunk - 1) / kChromaChunk * kChromaChunk;
  a.norm = job.norm;
  a.norm_p = job.norm_p;
  a.inv_p = job.norm == SMX_CHROMA_NORM_P ? 1.0 / job.norm_p : 1.0;
  a.tiny = job.elem_bytes == 8 ? DBL_MIN : (double)FLT_MIN;   // chroma.ml:42-55 smallest_normal
  const int64_t blocks = a.lead * a.ftiles;
  const int64_t chunks = (c.n_chroma + kChromaChunk - 1) / kChromaChunk;
  if (blocks > 2147483647LL || chunks > 65535) throw Failure("chroma: too many tiles for one launch");
  SMX_HIP_CHECK(smx::pool_malloc_async((void **)&a.raw, (size_t)job.lead * (size_t)c.n_chroma * (size_t)job.frames * sizeof(double),
                               job.stream));
  if (job.elem_bytes == 8) {
    SMX_LAUNCH(chroma_project_kernel<double>, dim3((unsigned)blocks, (unsigned)chunks), dim3(256), 0, job.stream, a);
    SMX_LAUNCH(chroma_normalise_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, job.stream, a);
  } else {
    SMX_LAUNCH(chroma_project_kernel<float>, dim3((unsigned)blocks, (unsigned)chunks), dim3(256), 0, job.stream, a);
    SMX_LAUNCH(chroma_normalise_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, job.stream, a);
  }
  SMX_HIP_CHECK(hipGetLastError());
  SMX_HIP_CHECK(hipFreeAsync(a.raw, job.stream));
}

}  // namespace smx
