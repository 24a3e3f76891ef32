// Log-mel / MFCC tail (Soundml.mfcc, soundml.ml:50-95; Convert.power_to_db, convert.ml:30-50):
//   db = 10 log10 (max (mel, amin)),  clamped at (max over the WHOLE tensor) - 80 dB,
//   cepstrum[k] = scale_k * sum_m db[m] * 2 cos (pi k (2 m + 1) / (2 n_mels)),  scale_0 = 1/sqrt(4 n_mels),
//   scale_k = 1/sqrt(2 n_mels), optionally times the sinusoidal lifter, rounded once to the audio dtype.
// The interior is float64 whatever the audio dtype (the reference's contract).  Two launches: a max reduction
// of the mel spectrogram (the logarithm is monotonic, so the maximum of db is db of the maximum), then one
// thread per (clip, frame) walking the mel axis with frames across lanes (coalesced).
#include "smx_internal.hpp"

namespace smx {
namespace {

template <typename T>
__global__ void __launch_bounds__(256) mel_max_kernel(const T *mel, int64_t total, unsigned long long *result) {
  double m = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const double v = (double)mel[i];
    m = v > m ? v : m;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(m, off);
    m = o > m ? o : m;
  }
  // non-negative doubles order like their bit patterns
  if ((threadIdx.x & 63) == 0) atomicMax(result, (unsigned long long)__double_as_longlong(m));
}

struct MfccArgs {
  const void *mel;          // [lead; n_mels; frames]
  void *out;                // [lead; n_mfcc; frames]
  const double *dct;        // [n_mfcc; n_mels] raw type-II rows 2 cos(pi k (2m+1) / (2 n_mels))
  const double *post;       // [n_mfcc; 2]: orthonormal scale, lifter weight (1 when absent)
  const unsigned long long *max_bits;
  int64_t lead, frames;
  int n_mels, n_mfcc;
};

constexpr int kChunk = 16;   // cepstral coefficients accumulated per pass over the mel axis

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

}  // namespace

void launch_mfcc(const MfccJob &job) {
  if (job.lead <= 0 || job.frames <= 0) return;
  const int n_mels = job.n_mels, n_mfcc = job.n_mfcc;
  // tables: raw DCT-II rows, orthonormal scales (soundml.ml:26-33), lifter weights (soundml.ml:35-42)
  std::vector<double> host((size_t)n_mfcc * n_mels + 2 * (size_t)n_mfcc);
  const double pi = 3.14159265358979323846;
  for (int k = 0; k < n_mfcc; ++k)
    for (int m = 0; m < n_mels; ++m)
      host[(size_t)k * n_mels + m] = 2.0 * std::cos(pi * (double)k * (double)(2 * m + 1) / (double)(2 * n_mels));
  double *post = host.data() + (size_t)n_mfcc * n_mels;
  for (int k = 0; k < n_mfcc; ++k) {
    post[2 * k] = k == 0 ? 1.0 / std::sqrt(4.0 * n_mels) : 1.0 / std::sqrt(2.0 * n_mels);
    post[2 * k + 1] = job.lifter > 0.0 ? 1.0 + job.lifter / 2.0 * std::sin(pi * (double)(k + 1) / job.lifter) : 1.0;
  }
  double *d_tab = nullptr;
  SMX_HIP_CHECK(hipMallocAsync((void **)&d_tab, (host.size() + 1) * sizeof(double), job.stream));
  SMX_HIP_CHECK(hipMemcpyAsync(d_tab, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice, job.stream));
  unsigned long long *d_max = reinterpret_cast<unsigned long long *>(d_tab + host.size());
  SMX_HIP_CHECK(hipMemsetAsync(d_max, 0, sizeof(unsigned long long), job.stream));
  SMX_HIP_CHECK(hipStreamSynchronize(job.stream));   // `host` is pageable memory that dies with this call
  const int64_t total = job.lead * (int64_t)n_mels * job.frames;
  const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
  if (job.elem_bytes == 8)
    hipLaunchKernelGGL(mel_max_kernel<double>, dim3(blocks), dim3(256), 0, job.stream, (const double *)job.mel, total, d_max);
  else
    hipLaunchKernelGGL(mel_max_kernel<float>, dim3(blocks), dim3(256), 0, job.stream, (const float *)job.mel, total, d_max);
  SMX_HIP_CHECK(hipGetLastError());
  if (job.lead > 65535) throw Failure("mfcc: too many leading slices for one launch");
  MfccArgs a{};
  a.mel = job.mel;
  a.out = job.out;
  a.dct = d_tab;
  a.post = d_tab + (size_t)n_mfcc * n_mels;
  a.max_bits = d_max;
  a.lead = job.lead;
  a.frames = job.frames;
  a.n_mels = n_mels;
  a.n_mfcc = n_mfcc;
  dim3 grid((unsigned)((job.frames + 255) / 256), (unsigned)job.lead);
  if (job.elem_bytes == 8) hipLaunchKernelGGL(mfcc_kernel<double>, grid, dim3(256), 0, job.stream, a);
  else hipLaunchKernelGGL(mfcc_kernel<float>, grid, dim3(256), 0, job.stream, a);
  SMX_HIP_CHECK(hipGetLastError());
  SMX_HIP_CHECK(hipFreeAsync(d_tab, job.stream));
}

}  // namespace smx
