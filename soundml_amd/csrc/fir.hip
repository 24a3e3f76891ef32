// FIR block convolution by overlap-save on the library's own FFT
// (BASELINE config 4; behavioural model: the reference's OLS resample executor,
// resample.ml:383-415 -- fixed block grid, zeros before the stream start, the
// wrap-carrying head of every circular result discarded).
//
//   y[c][i] = sum_k h[k] x[c][i - k],  i in [0, n)
//
// Two consecutive real blocks of one channel are packed as ONE complex signal
// z = a + i b: filtering with a real h commutes with the packing, so a single
// complex FFT(N) -> pointwise multiply with H -> inverse FFT(N) yields both
// blocks (real part / imaginary part) with no real-FFT post-pass.
//
// FFT: in-place Stockham (autosort) passes of radix 16 / 4 / 2 -- N = 16384 is
// 16.16.16.4.  One workgroup of N/16 threads owns a block pair; every thread keeps 16
// complex points in registers, so a pass is: read 16 (lane-contiguous, conflict free),
// barrier, twiddle + register DFT, write 16 to the autosort positions, barrier.  LDS
// addresses are XOR-swizzled (a ^ ((a >> 5) & 31)) which makes every pass's reads
// conflict free and its writes at most 2-way (tools/sim_fir_fft.py).  The first
// pass reads HBM straight into registers, the last inverse pass stores to HBM from
// registers, the H multiply happens in registers between the two transforms and the
// inverse is conj(FFT(conj(.))) with 1/N folded into H: 7 LDS round trips for
// N = 16384.  Twiddles: one table read per pass and radix group, powers by binary
// multiplication (depth <= 4).
//
// Algorithmic HBM bytes: 8 B per sample (4 in + 4 out) plus the (taps-1)/L halo.
#include <cmath>

#include "fft_device.hpp"
#include "smx_internal.hpp"

struct smx_fir_plan {
  int64_t taps = 0;
  int64_t nfft = 0;      // N
  int64_t valid = 0;     // L = N - taps + 1
  int log2n = 0;
  std::vector<double> h;
  struct Tables {
    float2 *h_nat = nullptr;  // H[k] / N, natural order
    float2 *tw = nullptr;     // exp(-2 pi i j / N), j < N/2
  };
  const Tables &tables() const;
  ~smx_fir_plan();

 private:
  mutable std::mutex mutex_;
  mutable std::map<int, Tables> tables_;
};

namespace smx {
namespace {

unsigned brev_host(unsigned v, int bits) {
  unsigned r = 0;
  for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1u) << (bits - 1 - i);
  return r;
}

struct FirArgs {
  const float *x;
  float *y;
  int64_t n, x_stride, y_stride;
  int64_t taps, nfft, valid;
  int log2n;
  int64_t pairs_per_channel;
  const float2 *h_nat;
  const float2 *tw;     // exp(-2 pi i j / N), j < N/2
};

using namespace fftdev;

template <int LOG2N>
__global__ void __launch_bounds__((1 << LOG2N) / 16) fir_ols_kernel(FirArgs a) {
  constexpr int N = 1 << LOG2N, T = N / 16;
  constexpr int RL = LastPass<LOG2N>::R, NSL = LastPass<LOG2N>::NS, GL = 16 / RL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2 *z = reinterpret_cast<float2 *>(smem);
  const int tid = threadIdx.x;
  const int64_t channel = blockIdx.x / a.pairs_per_channel;
  const int64_t pair = blockIdx.x % a.pairs_per_channel;
  const float *x = a.x + channel * a.x_stride;
  float *y = a.y + channel * a.y_stride;
  const int64_t base_a = (2 * pair) * a.valid - (a.taps - 1);      // first input of block a
  const int64_t base_b = base_a + a.valid;
  c32 r[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) {   // element tid + T*m: lane-contiguous HBM reads, zeros outside the stream
    const int64_t sa = base_a + tid + T * m, sb = base_b + tid + T * m;
    r[m].x = (sa >= 0 && sa < a.n) ? x[sa] : 0.0f;
    r[m].y = (sb >= 0 && sb < a.n) ? x[sb] : 0.0f;
  }
  fft_passes<LOG2N, true>(r, z, tid, a.tw);
  // Y = Z * H (1/N folded in); inverse = conj(FFT(conj(Y))): store conj(Y) for the second transform
#pragma unroll
  for (int i = 0; i < GL; ++i)
#pragma unroll
    for (int j = 0; j < RL; ++j) {
      const int idx = out_index<RL, NSL, T>(tid, i, j);
      const float2 hv = a.h_nat[idx];
      const c32 yv = cmul(r[i * RL + j], c32{hv.x, hv.y});
      z[swz(idx)] = make_float2(yv.x, -yv.y);
    }
  fft_passes<LOG2N, false>(r, z, tid, a.tw);
  const int64_t out_a = (2 * pair) * a.valid, out_b = out_a + a.valid;
  const int skip = (int)a.taps - 1;
#pragma unroll
  for (int i = 0; i < GL; ++i)
#pragma unroll
    for (int j = 0; j < RL; ++j) {
      const int idx = out_index<RL, NSL, T>(tid, i, j) - skip;   // position inside the valid span
      if (idx >= 0) {
        if (out_a + idx < a.n) y[out_a + idx] = r[i * RL + j].x;      // Re(conj(.)) =  Re
        if (out_b + idx < a.n) y[out_b + idx] = -r[i * RL + j].y;     // Im(conj(.)) = -Im
      }
    }
}

}  // namespace
}  // namespace smx

const smx_fir_plan::Tables &smx_fir_plan::tables() const {
  smx::init_device_pool();
  int device = 0;
  SMX_HIP_CHECK(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(mutex_);
  auto it = tables_.find(device);
  if (it != tables_.end()) return it->second;
  // H = DFT_N(h) in float64: iterative DIF (bit-reversed out), then unscrambled to natural order
  const int64_t N = nfft;
  std::vector<double> re((size_t)N, 0.0), im((size_t)N, 0.0);
  for (int64_t i = 0; i < taps; ++i) re[(size_t)i] = h[(size_t)i];
  for (int64_t half = N >> 1; half >= 1; half >>= 1) {
    const int64_t tstep = (N >> 1) / half;
    for (int64_t b = 0; b < (N >> 1); ++b) {
      const int64_t j = b & (half - 1);
      const int64_t i0 = ((b - j) << 1) + j, i1 = i0 + half;
      const double ang = -2.0 * M_PI * (double)(j * tstep) / (double)N;
      const double wr = std::cos(ang), wi = std::sin(ang);
      const double dr = re[(size_t)i0] - re[(size_t)i1], di = im[(size_t)i0] - im[(size_t)i1];
      re[(size_t)i0] += re[(size_t)i1];
      im[(size_t)i0] += im[(size_t)i1];
      re[(size_t)i1] = dr * wr - di * wi;
      im[(size_t)i1] = dr * wi + di * wr;
    }
  }
  std::vector<float2> hb((size_t)N), tw((size_t)(N / 2 > 0 ? N / 2 : 1));
  for (int64_t i = 0; i < N; ++i) {
    const unsigned k = smx::brev_host((unsigned)i, log2n);   // position i holds H[brev(i)]
    hb[k] = make_float2((float)(re[(size_t)i] / (double)N), (float)(im[(size_t)i] / (double)N));
  }
  for (int64_t j = 0; j < N / 2; ++j) {
    const double ang = -2.0 * M_PI * (double)j / (double)N;
    tw[(size_t)j] = make_float2((float)std::cos(ang), (float)std::sin(ang));
  }
  Tables t;
  SMX_HIP_CHECK(hipMalloc((void **)&t.h_nat, hb.size() * sizeof(float2)));
  SMX_HIP_CHECK(hipMemcpy(t.h_nat, hb.data(), hb.size() * sizeof(float2), hipMemcpyHostToDevice));
  SMX_HIP_CHECK(hipMalloc((void **)&t.tw, tw.size() * sizeof(float2)));
  SMX_HIP_CHECK(hipMemcpy(t.tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
  return tables_.emplace(device, t).first->second;
}

smx_fir_plan::~smx_fir_plan() {
  for (auto &kv : tables_) {
    (void)hipFree(kv.second.h_nat);
    (void)hipFree(kv.second.tw);
  }
}

using namespace smx;

namespace {
template <typename F>
int guarded_fir(F &&body) {
  try {
    body();
    return SMX_OK;
  } catch (const InvalidArgument &e) {
    set_last_error(e.what());
    return SMX_INVALID_ARGUMENT;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return SMX_FAILURE;
  }
}

void fir_apply_dev(const smx_fir_plan &p, const float *d_x, int64_t channels, int64_t n, int64_t x_stride,
                   float *d_y, int64_t y_stride, hipStream_t stream) {
  if (channels < 0 || n < 0) throw Failure("fir_apply: negative extent");
  if (channels == 0 || n == 0) return;
  if (x_stride < n || y_stride < n) throw Failure("fir_apply: stride smaller than the signal length");
  if (!d_x || !d_y) throw Failure("fir_apply: null device pointer");
  const smx_fir_plan::Tables &t = p.tables();
  FirArgs a{};
  a.x = d_x;
  a.y = d_y;
  a.n = n;
  a.x_stride = x_stride;
  a.y_stride = y_stride;
  a.taps = p.taps;
  a.nfft = p.nfft;
  a.valid = p.valid;
  a.log2n = p.log2n;
  const int64_t blocks = (n + p.valid - 1) / p.valid;
  a.pairs_per_channel = (blocks + 1) / 2;
  a.h_nat = t.h_nat;
  a.tw = t.tw;
  const int64_t grid = channels * a.pairs_per_channel;
  if (grid > 0x7fffffff) throw Failure("fir_apply: too many blocks for one launch");
  const size_t lds = (size_t)p.nfft * sizeof(float2);
  auto launch = [&](auto kernel, int threads) {
    SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SMX_LAUNCH(kernel, dim3((unsigned)grid), dim3(threads), lds, stream, a);
  };
  switch (p.log2n) {
    case 10: launch(fir_ols_kernel<10>, 64); break;
    case 11: launch(fir_ols_kernel<11>, 128); break;
    case 12: launch(fir_ols_kernel<12>, 256); break;
    case 13: launch(fir_ols_kernel<13>, 512); break;
    case 14: launch(fir_ols_kernel<14>, 1024); break;
    default: throw Failure("fir_apply: unsupported block size");
  }
  SMX_HIP_CHECK(hipGetLastError());
}
}  // namespace

extern "C" {

int smx_fir_kaiser_beta(double attenuation_db, double *out) {
  return guarded_fir([&] { *out = kaiser_beta(attenuation_db); });
}

int smx_fir_design_lowpass(int64_t taps, double cutoff, double beta, double *h) {
  return guarded_fir([&] {
    if (!h && taps > 0) throw Failure("design_lowpass: null output");
    design_lowpass(taps, cutoff, beta, h);
  });
}

int smx_fir_plan_create(const double *h, int64_t taps, smx_fir_plan **out) {
  return guarded_fir([&] {
    if (!out) throw Failure("fir_plan_create: null output handle");
    if (taps < 1)
      throw InvalidArgument(format("fir_plan_create: cannot filter with %lld taps (taps must be at least 1)",
                                   (long long)taps));
    if (taps > 8192)
      throw InvalidArgument(format(
          "fir_plan_create: cannot filter with %lld taps (this device path holds at most 8192 taps per "
          "16384-point block)",
          (long long)taps));
    if (!h) throw Failure("fir_plan_create: null taps");
    auto *p = new smx_fir_plan();
    p->taps = taps;
    int64_t n = 1024;
    while (n < 4 * taps && n < 16384) n *= 2;
    while (n < 2 * taps) n *= 2;
    p->nfft = n;
    p->valid = n - taps + 1;
    p->log2n = 0;
    while ((int64_t(1) << p->log2n) < n) ++p->log2n;
    p->h.assign(h, h + taps);
    *out = p;
  });
}

void smx_fir_plan_destroy(smx_fir_plan *p) { delete p; }
int64_t smx_fir_plan_block(const smx_fir_plan *p) { return p ? p->nfft : -1; }

int smx_fir_apply_f32_dev(const smx_fir_plan *p, const float *d_x, int64_t channels, int64_t n,
                          int64_t x_stride, float *d_y, int64_t y_stride, void *stream) {
  return guarded_fir([&] {
    if (!p) throw Failure("fir_apply: null plan");
    fir_apply_dev(*p, d_x, channels, n, x_stride, d_y, y_stride, (hipStream_t)stream);
  });
}

int smx_fir_apply_f32(const smx_fir_plan *p, const float *x, int64_t channels, int64_t n, float *y) {
  return guarded_fir([&] {
    if (!p) throw Failure("fir_apply: null plan");
    if (channels < 0 || n < 0) throw Failure("fir_apply: negative extent");
    if (channels == 0 || n == 0) return;
    if (!x || !y) throw Failure("fir_apply: null pointer");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1)
      throw Failure("soundml_amd: no HIP device is available (this library has no CPU fallback)");
    const size_t bytes = (size_t)channels * (size_t)n * sizeof(float);
    float *dx = nullptr, *dy = nullptr;
    SMX_HIP_CHECK(hipMalloc((void **)&dx, bytes));
    if (hipMalloc((void **)&dy, bytes) != hipSuccess) {
      (void)hipFree(dx);
      throw Failure("fir_apply: device allocation failed");
    }
    try {
      SMX_HIP_CHECK(hipMemcpy(dx, x, bytes, hipMemcpyHostToDevice));
      fir_apply_dev(*p, dx, channels, n, n, dy, n, nullptr);
      SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
      SMX_HIP_CHECK(hipMemcpy(y, dy, bytes, hipMemcpyDeviceToHost));
    } catch (...) {
      (void)hipFree(dx);
      (void)hipFree(dy);
      throw;
    }
    (void)hipFree(dx);
    (void)hipFree(dy);
  });
}

}  // extern "C"
