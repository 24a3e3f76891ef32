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
#include <cstdlib>

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
    // half-size real-transform kernel (M = N/2 complex points per block of N real samples)
    float2 *h_half = nullptr; // H[k] / (4 M), k = 0 .. M  (the 1/2 of the real post-pass, the 1/2 of the inverse
                              // pre-pass and the inverse transform's 1/M folded in)
    float2 *tw_m = nullptr;   // exp(-2 pi i j / M), j < M/2
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
  // half-size kernel
  const float2 *h_half; // H[k] / (4 M), k <= M
  const float2 *tw_m;   // exp(-2 pi i j / M), j < M/2
  int64_t lead;         // samples of every circular result that are discarded (even, >= taps - 1)
  int64_t step;         // block advance = N - lead (even)
  int64_t blocks_per_channel;
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

// ---- one real block per workgroup on the half-size transform --------------------------------------------------
// A block of N real samples is ONE complex transform of M = N/2 points over its (even, odd) sample pairs:
//   forward FFT_M  ->  real post-pass X[k] = (E - i w_k D) / 2  ->  Y = X H  ->  inverse pre-pass
//   Z'[k] = ((Y[k] + conj Y[M-k]) + i conj(w_k) (Y[k] - conj Y[M-k])) / 2  ->  inverse FFT_M  ->  (y[2n], y[2n+1])
// with E = Z[k] + conj Z[M-k], D = Z[k] - conj Z[M-k], w_k = exp(-2 pi i k / N).  Post-pass, product and pre-pass
// are one pointwise stage over the pairs (k, M - k).  The workgroup has M/16 threads and M float2 of LDS: for the
// 8192-tap plan 512 threads and 64 KB, so TWO workgroups share a CU and run out of phase -- one in its LDS
// exchanges while the other is in its butterflies -- where the packed-pair kernel above (1024 threads, 128 KB: one
// workgroup per CU, every wave in the same phase) pays VALU time plus LDS time.
// The block grid advances by an EVEN step (N - lead, lead = the even number >= taps - 1 of wrap-carrying samples
// that are discarded), so every window starts on an even sample: 8-byte loads and stores.
template <int LOG2M, bool ALIGNED>
__global__ void __launch_bounds__((1 << LOG2M) / 16, 4) fir_ols_real_kernel(FirArgs a) {   // 4 waves per SIMD: two 512-thread workgroups per CU
  constexpr int M = 1 << LOG2M, T = M / 16;
  constexpr int RL = LastPass<LOG2M>::R, NSL = LastPass<LOG2M>::NS, GL = 16 / RL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2 *z = reinterpret_cast<float2 *>(smem);
  const int tid = threadIdx.x;
  const int64_t channel = blockIdx.x / a.blocks_per_channel;
  const int64_t blk = blockIdx.x % a.blocks_per_channel;
  const float *x = a.x + channel * a.x_stride;
  float *y = a.y + channel * a.y_stride;
  const int64_t base = blk * a.step - a.lead;      // first sample of the window (even)
  c32 r[16];
  const bool inside = base >= 0 && base + 2 * M <= a.n;   // the whole window lies inside the stream (block-uniform)
  if (ALIGNED && inside) {
    const float2 *src = reinterpret_cast<const float2 *>(x + base) + tid;
#pragma unroll
    for (int m = 0; m < 16; ++m) {   // z[n] = (x[2n], x[2n+1]), n = tid + T m: lane-contiguous 8-byte reads
      const float2 v = src[T * m];
      r[m] = {v.x, v.y};
    }
  } else {
#pragma unroll
    for (int m = 0; m < 16; ++m) {   // the stream's first / last blocks: zeros outside the stream
      const int64_t g = base + 2 * (int64_t)(tid + T * m);
      r[m].x = (g >= 0 && g < a.n) ? x[g] : 0.0f;
      r[m].y = (g + 1 >= 0 && g + 1 < a.n) ? x[g + 1] : 0.0f;
    }
  }
  fft_passes<LOG2M, true, false, float, true>(r, z, tid, a.tw_m);
  __syncthreads();   // the last pass's reads of z are over
#pragma unroll
  for (int i = 0; i < GL; ++i)
#pragma unroll
    for (int j = 0; j < RL; ++j) {
      const c32 v = r[i * RL + j];
      z[swz(out_index<RL, NSL, T>(tid, i, j))] = make_float2(v.x, v.y);
    }
  __syncthreads();
  // pointwise stage: pairs (k, M - k), k = tid + T m < M/2 (k = 0 pairs with itself and carries bin M; thread 0 also takes k = M/2).
  // Every pair is read and written by one thread only, so the stage runs in place.
  auto pair = [&](int k) {
    const int kp = (M - k) & (M - 1);
    const float2 A = z[swz(k)], B = z[swz(kp)];
    const float2 w = a.tw[k];                               // exp(-2 pi i k / N), k <= M/2 < N/2
    const float2 hk = a.h_half[k], hp = a.h_half[M - k];
    const c32 E = {A.x + B.x, A.y - B.y}, D = {A.x - B.x, A.y + B.y};
    const c32 t = cmul(D, c32{w.x, w.y});                   // w D
    const c32 Xk = {E.x + t.y, E.y - t.x};                  // E - i w D          (= 2 X[k])
    const c32 Xp = {E.x - t.y, -(E.y + t.x)};               // conj(E + i w D)    (= 2 X[M-k])
    const c32 Yk = cmul(Xk, c32{hk.x, hk.y}), Yp = cmul(Xp, c32{hp.x, hp.y});
    const c32 P = {Yk.x + Yp.x, Yk.y - Yp.y}, Q = {Yk.x - Yp.x, Yk.y + Yp.y};
    const c32 u = cmul(Q, c32{w.x, -w.y});                  // conj(w) Q
    // Z'[k] = P + i conj(w) Q,  Z'[M-k] = conj(P - i conj(w) Q); the inverse runs as conj(FFT(conj .)): store the conjugates
    z[swz(k)] = make_float2(P.x - u.y, -(P.y + u.x));
    if (kp != k) z[swz(kp)] = make_float2(P.x + u.y, P.y - u.x);
  };
  int tp = tid;
  asm volatile("" : "+v"(tp));   // the pairs' addresses are formed here, not carried (spilled) from the top of the kernel
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    pair(tp + T * m);
    if (m == 3) __builtin_amdgcn_sched_barrier(0);   // four pairs in flight at a time: all eight (10 values each) spill
  }
  if (tid == 0) pair(M / 2);
  int ti = tid;
  asm volatile("" : "+v"(ti));   // the inverse re-derives its LDS addresses instead of carrying the forward transform's in (spilled) registers
  fft_passes<LOG2M, false, false, float, true>(r, z, ti, a.tw_m);
  const int64_t out0 = blk * a.step;
  int to = tid;
  asm volatile("" : "+v"(to));
  const bool whole = out0 + a.step <= a.n;   // every kept sample of this block exists (block-uniform)
#pragma unroll
  for (int i = 0; i < GL; ++i)
#pragma unroll
    for (int j = 0; j < RL; ++j) {
      const int64_t s = 2 * (int64_t)out_index<RL, NSL, T>(to, i, j) - a.lead;   // position inside the kept span (even)
      if (s >= 0 && s < a.step) {
        const int64_t o = out0 + s;
        const float re = r[i * RL + j].x, im = -r[i * RL + j].y;   // conj(FFT(conj .))
        if (ALIGNED && whole) {
          *reinterpret_cast<float2 *>(y + o) = make_float2(re, im);
        } else {
          if (o < a.n) y[o] = re;
          if (o + 1 < a.n) y[o + 1] = im;
        }
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
  {
    const int64_t M = N / 2;
    std::vector<float2> hh((size_t)M + 1), twm((size_t)(M / 2 > 0 ? M / 2 : 1));
    for (int64_t k = 0; k <= M; ++k) {   // position brev(k) of the DIF output holds H[k]
      const unsigned pos = smx::brev_host((unsigned)k, log2n);
      hh[(size_t)k] = make_float2((float)(re[pos] / (4.0 * (double)M)), (float)(im[pos] / (4.0 * (double)M)));
    }
    for (int64_t j = 0; j < M / 2; ++j) {
      const double ang = -2.0 * M_PI * (double)j / (double)M;
      twm[(size_t)j] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
    SMX_HIP_CHECK(hipMalloc((void **)&t.h_half, hh.size() * sizeof(float2)));
    SMX_HIP_CHECK(hipMemcpy(t.h_half, hh.data(), hh.size() * sizeof(float2), hipMemcpyHostToDevice));
    SMX_HIP_CHECK(hipMalloc((void **)&t.tw_m, twm.size() * sizeof(float2)));
    SMX_HIP_CHECK(hipMemcpy(t.tw_m, twm.data(), twm.size() * sizeof(float2), hipMemcpyHostToDevice));
  }
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
    (void)hipFree(kv.second.h_half);
    (void)hipFree(kv.second.tw_m);
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
  a.h_nat = t.h_nat;
  a.tw = t.tw;
  a.h_half = t.h_half;
  a.tw_m = t.tw_m;
  static const bool packed_env = [] { const char *e = std::getenv("SMX_FIR_PACKED"); return e && e[0] == '1'; }();
  const bool packed = packed_env && p.log2n <= 14;
  if (!packed) {   // one real block per workgroup, half-size transform
    a.lead = (p.taps - 1 + 1) & ~int64_t(1);                 // even, >= taps - 1
    a.step = p.nfft - a.lead;
    a.blocks_per_channel = (n + a.step - 1) / a.step;
    const int64_t grid = channels * a.blocks_per_channel;
    if (grid > 0x7fffffff) throw Failure("fir_apply: too many blocks for one launch");
    const size_t lds = (size_t)(p.nfft / 2) * sizeof(float2);
    const bool aligned = x_stride % 2 == 0 && y_stride % 2 == 0 && reinterpret_cast<uintptr_t>(d_x) % 8 == 0 &&
                         reinterpret_cast<uintptr_t>(d_y) % 8 == 0;
    auto launch = [&](auto kernel, int threads) {
      SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      SMX_LAUNCH(kernel, dim3((unsigned)grid), dim3(threads), lds, stream, a);
    };
    switch (p.log2n) {
      case 10: aligned ? launch(fir_ols_real_kernel<9, true>, 32) : launch(fir_ols_real_kernel<9, false>, 32); break;
      case 11: aligned ? launch(fir_ols_real_kernel<10, true>, 64) : launch(fir_ols_real_kernel<10, false>, 64); break;
      case 12: aligned ? launch(fir_ols_real_kernel<11, true>, 128) : launch(fir_ols_real_kernel<11, false>, 128); break;
      case 13: aligned ? launch(fir_ols_real_kernel<12, true>, 256) : launch(fir_ols_real_kernel<12, false>, 256); break;
      case 14: aligned ? launch(fir_ols_real_kernel<13, true>, 512) : launch(fir_ols_real_kernel<13, false>, 512); break;
      case 15: aligned ? launch(fir_ols_real_kernel<14, true>, 1024) : launch(fir_ols_real_kernel<14, false>, 1024); break;
      default: throw Failure("fir_apply: unsupported block size");
    }
    SMX_HIP_CHECK(hipGetLastError());
    return;
  }
  const int64_t blocks = (n + p.valid - 1) / p.valid;
  a.pairs_per_channel = (blocks + 1) / 2;
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
    if (taps > 16384)
      throw InvalidArgument(format(
          "fir_plan_create: cannot filter with %lld taps (this device path holds at most 16384 taps per "
          "32768-sample block)",
          (long long)taps));
    if (!h) throw Failure("fir_plan_create: null taps");
    auto *p = new smx_fir_plan();
    p->taps = taps;
    // block length: at least 4 x taps (75 % of every block is kept), at most 32768 real samples = 16384 complex points
    static const bool big = [] { const char *e = std::getenv("SMX_FIR_BIG"); return !(e && e[0] == '0'); }();
    int64_t n = 1024;
    while (n < 4 * taps && n < (big ? 32768 : 16384)) n *= 2;
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
