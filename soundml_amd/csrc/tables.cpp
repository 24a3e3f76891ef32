// Device-resident tables owned by the configuration handles: the analysis
// window, FFT twiddles and mel weights are built on the host in float64 (the
// reference's own precision for these tables, stft.ml:57-59 / mel.ml:31-33),
// rounded once where a kernel consumes float32, and uploaded lazily per HIP
// device.  Access is serialised by the handle's mutex.
#include <cmath>

#include "smx_internal.hpp"

namespace smx {
namespace {

template <typename T>
T *upload(const std::vector<T> &host) {
  T *dev = nullptr;
  SMX_HIP_CHECK(hipMalloc((void **)&dev, host.size() * sizeof(T) + 16));
  SMX_HIP_CHECK(hipMemcpy(dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  return dev;
}

bool is_pow2(int64_t n) { return n >= 1 && (n & (n - 1)) == 0; }

}  // namespace

// Scratch arrays (Griffin-Lim's spectra, the scratch spectrogram of the unfused compositions, small tables) come
// from the device's stream-ordered pool.  Left at its default the pool hands everything back to the driver at every
// synchronisation, and the next call pays the mapping again (measured: Griffin-Lim on the C2 batch 315 ms instead of
// 104).  Keep up to `bytes` of freed memory for reuse; the default, applied once per device, is 1/8 of its memory.
static std::mutex g_pool_mutex;
static std::map<int, bool> g_pool_done;
static int64_t g_pool_bytes = -1;   // < 0: the default

void set_scratch_retention(int64_t bytes) {
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  g_pool_bytes = bytes;
  g_pool_done.clear();
}

// The library's OWN stream-ordered pool, one per device (never the process-wide default pool, whose release threshold an
// embedding application may rely on): scratch arrays come from it and go back to it, and it keeps up to `threshold`
// bytes of freed scratch for the next call instead of returning them to the driver at every synchronisation.
static std::map<int, hipMemPool_t> g_pools;

static uint64_t pool_threshold_locked() {
  if (g_pool_bytes >= 0) return (uint64_t)g_pool_bytes;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
  const uint64_t eighth = (uint64_t)total_b / 8, cap = 16ull << 30;    // default: 1/8 of the device, at most 16 GiB
  return eighth < cap ? eighth : cap;
}

static hipMemPool_t device_pool_locked(int device) {
  auto it = g_pools.find(device);
  if (it == g_pools.end()) {
    hipMemPoolProps props = {};
    props.allocType = hipMemAllocationTypePinned;
    props.handleTypes = hipMemHandleTypeNone;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = device;
    hipMemPool_t pool = nullptr;
    if (hipMemPoolCreate(&pool, &props) != hipSuccess) pool = nullptr;   // fall back to plain stream-ordered allocation
    it = g_pools.emplace(device, pool).first;
    g_pool_done[device] = false;
  }
  if (it->second && !g_pool_done[device]) {
    uint64_t threshold = pool_threshold_locked();
    (void)hipMemPoolSetAttribute(it->second, hipMemPoolAttrReleaseThreshold, &threshold);
    g_pool_done[device] = true;
  }
  return it->second;
}

void init_device_pool() {
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return;
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  (void)device_pool_locked(device);
}

hipError_t pool_malloc_async(void **ptr, size_t bytes, hipStream_t stream) {
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  hipMemPool_t pool = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    pool = device_pool_locked(device);
  }
  return pool ? hipMallocFromPoolAsync(ptr, bytes, pool, stream) : hipMallocAsync(ptr, bytes, stream);
}

// Compute units of the current device: the size of every persistent grid.  Cached per device ordinal in atomics (rounds 1-5 kept
// one unsynchronised `static int` per launcher, filled by whichever device called first: a data race under concurrent host
// threads and the wrong grid after smx_set_device to a different part).
int device_cu_count() {
  static std::atomic<int> cached[64];
  int dev = 0;
  SMX_HIP_CHECK(hipGetDevice(&dev));
  const bool slot = dev >= 0 && dev < 64;
  if (slot) {
    const int c = cached[dev].load(std::memory_order_relaxed);
    if (c > 0) return c;
  }
  hipDeviceProp_t prop;
  SMX_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (slot) cached[dev].store(cus, std::memory_order_relaxed);
  return cus;
}

}  // namespace smx

const smx::StftTables &smx_stft_config::tables() const {
  smx::init_device_pool();
  int device = 0;
  SMX_HIP_CHECK(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(mutex_);
  auto it = tables_.find(device);
  if (it != tables_.end()) return it->second;

  smx::StftTables t;
  const int64_t n = fft_size;
  std::vector<float> w32((size_t)n);
  for (int64_t i = 0; i < n; ++i) w32[(size_t)i] = (float)analysis_window[(size_t)i];
  t.window_f64 = smx::upload(analysis_window);
  t.window_f32 = smx::upload(w32);

  // generic kernels: N twiddles exp(-2 pi i j / N) (the radix-2 passes read j < N/2,
  // the direct DFT all N)
  const int64_t tw = n;
  std::vector<double2> t64((size_t)tw);
  std::vector<float2> t32((size_t)tw);
  for (int64_t j = 0; j < tw; ++j) {
    const double a = -2.0 * M_PI * (double)j / (double)n;
    t64[(size_t)j] = make_double2(std::cos(a), std::sin(a));
    t32[(size_t)j] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  t.twiddle_f64 = smx::upload(t64);
  t.twiddle_f32 = smx::upload(t32);
  t.twiddle_len = tw;

  // chirp-z tables for sizes that are not powers of two (M = 256 .. 16384: N <= 8192)
  if (!smx::is_pow2(n) && n >= 2) {
    int log2m = 8;
    while ((int64_t(1) << log2m) < 2 * n - 1) ++log2m;
    if (log2m <= 14) {
      const int64_t m = int64_t(1) << log2m;
      auto chirp_angle = [&](int64_t i) {   // pi i^2 / N with i^2 reduced modulo 2 N (the chirp has that period)
        const int64_t r = (i * i) % (2 * n);
        return M_PI * (double)r / (double)n;
      };
      std::vector<float2> chirp((size_t)n), post((size_t)n), tw((size_t)(m / 2));
      for (int64_t i = 0; i < n; ++i) {
        const double a = chirp_angle(i), w = analysis_window[(size_t)i];
        chirp[(size_t)i] = make_float2((float)(w * std::cos(a)), (float)(-w * std::sin(a)));
        post[(size_t)i] = make_float2((float)std::cos(a), (float)(-std::sin(a)));
      }
      // filter b[j] = exp(+i pi j^2 / N) for |j| < N, wrapped into M points; its spectrum by an in-place
      // float64 radix-2 DIF transform, unscrambled, with the inverse transform's 1/M folded in
      std::vector<double> re((size_t)m, 0.0), im((size_t)m, 0.0);
      for (int64_t j = 0; j < n; ++j) {
        const double a = chirp_angle(j);
        re[(size_t)j] = std::cos(a);
        im[(size_t)j] = std::sin(a);
        if (j > 0) {
          re[(size_t)(m - j)] = std::cos(a);
          im[(size_t)(m - j)] = std::sin(a);
        }
      }
      for (int64_t half = m >> 1; half >= 1; half >>= 1) {
        const int64_t tstep = (m >> 1) / half;
        for (int64_t b = 0; b < (m >> 1); ++b) {
          const int64_t j = b & (half - 1);
          const int64_t i0 = ((b - j) << 1) + j, i1 = i0 + half;
          const double ang = -2.0 * M_PI * (double)(j * tstep) / (double)m;
          const double wr = std::cos(ang), wi = std::sin(ang);
          const double dr = re[(size_t)i0] - re[(size_t)i1], di = im[(size_t)i0] - im[(size_t)i1];
          re[(size_t)i0] += re[(size_t)i1];
          im[(size_t)i0] += im[(size_t)i1];
          re[(size_t)i1] = dr * wr - di * wi;
          im[(size_t)i1] = dr * wi + di * wr;
        }
      }
      std::vector<float2> filt((size_t)m);
      for (int64_t i = 0; i < m; ++i) {
        unsigned k = 0;
        for (int bit = 0; bit < log2m; ++bit) k |= ((unsigned)(i >> bit) & 1u) << (log2m - 1 - bit);   // position i holds B[brev(i)]
        filt[k] = make_float2((float)(re[(size_t)i] / (double)m), (float)(im[(size_t)i] / (double)m));
      }
      for (int64_t j = 0; j < m / 2; ++j) {
        const double a = -2.0 * M_PI * (double)j / (double)m;
        tw[(size_t)j] = make_float2((float)std::cos(a), (float)std::sin(a));
      }
      t.blu_chirp = smx::upload(chirp);
      t.blu_post = smx::upload(post);
      t.blu_filter = smx::upload(filt);
      t.blu_tw = smx::upload(tw);
      t.blu_log2m = log2m;
    }
    // even sizes: the frame as ONE chirp-z transform of length L = N/2 over its (even, odd) sample pairs, then the
    // real-input post-pass -- a quarter of the convolution work (fft 400: M = 512 instead of 1024)
    if (n % 2 == 0 && n >= 4) {
      const int64_t l = n / 2;
      int log2m2 = 8;
      while ((int64_t(1) << log2m2) < 2 * l - 1) ++log2m2;
      if (log2m2 <= 14) {
        const int64_t m = int64_t(1) << log2m2;
        auto angle = [&](int64_t i) { return M_PI * (double)((i * i) % (2 * l)) / (double)l; };
        std::vector<float2> chirp((size_t)l), tw((size_t)(m / 2));
        std::vector<float> hw((size_t)n);
        for (int64_t i = 0; i < n; ++i) hw[(size_t)i] = (float)(0.5 * analysis_window[(size_t)i]);
        for (int64_t i = 0; i < l; ++i) chirp[(size_t)i] = make_float2((float)std::cos(angle(i)), (float)(-std::sin(angle(i))));
        std::vector<double> re((size_t)m, 0.0), im((size_t)m, 0.0);
        for (int64_t j = 0; j < l; ++j) {
          re[(size_t)j] = std::cos(angle(j));
          im[(size_t)j] = std::sin(angle(j));
          if (j > 0) {
            re[(size_t)(m - j)] = re[(size_t)j];
            im[(size_t)(m - j)] = im[(size_t)j];
          }
        }
        for (int64_t half = m >> 1; half >= 1; half >>= 1) {   // in-place float64 radix-2 DIF, bit-reversed output
          const int64_t tstep = (m >> 1) / half;
          for (int64_t b = 0; b < (m >> 1); ++b) {
            const int64_t j = b & (half - 1);
            const int64_t i0 = ((b - j) << 1) + j, i1 = i0 + half;
            const double ang = -2.0 * M_PI * (double)(j * tstep) / (double)m;
            const double wr = std::cos(ang), wi = std::sin(ang);
            const double dr = re[(size_t)i0] - re[(size_t)i1], di = im[(size_t)i0] - im[(size_t)i1];
            re[(size_t)i0] += re[(size_t)i1];
            im[(size_t)i0] += im[(size_t)i1];
            re[(size_t)i1] = dr * wr - di * wi;
            im[(size_t)i1] = dr * wi + di * wr;
          }
        }
        std::vector<float2> filt((size_t)m);
        for (int64_t i = 0; i < m; ++i) {
          unsigned k = 0;
          for (int bit = 0; bit < log2m2; ++bit) k |= ((unsigned)(i >> bit) & 1u) << (log2m2 - 1 - bit);
          filt[k] = make_float2((float)(re[(size_t)i] / (double)m), (float)(im[(size_t)i] / (double)m));
        }
        for (int64_t j = 0; j < m / 2; ++j) {
          const double a = -2.0 * M_PI * (double)j / (double)m;
          tw[(size_t)j] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        t.blu2_chirp = smx::upload(chirp);
        t.blu2_filter = smx::upload(filt);
        t.blu2_tw = smx::upload(tw);
        t.blu2_window = smx::upload(hw);
        t.blu2_log2m = log2m2;
        // L = 2^a 3^b 5^c 7^d <= 1024 (and not a power of two: those have the Stockham kernels): radices 4, 2, 5, 3, 7
        if (l <= 1024 && (l & (l - 1)) != 0) {
          int64_t rest = l;
          int np = 0, radix[10];
          while (rest % 4 == 0 && np < 10) { radix[np++] = 4; rest /= 4; }
          while (rest % 2 == 0 && np < 10) { radix[np++] = 2; rest /= 2; }
          while (rest % 5 == 0 && np < 10) { radix[np++] = 5; rest /= 5; }
          while (rest % 3 == 0 && np < 10) { radix[np++] = 3; rest /= 3; }
          while (rest % 7 == 0 && np < 10) { radix[np++] = 7; rest /= 7; }
          if (rest == 1 && np > 0) {
            std::vector<float2> twl((size_t)l);
            for (int64_t j = 0; j < l; ++j) {
              const double a = -2.0 * M_PI * (double)j / (double)l;
              twl[(size_t)j] = make_float2((float)std::cos(a), (float)std::sin(a));
            }
            t.mixed_tw = smx::upload(twl);
            std::vector<double2> twd((size_t)l);
            for (int64_t j = 0; j < l; ++j) {
              const double a = -2.0 * M_PI * (double)j / (double)l;
              twd[(size_t)j] = make_double2(std::cos(a), std::sin(a));
            }
            t.mixed_tw_f64 = smx::upload(twd);
            t.mixed_npass = np;
            for (int i = 0; i < np; ++i) t.mixed_radix[i] = radix[i];
          }
        }
      }
    }
  }

  // odd sizes N = 3^b 5^c 7^d <= 1024 (441, 225, 375, 675, 945 ...): the same kernels on the frame itself as a complex signal of
  // N points (no half-size trick), plan for N, twiddles exp(-2 pi i j / N)
  if (n % 2 == 1 && n >= 3 && n <= 1024) {
    int64_t rest = n;
    int np = 0, radix[10];
    while (rest % 5 == 0 && np < 10) { radix[np++] = 5; rest /= 5; }
    while (rest % 3 == 0 && np < 10) { radix[np++] = 3; rest /= 3; }
    while (rest % 7 == 0 && np < 10) { radix[np++] = 7; rest /= 7; }
    if (rest == 1 && np > 0) {
      std::vector<float2> twl((size_t)n);
      std::vector<double2> twd((size_t)n);
      for (int64_t j = 0; j < n; ++j) {
        const double a = -2.0 * M_PI * (double)j / (double)n;
        twl[(size_t)j] = make_float2((float)std::cos(a), (float)std::sin(a));
        twd[(size_t)j] = make_double2(std::cos(a), std::sin(a));
      }
      t.mixed_tw = smx::upload(twl);
      t.mixed_tw_f64 = smx::upload(twd);
      t.mixed_npass = np;
      t.mixed_full = 1;
      for (int i = 0; i < np; ++i) t.mixed_radix[i] = radix[i];
    }
  }

  // small powers of two (N = 4 .. 256, below the Stockham frames kernels of istft.hip): the same plan, for the inverse only
  if (smx::is_pow2(n) && n >= 4 && n <= 256) {
    int64_t rest = n / 2;
    int np = 0;
    while (rest % 4 == 0 && np < 10) { t.mixed_radix[np++] = 4; rest /= 4; }
    while (rest % 2 == 0 && np < 10) { t.mixed_radix[np++] = 2; rest /= 2; }
    std::vector<float2> twl((size_t)(n / 2));
    for (int64_t j = 0; j < n / 2; ++j) {
      const double a = -2.0 * M_PI * (double)j / (double)(n / 2);
      twl[(size_t)j] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    t.mixed_tw = smx::upload(twl);
    std::vector<double2> twd((size_t)(n / 2));
    for (int64_t j = 0; j < n / 2; ++j) {
      const double a = -2.0 * M_PI * (double)j / (double)(n / 2);
      twd[(size_t)j] = make_double2(std::cos(a), std::sin(a));
    }
    t.mixed_tw_f64 = smx::upload(twd);
    t.mixed_npass = np;
  }

  // fast kernels (power-of-two N >= 64): half-scaled window and split tables
  if (smx::is_pow2(n) && n >= 64) {
    const int64_t m = n / 2;
    std::vector<float> hw((size_t)n);
    for (int64_t i = 0; i < n; ++i) hw[(size_t)i] = (float)(0.5 * analysis_window[(size_t)i]);
    std::vector<float2> wm((size_t)m), wn((size_t)m + 1);
    for (int64_t j = 0; j < m; ++j) {
      const double a = -2.0 * M_PI * (double)j / (double)m;
      wm[(size_t)j] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    for (int64_t k = 0; k <= m; ++k) {
      const double a = -2.0 * M_PI * (double)k / (double)n;
      wn[(size_t)k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    // synthesis: the window with the inverse transform's 1/(2M) and the conj(FFT(conj .)) sign folded in
    std::vector<float2> sw((size_t)m);
    for (int64_t j = 0; j < m; ++j)
      sw[(size_t)j] = make_float2((float)(analysis_window[(size_t)(2 * j)] / (double)(2 * m)),
                                  (float)(-analysis_window[(size_t)(2 * j + 1)] / (double)(2 * m)));
    if (n >= 512 && n <= 4096) {
      std::vector<double2> wm64((size_t)m);
      for (int64_t j = 0; j < m; ++j) {
        const double a = -2.0 * M_PI * (double)j / (double)m;
        wm64[(size_t)j] = make_double2(std::cos(a), std::sin(a));
      }
      t.fast_w_m_f64 = smx::upload(wm64);
      std::vector<double2> sw64((size_t)m);
      for (int64_t j = 0; j < m; ++j)
        sw64[(size_t)j] = make_double2(analysis_window[(size_t)(2 * j)] / (double)(2 * m), -analysis_window[(size_t)(2 * j + 1)] / (double)(2 * m));
      t.fast_synth_window_f64 = smx::upload(sw64);
    }
    t.fast_window = smx::upload(hw);
    t.fast_w_m = smx::upload(wm);
    t.fast_w_n = smx::upload(wn);
    t.fast_synth_window = smx::upload(sw);
  }
  return tables_.emplace(device, t).first->second;
}

smx::EnvelopeTable smx_stft_config::envelope(int64_t count) const {
  int device = 0;
  SMX_HIP_CHECK(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(mutex_);
  const auto key = std::make_pair(device, count);
  auto it = envelopes_.find(key);
  if (it != envelopes_.end()) return it->second;
  if (envelopes_.size() >= 64) {   // a caller cycling through many lengths: drop the oldest table.  The cache only gives up
    auto oldest = envelopes_.begin();   // ITS share: a caller that took the descriptor before keeps the device copy alive
    for (auto jt = envelopes_.begin(); jt != envelopes_.end(); ++jt)   // until its launch is enqueued (EnvelopeTable::owner)
      if (jt->second.serial < oldest->second.serial) oldest = jt;
    envelopes_.erase(oldest);
  }
  std::vector<double> head, period, tail;
  smx::EnvelopeTable e;
  smx::stft_envelope(*this, count, head, period, tail, e.head_n, e.stop);
  std::vector<double> packed;
  packed.reserve(head.size() + 2 * period.size() + tail.size() + 1);
  packed.insert(packed.end(), head.begin(), head.end());
  packed.insert(packed.end(), period.begin(), period.end());
  packed.insert(packed.end(), tail.begin(), tail.end());
  for (double v : period) packed.push_back(1.0 / v);   // the periodic part's reciprocals, behind the three pieces (the fused kernels multiply)
  packed.push_back(1.0);
  e.head = head.size();
  e.period = period.size();
  e.tail = tail.size();
  e.dev = smx::upload(packed);
  e.owner = std::shared_ptr<void>(e.dev, [](void *p) { (void)hipFree(p); });
  e.serial = ++envelope_serial_;
  return envelopes_.emplace(key, e).first->second;
}

smx_stft_config::~smx_stft_config() {
  envelopes_.clear();   // the cache's shares (the device copies go with their last holder)
  for (auto &kv : tables_) {
    smx::StftTables &t = kv.second;
    (void)hipFree(t.window_f64);
    (void)hipFree(t.window_f32);
    (void)hipFree(t.twiddle_f64);
    (void)hipFree(t.twiddle_f32);
    (void)hipFree(t.fast_window);
    (void)hipFree(t.fast_w_m);
    (void)hipFree(t.fast_w_m_f64);
    (void)hipFree(t.fast_synth_window_f64);
    (void)hipFree(t.fast_w_n);
    (void)hipFree(t.fast_synth_window);
    (void)hipFree(t.blu_chirp);
    (void)hipFree(t.blu_post);
    (void)hipFree(t.blu_filter);
    (void)hipFree(t.blu_tw);
    (void)hipFree(t.blu2_chirp);
    (void)hipFree(t.blu2_filter);
    (void)hipFree(t.blu2_tw);
    (void)hipFree(t.blu2_window);
    (void)hipFree(t.mixed_tw);
    (void)hipFree(t.mixed_tw_f64);
  }
}

const smx_mel_config::Tables &smx_mel_config::tables() const {
  smx::init_device_pool();
  int device = 0;
  SMX_HIP_CHECK(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(mutex_);
  auto it = tables_.find(device);
  if (it != tables_.end()) return it->second;

  Tables t;
  const int64_t nb = bins();
  t.w_f64 = smx::upload(weights);
  // MFMA operand image: rows padded to a multiple of 32 mel bands, K (bins)
  // padded to a multiple of 32, zero filled, float32.
  t.n_mels_pad = (n_mels + 31) / 32 * 32;
  t.k_pad = (nb + 31) / 32 * 32;
  std::vector<float> w32((size_t)(t.n_mels_pad * t.k_pad), 0.0f);
  std::vector<int> lo((size_t)t.n_mels_pad, 0), hi((size_t)t.n_mels_pad, 0);
  for (int64_t m = 0; m < n_mels; ++m) {
    int first = -1, last = -1;
    for (int64_t b = 0; b < nb; ++b) {
      const double w = weights[(size_t)(m * nb + b)];
      w32[(size_t)(m * t.k_pad + b)] = (float)w;
      if (w != 0.0) {
        if (first < 0) first = (int)b;
        last = (int)b;
      }
    }
    lo[(size_t)m] = first < 0 ? 0 : first;
    hi[(size_t)m] = last < 0 ? 0 : last + 1;
  }
  t.w_f32 = smx::upload(w32);
  {
    // the same weights in MFMA-operand order for the 16-frame kernels' tail: per 16-row tile and 4-bin k-step the 64
    // lane values of v_mfma_f32_16x16x4_f32's A operand (lane l: row l & 15, bin l >> 4) lie contiguously, so a wave's
    // load is one 256-byte run instead of sixteen 16-byte pieces of sixteen rows
    const int64_t tiles = t.n_mels_pad / 16, steps = t.k_pad / 4;
    std::vector<float> wt((size_t)(tiles * steps * 64), 0.0f);
    for (int64_t tt = 0; tt < tiles; ++tt)
      for (int64_t k4 = 0; k4 < steps; ++k4)
        for (int l = 0; l < 64; ++l)
          wt[(size_t)((tt * steps + k4) * 64 + l)] = w32[(size_t)((16 * tt + (l & 15)) * t.k_pad + 4 * k4 + (l >> 4))];
    t.w_tile = smx::upload(wt);
  }
  {
    // and for Mel.apply's v_mfma_f32_32x32x2_f32 (lane l: row l & 31, bin l >> 5): per 32-row block and 2-bin k-step the
    // 64 lane values contiguous, plus each block's band
    const int64_t blocks = t.n_mels_pad / 32, steps = t.k_pad / 2;
    std::vector<float> wb((size_t)(blocks * steps * 64), 0.0f);
    for (int64_t bb = 0; bb < blocks; ++bb)
      for (int64_t k2 = 0; k2 < steps; ++k2)
        for (int l = 0; l < 64; ++l)
          wb[(size_t)((bb * steps + k2) * 64 + l)] = w32[(size_t)((32 * bb + (l & 31)) * t.k_pad + 2 * k2 + (l >> 5))];
    t.w_block = smx::upload(wb);
    std::vector<int> blo((size_t)blocks, 0), bhi((size_t)blocks, 0);
    for (int64_t bb = 0; bb < blocks; ++bb) {
      int l = 0x7fffffff, h = 0;
      for (int64_t r = 32 * bb; r < 32 * bb + 32; ++r)
        if (hi[(size_t)r] > lo[(size_t)r]) {
          l = std::min(l, lo[(size_t)r]);
          h = std::max(h, hi[(size_t)r]);
        }
      if (h > 0) {
        blo[(size_t)bb] = l;
        bhi[(size_t)bb] = h;
      }
    }
    t.block_lo = smx::upload(blo);
    t.block_hi = smx::upload(bhi);
  }
  t.band_lo = smx::upload(lo);
  t.band_hi = smx::upload(hi);
  {
    const int64_t tiles = t.n_mels_pad / 16;
    std::vector<int> tlo((size_t)tiles, 0), thi((size_t)tiles, 0);
    for (int64_t tt = 0; tt < tiles; ++tt) {
      int l = 0x7fffffff, h = 0;
      for (int64_t r = 16 * tt; r < 16 * tt + 16; ++r)
        if (hi[(size_t)r] > lo[(size_t)r]) {
          l = std::min(l, lo[(size_t)r]);
          h = std::max(h, hi[(size_t)r]);
        }
      if (h > 0) {
        tlo[(size_t)tt] = l;
        thi[(size_t)tt] = h;
      }
    }
    t.tile_lo = smx::upload(tlo);
    t.tile_hi = smx::upload(thi);
  }
  return tables_.emplace(device, t).first->second;
}

smx_mel_config::~smx_mel_config() {
  for (auto &kv : tables_) {
    (void)hipFree(kv.second.w_f64);
    (void)hipFree(kv.second.w_f32);
    (void)hipFree(kv.second.band_lo);
    (void)hipFree(kv.second.band_hi);
    (void)hipFree(kv.second.tile_lo);
    (void)hipFree(kv.second.w_tile);
    (void)hipFree(kv.second.w_block);
    (void)hipFree(kv.second.block_lo);
    (void)hipFree(kv.second.block_hi);
    (void)hipFree(kv.second.tile_hi);
  }
  for (auto &kv : fused32_) {
    (void)hipFree(kv.second.items);
    (void)hipFree(kv.second.w_mfma);
  }
  for (auto &kv : fused4_) {
    (void)hipFree(kv.second.items);
    (void)hipFree(kv.second.w_mfma);
  }
}

const double *smx_chroma_config::device_weights() const {
  smx::init_device_pool();
  int device = 0;
  SMX_HIP_CHECK(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(mutex_);
  auto it = tables_.find(device);
  if (it != tables_.end()) return it->second;
  // transposed [bins; rows_pad] (rows padded with zeros to a multiple of the kernel's chunk of 12): the projection
  // kernel reads one bin's column of weights with scalar loads
  const int64_t nb = bins(), rows_pad = (n_chroma + 11) / 12 * 12;
  std::vector<double> wt((size_t)(nb * rows_pad), 0.0);
  for (int64_t c = 0; c < n_chroma; ++c)
    for (int64_t j = 0; j < nb; ++j) wt[(size_t)(j * rows_pad + c)] = weights[(size_t)(c * nb + j)];
  double *dev = smx::upload(wt);
  tables_.emplace(device, dev);
  return dev;
}

smx_chroma_config::~smx_chroma_config() {
  for (auto &kv : tables_) (void)hipFree(kv.second);
}
