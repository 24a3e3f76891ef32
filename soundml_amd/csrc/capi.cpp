// extern "C" entry points of include/soundml_amd.h.  Every function validates
// its arguments (raising the reference's Invalid_argument messages as
// SMX_INVALID_ARGUMENT) before any pointer is formed or device work enqueued,
// mirroring the discipline of the reference's stub layer
// (resample_stubs.c:228-276).  There is no CPU compute path here: the host
// entry points upload, launch the HIP kernels and download.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

#include <algorithm>

#include "smx_internal.hpp"

namespace smx {

static thread_local std::string g_last_error;
static std::atomic<int> g_interior{SMX_INTERIOR_F32};

void set_last_error(const std::string &message) { g_last_error = message; }

std::atomic<unsigned long long> g_kernel_launches{0};

int env_flag(const char *name) {
  const char *e = std::getenv(name);
  if (!e) return -1;
  if (e[0] == '\0' || !std::strcmp(e, "0") || !std::strcmp(e, "false") || !std::strcmp(e, "off")) return 0;
  return 1;
}
long env_int(const char *name, long fallback) {
  const char *e = std::getenv(name);
  return e && e[0] ? std::atol(e) : fallback;
}

bool fast_path_disabled() { return env_flag("SMX_DISABLE_FAST") == 1; }

void launch_stft(const StftJob &job) {
  if (job.count <= 0 || job.lead <= 0) return;
  if (launch_stft_fast(job)) return;
  launch_stft_generic(job);
}

namespace {

template <typename F>
int guarded(F &&body) {
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

void require_device() {
  int count = 0;
  hipError_t err = hipGetDeviceCount(&count);
  if (err != hipSuccess || count < 1)
    throw Failure("soundml_amd: no HIP device is available (this library has no CPU fallback)");
}

struct DeviceScratch {  // RAII device allocation for the host-pointer entry points, from the stream-ordered pool:
  void *ptr = nullptr;  // hipFree of a GB-sized array costs more than the kernels; the pool keeps it for the next call
  explicit DeviceScratch(size_t bytes) {
    if (bytes == 0) bytes = 16;
    init_device_pool();
    SMX_HIP_CHECK(smx::pool_malloc_async(&ptr, bytes, nullptr));
  }
  ~DeviceScratch() { (void)hipFreeAsync(ptr, nullptr); }
  DeviceScratch(const DeviceScratch &) = delete;
  DeviceScratch &operator=(const DeviceScratch &) = delete;
};

// ---- the device list of the host-pointer batch calls (smx_set_devices; round 6) ----------------------------------------------
// The reference's caller is ONE process handing over host tensors, "a batch of clips is one call" (stft.mli:211-250); its leading
// axes are independent by contract (stft.mli:214-218, tested per slice: stft_grid.ml:180-205).  With a device list set, a
// host-pointer batch call splits `lead` into contiguous clip ranges by the rule of soundml_amd/shard.py clip_range (the first
// lead % S shards own one clip more) and runs each range on its device from a host thread of its own: its own HIP streams, its own
// staging rings (transfer.cpp: per device), its own PCIe link; tables are built lazily per device as before.  No collective: every
// shard writes its slice of the caller's result.  A device may be listed more than once (virtual shards: two uploads of one device
// in flight at a time).
std::mutex g_devices_mutex;
std::vector<int> g_devices;   // empty: the calling thread's current device (smx_set_device), as rounds 1-5

std::vector<int> device_list() {
  std::lock_guard<std::mutex> g(g_devices_mutex);
  return g_devices;
}

// clips [lo, hi) of shard `rank` of `world` (soundml_amd/shard.py: clip_range)
void clip_range(int64_t total, int64_t world, int64_t rank, int64_t &lo, int64_t &hi) {
  const int64_t base = total / world, extra = total % world;
  lo = rank * base + std::min(rank, extra);
  hi = lo + base + (rank < extra ? 1 : 0);
}

// body(clip0, nclips) for every shard, on the shard's device; the first failure is rethrown as it was thrown (InvalidArgument
// stays InvalidArgument).  Without a device list: body(0, lead) on the caller's thread and device.
template <class F>
void for_each_shard(int64_t lead, F &&body) {
  const std::vector<int> devices = device_list();
  if (devices.empty() || lead <= 0) {
    body((int64_t)0, lead);
    return;
  }
  const int64_t shards = std::min<int64_t>((int64_t)devices.size(), lead);
  int caller_device = 0;
  SMX_HIP_CHECK(hipGetDevice(&caller_device));
  std::vector<std::exception_ptr> errors((size_t)shards);
  // the shards start their transfers together (a thread's first HIP call can take milliseconds: without the rendezvous the first
  // shard's upload was over before the last one's began, and the link sat idle in between)
  std::mutex gate_mutex;
  std::condition_variable gate_cv;
  int64_t arrived = 0;
  auto rendezvous = [&] {
    std::unique_lock<std::mutex> g(gate_mutex);
    if (++arrived == shards) gate_cv.notify_all();
    else gate_cv.wait(g, [&] { return arrived == shards; });
  };
  auto run = [&](int64_t s) {
    bool met = false;
    try {
      const hipError_t set = hipSetDevice(devices[(size_t)s]);
      if (set == hipSuccess) (void)hipFree(nullptr);   // (the thread's HIP state for the device exists before the rendezvous)
      met = true;
      rendezvous();
      SMX_HIP_CHECK(set);
      set_transfer_share((int)shards);
      int64_t lo, hi;
      clip_range(lead, shards, s, lo, hi);
      if (hi > lo) body(lo, hi - lo);
    } catch (...) {
      errors[(size_t)s] = std::current_exception();
      if (!met) rendezvous();   // (nobody waits for a shard that failed early)
    }
    set_transfer_share(1);
  };
  if (shards == 1) {
    run(0);
    (void)hipSetDevice(caller_device);
  } else {
    std::vector<std::thread> threads;
    for (int64_t s = 1; s < shards; ++s) threads.emplace_back(run, s);
    run(0);   // the caller's thread takes the first shard
    for (auto &t : threads) t.join();
    (void)hipSetDevice(caller_device);
  }
  for (auto &e : errors)
    if (e) std::rethrow_exception(e);
}

void check_config(const void *c, const char *fn) {
  if (!c) throw Failure(format("%s: configuration handle is null", fn));
}

void check_rank_extents(const char *fn, int64_t lead, int64_t n) {
  if (lead < 0 || n < 0)
    throw Failure(format("%s: negative extent (lead %lld, n %lld)", fn, (long long)lead, (long long)n));
}

void check_range(const smx_stft_config &c, int64_t n, int64_t p0, int64_t p1) {
  const int64_t total = c.frames(n);
  if (p0 < 0 || p0 > p1 || p1 > total)  // stft.ml:655-661
    throw InvalidArgument(format(
        "transform_range: cannot take frames [%lld, %lld) of a %lld-frame transform (the range must "
        "satisfy 0 <= p0 <= p1 <= frames)",
        (long long)p0, (long long)p1, (long long)total));
}

// device-resident analysis of frames [p0, p1)
void stft_range_dev(const smx_stft_config &c, const void *d_x, int in_bytes, int64_t lead, int64_t n,
                    int64_t x_stride, int64_t p0, int64_t p1, OutMode mode, double power, void *d_out,
                    hipStream_t stream) {
  check_rank_extents("transform_range", lead, n);
  if (x_stride < n) throw Failure("transform_range: x_stride is smaller than the signal length");
  check_range(c, n, p0, p1);
  if (p0 == p1 || lead == 0) return;  // frameless_spectrum: nothing to write (stft.ml:629-630)
  if (!d_x || !d_out) throw Failure("transform_range: null device pointer");
  StftJob job;
  job.cfg = &c;
  job.x = d_x;
  job.in_bytes = in_bytes;
  job.interior = in_bytes == 8 ? SMX_INTERIOR_F64 : g_interior.load();
  job.lead = lead;
  job.n = n;
  job.x_stride = x_stride;
  job.left = c.left_width();
  job.pad = c.pad;
  job.pad_value = c.pad_value;
  job.p0 = p0;
  job.count = p1 - p0;
  job.mode = mode;
  job.power = power;
  job.out = d_out;
  job.out_stride = p1 - p0;
  job.out_offset = 0;
  job.stream = stream;
  launch_stft(job);
}

// host-pointer analysis on the current device: upload, run, download
void stft_range_host_one(const smx_stft_config &c, const void *x, int in_bytes, int64_t lead, int64_t n,
                         int64_t p0, int64_t p1, OutMode mode, double power, void *out) {
  check_rank_extents("transform", lead, n);
  check_range(c, n, p0, p1);
  const int64_t count = p1 - p0;
  const size_t out_elems = (size_t)lead * (size_t)c.bins() * (size_t)count * (mode == OUT_COMPLEX ? 2 : 1);
  if (count == 0 || lead == 0) return;
  if (!x || !out) throw Failure("transform: null pointer");
  require_device();
  static const bool trace = env_flag("SMX_HOST_TRACE") == 1;   // diagnostic: where a host call's time goes
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  DeviceScratch dx((size_t)lead * (size_t)n * (size_t)in_bytes);
  DeviceScratch dout(out_elems * (size_t)in_bytes);
  const double t1 = now();
  // A large batch is cut into units of clips whose upload, kernels and download overlap (transfer.cpp; a unit's result is the
  // slice of the whole call's bit for bit: the reference's per-slice law, stft_grid.ml:180-205).  SMX_HOST_PIPELINE=0: serially.
  const size_t in_clip = (size_t)n * (size_t)in_bytes, out_clip = out_elems / (size_t)lead * (size_t)in_bytes;
  // (without page-locked staging memory the pipelined form cannot run: the serial path's plain hipMemcpy still completes the call)
  if (lead >= 8 && (size_t)lead * (in_clip + out_clip) >= ((size_t)128 << 20) && in_clip > 0 && env_flag("SMX_HOST_PIPELINE") != 0 &&
      staging_available()) {
    int64_t unit = (int64_t)(((size_t)48 << 20) / std::max(in_clip, out_clip));   // ~48 MB of the larger side per unit
    unit = std::max<int64_t>(1, std::min<int64_t>(unit, (lead + 3) / 4));
    SMX_HIP_CHECK(hipStreamSynchronize(nullptr));   // the scratch arrays come from the null stream's pool
    (void)c.tables();                              // (lazy tables are built before the threads start)
    pipelined_host_call(x, in_clip, out, out_clip, lead, unit, dx.ptr, dout.ptr, [&](int64_t clip0, int64_t nc, hipStream_t stream) {
      stft_range_dev(c, reinterpret_cast<const unsigned char *>(dx.ptr) + (size_t)clip0 * in_clip, in_bytes, nc, n, n, p0, p1, mode, power,
                     reinterpret_cast<unsigned char *>(dout.ptr) + (size_t)clip0 * out_clip, stream);
    });
    if (trace) fprintf(stderr, "[smx] host transform: allocate %.2f ms, pipelined upload / kernels / download %.2f (units of %lld clips)\n", t1 - t0, now() - t1, (long long)unit);
    return;
  }
  copy_to_device(dx.ptr, x, (size_t)lead * (size_t)n * (size_t)in_bytes);
  const double t2 = now();
  stft_range_dev(c, dx.ptr, in_bytes, lead, n, n, p0, p1, mode, power, dout.ptr, nullptr);
  SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
  const double t3 = now();
  copy_to_host(out, dout.ptr, out_elems * (size_t)in_bytes);
  if (trace)
    fprintf(stderr, "[smx] host transform: allocate %.2f ms, upload %.2f, kernels %.2f, download %.2f\n", t1 - t0, t2 - t1,
            t3 - t2, now() - t3);
}

// host-pointer analysis: the whole batch on the current device, or -- with a device list (smx_set_devices) -- contiguous clip
// ranges on the listed devices side by side (a range's result is the slice of the whole call's bit for bit: stft_grid.ml:180-205)
void stft_range_host(const smx_stft_config &c, const void *x, int in_bytes, int64_t lead, int64_t n,
                     int64_t p0, int64_t p1, OutMode mode, double power, void *out) {
  check_rank_extents("transform", lead, n);
  check_range(c, n, p0, p1);
  if (p1 == p0 || lead == 0) return;
  if (!x || !out) throw Failure("transform: null pointer");
  require_device();
  const size_t in_clip = (size_t)n * (size_t)in_bytes;
  const size_t out_clip = (size_t)c.bins() * (size_t)(p1 - p0) * (mode == OUT_COMPLEX ? 2 : 1) * (size_t)in_bytes;
  for_each_shard(lead, [&](int64_t clip0, int64_t nc) {
    stft_range_host_one(c, reinterpret_cast<const unsigned char *>(x) + (size_t)clip0 * in_clip, in_bytes, nc, n, p0, p1, mode, power,
                        reinterpret_cast<unsigned char *>(out) + (size_t)clip0 * out_clip);
  });
}

// Stft.invert's checks (stft.ml:745-786), in the reference's order and wording
void check_synthesis(const smx_stft_config &c, int64_t bins, int64_t length, bool has_length) {
  if (bins != c.bins())
    throw InvalidArgument(format(
        "invert: cannot invert %lld frequency bins of a %lld-point transform (the bin axis must hold "
        "fft_size / 2 + 1 = %lld values)",
        (long long)bins, (long long)c.fft_size, (long long)c.bins()));
  if (has_length && length < 0)
    throw InvalidArgument(format("invert: cannot synthesise a signal of length %lld (length must be non-negative)",
                                 (long long)length));
  if (!stft_nola(c))
    throw InvalidArgument(format(
        "invert: cannot invert a %lld-point window advanced by %lld samples inside a %lld-point frame (the "
        "overlap-added squared window must stay above 1e-10 of its largest value at every position)",
        (long long)c.win_length, (long long)c.hop, (long long)c.fft_size));
}

// length < 0 means "not given" at the ABI only after the explicit has_length flag said so
void invert_dev(const smx_stft_config &c, const void *d_z, int z_bytes, int64_t lead, int64_t bins, int64_t frames,
                int has_length, int64_t length, void *d_out, hipStream_t stream) {
  if (lead < 0 || frames < 0) throw Failure("invert: negative extent");
  check_synthesis(c, bins, length, has_length != 0);
  const int64_t out_len = has_length ? length : stft_output_length(c, frames);
  if (lead == 0 || out_len == 0) return;
  if (!d_out || (frames > 0 && !d_z)) throw Failure("invert: null device pointer");
  IstftJob job;
  job.cfg = &c;
  job.z = d_z;
  job.z_bytes = z_bytes;
  job.interior = z_bytes == 16 ? SMX_INTERIOR_F64 : g_interior.load();
  job.lead = lead;
  job.frames = frames;
  // only the frames the output can reach are inverted (stft.ml:915-921)
  const int64_t left = c.left_width();
  job.count = has_length ? std::min<int64_t>(frames, (length + left + c.hop - 1) / c.hop) : frames;
  job.out_len = out_len;
  job.out = d_out;
  job.stream = stream;
  launch_istft(job);
}

void invert_host_one(const smx_stft_config &c, const void *z, int z_bytes, int64_t lead, int64_t bins, int64_t frames,
                     int has_length, int64_t length, void *out) {
  if (lead < 0 || frames < 0) throw Failure("invert: negative extent");
  check_synthesis(c, bins, length, has_length != 0);
  const int64_t out_len = has_length ? length : stft_output_length(c, frames);
  if (lead == 0 || out_len == 0) return;
  if (!out || (frames > 0 && !z)) throw Failure("invert: null pointer");
  require_device();
  const size_t zb = (size_t)lead * (size_t)bins * (size_t)frames * (size_t)z_bytes;
  const size_t ob = (size_t)lead * (size_t)out_len * (size_t)(z_bytes / 2);
  DeviceScratch dz(zb), dout(ob);
  // a large batch in units of clips whose upload, synthesis and download overlap (as stft_range_host; every clip is synthesised
  // on its own: stft.mli:214-218)
  const size_t z_clip = (size_t)bins * (size_t)frames * (size_t)z_bytes, o_clip = (size_t)out_len * (size_t)(z_bytes / 2);
  if (lead >= 8 && zb + ob >= ((size_t)128 << 20) && z_clip > 0 && o_clip > 0 && env_flag("SMX_HOST_PIPELINE") != 0 && staging_available()) {
    int64_t unit = (int64_t)(((size_t)48 << 20) / std::max(z_clip, o_clip));
    unit = std::max<int64_t>(1, std::min<int64_t>(unit, (lead + 3) / 4));
    SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
    (void)c.tables();
    pipelined_host_call(z, z_clip, out, o_clip, lead, unit, dz.ptr, dout.ptr, [&](int64_t clip0, int64_t nc, hipStream_t stream) {
      invert_dev(c, reinterpret_cast<const unsigned char *>(dz.ptr) + (size_t)clip0 * z_clip, z_bytes, nc, bins, frames, has_length, length,
                 reinterpret_cast<unsigned char *>(dout.ptr) + (size_t)clip0 * o_clip, stream);
    });
    return;
  }
  if (zb) copy_to_device(dz.ptr, z, zb);
  invert_dev(c, dz.ptr, z_bytes, lead, bins, frames, has_length, length, dout.ptr, nullptr);
  SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
  copy_to_host(out, dout.ptr, ob);
}

// host-pointer synthesis: the current device, or the device list's clip ranges side by side (every clip is synthesised on its
// own: stft.mli:214-218)
void invert_host(const smx_stft_config &c, const void *z, int z_bytes, int64_t lead, int64_t bins, int64_t frames,
                 int has_length, int64_t length, void *out) {
  if (lead < 0 || frames < 0) throw Failure("invert: negative extent");
  check_synthesis(c, bins, length, has_length != 0);
  const int64_t out_len = has_length ? length : stft_output_length(c, frames);
  if (lead == 0 || out_len == 0) return;
  if (!out || (frames > 0 && !z)) throw Failure("invert: null pointer");
  require_device();
  const size_t z_clip = (size_t)bins * (size_t)frames * (size_t)z_bytes, o_clip = (size_t)out_len * (size_t)(z_bytes / 2);
  for_each_shard(lead, [&](int64_t clip0, int64_t nc) {
    invert_host_one(c, z ? reinterpret_cast<const unsigned char *>(z) + (size_t)clip0 * z_clip : nullptr, z_bytes, nc, bins, frames, has_length,
                    length, reinterpret_cast<unsigned char *>(out) + (size_t)clip0 * o_clip);
  });
}

}  // namespace
}  // namespace smx

using namespace smx;

// =============================== library =====================================
extern "C" {

const char *smx_last_error(void) { return g_last_error.c_str(); }
int smx_version(void) { return 100; }
unsigned long long smx_debug_kernel_launches(void) { return g_kernel_launches.load(); }

int smx_device_count(int *count) {
  return guarded([&] {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    if (count) *count = n;
  });
}

int smx_set_device(int device) {
  return guarded([&] { SMX_HIP_CHECK(hipSetDevice(device)); });
}
int smx_set_devices(const int *devices, int n) {
  return guarded([&] {
    if (n < 0 || n > 1024) throw Failure(format("smx_set_devices: cannot list %d devices", n));
    if (n > 0 && !devices) throw Failure("smx_set_devices: null device list");
    int count = 0;
    if (n > 0 && (hipGetDeviceCount(&count) != hipSuccess || count < 1))
      throw Failure("soundml_amd: no HIP device is available (this library has no CPU fallback)");
    for (int i = 0; i < n; ++i)
      if (devices[i] < 0 || devices[i] >= count)
        throw Failure(format("smx_set_devices: device %d is not one of the %d visible devices", devices[i], count));
    std::lock_guard<std::mutex> g(g_devices_mutex);
    g_devices.assign(devices, devices + n);
  });
}
int smx_get_devices(int *devices, int capacity, int *n) {
  return guarded([&] {
    if (!n) throw Failure("smx_get_devices: null count pointer");
    const std::vector<int> d = device_list();
    *n = (int)d.size();
    if (devices)
      for (int i = 0; i < (int)d.size() && i < capacity; ++i) devices[i] = d[(size_t)i];
  });
}
int smx_shard_clip_range(int64_t total_clips, int64_t shards, int64_t shard, int64_t *lo, int64_t *hi) {
  return guarded([&] {
    if (!lo || !hi) throw Failure("smx_shard_clip_range: null result pointer");
    if (total_clips < 0) throw Failure("smx_shard_clip_range: negative clip count");
    if (shards < 1 || shard < 0 || shard >= shards)
      throw Failure(format("smx_shard_clip_range: shard %lld outside a list of %lld", (long long)shard, (long long)shards));
    clip_range(total_clips, shards, shard, *lo, *hi);
  });
}
int smx_debug_staging_peak(int *uploads, int *downloads, int reset) {
  return guarded([&] { staging_peak(uploads, downloads, reset != 0); });
}

int smx_set_scratch_retention(int64_t bytes) {
  return guarded([&] {
    if (bytes < -1) throw Failure("set_scratch_retention: bytes must be >= 0, or -1 for the default");
    set_scratch_retention(bytes);
  });
}

int smx_set_interior(int interior) {
  return guarded([&] {
    if (interior != SMX_INTERIOR_F32 && interior != SMX_INTERIOR_F64)
      throw InvalidArgument(format("set_interior: unknown interior %d", interior));
    g_interior.store(interior);
  });
}
int smx_get_interior(void) { return g_interior.load(); }

int smx_synchronize(void *stream) {
  return guarded([&] { SMX_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream)); });
}

// diagnostics (tests): Stft.transform of device-resident float32 audio into Griffin-Lim's internal frame-major layout
int smx_debug_stft_transform_frame_major_f32_dev(const smx_stft_config *c, const float *d_x, int64_t lead, int64_t n, float *d_out,
                                                 int64_t pitch_floats, int64_t rows_per_clip, void *stream) {
  return guarded([&] {
    check_config(c, "transform");
    StftJob job;
    job.cfg = c;
    job.x = d_x;
    job.in_bytes = 4;
    job.interior = g_interior.load();
    job.lead = lead;
    job.n = n;
    job.x_stride = n;
    job.left = c->left_width();
    job.pad = c->pad;
    job.pad_value = c->pad_value;
    job.p0 = 0;
    job.count = c->frames(n);
    job.mode = OUT_COMPLEX;
    job.power = 0.0;
    job.out = d_out;
    job.out_stride = job.count;
    job.out_offset = 0;
    job.stream = (hipStream_t)stream;
    if (!launch_stft_complex_fm(job, d_out, pitch_floats, rows_per_clip)) throw Failure("transform (frame-major): not eligible");
  });
}

int smx_host_alloc(size_t bytes, void **ptr) {
  return guarded([&] {
    if (!ptr) throw Failure("smx_host_alloc: null result pointer");
    *ptr = nullptr;
    require_device();
    *ptr = host_alloc(bytes);
  });
}
int smx_host_free(void *ptr) {
  return guarded([&] { host_free(ptr); });
}

// =============================== Window ======================================
int smx_window_make(int kind, int periodic, int64_t n, double *out) {
  return guarded([&] {
    if (n >= 1 && !out) throw Failure("make: null output");
    window_make(kind, periodic != 0, n, out);
  });
}

int smx_window_make_param(int kind, double param, int periodic, int64_t n, double *out) {
  return guarded([&] {
    if (n >= 1 && !out) throw Failure("make: null output");
    window_make_param(kind, param, periodic != 0, n, out);
  });
}

int smx_window_cola(int kind, double param, int64_t length, int64_t hop, int *cola) {
  return guarded([&] {
    if (!cola) throw Failure("cola: null output");
    *cola = window_cola(kind, param, length, hop) ? 1 : 0;
  });
}

// Convert.hz_to_mel / mel_to_hz (convert.ml:70-102): the scalar maps the mel breakpoints are built from, on host values
int smx_hz_to_mel(int scale, const double *f, int64_t n, double *out) {
  return guarded([&] {
    if (scale != SMX_MEL_SLANEY && scale != SMX_MEL_HTK) throw InvalidArgument(format("hz_to_mel: unknown mel scale %d", scale));
    if (n > 0 && (!f || !out)) throw Failure("hz_to_mel: null pointer");
    for (int64_t i = 0; i < n; ++i) out[i] = hz_to_mel(f[i], scale);
  });
}
int smx_mel_to_hz(int scale, const double *m, int64_t n, double *out) {
  return guarded([&] {
    if (scale != SMX_MEL_SLANEY && scale != SMX_MEL_HTK) throw InvalidArgument(format("mel_to_hz: unknown mel scale %d", scale));
    if (n > 0 && (!m || !out)) throw Failure("mel_to_hz: null pointer");
    for (int64_t i = 0; i < n; ++i) out[i] = mel_to_hz(m[i], scale);
  });
}

// =============================== Stft.Config =================================
int smx_stft_config_create(int64_t fft_size, int64_t win_length, int64_t hop, int alignment, int pad,
                           double pad_value, int scale, int window_kind, const double *custom_window,
                           smx_stft_config **out) {
  return guarded([&] {
    if (!out) throw Failure("create: null output handle");
    *out = stft_config_create(fft_size, win_length, hop, alignment, pad, pad_value, scale, window_kind,
                              custom_window);
  });
}
void smx_stft_config_destroy(smx_stft_config *c) { delete c; }
int64_t smx_stft_config_fft_size(const smx_stft_config *c) { return c ? c->fft_size : -1; }
int64_t smx_stft_config_hop(const smx_stft_config *c) { return c ? c->hop : -1; }
int64_t smx_stft_config_win_length(const smx_stft_config *c) { return c ? c->win_length : -1; }
int64_t smx_stft_config_bins(const smx_stft_config *c) { return c ? c->bins() : -1; }
int64_t smx_stft_config_left_width(const smx_stft_config *c) { return c ? c->left_width() : -1; }
int64_t smx_stft_config_right_width(const smx_stft_config *c) { return c ? c->right_width() : -1; }
int64_t smx_stft_config_latency(const smx_stft_config *c) {
  return c ? (c->alignment == SMX_ALIGN_CENTERED ? c->fft_size / 2 : 0) : -1;
}
int smx_stft_config_analysis_window(const smx_stft_config *c, double *out) {
  return guarded([&] {
    check_config(c, "analysis_window");
    std::memcpy(out, c->analysis_window.data(), c->analysis_window.size() * sizeof(double));
  });
}

// =============================== frame grid ==================================
int smx_stft_frames(const smx_stft_config *c, int64_t n, int64_t *out) {
  return guarded([&] {
    check_config(c, "frames");
    *out = c->frames(n);
  });
}
int smx_stft_first_complete(const smx_stft_config *c, int64_t *out) {
  return guarded([&] {
    check_config(c, "first_complete");
    *out = stft_first_complete(*c);
  });
}
int smx_stft_last_complete(const smx_stft_config *c, int64_t n, int64_t *out) {
  return guarded([&] {
    check_config(c, "last_complete");
    *out = stft_last_complete(*c, n);
  });
}
static void check_sample_rate(const char *op, int64_t sample_rate) {  // stft.ml:237-243
  if (sample_rate < 1)
    throw InvalidArgument(format(
        "%s: cannot use a sample rate of %lld Hz (sample_rate must be at least 1)", op,
        (long long)sample_rate));
}
int smx_stft_times(const smx_stft_config *c, int64_t sample_rate, int64_t n, double *out) {
  return guarded([&] {
    check_config(c, "times");
    check_sample_rate("times", sample_rate);
    if (n < 0)
      throw InvalidArgument(format(
          "times: cannot analyse a signal of length %lld (length must be non-negative)", (long long)n));
    const int64_t count = c->frames(n);
    for (int64_t p = 0; p < count; ++p)  // stft.ml:245-254: p*hop exact, one rounding
      out[p] = ((double)p * (double)c->hop) / (double)sample_rate;
  });
}
int smx_stft_frequencies(const smx_stft_config *c, int64_t sample_rate, double *out) {
  return guarded([&] {
    check_config(c, "frequencies");
    check_sample_rate("frequencies", sample_rate);
    const double step = (double)sample_rate / (double)c->fft_size;  // stft.ml:256-261
    for (int64_t k = 0; k < c->bins(); ++k) out[k] = (double)k * step;
  });
}

// =============================== transforms ==================================
int smx_stft_transform_f32(const smx_stft_config *c, const float *x, int64_t lead, int64_t n,
                           float *out) {
  return guarded([&] {
    check_config(c, "transform");
    check_rank_extents("transform", lead, n);
    stft_range_host(*c, x, 4, lead, n, 0, c->frames(n), OUT_COMPLEX, 0.0, out);
  });
}
int smx_stft_transform_f64(const smx_stft_config *c, const double *x, int64_t lead, int64_t n,
                           double *out) {
  return guarded([&] {
    check_config(c, "transform");
    check_rank_extents("transform", lead, n);
    stft_range_host(*c, x, 8, lead, n, 0, c->frames(n), OUT_COMPLEX, 0.0, out);
  });
}
int smx_stft_transform_range_f32(const smx_stft_config *c, const float *x, int64_t lead, int64_t n,
                                 int64_t p0, int64_t p1, float *out) {
  return guarded([&] {
    check_config(c, "transform_range");
    stft_range_host(*c, x, 4, lead, n, p0, p1, OUT_COMPLEX, 0.0, out);
  });
}
int smx_stft_transform_range_f64(const smx_stft_config *c, const double *x, int64_t lead, int64_t n,
                                 int64_t p0, int64_t p1, double *out) {
  return guarded([&] {
    check_config(c, "transform_range");
    stft_range_host(*c, x, 8, lead, n, p0, p1, OUT_COMPLEX, 0.0, out);
  });
}
int smx_stft_power_spectrum_f32(const smx_stft_config *c, const float *x, int64_t lead, int64_t n,
                                double power, float *out) {
  return guarded([&] {
    check_config(c, "power_spectrum");
    check_rank_extents("power_spectrum", lead, n);
    stft_range_host(*c, x, 4, lead, n, 0, c->frames(n), OUT_POWER, power, out);
  });
}
int smx_stft_power_spectrum_f64(const smx_stft_config *c, const double *x, int64_t lead, int64_t n,
                                double power, double *out) {
  return guarded([&] {
    check_config(c, "power_spectrum");
    check_rank_extents("power_spectrum", lead, n);
    stft_range_host(*c, x, 8, lead, n, 0, c->frames(n), OUT_POWER, power, out);
  });
}

// |frames [p0, p1)|^power from host memory: what an nx-tensor caller of a ranged power face gets (the OCaml stub's
// mode 1 honours its range through these instead of writing every frame)
int smx_stft_power_range_f32(const smx_stft_config *c, const float *x, int64_t lead, int64_t n, int64_t p0, int64_t p1,
                             double power, float *out) {
  return guarded([&] {
    check_config(c, "transform_range");
    stft_range_host(*c, x, 4, lead, n, p0, p1, OUT_POWER, power, out);
  });
}
int smx_stft_power_range_f64(const smx_stft_config *c, const double *x, int64_t lead, int64_t n, int64_t p0, int64_t p1,
                             double power, double *out) {
  return guarded([&] {
    check_config(c, "transform_range");
    stft_range_host(*c, x, 8, lead, n, p0, p1, OUT_POWER, power, out);
  });
}

int smx_stft_transform_range_f32_dev(const smx_stft_config *c, const float *d_x, int64_t lead,
                                     int64_t n, int64_t x_stride, int64_t p0, int64_t p1,
                                     float *d_out, void *stream) {
  return guarded([&] {
    check_config(c, "transform_range");
    stft_range_dev(*c, d_x, 4, lead, n, x_stride, p0, p1, OUT_COMPLEX, 0.0, d_out, (hipStream_t)stream);
  });
}
int smx_stft_transform_range_f64_dev(const smx_stft_config *c, const double *d_x, int64_t lead,
                                     int64_t n, int64_t x_stride, int64_t p0, int64_t p1,
                                     double *d_out, void *stream) {
  return guarded([&] {
    check_config(c, "transform_range");
    stft_range_dev(*c, d_x, 8, lead, n, x_stride, p0, p1, OUT_COMPLEX, 0.0, d_out, (hipStream_t)stream);
  });
}
int smx_stft_power_range_f32_dev(const smx_stft_config *c, const float *d_x, int64_t lead, int64_t n,
                                 int64_t x_stride, int64_t p0, int64_t p1, double power, float *d_out,
                                 void *stream) {
  return guarded([&] {
    check_config(c, "power_spectrum");
    stft_range_dev(*c, d_x, 4, lead, n, x_stride, p0, p1, OUT_POWER, power, d_out, (hipStream_t)stream);
  });
}
int smx_stft_power_range_f64_dev(const smx_stft_config *c, const double *d_x, int64_t lead, int64_t n,
                                 int64_t x_stride, int64_t p0, int64_t p1, double power,
                                 double *d_out, void *stream) {
  return guarded([&] {
    check_config(c, "power_spectrum");
    stft_range_dev(*c, d_x, 8, lead, n, x_stride, p0, p1, OUT_POWER, power, d_out, (hipStream_t)stream);
  });
}

}  // extern "C"

// ---- Stft.Synthesis (stft.ml:1029-1298): incremental least-squares synthesis ---------------------------------------------
// The reference keeps the last blocks - 1 WINDOWED frames; here the state is the last blocks - 1 SPECTRA (device,
// [channels; bins; blocks - 1], frames fastest) and a step runs the offline synthesis kernels over [history ++ new frames]
// with the envelope of the whole stream (IstftJob::env_q0 / env_open): a frame is inverted by the code that inverts it
// offline and a position sums its taps in the offline order, so any chunking of a stream totals Stft.invert bit for bit
// (the reference's law, soundml/test/istft/istft_law.ml) at the price of re-inverting blocks - 1 frames per step.
// The held quotients (`carry`, stft.ml:1063) and the head-trim counter are as in the reference; absent frames before the
// stream are zero spectra, whose frames are the exact zeros the reference's overlap-add pads with.
struct smx_stft_synthesis {
  const smx_stft_config *cfg = nullptr;
  int z_bytes = 8;                     // 8 = complex64 in / float32 out, 16 = complex128 / float64
  int64_t channels = 0, max_block = 0;
  int64_t left = 0, hold = 0, blocks = 0;
  int64_t fed = 0, drop = 0;
  bool drained = false;
  void *d_hist = nullptr;              // [channels; bins; blocks - 1] complex
  void *d_local = nullptr;             // [channels; bins; blocks - 1 + cap] complex
  void *d_quot = nullptr;              // [channels; cap * hop] real
  void *d_carry = nullptr;             // [channels; hold] real (valid when carry_len == hold)
  void *d_stage = nullptr;             // host-pointer steps: the chunk and the released samples
  int64_t cap = 0, carry_len = 0, stage_bytes = 0;
  hipStream_t last_stream = nullptr;   // the stream of the latest step / flush: reset orders its memset behind it
  int64_t elem() const { return z_bytes / 2; }
  ~smx_stft_synthesis() {
    (void)hipFree(d_hist); (void)hipFree(d_local); (void)hipFree(d_quot); (void)hipFree(d_carry); (void)hipFree(d_stage);
  }
};

// =============================== Stft.Kernel ==================================
// Streaming analysis (stft.ml:366-622).  The state machine is the reference's:
// a prelude until the left extension is computable, the pending padded suffix,
// the last right+1 raw samples, and `skip` when hop > fft.  The carry (pending
// suffix, tail) lives in DEVICE memory; each step uploads only the new chunk and
// runs the same analysis kernels over [pending ++ extension ++ chunk].
struct smx_stft_kernel {
  const smx_stft_config *cfg = nullptr;
  int dtype_bytes = 4;
  int64_t channels = 0, max_block = 0;
  int64_t left = 0, right = 0;
  bool started = false, drained = false;
  int64_t received = 0;
  int64_t skip = 0;
  // the small carried pieces live on the device too (prelude and tail are at most left + 1 / right + 1 samples per
  // channel), so a step on a device-resident chunk moves nothing over the host link
  void *d_prelude = nullptr;            // [channels][prelude_cap], prelude_len valid
  int64_t prelude_len = 0, prelude_cap = 0;
  void *d_tail = nullptr;               // [channels][right + 1], tail_len valid
  int64_t tail_len = 0;
  void *d_stream = nullptr;             // device [channels][cap]: pending ++ new padded samples
  int64_t cap = 0;
  int64_t pending_len = 0;
  void *d_out = nullptr;
  int64_t out_cap = 0;                  // frames
  // what a step emits: the complex spectrum (Stft.Kernel / Stft.stage) or |.|^power of it in the chunk's dtype
  // (Stft.power_stage, stft.ml:1364-1409)
  OutMode mode = OUT_COMPLEX;
  double power = 2.0;
  int64_t values() const { return mode == OUT_COMPLEX ? 2 : 1; }   // scalars per emitted element
  ~smx_stft_kernel() {
    (void)hipFree(d_stream);
    (void)hipFree(d_out);
    (void)hipFree(d_prelude);
    (void)hipFree(d_tail);
  }
};

namespace smx {
namespace {

int64_t frame_bound(const smx_stft_config &c, int64_t max_block) {  // stft.ml:1316-1317 + :1307-1308
  int64_t lat = c.alignment == SMX_ALIGN_CENTERED ? c.fft_size / 2 : 0;
  if (c.pad == SMX_PAD_REFLECT && c.left_width() > lat) lat = c.left_width();
  return (max_block + lat + c.hop - 1) / c.hop + 1;
}

// (the copy into the grown buffer runs on the caller's stream, behind the previous step's asynchronous move of the pending
// suffix; the old buffer is released only once that copy has run)
void ensure_stream(smx_stft_kernel &k, int64_t need, hipStream_t stream) {
  if (need <= k.cap) return;
  int64_t cap = k.cap ? k.cap : 1024;
  while (cap < need) cap *= 2;
  void *fresh = nullptr;
  const size_t es = (size_t)k.dtype_bytes;
  SMX_HIP_CHECK(hipMalloc(&fresh, (size_t)k.channels * (size_t)cap * es));
  if (k.d_stream && k.pending_len > 0)
    SMX_HIP_CHECK(hipMemcpy2DAsync(fresh, (size_t)cap * es, k.d_stream, (size_t)k.cap * es,
                                   (size_t)k.pending_len * es, (size_t)k.channels, hipMemcpyDeviceToDevice, stream));
  SMX_HIP_CHECK(hipStreamSynchronize(stream));
  (void)hipFree(k.d_stream);
  k.d_stream = fresh;
  k.cap = cap;
}

// One row-wise copy between device arrays: len elements per channel from src[ch][src_off ..] to dst[ch][dst_off ..]
// (strides in elements).  Ranges of one array may not overlap.
void rows_copy(void *dst, int64_t dst_stride, int64_t dst_off, const void *src, int64_t src_stride, int64_t src_off, int64_t len,
               int64_t channels, size_t es, hipStream_t stream) {
  if (len <= 0 || channels <= 0) return;
  SMX_HIP_CHECK(hipMemcpy2DAsync(reinterpret_cast<unsigned char *>(dst) + (size_t)dst_off * es, (size_t)dst_stride * es,
                                 reinterpret_cast<const unsigned char *>(src) + (size_t)src_off * es, (size_t)src_stride * es,
                                 (size_t)len * es, (size_t)channels, hipMemcpyDeviceToDevice, stream));
}

// The three border gathers of the state machine, on the device (kind 0: the left extension at install, stft.ml:457-471
// `left_pad`: reflect reads x[left - j], edge x[0]; kind 1: `pad_signal` of a stream that never reached the install
// threshold, stft.ml:318-338; kind 2: the right extension at the drain, stft.ml:506-519 `right_pad`: reflect reads
// tail[len - 2 - i], edge tail[len - 1]); constant padding writes the value.
template <typename T>
__global__ void __launch_bounds__(256) stream_pad_kernel(T *dst, int64_t dst_stride, int64_t count, const T *src, int64_t src_stride,
                                                         int64_t src_len, int kind, int pad, T value, int64_t left) {
  const int64_t ch = blockIdx.y;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < count; j += (int64_t)gridDim.x * 256) {
    int64_t idx = -1;
    if (pad != SMX_PAD_CONSTANT) {
      if (kind == 0) idx = pad == SMX_PAD_REFLECT ? left - j : 0;
      else if (kind == 2) idx = pad == SMX_PAD_REFLECT ? src_len - 2 - j : src_len - 1;
      else {
        const int64_t q = j - left;
        if (q >= 0 && q < src_len) idx = q;
        else if (pad == SMX_PAD_REFLECT) {
          if (src_len == 1) idx = 0;
          else {
            const int64_t period = 2 * (src_len - 1);
            const int64_t m = ((q % period) + period) % period;
            idx = m < src_len ? m : period - m;
          }
        } else idx = q < 0 ? 0 : src_len - 1;
      }
    } else if (kind == 1) {
      const int64_t q = j - left;
      if (q >= 0 && q < src_len) idx = q;
    }
    dst[ch * dst_stride + j] = idx < 0 ? value : src[ch * src_stride + idx];
  }
}

void stream_pad(smx_stft_kernel &k, void *dst, int64_t dst_stride, int64_t count, const void *src, int64_t src_stride, int64_t src_len,
                int kind, hipStream_t stream) {
  if (count <= 0) return;
  if (k.channels > 65535) throw Failure("step: more than 65535 channels in one stream");
  dim3 grid((unsigned)std::min<int64_t>((count + 255) / 256, 256), (unsigned)k.channels);
  if (k.dtype_bytes == 4)
    SMX_LAUNCH(stream_pad_kernel<float>, grid, dim3(256), 0, stream, reinterpret_cast<float *>(dst), dst_stride, count,
               reinterpret_cast<const float *>(src), src_stride, src_len, kind, k.cfg->pad, (float)k.cfg->pad_value, k.left);
  else
    SMX_LAUNCH(stream_pad_kernel<double>, grid, dim3(256), 0, stream, reinterpret_cast<double *>(dst), dst_stride, count,
               reinterpret_cast<const double *>(src), src_stride, src_len, kind, k.cfg->pad, (double)k.cfg->pad_value, k.left);
  SMX_HIP_CHECK(hipGetLastError());
}

// where a call's frames go: the caller's window [channels; bins; capacity], in host or device memory
struct StreamOut {
  void *ptr;
  int64_t capacity;
  bool on_device;
};

// stft.ml:415-442 `process`: append `extra` (device rows, extra_stride apart, from element extra_off) to the pending padded
// stream and emit every frame that became complete.
int64_t process(smx_stft_kernel &k, const void *extra, int64_t extra_stride, int64_t extra_off, int64_t extra_len, const StreamOut &out,
                hipStream_t stream) {
  const smx_stft_config &c = *k.cfg;
  const int64_t fft = c.fft_size, hop = c.hop;
  const size_t es = (size_t)k.dtype_bytes;
  const int64_t total = k.pending_len + extra_len;
  ensure_stream(k, total, stream);
  rows_copy(k.d_stream, k.cap, k.pending_len, extra, extra_stride, extra_off, extra_len, k.channels, es, stream);
  const int64_t count = total < fft ? 0 : 1 + (total - fft) / hop;
  if (count == 0) {
    k.pending_len = total;
    return 0;
  }
  if (count > out.capacity)
    throw Failure(format("step: %lld frames do not fit the caller's %lld-frame output window",
                         (long long)count, (long long)out.capacity));
  const int64_t bins = c.bins();
  if (count > k.out_cap) {
    SMX_HIP_CHECK(hipStreamSynchronize(stream));
    (void)hipFree(k.d_out);
    k.d_out = nullptr;
    int64_t cap = k.out_cap ? k.out_cap : 16;
    while (cap < count) cap *= 2;
    SMX_HIP_CHECK(hipMalloc(&k.d_out, (size_t)k.channels * (size_t)bins * (size_t)cap * 2 * es));   // sized for either face
    k.out_cap = cap;
  }
  StftJob job;
  job.cfg = &c;
  job.x = k.d_stream;
  job.in_bytes = k.dtype_bytes;
  job.interior = k.dtype_bytes == 8 ? SMX_INTERIOR_F64 : smx_get_interior();
  job.lead = k.channels;
  job.n = total;
  job.x_stride = k.cap;
  job.left = 0;                 // the stream is already padded
  job.pad = SMX_PAD_CONSTANT;
  job.pad_value = 0.0;
  job.p0 = 0;
  job.count = count;
  job.mode = k.mode;
  job.power = k.power;
  job.out = k.d_out;
  job.out_stride = count;
  job.out_offset = 0;
  job.stream = stream;
  launch_stft(job);
  // [channels; bins; count] (dense) -> caller's [channels; bins; capacity] window
  const size_t vs = (size_t)k.values() * es;
  if (out.on_device) {
    SMX_HIP_CHECK(hipMemcpy2DAsync(out.ptr, (size_t)out.capacity * vs, k.d_out, (size_t)count * vs, (size_t)count * vs,
                                   (size_t)(k.channels * bins), hipMemcpyDeviceToDevice, stream));
  } else {
    SMX_HIP_CHECK(hipStreamSynchronize(stream));
    SMX_HIP_CHECK(hipMemcpy2D(out.ptr, (size_t)out.capacity * vs, k.d_out, (size_t)count * vs, (size_t)count * vs,
                              (size_t)(k.channels * bins), hipMemcpyDeviceToHost));
  }
  const int64_t next_start = count * hop;
  if (next_start >= total) {
    k.skip += next_start - total;
    k.pending_len = 0;
  } else {
    const int64_t keep = total - next_start;
    // move the suffix to the front (rows are independent; overlapping ranges -> staged copy)
    void *tmp = nullptr;
    SMX_HIP_CHECK(smx::pool_malloc_async(&tmp, (size_t)k.channels * (size_t)keep * es, stream));
    rows_copy(tmp, keep, 0, k.d_stream, k.cap, next_start, keep, k.channels, es, stream);
    rows_copy(k.d_stream, k.cap, 0, tmp, keep, 0, keep, k.channels, es, stream);
    SMX_HIP_CHECK(hipFreeAsync(tmp, stream));
    k.pending_len = keep;
  }
  return count;
}

int64_t install_threshold(const smx_stft_config &c) {  // stft.ml:447-452
  return c.pad == SMX_PAD_REFLECT ? c.left_width() + 1 : 1;
}

// the last `right + 1` samples of [old tail ++ chunk] (stft.ml:492-502 `update_tail`; stft.ml:480-483 at install)
void update_tail(smx_stft_kernel &k, const void *chunk, int64_t chunk_stride, int64_t m, hipStream_t stream) {
  const size_t es = (size_t)k.dtype_bytes;
  const int64_t keep = k.right + 1;
  if (!k.d_tail) SMX_HIP_CHECK(hipMalloc(&k.d_tail, (size_t)k.channels * (size_t)keep * es));
  if (m >= keep) {
    rows_copy(k.d_tail, keep, 0, chunk, chunk_stride, m - keep, keep, k.channels, es, stream);
    k.tail_len = keep;
    return;
  }
  const int64_t cm = k.tail_len + m, start = cm - keep > 0 ? cm - keep : 0, kept_old = k.tail_len - start;
  if (start > 0 && kept_old > 0) {   // shift the surviving part of the old tail to the front (staged: the ranges overlap)
    void *tmp = nullptr;
    SMX_HIP_CHECK(smx::pool_malloc_async(&tmp, (size_t)k.channels * (size_t)kept_old * es, stream));
    rows_copy(tmp, kept_old, 0, k.d_tail, keep, start, kept_old, k.channels, es, stream);
    rows_copy(k.d_tail, keep, 0, tmp, kept_old, 0, kept_old, k.channels, es, stream);
    SMX_HIP_CHECK(hipFreeAsync(tmp, stream));
  }
  const int64_t old = kept_old > 0 ? kept_old : 0;
  // (start >= tail_len cannot happen: then m >= keep)
  rows_copy(k.d_tail, keep, old, chunk, chunk_stride, 0, m, k.channels, es, stream);
  k.tail_len = old + m;
}

// stft.ml:476-488 `install`: x = [prelude ++ chunk] (device rows, x_stride apart), n samples
int64_t install(smx_stft_kernel &k, const void *x, int64_t x_stride, int64_t n, const StreamOut &out, hipStream_t stream) {
  const size_t es = (size_t)k.dtype_bytes;
  if (k.right > 0) {
    k.tail_len = 0;
    const int64_t keep = k.right + 1 < n ? k.right + 1 : n;
    if (!k.d_tail) SMX_HIP_CHECK(hipMalloc(&k.d_tail, (size_t)k.channels * (size_t)(k.right + 1) * es));
    rows_copy(k.d_tail, k.right + 1, 0, x, x_stride, n - keep, keep, k.channels, es, stream);
    k.tail_len = keep;
  }
  k.started = true;
  k.prelude_len = 0;
  if (k.left == 0) return process(k, x, x_stride, 0, n, out, stream);
  void *both = nullptr;   // [left extension ++ x]
  const int64_t bl = k.left + n;
  SMX_HIP_CHECK(smx::pool_malloc_async(&both, (size_t)k.channels * (size_t)bl * es, stream));
  stream_pad(k, both, bl, k.left, x, x_stride, n, 0, stream);
  rows_copy(both, bl, k.left, x, x_stride, 0, n, k.channels, es, stream);
  const int64_t emitted = process(k, both, bl, 0, bl, out, stream);
  SMX_HIP_CHECK(hipFreeAsync(both, stream));
  return emitted;
}

// Kernel.step on a device-resident chunk [channels][m] (rows chunk_stride apart): stft.ml:521-559
int64_t step_core(smx_stft_kernel &k, const void *chunk, int64_t chunk_stride, int64_t m, const StreamOut &out, hipStream_t stream) {
  if (k.drained)
    throw InvalidArgument("step: cannot feed a drained kernel (flush consumed the tail; reset before reusing)");
  if (m < 0) throw Failure("step: negative chunk length");
  if (m == 0) return 0;
  if (!chunk) throw Failure("step: null chunk");
  const size_t es = (size_t)k.dtype_bytes;
  k.received += m;
  if (!k.started) {
    const int64_t n = k.prelude_len + m;
    if (k.received >= install_threshold(*k.cfg)) {
      if (k.prelude_len == 0) return install(k, chunk, chunk_stride, m, out, stream);
      void *x = nullptr;
      SMX_HIP_CHECK(smx::pool_malloc_async(&x, (size_t)k.channels * (size_t)n * es, stream));
      rows_copy(x, n, 0, k.d_prelude, k.prelude_cap, 0, k.prelude_len, k.channels, es, stream);
      rows_copy(x, n, k.prelude_len, chunk, chunk_stride, 0, m, k.channels, es, stream);
      const int64_t emitted = install(k, x, n, n, out, stream);
      SMX_HIP_CHECK(hipFreeAsync(x, stream));
      return emitted;
    }
    if (n > k.prelude_cap) {   // below the install threshold: at most left samples ever wait here
      const int64_t cap = std::max<int64_t>(n, k.left + 1);
      void *fresh = nullptr;
      SMX_HIP_CHECK(hipMalloc(&fresh, (size_t)k.channels * (size_t)cap * es));
      rows_copy(fresh, cap, 0, k.d_prelude, k.prelude_cap, 0, k.prelude_len, k.channels, es, stream);
      SMX_HIP_CHECK(hipStreamSynchronize(stream));
      (void)hipFree(k.d_prelude);
      k.d_prelude = fresh;
      k.prelude_cap = cap;
    }
    rows_copy(k.d_prelude, k.prelude_cap, k.prelude_len, chunk, chunk_stride, 0, m, k.channels, es, stream);
    k.prelude_len = n;
    return 0;
  }
  if (k.right > 0) update_tail(k, chunk, chunk_stride, m, stream);
  if (k.skip >= m) {
    k.skip -= m;
    return 0;
  }
  const int64_t dropped = k.skip;
  k.skip = 0;
  return process(k, chunk, chunk_stride, dropped, m - dropped, out, stream);
}

// Kernel.flush: stft.ml:561-595
int64_t flush_core(smx_stft_kernel &k, const StreamOut &out, hipStream_t stream) {
  if (k.drained) return 0;
  k.drained = true;
  const size_t es = (size_t)k.dtype_bytes;
  int64_t emitted = 0;
  if (!k.started) {
    if (k.received != 0) {   // the whole stream sits in the prelude: pad_signal, stft.ml:318-338
      const int64_t n = k.prelude_len, pl = k.left + n + k.right;
      void *padded = nullptr;
      SMX_HIP_CHECK(smx::pool_malloc_async(&padded, (size_t)k.channels * (size_t)pl * es, stream));
      stream_pad(k, padded, pl, pl, k.d_prelude, k.prelude_cap, n, 1, stream);
      k.started = true;
      k.prelude_len = 0;
      emitted = process(k, padded, pl, 0, pl, out, stream);
      SMX_HIP_CHECK(hipFreeAsync(padded, stream));
    }
  } else if (k.right > 0) {
    const int64_t r = k.right;
    if (k.skip >= r) {
      k.skip -= r;
    } else {
      void *rp = nullptr;   // stft.ml:506-519 right_pad
      SMX_HIP_CHECK(smx::pool_malloc_async(&rp, (size_t)k.channels * (size_t)r * es, stream));
      stream_pad(k, rp, r, r, k.d_tail, k.right + 1, k.tail_len, 2, stream);
      const int64_t dropped = k.skip;
      k.skip = 0;
      emitted = process(k, rp, r, dropped, r - dropped, out, stream);
      SMX_HIP_CHECK(hipFreeAsync(rp, stream));
    }
  }
  k.pending_len = 0;
  return emitted;
}

}  // namespace
}  // namespace smx

extern "C" {

int smx_stft_kernel_prepare(const smx_stft_config *c, int dtype_bytes, int64_t channels,
                            int64_t max_block, smx_stft_kernel **out) {
  return guarded([&] {
    check_config(c, "prepare");
    if (channels < 1)  // stft.ml:604-608
      throw InvalidArgument(format(
          "prepare: cannot analyse %lld channels (channels must be at least 1)", (long long)channels));
    if (max_block < 1)  // stft.ml:609-614
      throw InvalidArgument(format(
          "prepare: cannot accept blocks of %lld samples (max_block must be at least 1)",
          (long long)max_block));
    if (dtype_bytes != 4 && dtype_bytes != 8) throw Failure("prepare: dtype_bytes must be 4 or 8");
    if (!out) throw Failure("prepare: null output handle");
    require_device();
    auto k = std::make_unique<smx_stft_kernel>();
    k->cfg = c;
    k->dtype_bytes = dtype_bytes;
    k->channels = channels;
    k->max_block = max_block;
    k->left = c->left_width();
    k->right = c->right_width();
    *out = k.release();
  });
}

int smx_stft_kernel_prepare_power(const smx_stft_config *c, int dtype_bytes, int64_t channels, int64_t max_block,
                                  double power, smx_stft_kernel **out) {
  const int status = smx_stft_kernel_prepare(c, dtype_bytes, channels, max_block, out);
  if (status == SMX_OK) {
    (*out)->mode = OUT_POWER;
    (*out)->power = power;
  }
  return status;
}

int64_t smx_stft_stage_latency(const smx_stft_config *c) {   // max (Config.latency c) (install_threshold c - 1)
  if (!c) return -1;
  int64_t lat = c->alignment == SMX_ALIGN_CENTERED ? c->fft_size / 2 : 0;
  if (c->pad == SMX_PAD_REFLECT && c->left_width() > lat) lat = c->left_width();
  return lat;
}
int64_t smx_stft_frame_bound(const smx_stft_config *c, int64_t max_items) { return c ? frame_bound(*c, max_items) : -1; }

void smx_stft_kernel_destroy(smx_stft_kernel *k) { delete k; }

int smx_stft_kernel_frame_bound(const smx_stft_kernel *k, int64_t *out) {
  return guarded([&] {
    if (!k) throw Failure("frame_bound: null kernel");
    *out = frame_bound(*k->cfg, k->max_block);
  });
}

int smx_stft_kernel_channels(const smx_stft_kernel *k, int64_t *out) {
  return guarded([&] {
    if (!k || !out) throw Failure("channels: null argument");
    *out = k->channels;
  });
}

const smx_stft_config *smx_stft_kernel_config(const smx_stft_kernel *k) { return k ? k->cfg : nullptr; }

int smx_stft_kernel_set_channels(smx_stft_kernel *k, int64_t channels) {
  return guarded([&] {
    if (!k) throw Failure("set_channels: null kernel");
    if (channels < 1)  // stft.ml:604-608
      throw InvalidArgument(format(
          "prepare: cannot analyse %lld channels (channels must be at least 1)", (long long)channels));
    if (channels == k->channels) return;
    if (k->received != 0 || k->started || k->drained)
      throw InvalidArgument(format("step: cannot feed %lld channels to a stream of %lld (the chunks of one stream share their "
                                   "leading shape; reset before changing it)", (long long)channels, (long long)k->channels));
    // nothing carried yet: the device buffers are sized on first use
    SMX_HIP_CHECK(hipFree(k->d_stream));
    SMX_HIP_CHECK(hipFree(k->d_out));
    SMX_HIP_CHECK(hipFree(k->d_prelude));
    SMX_HIP_CHECK(hipFree(k->d_tail));
    k->d_stream = k->d_out = k->d_prelude = k->d_tail = nullptr;
    k->cap = k->out_cap = k->prelude_cap = 0;
    k->channels = channels;
  });
}

int smx_stft_kernel_reset(smx_stft_kernel *k) {  // stft.ml:401-409
  return guarded([&] {
    if (!k) throw Failure("reset: null kernel");
    k->started = k->drained = false;
    k->received = k->skip = 0;
    k->prelude_len = 0;
    k->tail_len = 0;
    k->pending_len = 0;
  });
}

// host chunk [channels; m] -> frames in the caller's host window [channels; bins; capacity]
int smx_stft_kernel_step(smx_stft_kernel *k, const void *chunk, int64_t m, void *out, int64_t capacity,
                         int64_t *emitted) {
  return guarded([&] {  // stft.ml:521-559
    if (!k || !emitted) throw Failure("step: null argument");
    *emitted = 0;
    if (m > 0 && !chunk) throw Failure("step: null chunk");
    if (m <= 0 || k->drained) {   // nothing to move: the core states the error or the empty result
      *emitted = step_core(*k, chunk, m, m, StreamOut{out, capacity, false}, nullptr);
      return;
    }
    DeviceScratch d((size_t)k->channels * (size_t)m * (size_t)k->dtype_bytes);
    copy_to_device(d.ptr, chunk, (size_t)k->channels * (size_t)m * (size_t)k->dtype_bytes);
    *emitted = step_core(*k, d.ptr, m, m, StreamOut{out, capacity, false}, nullptr);
    SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
  });
}

int smx_stft_kernel_flush(smx_stft_kernel *k, void *out, int64_t capacity, int64_t *emitted) {
  return guarded([&] {  // stft.ml:561-595
    if (!k || !emitted) throw Failure("flush: null argument");
    *emitted = 0;
    *emitted = flush_core(*k, StreamOut{out, capacity, false}, nullptr);
    SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
  });
}

// the same on a device-resident chunk (rows x_stride samples apart) and a device-resident window: nothing crosses the host link
int smx_stft_kernel_step_dev(smx_stft_kernel *k, const void *d_chunk, int64_t m, int64_t x_stride, void *d_out, int64_t capacity,
                             int64_t *emitted, void *stream) {
  return guarded([&] {
    if (!k || !emitted) throw Failure("step: null argument");
    *emitted = 0;
    if (m > 0 && x_stride < m) throw Failure("step: stride smaller than the chunk length");
    *emitted = step_core(*k, d_chunk, x_stride, m, StreamOut{d_out, capacity, true}, (hipStream_t)stream);
  });
}

int smx_stft_kernel_flush_dev(smx_stft_kernel *k, void *d_out, int64_t capacity, int64_t *emitted, void *stream) {
  return guarded([&] {
    if (!k || !emitted) throw Failure("flush: null argument");
    *emitted = 0;
    *emitted = flush_core(*k, StreamOut{d_out, capacity, true}, (hipStream_t)stream);
  });
}

}  // extern "C"

// =============================== Stft.Synthesis (stft.ml:1029-1298) ============
namespace smx {
namespace {

void synth_grow(smx_stft_synthesis &s, int64_t k) {
  if (k <= s.cap) return;
  const int64_t cap = std::max<int64_t>(k, std::max<int64_t>(s.max_block, s.blocks));
  const int64_t bins = s.cfg->bins();
  SMX_HIP_CHECK(hipFree(s.d_local));
  SMX_HIP_CHECK(hipFree(s.d_quot));
  s.d_local = s.d_quot = nullptr;
  SMX_HIP_CHECK(hipMalloc(&s.d_local, (size_t)s.channels * (size_t)bins * (size_t)(2 * s.blocks + cap) * (size_t)s.z_bytes));
  SMX_HIP_CHECK(hipMalloc(&s.d_quot, (size_t)s.channels * (size_t)(cap * s.cfg->hop + s.cfg->fft_size) * (size_t)s.elem()));
  s.cap = cap;
}

void synth_clear(smx_stft_synthesis &s, hipStream_t stream) {   // stft.ml:1091-1096 `synthesis_reset`
  s.fed = 0;
  s.drop = s.left;
  s.drained = false;
  s.carry_len = 0;
  if (s.blocks > 1)   // absent frames before the stream: zero spectra
    SMX_HIP_CHECK(hipMemsetAsync(s.d_hist, 0, (size_t)s.channels * (size_t)s.cfg->bins() * (size_t)(s.blocks - 1) * (size_t)s.z_bytes, stream));
}

// quotients of the array's hops [blocks - 1, blocks - 1 + count_out / hop) -> d_quot, released through the carry into d_out
int64_t synth_emit(smx_stft_synthesis &s, int64_t frames_local, int64_t nq, int64_t env_count, bool open_end, int64_t keep, void *d_out,
                   int64_t capacity, hipStream_t stream) {
  const smx_stft_config &c = *s.cfg;
  IstftJob job;
  job.cfg = &c;
  job.z = s.d_local;
  job.z_bytes = s.z_bytes;
  job.interior = s.z_bytes == 16 ? SMX_INTERIOR_F64 : g_interior.load();
  job.lead = s.channels;
  job.frames = frames_local;
  job.count = frames_local;
  job.out_len = nq;
  job.out = s.d_quot;
  job.stream = stream;
  job.left = (s.blocks - 1) * c.hop;
  job.env_q0 = (s.fed - (s.blocks - 1)) * c.hop;
  job.env_count = env_count;
  job.env_open = open_end;
  const int64_t total = s.carry_len + nq, release = total - keep;
  const int64_t dropped = std::min<int64_t>(s.drop, release);
  const int64_t emit = release - dropped;
  // checked before anything is launched or the head trim is consumed: a retry with a larger buffer starts from the same state
  if (emit > capacity) throw Failure("synthesis: output capacity below the samples this call releases");
  launch_istft(job);
  s.drop -= dropped;
  void *carry_next = nullptr;
  if (keep > 0) SMX_HIP_CHECK(smx::pool_malloc_async(&carry_next, (size_t)s.channels * (size_t)s.hold * (size_t)s.elem(), stream));
  launch_synthesis_release(s.d_carry, s.carry_len, s.d_quot, nq, s.channels, dropped, release, s.hold > 0 ? s.hold : 1, d_out, capacity,
                           carry_next, (int)s.elem(), stream);
  if (keep > 0) {
    SMX_HIP_CHECK(hipMemcpyAsync(s.d_carry, carry_next, (size_t)s.channels * (size_t)s.hold * (size_t)s.elem(), hipMemcpyDeviceToDevice, stream));
    SMX_HIP_CHECK(hipFreeAsync(carry_next, stream));
  }
  s.carry_len = keep;
  return emit;
}

int64_t synth_step_dev(smx_stft_synthesis &s, const void *d_z, int64_t bins, int64_t k, void *d_out, int64_t capacity, hipStream_t stream) {
  const smx_stft_config &c = *s.cfg;
  if (s.drained)   // stft.ml:1182-1186
    throw InvalidArgument("step: cannot feed a drained kernel (flush consumed the tail; reset before reusing)");
  if (bins != c.bins())   // check_frames "step", stft.ml:753-766
    throw InvalidArgument(format("step: cannot invert %lld frequency bins of a %lld-point transform (the bin axis must hold "
                                 "fft_size / 2 + 1 = %lld values)", (long long)bins, (long long)c.fft_size, (long long)c.bins()));
  if (k < 0) throw Failure("step: negative frame count");
  if (k == 0) return 0;
  if (!d_z || !d_out) throw Failure("step: null device pointer");
  synth_grow(s, k);
  const int64_t b = s.blocks, rows = s.channels * c.bins();
  const size_t zb = (size_t)s.z_bytes, lpitch = (size_t)(b - 1 + k) * zb;
  if (b > 1)
    SMX_HIP_CHECK(hipMemcpy2DAsync(s.d_local, lpitch, s.d_hist, (size_t)(b - 1) * zb, (size_t)(b - 1) * zb, (size_t)rows, hipMemcpyDeviceToDevice, stream));
  SMX_HIP_CHECK(hipMemcpy2DAsync(reinterpret_cast<unsigned char *>(s.d_local) + (size_t)(b - 1) * zb, lpitch, d_z, (size_t)k * zb, (size_t)k * zb,
                                 (size_t)rows, hipMemcpyDeviceToDevice, stream));
  const int64_t emit = synth_emit(s, b - 1 + k, k * c.hop, 2 * b + 2, true, s.hold, d_out, capacity, stream);
  if (b > 1)   // the last blocks - 1 spectra of [history ++ chunk]
    SMX_HIP_CHECK(hipMemcpy2DAsync(s.d_hist, (size_t)(b - 1) * zb, reinterpret_cast<unsigned char *>(s.d_local) + (size_t)k * zb, lpitch,
                                   (size_t)(b - 1) * zb, (size_t)rows, hipMemcpyDeviceToDevice, stream));
  s.fed += k;
  return emit;
}

int64_t synth_flush_dev(smx_stft_synthesis &s, void *d_out, int64_t capacity, hipStream_t stream) {   // stft.ml:1243-1269
  if (s.drained) return 0;
  s.drained = true;
  const smx_stft_config &c = *s.cfg;
  const int64_t b = s.blocks, cut = std::max<int64_t>(0, c.fft_size - c.hop - c.right_width());
  s.carry_len = 0;   // a held tail is a tail the trim discards
  if (s.fed == 0 || cut == 0) return 0;
  if (!d_out) throw Failure("flush: null device pointer");
  synth_grow(s, b);
  const int64_t rows = s.channels * c.bins();
  const size_t zb = (size_t)s.z_bytes, lpitch = (size_t)(2 * (b - 1)) * zb;
  SMX_HIP_CHECK(hipMemsetAsync(s.d_local, 0, lpitch * (size_t)rows, stream));
  SMX_HIP_CHECK(hipMemcpy2DAsync(s.d_local, lpitch, s.d_hist, (size_t)(b - 1) * zb, (size_t)(b - 1) * zb, (size_t)rows, hipMemcpyDeviceToDevice, stream));
  return synth_emit(s, 2 * (b - 1), cut, s.fed, false, 0, d_out, capacity, stream);
}

}  // namespace
}  // namespace smx

extern "C" {

int smx_stft_synthesis_prepare(const smx_stft_config *c, int dtype_bytes, int64_t channels, int64_t max_block, smx_stft_synthesis **out) {
  return guarded([&] {
    check_config(c, "prepare");
    if (channels < 1)   // stft.ml:1278-1283
      throw InvalidArgument(format("prepare: cannot synthesise %lld channels (channels must be at least 1)", (long long)channels));
    if (max_block < 1)  // stft.ml:1284-1289
      throw InvalidArgument(format("prepare: cannot accept blocks of %lld frames (max_block must be at least 1)", (long long)max_block));
    if (!stft_nola(*c))  // check_invertible "prepare", stft.ml:742-749
      throw InvalidArgument(format(
          "prepare: cannot invert a %lld-point window advanced by %lld samples inside a %lld-point frame (the "
          "overlap-added squared window must stay above 1e-10 of its largest value at every position)",
          (long long)c->win_length, (long long)c->hop, (long long)c->fft_size));
    if (dtype_bytes != 4 && dtype_bytes != 8) throw Failure("prepare: dtype_bytes must be 4 or 8");
    if (!out) throw Failure("prepare: null output handle");
    require_device();
    auto s = std::make_unique<smx_stft_synthesis>();
    s->cfg = c;
    s->z_bytes = 2 * dtype_bytes;
    s->channels = channels;
    s->max_block = max_block;
    s->left = c->left_width();
    s->hold = std::max<int64_t>(0, c->hop + c->right_width() - c->fft_size);   // stft.ml:1073
    s->blocks = (c->fft_size + c->hop - 1) / c->hop;
    if (s->blocks > 1)
      SMX_HIP_CHECK(hipMalloc(&s->d_hist, (size_t)channels * (size_t)c->bins() * (size_t)(s->blocks - 1) * (size_t)s->z_bytes));
    if (s->hold > 0) SMX_HIP_CHECK(hipMalloc(&s->d_carry, (size_t)channels * (size_t)s->hold * (size_t)s->elem()));
    synth_clear(*s, nullptr);
    SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
    *out = s.release();
  });
}

void smx_stft_synthesis_destroy(smx_stft_synthesis *s) { delete s; }

int64_t smx_stft_synthesis_latency(const smx_stft_config *c) {   // Config.synthesis_latency, stft.ml:152-153
  return c ? c->left_width() + std::max<int64_t>(0, c->hop + c->right_width() - c->fft_size) : -1;
}

int smx_stft_synthesis_sample_bound(const smx_stft_synthesis *s, int64_t *out) {   // synthesis_stage's out_format, stft.ml:1421-1427
  return guarded([&] {
    if (!s || !out) throw Failure("sample_bound: null argument");
    *out = std::max<int64_t>(s->max_block * s->cfg->hop, s->cfg->fft_size - s->cfg->hop);
  });
}

int smx_stft_synthesis_reset(smx_stft_synthesis *s) {
  return guarded([&] {
    if (!s) throw Failure("reset: null kernel");
    // on the stream the kernel last ran on (a non-blocking side stream is not ordered with the null stream), then drained:
    // the next step may come on any stream
    synth_clear(*s, s->last_stream);
    SMX_HIP_CHECK(hipStreamSynchronize(s->last_stream));
  });
}

int smx_stft_synthesis_step_dev(smx_stft_synthesis *s, const void *d_z, int64_t bins, int64_t k, void *d_out, int64_t capacity,
                                int64_t *emitted, void *stream) {
  return guarded([&] {
    if (!s || !emitted) throw Failure("step: null argument");
    *emitted = 0;
    s->last_stream = (hipStream_t)stream;
    *emitted = synth_step_dev(*s, d_z, bins, k, d_out, capacity, (hipStream_t)stream);
  });
}

int smx_stft_synthesis_flush_dev(smx_stft_synthesis *s, void *d_out, int64_t capacity, int64_t *emitted, void *stream) {
  return guarded([&] {
    if (!s || !emitted) throw Failure("flush: null argument");
    *emitted = 0;
    s->last_stream = (hipStream_t)stream;
    *emitted = synth_flush_dev(*s, d_out, capacity, (hipStream_t)stream);
  });
}

// host chunks: [channels; bins; k] complex in, [channels; capacity] real out (row stride = capacity)
int smx_stft_synthesis_step(smx_stft_synthesis *s, const void *z, int64_t bins, int64_t k, void *out, int64_t capacity, int64_t *emitted) {
  return guarded([&] {
    if (!s || !emitted) throw Failure("step: null argument");
    *emitted = 0;
    if (k > 0 && (!z || !out)) throw Failure("step: null pointer");
    if (k <= 0 || bins != s->cfg->bins() || s->drained) {   // nothing to move: the device entry states the error or the empty result
      *emitted = synth_step_dev(*s, z, bins, k, out, capacity, nullptr);
      return;
    }
    const size_t zb = (size_t)s->channels * (size_t)bins * (size_t)k * (size_t)s->z_bytes;
    const size_t ob = (size_t)s->channels * (size_t)capacity * (size_t)s->elem();
    DeviceScratch dz(zb), dout(ob);
    copy_to_device(dz.ptr, z, zb);
    *emitted = synth_step_dev(*s, dz.ptr, bins, k, dout.ptr, capacity, nullptr);
    SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
    if (*emitted > 0) copy_to_host(out, dout.ptr, ob);
  });
}

int smx_stft_synthesis_flush(smx_stft_synthesis *s, void *out, int64_t capacity, int64_t *emitted) {
  return guarded([&] {
    if (!s || !emitted) throw Failure("flush: null argument");
    *emitted = 0;
    const size_t ob = (size_t)s->channels * (size_t)std::max<int64_t>(capacity, 1) * (size_t)s->elem();
    DeviceScratch dout(ob);
    *emitted = synth_flush_dev(*s, dout.ptr, capacity, nullptr);
    SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
    if (*emitted > 0) {
      if (!out) throw Failure("flush: null pointer");
      copy_to_host(out, dout.ptr, ob);
    }
  });
}

}  // extern "C"

// =============================== Mel ==========================================
namespace smx {
namespace {

void check_mel_shape(const smx_mel_config &c, int64_t lead, int64_t bins, int64_t frames) {
  if (lead < 0 || frames < 0 || bins < 0) throw Failure("apply: negative extent");
  if (bins != c.bins())  // mel.ml:210-218
    throw InvalidArgument(format(
        "apply: cannot project %lld frequency bins through a filterbank built for an FFT of size %lld "
        "(%lld bins)",
        (long long)bins, (long long)c.fft_size, (long long)c.bins()));
}

void mel_apply_dev(const smx_mel_config &c, const void *d_s, int elem_bytes, int64_t lead, int64_t bins,
                   int64_t frames, void *d_out, hipStream_t stream) {
  check_mel_shape(c, lead, bins, frames);
  if (lead == 0 || frames == 0) return;  // mel.ml:220-226: nothing to reduce over
  if (!d_s || !d_out) throw Failure("apply: null device pointer");
  MelJob job;
  job.cfg = &c;
  job.s = d_s;
  job.elem_bytes = elem_bytes;
  job.lead = lead;
  job.frames = frames;
  job.out = d_out;
  job.stream = stream;
  launch_mel_apply(job);
}

void mel_apply_host(const smx_mel_config &c, const void *s, int elem_bytes, int64_t lead, int64_t bins,
                    int64_t frames, void *out) {
  check_mel_shape(c, lead, bins, frames);
  if (lead == 0 || frames == 0) return;
  if (!s || !out) throw Failure("apply: null pointer");
  require_device();
  const size_t in_bytes = (size_t)lead * (size_t)bins * (size_t)frames * (size_t)elem_bytes;
  const size_t out_bytes = (size_t)lead * (size_t)c.n_mels * (size_t)frames * (size_t)elem_bytes;
  DeviceScratch ds(in_bytes), dout(out_bytes);
  copy_to_device(ds.ptr, s, in_bytes);
  mel_apply_dev(c, ds.ptr, elem_bytes, lead, bins, frames, dout.ptr, nullptr);
  SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
  copy_to_host(out, dout.ptr, out_bytes);
}

void check_fft_sizes(const smx_stft_config &sc, const smx_mel_config &mc) {  // soundml.ml:12-20
  if (sc.fft_size != mc.fft_size)
    throw InvalidArgument(format(
        "mel_spectrogram: cannot project a %lld-point STFT through a filterbank built for an FFT of "
        "size %lld (the two configurations must agree on fft_size)",
        (long long)sc.fft_size, (long long)mc.fft_size));
}

// Soundml.mel_spectrogram on device-resident audio.  The fused kernel keeps the
// power tile on chip; other geometries run power_spectrum into a scratch
// spectrogram followed by Mel.apply (the reference's own composition).
void mel_spectrogram_dev(const smx_stft_config &sc, const smx_mel_config &mc, const void *d_x,
                         int in_bytes, int64_t lead, int64_t n, int64_t x_stride, double power,
                         void *d_out, hipStream_t stream) {
  check_fft_sizes(sc, mc);
  check_rank_extents("mel_spectrogram", lead, n);
  if (x_stride < n) throw Failure("mel_spectrogram: x_stride is smaller than the signal length");
  const int64_t count = sc.frames(n);
  if (lead == 0 || count == 0) return;
  if (!d_x || !d_out) throw Failure("mel_spectrogram: null device pointer");
  MelSpecJob job;
  job.stft.cfg = &sc;
  job.stft.x = d_x;
  job.stft.in_bytes = in_bytes;
  job.stft.interior = in_bytes == 8 ? SMX_INTERIOR_F64 : g_interior.load();
  job.stft.lead = lead;
  job.stft.n = n;
  job.stft.x_stride = x_stride;
  job.stft.left = sc.left_width();
  job.stft.pad = sc.pad;
  job.stft.pad_value = sc.pad_value;
  job.stft.p0 = 0;
  job.stft.count = count;
  job.stft.mode = OUT_POWER;
  job.stft.power = power;
  job.stft.stream = stream;
  job.mel = &mc;
  job.out = d_out;
  if (launch_mel_spectrogram_fused(job)) return;
  // two-step form; scratch is stream-ordered
  void *scratch = nullptr;
  const size_t bytes = (size_t)lead * (size_t)sc.bins() * (size_t)count * (size_t)in_bytes;
  SMX_HIP_CHECK(smx::pool_malloc_async(&scratch, bytes, stream));
  job.stft.out = scratch;
  job.stft.out_stride = count;
  job.stft.out_offset = 0;
  launch_stft(job.stft);
  mel_apply_dev(mc, scratch, in_bytes, lead, sc.bins(), count, d_out, stream);
  SMX_HIP_CHECK(hipFreeAsync(scratch, stream));
}

void mel_spectrogram_host_one(const smx_stft_config &sc, const smx_mel_config &mc, const void *x, int in_bytes,
                              int64_t lead, int64_t n, double power, void *out) {
  check_fft_sizes(sc, mc);
  check_rank_extents("mel_spectrogram", lead, n);
  const int64_t count = sc.frames(n);
  if (lead == 0 || count == 0) return;
  if (!x || !out) throw Failure("mel_spectrogram: null pointer");
  require_device();
  const size_t in_total = (size_t)lead * (size_t)n * (size_t)in_bytes;
  const size_t out_total = (size_t)lead * (size_t)mc.n_mels * (size_t)count * (size_t)in_bytes;
  DeviceScratch dx(in_total), dout(out_total);
  copy_to_device(dx.ptr, x, in_total);
  mel_spectrogram_dev(sc, mc, dx.ptr, in_bytes, lead, n, n, power, dout.ptr, nullptr);
  SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
  copy_to_host(out, dout.ptr, out_total);
}

// Soundml.mel_spectrogram from host memory: the current device, or the device list's clip ranges side by side (soundml.mli:85-101:
// leading axes broadcast; mel_props.ml:136-155 tests the slices)
void mel_spectrogram_host(const smx_stft_config &sc, const smx_mel_config &mc, const void *x, int in_bytes,
                          int64_t lead, int64_t n, double power, void *out) {
  check_fft_sizes(sc, mc);
  check_rank_extents("mel_spectrogram", lead, n);
  const int64_t count = sc.frames(n);
  if (lead == 0 || count == 0) return;
  if (!x || !out) throw Failure("mel_spectrogram: null pointer");
  require_device();
  const size_t in_clip = (size_t)n * (size_t)in_bytes, out_clip = (size_t)mc.n_mels * (size_t)count * (size_t)in_bytes;
  for_each_shard(lead, [&](int64_t clip0, int64_t nc) {
    mel_spectrogram_host_one(sc, mc, reinterpret_cast<const unsigned char *>(x) + (size_t)clip0 * in_clip, in_bytes, nc, n, power,
                             reinterpret_cast<unsigned char *>(out) + (size_t)clip0 * out_clip);
  });
}


// Stft.griffin_lim (stft.ml:961-1017) on device-resident data: the reference's loop, its transforms replaced by
// this library's synthesis and analysis launches, everything on `stream`.
void griffin_lim_dev(const smx_stft_config &c, const void *d_s, int elem_bytes, int64_t lead, int64_t bins,
                     int64_t frames, int64_t n_iter, double momentum, const void *d_phase, int has_length,
                     int64_t length, void *d_out, hipStream_t stream) {
  if (lead < 0 || frames < 0) throw Failure("griffin_lim: negative extent");
  // check_synthesis "griffin_lim" (stft.ml:963), then the loop's own preconditions
  if (bins != c.bins())
    throw InvalidArgument(format(
        "griffin_lim: cannot invert %lld frequency bins of a %lld-point transform (the bin axis must hold "
        "fft_size / 2 + 1 = %lld values)",
        (long long)bins, (long long)c.fft_size, (long long)c.bins()));
  if (has_length && length < 0)
    throw InvalidArgument(format("griffin_lim: cannot synthesise a signal of length %lld (length must be non-negative)",
                                 (long long)length));
  if (!stft_nola(c))
    throw InvalidArgument(format(
        "griffin_lim: cannot invert a %lld-point window advanced by %lld samples inside a %lld-point frame (the "
        "overlap-added squared window must stay above 1e-10 of its largest value at every position)",
        (long long)c.win_length, (long long)c.hop, (long long)c.fft_size));
  if (n_iter < 1)
    throw InvalidArgument(format("griffin_lim: cannot run %lld iterations (n_iter must be at least 1)", (long long)n_iter));
  if (momentum < 0.0)
    throw InvalidArgument(format("griffin_lim: cannot use a momentum of %g (momentum must be non-negative)", momentum));
  const int64_t natural = stft_output_length(c, frames);
  const int64_t out_len = has_length ? length : natural;
  if (lead == 0 || out_len == 0) return;
  if (!d_out || (frames > 0 && !d_s)) throw Failure("griffin_lim: null device pointer");
  const int64_t total = lead * bins * frames;
  if (elem_bytes == 4 && g_interior.load() == SMX_INTERIOR_F64) {
    // the reference's loop is float64 throughout whatever the dtype of s (stft.ml:977, 1017): widen, run, narrow
    double *s64 = nullptr, *p64 = nullptr, *o64 = nullptr;
    const size_t n64 = (size_t)std::max<int64_t>(total, 1) * sizeof(double);
    SMX_HIP_CHECK(smx::pool_malloc_async((void **)&s64, n64, stream));
    launch_gl_widen((const float *)d_s, s64, total, stream);
    if (d_phase) {
      SMX_HIP_CHECK(smx::pool_malloc_async((void **)&p64, n64, stream));
      launch_gl_widen((const float *)d_phase, p64, total, stream);
    }
    SMX_HIP_CHECK(smx::pool_malloc_async((void **)&o64, (size_t)lead * (size_t)out_len * sizeof(double), stream));
    griffin_lim_dev(c, s64, 8, lead, bins, frames, n_iter, momentum, p64, has_length, length, o64, stream);
    launch_gl_narrow(o64, (float *)d_out, lead * out_len, stream);
    for (void *ptr : {(void *)s64, (void *)p64, (void *)o64})
      if (ptr) SMX_HIP_CHECK(hipFreeAsync(ptr, stream));
    return;
  }
  const int z_bytes = 2 * elem_bytes;
  void *zbuf = nullptr;
  const size_t cbytes = (size_t)std::max<int64_t>(total, 1) * (size_t)z_bytes;
  auto make_job = [&](const void *z, int has_len, int64_t len, void *out, int64_t olen) {
    IstftJob job;
    job.cfg = &c;
    job.z = z;
    job.z_bytes = z_bytes;
    job.interior = elem_bytes == 8 ? SMX_INTERIOR_F64 : g_interior.load();
    job.lead = lead;
    job.frames = frames;
    job.count = has_len ? std::min<int64_t>(frames, (len + c.left_width() + c.hop - 1) / c.hop) : frames;
    job.out_len = olen;
    job.out = out;
    job.stream = stream;
    return job;
  };
  // S * angles: multiplied in by the fused synthesis kernel as it stages the spectrum, or materialised for the others
  auto synth = [&](const void *angles_in, int has_len, int64_t len, void *out, int64_t olen) {
    IstftJob job = make_job(angles_in, has_len, len, out, olen);
    if (istft_takes_factors(job)) {
      job.mag = d_s;
    } else {
      if (!zbuf) SMX_HIP_CHECK(smx::pool_malloc_async(&zbuf, cbytes, stream));
      launch_gl_apply(d_s, angles_in, zbuf, total, elem_bytes, stream);
      job.z = zbuf;
    }
    launch_istft(job);
  };
  void *angles = nullptr, *rebuilt = nullptr, *previous = nullptr, *signal = nullptr;
  const bool iterate = natural > 0 && frames > 0;   // stft.ml:993-996
  const double beta = momentum / (1.0 + momentum);
  const bool folded = iterate && istft_takes_factors(make_job(nullptr, 0, 0, nullptr, natural));
  // Round 5 (late): the loop's spectra FRAME-MAJOR where both fft-2048 pipelines take them so -- c_k never leaves the library, so its
  // layout is free: the analysis stores a frame's bins straight from its registers (no output tile, no flush, waves that never
  // wait for each other: stft2048_complex_fm_kernel) and the synthesis stages whole rows (istft2048_pipe_kernel<.., true>).  The
  // arithmetic of every value is that of the reference-layout kernels: same bits.  SMX_GL_FRAME_MAJOR=0: the reference layout.
  bool frame_major = false;
  const int64_t fm_rows = (frames + 15) / 16 * 16, fm_pitch = 1032;   // (rows of 8256 bytes: whole 64-byte blocks)
  auto fm_analysis_job = [&](const void *x, void *out) {
    StftJob job;
    job.cfg = &c;
    job.x = x;
    job.in_bytes = 4;
    job.interior = g_interior.load();
    job.lead = lead;
    job.n = natural;
    job.x_stride = natural;
    job.left = c.left_width();
    job.pad = c.pad;
    job.pad_value = c.pad_value;
    job.p0 = 0;
    job.count = frames;
    job.mode = OUT_COMPLEX;
    job.power = 0.0;
    job.out = out;
    job.out_stride = frames;
    job.out_offset = 0;
    job.stream = stream;
    return job;
  };
  if (folded && elem_bytes == 4 && env_flag("SMX_GL_FRAME_MAJOR") != 0 && c.frames(natural) == frames && lead <= 65535) {   // (the transposition's grid: a clip per blockIdx.z)
    IstftJob probe = make_job(nullptr, 0, 0, nullptr, natural);
    probe.fm_pitch = fm_pitch;
    probe.fm_rows = fm_rows;
    frame_major = istft_frame_major_ok(probe) && launch_stft_complex_fm(fm_analysis_job(nullptr, nullptr), nullptr, 2 * fm_pitch, fm_rows, /*only_ask=*/true);
  }
  if (!frame_major || d_phase) {   // the initial angles in the reference layout (the frame-major loop without a phase fills its own)
    SMX_HIP_CHECK(smx::pool_malloc_async(&angles, cbytes, stream));
    launch_gl_init(d_phase, angles, total, elem_bytes, stream);
  }
  if (frame_major) {
    const size_t fm_elems = (size_t)lead * (size_t)fm_rows * (size_t)fm_pitch;
    void *mag_fm = nullptr, *a_fm = nullptr, *b_fm = nullptr, *c_fm = nullptr;
    SMX_HIP_CHECK(smx::pool_malloc_async(&mag_fm, fm_elems * 4, stream));
    SMX_HIP_CHECK(smx::pool_malloc_async(&a_fm, fm_elems * 8, stream));
    SMX_HIP_CHECK(smx::pool_malloc_async(&b_fm, fm_elems * 8, stream));
    SMX_HIP_CHECK(smx::pool_malloc_async(&c_fm, fm_elems * 8, stream));
    SMX_HIP_CHECK(smx::pool_malloc_async(&signal, (size_t)lead * (size_t)natural * 4, stream));
    launch_gl_to_frame_major(d_s, mag_fm, lead, bins, frames, fm_rows, fm_pitch, 4, stream);
    if (d_phase) launch_gl_to_frame_major(angles, a_fm, lead, bins, frames, fm_rows, fm_pitch, 8, stream);
    else launch_gl_init(nullptr, a_fm, (int64_t)fm_elems, 4, stream);   // (1, 0) everywhere
    auto synth_fm = [&](const void *z, const void *prev, bool unit, int has_len, int64_t len, void *out, int64_t olen) {
      IstftJob job = make_job(z, has_len, len, out, olen);
      job.mag = mag_fm;
      job.unit = unit;
      job.prev = prev;
      job.beta = beta;
      job.fm_pitch = fm_pitch;
      job.fm_rows = fm_rows;
      launch_istft(job);
    };
    void *cur = nullptr, *old = nullptr;          // c_k, c_(k-1)
    void *spare[2] = {b_fm, c_fm};
    for (int64_t k = 0; k < n_iter; ++k) {
      if (!cur) synth_fm(a_fm, nullptr, false, 0, 0, signal, natural);
      else synth_fm(cur, old, true, 0, 0, signal, natural);
      void *target = !cur ? spare[0] : (old ? old : spare[1]);
      if (!launch_stft_complex_fm(fm_analysis_job(signal, target), target, 2 * fm_pitch, fm_rows)) throw Failure("griffin_lim: the frame-major analysis refused a job it had accepted");
      old = cur;
      cur = target;
    }
    synth_fm(cur, old, true, has_length, length, d_out, out_len);
    for (void *ptr : {mag_fm, a_fm, b_fm, c_fm})
      SMX_HIP_CHECK(hipFreeAsync(ptr, stream));
  } else if (folded) {
    // fused fft-2048 kernels: neither S * angles nor the angles themselves are materialised after the first pass --
    // the synthesis kernel forms S * unit(c_k - beta c_(k-1)) from the two latest rebuilt spectra as it stages them
    // (stft.ml:1003-1012), and the analysis of iteration k + 1 overwrites c_(k-1), which nothing reads any more
    SMX_HIP_CHECK(smx::pool_malloc_async(&rebuilt, cbytes, stream));
    SMX_HIP_CHECK(smx::pool_malloc_async(&previous, cbytes, stream));
    SMX_HIP_CHECK(smx::pool_malloc_async(&signal, (size_t)lead * (size_t)natural * (size_t)elem_bytes, stream));
    void *cur = nullptr, *old = nullptr;          // c_k, c_(k-1)
    auto synth_unit = [&](int has_len, int64_t len, void *out, int64_t olen) {
      IstftJob job = make_job(cur, has_len, len, out, olen);
      job.mag = d_s;
      job.unit = true;
      job.prev = old;
      job.beta = beta;
      launch_istft(job);
    };
    for (int64_t k = 0; k < n_iter; ++k) {
      if (!cur) synth(angles, 0, 0, signal, natural);
      else synth_unit(0, 0, signal, natural);
      void *target = !cur ? rebuilt : (old ? old : previous);
      stft_range_dev(c, signal, elem_bytes, lead, natural, natural, 0, frames, OUT_COMPLEX, 0.0, target, stream);
      old = cur;
      cur = target;
    }
    synth_unit(has_length, length, d_out, out_len);
  } else {
    if (iterate) {
      SMX_HIP_CHECK(smx::pool_malloc_async(&rebuilt, cbytes, stream));
      SMX_HIP_CHECK(smx::pool_malloc_async(&previous, cbytes, stream));
      SMX_HIP_CHECK(smx::pool_malloc_async(&signal, (size_t)lead * (size_t)natural * (size_t)elem_bytes, stream));
      bool has_prev = false;
      for (int64_t k = 0; k < n_iter; ++k) {
        synth(angles, 0, 0, signal, natural);
        stft_range_dev(c, signal, elem_bytes, lead, natural, natural, 0, frames, OUT_COMPLEX, 0.0, rebuilt, stream);
        launch_gl_update(rebuilt, has_prev ? previous : nullptr, beta, angles, total, elem_bytes, stream);
        std::swap(rebuilt, previous);   // previous <- rebuilt
        has_prev = true;
      }
    }
    synth(angles, has_length, length, d_out, out_len);
  }
  for (void *ptr : {angles, zbuf, rebuilt, previous, signal})
    if (ptr) SMX_HIP_CHECK(hipFreeAsync(ptr, stream));
}

void griffin_lim_host(const smx_stft_config &c, const void *s, int elem_bytes, int64_t lead, int64_t bins,
                      int64_t frames, int64_t n_iter, double momentum, const void *phase, int has_length,
                      int64_t length, void *out) {
  if (lead < 0 || frames < 0) throw Failure("griffin_lim: negative extent");
  const int64_t out_len = has_length ? length : (frames >= 0 ? stft_output_length(c, frames) : 0);
  const bool run = lead > 0 && out_len > 0 && bins == c.bins() && stft_nola(c) && n_iter >= 1 && momentum >= 0.0 &&
                   !(has_length && length < 0);
  if (!run) {   // the checks (and nothing else) run without a device
    griffin_lim_dev(c, nullptr, elem_bytes, 0, bins, frames, n_iter, momentum, nullptr, has_length, length, nullptr, nullptr);
    return;
  }
  if (!out || (frames > 0 && !s)) throw Failure("griffin_lim: null pointer");
  require_device();
  const size_t sb = (size_t)lead * (size_t)bins * (size_t)frames * (size_t)elem_bytes;
  const size_t ob = (size_t)lead * (size_t)out_len * (size_t)elem_bytes;
  DeviceScratch ds(sb), dp(phase ? sb : 16), dout(ob);
  if (sb) copy_to_device(ds.ptr, s, sb);
  if (phase && sb) copy_to_device(dp.ptr, phase, sb);
  griffin_lim_dev(c, ds.ptr, elem_bytes, lead, bins, frames, n_iter, momentum, phase ? dp.ptr : nullptr, has_length,
                  length, dout.ptr, nullptr);
  SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
  copy_to_host(out, dout.ptr, ob);
}

// Soundml.mfcc (soundml.ml:50-95): checks in the reference's order and words, then mel_spectrogram + the tail
void check_mfcc(const smx_stft_config &sc, const smx_mel_config &mc, int64_t n_mfcc, int has_lifter, double lifter) {
  if (sc.fft_size != mc.fft_size)
    throw InvalidArgument(format(
        "mfcc: cannot project a %lld-point STFT through a filterbank built for an FFT of size %lld (the two "
        "configurations must agree on fft_size)",
        (long long)sc.fft_size, (long long)mc.fft_size));
  if (n_mfcc < 1 || n_mfcc > mc.n_mels)
    throw InvalidArgument(format(
        "mfcc: cannot keep %lld cepstral coefficients of %lld mel bands (n_mfcc must lie in [1, n_mels])",
        (long long)n_mfcc, (long long)mc.n_mels));
  if (has_lifter && !(std::isfinite(lifter) && lifter >= 0.0))
    throw InvalidArgument(format("mfcc: cannot lifter with a coefficient of %g (lifter must be finite and non-negative)",
                                 lifter));
}

void mfcc_dev(const smx_stft_config &sc, const smx_mel_config &mc, const void *d_x, int in_bytes, int64_t lead,
              int64_t n, int64_t x_stride, int64_t n_mfcc, int has_lifter, double lifter, void *d_out,
              hipStream_t stream) {
  check_mfcc(sc, mc, n_mfcc, has_lifter, lifter);
  check_rank_extents("mfcc", lead, n);
  const int64_t count = sc.frames(n);
  if (lead == 0 || count == 0) return;
  if (!d_x || !d_out) throw Failure("mfcc: null device pointer");
  void *mel = nullptr;
  SMX_HIP_CHECK(smx::pool_malloc_async(&mel, (size_t)lead * (size_t)mc.n_mels * (size_t)count * (size_t)in_bytes, stream));
  mel_spectrogram_dev(sc, mc, d_x, in_bytes, lead, n, x_stride, 2.0, mel, stream);
  MfccJob job;
  job.mel = mel;
  job.elem_bytes = in_bytes;
  job.lead = lead;
  job.frames = count;
  job.n_mels = (int)mc.n_mels;
  job.n_mfcc = (int)n_mfcc;
  job.lifter = has_lifter ? lifter : 0.0;
  job.out = d_out;
  job.stream = stream;
  launch_mfcc(job);
  SMX_HIP_CHECK(hipFreeAsync(mel, stream));
}

void mfcc_host(const smx_stft_config &sc, const smx_mel_config &mc, const void *x, int in_bytes, int64_t lead,
               int64_t n, int64_t n_mfcc, int has_lifter, double lifter, void *out) {
  check_mfcc(sc, mc, n_mfcc, has_lifter, lifter);
  check_rank_extents("mfcc", lead, n);
  const int64_t count = sc.frames(n);
  if (lead == 0 || count == 0) return;
  if (!x || !out) throw Failure("mfcc: null pointer");
  require_device();
  const size_t in_total = (size_t)lead * (size_t)n * (size_t)in_bytes;
  const size_t out_total = (size_t)lead * (size_t)n_mfcc * (size_t)count * (size_t)in_bytes;
  DeviceScratch dx(in_total), dout(out_total);
  copy_to_device(dx.ptr, x, in_total);
  mfcc_dev(sc, mc, dx.ptr, in_bytes, lead, n, n, n_mfcc, has_lifter, lifter, dout.ptr, nullptr);
  SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
  copy_to_host(out, dout.ptr, out_total);
}

// ---- Convert.power_to_db / amplitude_to_db (convert.ml:3-62) ------------------------------------------------
void to_db_checks(const char *fn, double reference, double amin, int has_top_db, double top_db) {
  if (!(std::isfinite(reference) && reference > 0.0))
    throw InvalidArgument(format("Soundml.Convert.%s: reference must be finite and positive", fn));
  if (!(std::isfinite(amin) && amin > 0.0))
    throw InvalidArgument(format("Soundml.Convert.%s: amin must be finite and positive", fn));
  if (has_top_db && !(std::isfinite(top_db) && top_db >= 0.0))
    throw InvalidArgument(format("Soundml.Convert.%s: top_db must be finite and non-negative", fn));
}

void to_db_dev(bool amplitude, const void *d_s, int elem_bytes, int64_t total, double reference, double amin, int has_top_db,
               double top_db, void *d_out, hipStream_t stream) {
  const char *fn = amplitude ? "amplitude_to_db" : "power_to_db";
  to_db_checks(fn, reference, amin, has_top_db, top_db);
  if (total < 0) throw Failure(format("%s: negative extent", fn));
  if (total == 0) return;
  if (!d_s || !d_out) throw Failure(format("%s: null device pointer", fn));
  ToDbJob job;
  job.s = d_s;
  job.out = d_out;
  job.elem_bytes = elem_bytes;
  job.total = total;
  job.gain = amplitude ? 20.0 : 10.0;
  job.magnitude = amplitude;
  job.reference = reference;
  job.amin = amin;
  job.has_top_db = has_top_db != 0;
  job.top_db = top_db;
  job.stream = stream;
  launch_to_db(job);
}

void to_db_host(bool amplitude, const void *s, int elem_bytes, int64_t total, double reference, double amin, int has_top_db,
                double top_db, void *out) {
  to_db_checks(amplitude ? "amplitude_to_db" : "power_to_db", reference, amin, has_top_db, top_db);
  if (total <= 0) return;
  if (!s || !out) throw Failure("to_db: null pointer");
  require_device();
  const size_t bytes = (size_t)total * (size_t)elem_bytes;
  DeviceScratch d(bytes);
  copy_to_device(d.ptr, s, bytes);
  to_db_dev(amplitude, d.ptr, elem_bytes, total, reference, amin, has_top_db, top_db, d.ptr, nullptr);
  SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
  copy_to_host(out, d.ptr, bytes);
}

// ---- Spectral.* (spectral.ml:27-255): checks in the reference's order and words, then one launch -------------
const char *spectral_op(int feature) {
  switch (feature) {
    case SPECTRAL_CENTROID: return "spectral_centroid";
    case SPECTRAL_BANDWIDTH: return "spectral_bandwidth";
    case SPECTRAL_ROLLOFF: return "spectral_rolloff";
    default: return "spectral_flatness";
  }
}

struct SpectralParams {
  int feature = SPECTRAL_CENTROID;
  double p = 2.0, power = 2.0;          // p: bandwidth exponent | roll_percent | amin
  const double *freqs = nullptr;        // host
  int64_t n_freqs = 0, sample_rate = 0;
  bool has_centroid = false;
  int64_t c_rows = 0, c_frames = 0;
};

// everything that does not read the data; returns the FFT-grid step (0 with a custom grid)
double check_spectral(const SpectralParams &q, int64_t lead, int64_t bins, int64_t frames) {
  const char *op = spectral_op(q.feature);
  if (lead < 0 || bins < 0 || frames < 0)
    throw Failure(format("%s: negative extent (lead %lld, bins %lld, frames %lld)", op, (long long)lead,
                         (long long)bins, (long long)frames));
  if (q.feature == SPECTRAL_BANDWIDTH && !(std::isfinite(q.p) && q.p > 0.0))
    throw InvalidArgument(format("%s: cannot raise deviations to the power %g (p must be finite and positive)", op, q.p));
  if (q.feature == SPECTRAL_ROLLOFF && !(q.p > 0.0 && q.p < 1.0))
    throw InvalidArgument(format(
        "%s: cannot keep %g of the spectral energy (roll_percent must lie strictly between 0 and 1)", op, q.p));
  double step = 0.0;
  if (q.feature == SPECTRAL_FLATNESS) {
    if (!(std::isfinite(q.p) && q.p > 0.0))
      throw InvalidArgument(format("%s: cannot floor the spectrum at %g (amin must be finite and positive)", op, q.p));
    if (!(std::isfinite(q.power) && q.power > 0.0))
      throw InvalidArgument(format(
          "%s: cannot raise magnitudes to the power %g (power must be finite and positive)", op, q.power));
    return step;
  }
  // grid (spectral.ml:105-136)
  if (q.sample_rate < 1)
    throw InvalidArgument(format("%s: cannot use a sample rate of %lld Hz (sample_rate must be at least 1)", op,
                                 (long long)q.sample_rate));
  if (q.freqs) {
    if (q.n_freqs != bins)
      throw InvalidArgument(format(
          "%s: cannot pair %lld bin frequencies with %lld bins (freqs holds one frequency per bin)", op,
          (long long)q.n_freqs, (long long)bins));
  } else {
    if (bins < 2)
      throw InvalidArgument(format(
          "%s: cannot derive bin frequencies for a %lld-bin spectrogram (the implied FFT size is %lld; pass "
          "freqs explicitly)", op, (long long)bins, (long long)(2 * (bins - 1))));
    const int64_t fft_size = 2 * (bins - 1);
    step = 1.0 / ((double)fft_size * (1.0 / (double)q.sample_rate));
  }
  if (q.feature == SPECTRAL_BANDWIDTH && q.has_centroid && (q.c_rows != 1 || q.c_frames != frames))
    throw InvalidArgument(format(
        "%s: cannot reuse a centroid with %lld rows over %lld frames for a %lld-frame spectrogram (centroid must "
        "be [...; 1; frames], one frequency per frame)", op, (long long)q.c_rows, (long long)q.c_frames,
        (long long)frames));
  return step;
}

void spectral_dev(const SpectralParams &q, const void *d_s, int elem_bytes, int64_t lead, int64_t bins,
                  int64_t frames, const void *d_centroid, void *d_out, hipStream_t stream) {
  const double step = check_spectral(q, lead, bins, frames);
  if (lead == 0 || bins == 0 || frames == 0) return;   // empty_feature: all zero by contract, nothing to reduce
  if (!d_s || !d_out) throw Failure(format("%s: null device pointer", spectral_op(q.feature)));
  SpectralJob job;
  job.feature = q.feature;
  job.s = d_s;
  job.elem_bytes = elem_bytes;
  job.lead = lead;
  job.bins = bins;
  job.frames = frames;
  job.freqs = q.freqs;
  job.step = step;
  job.p = q.p;
  job.power = q.power;
  job.centroid = q.has_centroid ? d_centroid : nullptr;
  job.out = d_out;
  job.stream = stream;
  if (!launch_spectral(job))
    throw InvalidArgument(format(
        "%s: cannot analyse a spectrogram with negative or NaN values (a magnitude spectrogram is non-negative)",
        spectral_op(q.feature)));
}

void spectral_host(const SpectralParams &q, const void *s, int elem_bytes, int64_t lead, int64_t bins,
                   int64_t frames, const void *centroid, void *out) {
  check_spectral(q, lead, bins, frames);
  if (lead == 0 || bins == 0 || frames == 0) return;
  if (!s || !out) throw Failure(format("%s: null pointer", spectral_op(q.feature)));
  require_device();
  const size_t in_total = (size_t)lead * (size_t)bins * (size_t)frames * (size_t)elem_bytes;
  const size_t out_total = (size_t)lead * (size_t)frames * (size_t)elem_bytes;
  DeviceScratch ds(in_total), dout(out_total), dc(q.has_centroid ? out_total : 0);
  copy_to_device(ds.ptr, s, in_total);
  if (q.has_centroid) copy_to_device(dc.ptr, centroid, out_total);
  spectral_dev(q, ds.ptr, elem_bytes, lead, bins, frames, dc.ptr, dout.ptr, nullptr);
  SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
  copy_to_host(out, dout.ptr, out_total);
}

// ---- Chroma.apply / Soundml.chroma_stft (chroma.ml:285-317, soundml.ml:97-107) ------------------------------
void check_chroma_norm(const char *op, int norm, double norm_p) {   // chroma.ml:30-40
  if (norm == SMX_CHROMA_NORM_NONE || norm == SMX_CHROMA_NORM_INF) return;
  if (norm != SMX_CHROMA_NORM_P) throw Failure(format("%s: unknown norm %d", op, norm));
  if (!(std::isfinite(norm_p) && norm_p > 0.0))
    throw InvalidArgument(format(
        "%s: cannot normalise in the %g-norm (the exponent must be finite and positive)", op, norm_p));
}

void chroma_apply_dev(const smx_chroma_config &c, const void *d_s, int elem_bytes, int64_t lead, int64_t bins,
                      int64_t frames, int norm, double norm_p, void *d_out, hipStream_t stream) {
  check_chroma_norm("apply", norm, norm_p);
  if (lead < 0 || bins < 0 || frames < 0)
    throw Failure(format("apply: negative extent (lead %lld, bins %lld, frames %lld)", (long long)lead,
                         (long long)bins, (long long)frames));
  if (bins != c.bins())
    throw InvalidArgument(format(
        "apply: cannot project %lld frequency bins through a matrix built for an FFT of size %lld (%lld bins)",
        (long long)bins, (long long)c.fft_size, (long long)c.bins()));
  if (lead == 0 || frames == 0) return;
  if (!d_s || !d_out) throw Failure("apply: null device pointer");
  ChromaJob job;
  job.config = &c;
  job.s = d_s;
  job.elem_bytes = elem_bytes;
  job.lead = lead;
  job.frames = frames;
  job.norm = norm;
  job.norm_p = norm_p;
  job.out = d_out;
  job.stream = stream;
  launch_chroma(job);
}

void chroma_apply_host(const smx_chroma_config &c, const void *s, int elem_bytes, int64_t lead, int64_t bins,
                       int64_t frames, int norm, double norm_p, void *out) {
  check_chroma_norm("apply", norm, norm_p);
  if (bins != c.bins() || lead <= 0 || frames <= 0) {   // the checks and the empty case, nothing to upload
    chroma_apply_dev(c, nullptr, elem_bytes, lead, bins, frames, norm, norm_p, nullptr, nullptr);
    return;
  }
  if (!s || !out) throw Failure("apply: null pointer");
  require_device();
  const size_t in_total = (size_t)lead * (size_t)bins * (size_t)frames * (size_t)elem_bytes;
  const size_t out_total = (size_t)lead * (size_t)c.n_chroma * (size_t)frames * (size_t)elem_bytes;
  DeviceScratch ds(in_total), dout(out_total);
  copy_to_device(ds.ptr, s, in_total);
  chroma_apply_dev(c, ds.ptr, elem_bytes, lead, bins, frames, norm, norm_p, dout.ptr, nullptr);
  SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
  copy_to_host(out, dout.ptr, out_total);
}

void check_chroma_stft(const smx_stft_config &sc, const smx_chroma_config &cc) {   // soundml.ml:98-106
  if (sc.fft_size != cc.fft_size)
    throw InvalidArgument(format(
        "chroma_stft: cannot project a %lld-point STFT through a filterbank built for an FFT of size %lld (the two "
        "configurations must agree on fft_size)", (long long)sc.fft_size, (long long)cc.fft_size));
}

void chroma_stft_dev(const smx_stft_config &sc, const smx_chroma_config &cc, const void *d_x, int in_bytes,
                     int64_t lead, int64_t n, int64_t x_stride, double power, int norm, double norm_p, void *d_out,
                     hipStream_t stream) {
  check_chroma_stft(sc, cc);
  check_chroma_norm("apply", norm, norm_p);
  check_rank_extents("chroma_stft", lead, n);
  const int64_t count = sc.frames(n);
  if (lead == 0 || count == 0) return;
  if (!d_x || !d_out) throw Failure("chroma_stft: null device pointer");
  void *spec = nullptr;
  SMX_HIP_CHECK(smx::pool_malloc_async(&spec, (size_t)lead * (size_t)sc.bins() * (size_t)count * (size_t)in_bytes, stream));
  stft_range_dev(sc, d_x, in_bytes, lead, n, x_stride, 0, count, OUT_POWER, power, spec, stream);
  chroma_apply_dev(cc, spec, in_bytes, lead, sc.bins(), count, norm, norm_p, d_out, stream);
  SMX_HIP_CHECK(hipFreeAsync(spec, stream));
}

void chroma_stft_host(const smx_stft_config &sc, const smx_chroma_config &cc, const void *x, int in_bytes,
                      int64_t lead, int64_t n, double power, int norm, double norm_p, void *out) {
  check_chroma_stft(sc, cc);
  check_chroma_norm("apply", norm, norm_p);
  check_rank_extents("chroma_stft", lead, n);
  const int64_t count = sc.frames(n);
  if (lead == 0 || count == 0) return;
  if (!x || !out) throw Failure("chroma_stft: null pointer");
  require_device();
  const size_t in_total = (size_t)lead * (size_t)n * (size_t)in_bytes;
  const size_t out_total = (size_t)lead * (size_t)cc.n_chroma * (size_t)count * (size_t)in_bytes;
  DeviceScratch dx(in_total), dout(out_total);
  copy_to_device(dx.ptr, x, in_total);
  chroma_stft_dev(sc, cc, dx.ptr, in_bytes, lead, n, n, power, norm, norm_p, dout.ptr, nullptr);
  SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
  copy_to_host(out, dout.ptr, out_total);
}

}  // namespace
}  // namespace smx

extern "C" {

int smx_mel_config_create(int64_t n_mels, int64_t sample_rate, int64_t fft_size, double f_min,
                          int has_f_max, double f_max, int scale, int norm, smx_mel_config **out) {
  return guarded([&] {
    if (!out) throw Failure("create: null output handle");
    *out = mel_config_create(n_mels, sample_rate, fft_size, f_min, has_f_max != 0, f_max, scale, norm);
  });
}
int smx_mel_config_from_weights(int64_t rows, int64_t fft_size, const double *weights, smx_mel_config **out) {
  return guarded([&] {
    if (!out || !weights) throw Failure("from_weights: null pointer");
    if (rows < 1) throw InvalidArgument(format("from_weights: cannot build %lld filters (rows must be at least 1)", (long long)rows));
    if (fft_size < 1)
      throw InvalidArgument(format("from_weights: cannot use an FFT of size %lld (fft_size must be at least 1)", (long long)fft_size));
    smx_mel_config *c = new smx_mel_config();
    c->n_mels = rows;
    c->fft_size = fft_size;
    c->weights.assign(weights, weights + (size_t)rows * (size_t)(fft_size / 2 + 1));
    *out = c;
  });
}
void smx_mel_config_destroy(smx_mel_config *c) { delete c; }
int64_t smx_mel_config_n_mels(const smx_mel_config *c) { return c ? c->n_mels : -1; }
int64_t smx_mel_config_bins(const smx_mel_config *c) { return c ? c->bins() : -1; }
int64_t smx_mel_config_fft_size(const smx_mel_config *c) { return c ? c->fft_size : -1; }
double smx_mel_config_f_max(const smx_mel_config *c) { return c ? c->f_max : -1.0; }
int smx_mel_filterbank(const smx_mel_config *c, double *out) {
  return guarded([&] {
    check_config(c, "filterbank");
    std::memcpy(out, c->weights.data(), c->weights.size() * sizeof(double));
  });
}
int smx_mel_apply_f32(const smx_mel_config *c, const float *s, int64_t lead, int64_t bins,
                      int64_t frames, float *out) {
  return guarded([&] {
    check_config(c, "apply");
    mel_apply_host(*c, s, 4, lead, bins, frames, out);
  });
}
int smx_mel_apply_f64(const smx_mel_config *c, const double *s, int64_t lead, int64_t bins,
                      int64_t frames, double *out) {
  return guarded([&] {
    check_config(c, "apply");
    mel_apply_host(*c, s, 8, lead, bins, frames, out);
  });
}
int smx_mel_apply_f32_dev(const smx_mel_config *c, const float *d_s, int64_t lead, int64_t bins,
                          int64_t frames, float *d_out, void *stream) {
  return guarded([&] {
    check_config(c, "apply");
    mel_apply_dev(*c, d_s, 4, lead, bins, frames, d_out, (hipStream_t)stream);
  });
}
int smx_mel_apply_f64_dev(const smx_mel_config *c, const double *d_s, int64_t lead, int64_t bins,
                          int64_t frames, double *d_out, void *stream) {
  return guarded([&] {
    check_config(c, "apply");
    mel_apply_dev(*c, d_s, 8, lead, bins, frames, d_out, (hipStream_t)stream);
  });
}
int smx_mel_spectrogram_f32(const smx_stft_config *sc, const smx_mel_config *mc, const float *x,
                            int64_t lead, int64_t n, double power, float *out) {
  return guarded([&] {
    check_config(sc, "mel_spectrogram");
    check_config(mc, "mel_spectrogram");
    mel_spectrogram_host(*sc, *mc, x, 4, lead, n, power, out);
  });
}
int smx_mel_spectrogram_f64(const smx_stft_config *sc, const smx_mel_config *mc, const double *x,
                            int64_t lead, int64_t n, double power, double *out) {
  return guarded([&] {
    check_config(sc, "mel_spectrogram");
    check_config(mc, "mel_spectrogram");
    mel_spectrogram_host(*sc, *mc, x, 8, lead, n, power, out);
  });
}
int smx_mel_spectrogram_f32_dev(const smx_stft_config *sc, const smx_mel_config *mc, const float *d_x,
                                int64_t lead, int64_t n, int64_t x_stride, double power, float *d_out,
                                void *stream) {
  return guarded([&] {
    check_config(sc, "mel_spectrogram");
    check_config(mc, "mel_spectrogram");
    mel_spectrogram_dev(*sc, *mc, d_x, 4, lead, n, x_stride, power, d_out, (hipStream_t)stream);
  });
}


// ---- Stft.griffin_lim (stft.ml:961-1017) -----------------------------------------------------------
int smx_stft_griffin_lim_f32(const smx_stft_config *c, const float *s, int64_t lead, int64_t bins, int64_t frames,
                             int64_t n_iter, double momentum, const float *init_phase, int has_length,
                             int64_t length, float *out) {
  return guarded([&] {
    check_config(c, "griffin_lim");
    griffin_lim_host(*c, s, 4, lead, bins, frames, n_iter, momentum, init_phase, has_length, length, out);
  });
}
int smx_stft_griffin_lim_f64(const smx_stft_config *c, const double *s, int64_t lead, int64_t bins, int64_t frames,
                             int64_t n_iter, double momentum, const double *init_phase, int has_length,
                             int64_t length, double *out) {
  return guarded([&] {
    check_config(c, "griffin_lim");
    griffin_lim_host(*c, s, 8, lead, bins, frames, n_iter, momentum, init_phase, has_length, length, out);
  });
}
int smx_stft_griffin_lim_f32_dev(const smx_stft_config *c, const float *d_s, int64_t lead, int64_t bins,
                                 int64_t frames, int64_t n_iter, double momentum, const float *d_init_phase,
                                 int has_length, int64_t length, float *d_out, void *stream) {
  return guarded([&] {
    check_config(c, "griffin_lim");
    griffin_lim_dev(*c, d_s, 4, lead, bins, frames, n_iter, momentum, d_init_phase, has_length, length, d_out,
                    (hipStream_t)stream);
  });
}

// ---- Spectral.* (spectral.ml:171-255) ---------------------------------------------------------------
int smx_spectral_centroid_f32(const float *s, int64_t lead, int64_t bins, int64_t frames, const double *freqs, int64_t n_freqs, int64_t sample_rate, float *out) {
  return guarded([&] {
    SpectralParams q;
    q.feature = SPECTRAL_CENTROID;
    q.freqs = freqs;
    q.n_freqs = n_freqs;
    q.sample_rate = sample_rate;
    spectral_host(q, s, 4, lead, bins, frames, nullptr, out);
  });
}
int smx_spectral_centroid_f64(const double *s, int64_t lead, int64_t bins, int64_t frames, const double *freqs, int64_t n_freqs, int64_t sample_rate, double *out) {
  return guarded([&] {
    SpectralParams q;
    q.feature = SPECTRAL_CENTROID;
    q.freqs = freqs;
    q.n_freqs = n_freqs;
    q.sample_rate = sample_rate;
    spectral_host(q, s, 8, lead, bins, frames, nullptr, out);
  });
}
int smx_spectral_centroid_f32_dev(const float *d_s, int64_t lead, int64_t bins, int64_t frames, const double *freqs, int64_t n_freqs, int64_t sample_rate, float *d_out, void *stream) {
  return guarded([&] {
    SpectralParams q;
    q.feature = SPECTRAL_CENTROID;
    q.freqs = freqs;
    q.n_freqs = n_freqs;
    q.sample_rate = sample_rate;
    spectral_dev(q, d_s, 4, lead, bins, frames, nullptr, d_out, (hipStream_t)stream);
  });
}
int smx_spectral_bandwidth_f32(const float *s, int64_t lead, int64_t bins, int64_t frames, double p, const double *freqs, int64_t n_freqs, const float *centroid, int64_t c_rows, int64_t c_frames, int64_t sample_rate, float *out) {
  return guarded([&] {
    SpectralParams q;
    q.feature = SPECTRAL_BANDWIDTH;
    q.p = p;
    q.freqs = freqs;
    q.n_freqs = n_freqs;
    q.sample_rate = sample_rate;
    q.has_centroid = centroid != nullptr;
    q.c_rows = c_rows;
    q.c_frames = c_frames;
    spectral_host(q, s, 4, lead, bins, frames, centroid, out);
  });
}
int smx_spectral_bandwidth_f64(const double *s, int64_t lead, int64_t bins, int64_t frames, double p, const double *freqs, int64_t n_freqs, const double *centroid, int64_t c_rows, int64_t c_frames, int64_t sample_rate, double *out) {
  return guarded([&] {
    SpectralParams q;
    q.feature = SPECTRAL_BANDWIDTH;
    q.p = p;
    q.freqs = freqs;
    q.n_freqs = n_freqs;
    q.sample_rate = sample_rate;
    q.has_centroid = centroid != nullptr;
    q.c_rows = c_rows;
    q.c_frames = c_frames;
    spectral_host(q, s, 8, lead, bins, frames, centroid, out);
  });
}
int smx_spectral_bandwidth_f32_dev(const float *d_s, int64_t lead, int64_t bins, int64_t frames, double p, const double *freqs, int64_t n_freqs, const float *d_centroid, int64_t c_rows, int64_t c_frames, int64_t sample_rate, float *d_out, void *stream) {
  return guarded([&] {
    SpectralParams q;
    q.feature = SPECTRAL_BANDWIDTH;
    q.p = p;
    q.freqs = freqs;
    q.n_freqs = n_freqs;
    q.sample_rate = sample_rate;
    q.has_centroid = d_centroid != nullptr;
    q.c_rows = c_rows;
    q.c_frames = c_frames;
    spectral_dev(q, d_s, 4, lead, bins, frames, d_centroid, d_out, (hipStream_t)stream);
  });
}
int smx_spectral_rolloff_f32(const float *s, int64_t lead, int64_t bins, int64_t frames, double roll_percent, const double *freqs, int64_t n_freqs, int64_t sample_rate, float *out) {
  return guarded([&] {
    SpectralParams q;
    q.feature = SPECTRAL_ROLLOFF;
    q.p = roll_percent;
    q.freqs = freqs;
    q.n_freqs = n_freqs;
    q.sample_rate = sample_rate;
    spectral_host(q, s, 4, lead, bins, frames, nullptr, out);
  });
}
int smx_spectral_rolloff_f64(const double *s, int64_t lead, int64_t bins, int64_t frames, double roll_percent, const double *freqs, int64_t n_freqs, int64_t sample_rate, double *out) {
  return guarded([&] {
    SpectralParams q;
    q.feature = SPECTRAL_ROLLOFF;
    q.p = roll_percent;
    q.freqs = freqs;
    q.n_freqs = n_freqs;
    q.sample_rate = sample_rate;
    spectral_host(q, s, 8, lead, bins, frames, nullptr, out);
  });
}
int smx_spectral_rolloff_f32_dev(const float *d_s, int64_t lead, int64_t bins, int64_t frames, double roll_percent, const double *freqs, int64_t n_freqs, int64_t sample_rate, float *d_out, void *stream) {
  return guarded([&] {
    SpectralParams q;
    q.feature = SPECTRAL_ROLLOFF;
    q.p = roll_percent;
    q.freqs = freqs;
    q.n_freqs = n_freqs;
    q.sample_rate = sample_rate;
    spectral_dev(q, d_s, 4, lead, bins, frames, nullptr, d_out, (hipStream_t)stream);
  });
}
int smx_spectral_flatness_f32(const float *s, int64_t lead, int64_t bins, int64_t frames, double amin, double power, float *out) {
  return guarded([&] {
    SpectralParams q;
    q.feature = SPECTRAL_FLATNESS;
    q.p = amin;
    q.power = power;
    spectral_host(q, s, 4, lead, bins, frames, nullptr, out);
  });
}
int smx_spectral_flatness_f64(const double *s, int64_t lead, int64_t bins, int64_t frames, double amin, double power, double *out) {
  return guarded([&] {
    SpectralParams q;
    q.feature = SPECTRAL_FLATNESS;
    q.p = amin;
    q.power = power;
    spectral_host(q, s, 8, lead, bins, frames, nullptr, out);
  });
}
int smx_spectral_flatness_f32_dev(const float *d_s, int64_t lead, int64_t bins, int64_t frames, double amin, double power, float *d_out, void *stream) {
  return guarded([&] {
    SpectralParams q;
    q.feature = SPECTRAL_FLATNESS;
    q.p = amin;
    q.power = power;
    spectral_dev(q, d_s, 4, lead, bins, frames, nullptr, d_out, (hipStream_t)stream);
  });
}

// ---- Chroma (chroma.ml:95-317, soundml.ml:97-107) ---------------------------------------------------
int smx_chroma_config_create(int64_t n_chroma, double tuning, double ctroct, int has_octwidth, double octwidth,
                             int base_c, int64_t sample_rate, int64_t fft_size, smx_chroma_config **out) {
  return guarded([&] {
    if (!out) throw Failure("create: null output handle");
    *out = chroma_config_create(n_chroma, tuning, ctroct, has_octwidth != 0, octwidth, base_c != 0, sample_rate, fft_size);
  });
}
void smx_chroma_config_destroy(smx_chroma_config *c) { delete c; }
int64_t smx_chroma_config_n_chroma(const smx_chroma_config *c) { return c ? c->n_chroma : -1; }
int64_t smx_chroma_config_bins(const smx_chroma_config *c) { return c ? c->bins() : -1; }
int64_t smx_chroma_config_fft_size(const smx_chroma_config *c) { return c ? c->fft_size : -1; }
int smx_chroma_filterbank(const smx_chroma_config *c, double *out) {
  return guarded([&] {
    check_config(c, "filterbank");
    if (!out) throw Failure("filterbank: null pointer");
    std::memcpy(out, c->weights.data(), c->weights.size() * sizeof(double));
  });
}
int smx_chroma_apply_f32(const smx_chroma_config *c, const float *s, int64_t lead, int64_t bins, int64_t frames,
                         int norm, double norm_p, float *out) {
  return guarded([&] {
    check_config(c, "apply");
    chroma_apply_host(*c, s, 4, lead, bins, frames, norm, norm_p, out);
  });
}
int smx_chroma_apply_f64(const smx_chroma_config *c, const double *s, int64_t lead, int64_t bins, int64_t frames,
                         int norm, double norm_p, double *out) {
  return guarded([&] {
    check_config(c, "apply");
    chroma_apply_host(*c, s, 8, lead, bins, frames, norm, norm_p, out);
  });
}
int smx_chroma_apply_f32_dev(const smx_chroma_config *c, const float *d_s, int64_t lead, int64_t bins,
                             int64_t frames, int norm, double norm_p, float *d_out, void *stream) {
  return guarded([&] {
    check_config(c, "apply");
    chroma_apply_dev(*c, d_s, 4, lead, bins, frames, norm, norm_p, d_out, (hipStream_t)stream);
  });
}
int smx_chroma_stft_f32(const smx_stft_config *sc, const smx_chroma_config *cc, const float *x, int64_t lead,
                        int64_t n, double power, int norm, double norm_p, float *out) {
  return guarded([&] {
    check_config(sc, "chroma_stft");
    check_config(cc, "chroma_stft");
    chroma_stft_host(*sc, *cc, x, 4, lead, n, power, norm, norm_p, out);
  });
}
int smx_chroma_stft_f64(const smx_stft_config *sc, const smx_chroma_config *cc, const double *x, int64_t lead,
                        int64_t n, double power, int norm, double norm_p, double *out) {
  return guarded([&] {
    check_config(sc, "chroma_stft");
    check_config(cc, "chroma_stft");
    chroma_stft_host(*sc, *cc, x, 8, lead, n, power, norm, norm_p, out);
  });
}
int smx_chroma_stft_f32_dev(const smx_stft_config *sc, const smx_chroma_config *cc, const float *d_x,
                            int64_t lead, int64_t n, int64_t x_stride, double power, int norm, double norm_p,
                            float *d_out, void *stream) {
  return guarded([&] {
    check_config(sc, "chroma_stft");
    check_config(cc, "chroma_stft");
    chroma_stft_dev(*sc, *cc, d_x, 4, lead, n, x_stride, power, norm, norm_p, d_out, (hipStream_t)stream);
  });
}

// ---- Convert.power_to_db / amplitude_to_db (convert.ml:52-62) ------------------------------------------
int smx_power_to_db_f32(const float *s, int64_t total, double reference, double amin, int has_top_db, double top_db, float *out) {
  return guarded([&] { to_db_host(false, s, 4, total, reference, amin, has_top_db, top_db, out); });
}
int smx_power_to_db_f64(const double *s, int64_t total, double reference, double amin, int has_top_db, double top_db, double *out) {
  return guarded([&] { to_db_host(false, s, 8, total, reference, amin, has_top_db, top_db, out); });
}
int smx_power_to_db_f32_dev(const float *d_s, int64_t total, double reference, double amin, int has_top_db, double top_db,
                            float *d_out, void *stream) {
  return guarded([&] { to_db_dev(false, d_s, 4, total, reference, amin, has_top_db, top_db, d_out, (hipStream_t)stream); });
}
int smx_amplitude_to_db_f32(const float *s, int64_t total, double reference, double amin, int has_top_db, double top_db, float *out) {
  return guarded([&] { to_db_host(true, s, 4, total, reference, amin, has_top_db, top_db, out); });
}
int smx_amplitude_to_db_f64(const double *s, int64_t total, double reference, double amin, int has_top_db, double top_db,
                            double *out) {
  return guarded([&] { to_db_host(true, s, 8, total, reference, amin, has_top_db, top_db, out); });
}
int smx_amplitude_to_db_f32_dev(const float *d_s, int64_t total, double reference, double amin, int has_top_db, double top_db,
                                float *d_out, void *stream) {
  return guarded([&] { to_db_dev(true, d_s, 4, total, reference, amin, has_top_db, top_db, d_out, (hipStream_t)stream); });
}

// ---- Soundml.mfcc (soundml.ml:50-95) --------------------------------------------------------------
int smx_mfcc_f32(const smx_stft_config *sc, const smx_mel_config *mc, const float *x, int64_t lead, int64_t n,
                 int64_t n_mfcc, int has_lifter, double lifter, float *out) {
  return guarded([&] {
    check_config(sc, "mfcc");
    check_config(mc, "mfcc");
    mfcc_host(*sc, *mc, x, 4, lead, n, n_mfcc, has_lifter, lifter, out);
  });
}
int smx_mfcc_f64(const smx_stft_config *sc, const smx_mel_config *mc, const double *x, int64_t lead, int64_t n,
                 int64_t n_mfcc, int has_lifter, double lifter, double *out) {
  return guarded([&] {
    check_config(sc, "mfcc");
    check_config(mc, "mfcc");
    mfcc_host(*sc, *mc, x, 8, lead, n, n_mfcc, has_lifter, lifter, out);
  });
}
int smx_mfcc_f32_dev(const smx_stft_config *sc, const smx_mel_config *mc, const float *d_x, int64_t lead, int64_t n,
                     int64_t x_stride, int64_t n_mfcc, int has_lifter, double lifter, float *d_out, void *stream) {
  return guarded([&] {
    check_config(sc, "mfcc");
    check_config(mc, "mfcc");
    mfcc_dev(*sc, *mc, d_x, 4, lead, n, x_stride, n_mfcc, has_lifter, lifter, d_out, (hipStream_t)stream);
  });
}

// ---- least-squares synthesis: Stft.invert (stft.ml:902-939) ----------------------------------------
int smx_stft_nola(const smx_stft_config *c, int *invertible) {
  return guarded([&] {
    check_config(c, "nola");
    if (invertible) *invertible = stft_nola(*c) ? 1 : 0;
  });
}
int smx_stft_output_length(const smx_stft_config *c, int64_t frames, int64_t *length) {
  return guarded([&] {
    check_config(c, "output_length");
    if (frames < 0) throw Failure("output_length: negative frame count");
    if (length) *length = stft_output_length(*c, frames);
  });
}
int smx_stft_invert_f32(const smx_stft_config *c, const float *z, int64_t lead, int64_t bins, int64_t frames,
                        int has_length, int64_t length, float *out) {
  return guarded([&] {
    check_config(c, "invert");
    invert_host(*c, z, 8, lead, bins, frames, has_length, length, out);
  });
}
int smx_stft_invert_f64(const smx_stft_config *c, const double *z, int64_t lead, int64_t bins, int64_t frames,
                        int has_length, int64_t length, double *out) {
  return guarded([&] {
    check_config(c, "invert");
    invert_host(*c, z, 16, lead, bins, frames, has_length, length, out);
  });
}
int smx_stft_invert_f32_dev(const smx_stft_config *c, const float *d_z, int64_t lead, int64_t bins, int64_t frames,
                            int has_length, int64_t length, float *d_out, void *stream) {
  return guarded([&] {
    check_config(c, "invert");
    invert_dev(*c, d_z, 8, lead, bins, frames, has_length, length, d_out, (hipStream_t)stream);
  });
}
int smx_stft_invert_f64_dev(const smx_stft_config *c, const double *d_z, int64_t lead, int64_t bins, int64_t frames,
                            int has_length, int64_t length, double *d_out, void *stream) {
  return guarded([&] {
    check_config(c, "invert");
    invert_dev(*c, d_z, 16, lead, bins, frames, has_length, length, d_out, (hipStream_t)stream);
  });
}

}  // extern "C"
