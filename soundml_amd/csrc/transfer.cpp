// Host <-> device transfers of the host-pointer entry points (the drop-in boundary: nx hands over pageable host
// tensors, soundml/lib/stft.ml works on host memory throughout).  A plain hipMemcpy of pageable memory runs at
// 10-13 GB/s on this platform and most of a C2 call's 116 ms went there; large transfers go instead through a ring
// of pinned staging buffers, the DMA engine running asynchronously while a few host threads move the previous chunk
// between the staging buffer and the caller's (possibly never-touched, page-faulting) memory.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "smx_internal.hpp"

namespace smx {
namespace {

constexpr size_t kChunk = (size_t)16 << 20;   // bytes per staging buffer
constexpr int kRing = 4;                      // staging buffers in flight
constexpr size_t kDirect = (size_t)4 << 20;   // below this a plain hipMemcpy is as fast

// Staging rings: a pool PER DEVICE AND DIRECTION (round 6; rounds 2-5 kept one ring pair per process behind one mutex, so eight host
// threads on eight GPUs -- smx_set_devices -- would have taken turns on every upload).  A staged transfer leases a ring of its
// device for its duration; a second concurrent transfer on the same device (two host threads, or one device listed twice in
// smx_set_devices) gets a ring of its own, up to kMaxRings per device and direction, after which callers wait for a free one.
// Every MI355X has its own PCIe Gen5 link: transfers of different devices never meet here.
struct Ring {
  void *buf[kRing] = {};
  bool busy = false;
  bool allocate() {
    for (int i = 0; i < kRing; ++i)
      if (hipHostMalloc(&buf[i], kChunk, hipHostMallocPortable) != hipSuccess) {
        for (int j = 0; j < i; ++j) (void)hipHostFree(buf[j]);
        buf[0] = nullptr;
        (void)hipGetLastError();
        return false;
      }
    return true;
  }
};
constexpr int kMaxRings = 4;       // per device and direction (64 MB of page-locked memory each, allocated on first use)
constexpr int kMaxDevices = 64;
struct RingPool {
  std::mutex mu;
  std::condition_variable freed;
  std::vector<std::unique_ptr<Ring>> rings;
};
RingPool &ring_pool(int device, bool to_host) {
  static RingPool *pools = new RingPool[2 * kMaxDevices];   // (never destroyed: see pinned_pool)
  return pools[2 * (device >= 0 && device < kMaxDevices ? device : 0) + (to_host ? 1 : 0)];
}
// how many staged transfers hold a ring right now, and the most that ever did at once (smx_debug_staging_peak: the tests of the
// per-device staging look at it)
std::atomic<int> g_staged_now[2] = {{0}, {0}}, g_staged_peak[2] = {{0}, {0}};
struct RingLease {
  RingPool *pool = nullptr;
  Ring *ring = nullptr;
  int dir = 0;
  // a free ring of the current device, a new one while the pool is below kMaxRings, else the next one released; nullptr when no
  // page-locked memory is to be had at all
  RingLease(bool to_host) : dir(to_host ? 1 : 0) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); device = 0; }
    pool = &ring_pool(device, to_host);
    std::unique_lock<std::mutex> g(pool->mu);
    for (;;) {
      for (auto &r : pool->rings)
        if (!r->busy) { ring = r.get(); break; }
      if (ring) break;
      if ((int)pool->rings.size() < kMaxRings) {
        auto r = std::make_unique<Ring>();
        if (r->allocate()) {
          ring = r.get();
          pool->rings.push_back(std::move(r));
          break;
        }
        if (pool->rings.empty()) return;   // nothing allocated, nothing to wait for
      }
      pool->freed.wait(g);
    }
    ring->busy = true;
    const int now = g_staged_now[dir].fetch_add(1) + 1;
    int peak = g_staged_peak[dir].load();
    while (now > peak && !g_staged_peak[dir].compare_exchange_weak(peak, now)) {}
  }
  ~RingLease() {
    if (!ring) return;
    g_staged_now[dir].fetch_sub(1);
    { std::lock_guard<std::mutex> g(pool->mu); ring->busy = false; }
    pool->freed.notify_one();
  }
  RingLease(const RingLease &) = delete;
  RingLease &operator=(const RingLease &) = delete;
};

// What a pipelined host call hangs on a staged transfer (pipelined_host_call below): the upload reports every DMA it has
// enqueued, the download asks before it enqueues one.
struct Hooks {
  std::function<void(long chunk, hipStream_t stream)> enqueued;   // upload: the DMA of chunk `chunk` is on `stream`
  std::function<void(long chunk, hipStream_t stream)> before;     // download: called before the DMA of chunk `chunk` goes on `stream`
};

// what a stage of a pipelined call throws when it stops because ANOTHER stage failed (never the root cause)
struct StageAborted : Failure {
  using Failure::Failure;
};

// the CPU quota of the process's cgroup in cores (cgroup v2 cpu.max = "<quota> <period>"; 0 = none or unreadable).  The GPU boxes
// show 256 logical CPUs and grant 16: threads beyond the quota are throttled for whole scheduler periods (a 25 ms call took 80).
int cgroup_cpu_quota() {
  FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r");
  if (!f) return 0;
  long long quota = 0, period = 0;
  char first[32] = {0};
  int cores = 0;
  if (std::fscanf(f, "%31s %lld", first, &period) == 2 && std::strcmp(first, "max") != 0 && period > 0) {
    quota = std::atoll(first);
    cores = (int)((quota + period - 1) / period);
  }
  std::fclose(f);
  return cores;
}

int worker_count_whole() {
  static const int n = [] {
    if (env_flag("SMX_COPY_THREADS") >= 0) return std::max(1, std::min(64, (int)env_int("SMX_COPY_THREADS", 16)));
    const unsigned hw = std::thread::hardware_concurrency();
    int t = (int)std::max(1u, std::min(16u, hw / 2));
    const int quota = cgroup_cpu_quota();
    if (quota > 0) t = std::max(1, std::min(t, quota - 4));   // the DMA drivers, the launching thread and the HIP runtime's own keep a share
    return t;
  }();
  return n;
}
// the shards of a device-list call (capi.cpp: for_each_shard) run side by side on the same host cores: each takes its share of
// the copying threads
thread_local int t_transfer_share = 1;
int worker_count() { return std::max(1, worker_count_whole() / std::max(1, t_transfer_share)); }

struct StreamAndEvents {
  hipStream_t stream = nullptr;
  hipEvent_t ev[kRing] = {};
  StreamAndEvents() {
    SMX_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    // (blocking waits: a thread that spins in hipEventSynchronize through a DMA spends CPU quota the copying threads need)
    for (auto &e : ev) SMX_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync));
  }
  ~StreamAndEvents() {
    for (auto &e : ev)
      if (e) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

// (a short spin, then sleeps: the GPU boxes give a process a CPU quota -- 16 cores of 256 -- and a thread that spins through a DMA's
// milliseconds spends the quota its copying neighbours need; with two transfers in flight that throttled the whole call)
inline void wait_for(const std::atomic<long> &v, long target) {
  int spins = 0;
  while (v.load(std::memory_order_acquire) < target) {
    if (++spins > 256) std::this_thread::sleep_for(std::chrono::microseconds(30));
    else if (spins > 64) std::this_thread::yield();
  }
}

// The slice of chunk `i` that worker `w` of `t` moves (64-byte aligned cuts).
inline void slice(size_t len, int w, int t, size_t &lo, size_t &hi) {
  const size_t per = ((len + (size_t)t - 1) / (size_t)t + 63) & ~(size_t)63;
  lo = std::min(len, per * (size_t)w);
  hi = std::min(len, lo + per);
}

// A freshly allocated result array is untouched memory: every 4 KB page of it faults as the copying threads reach it, and that,
// not PCIe, bounds the download (0.98 GB of C2 spectrogram: 240 000 faults).  Ask for transparent huge pages on the whole 2 MB
// pieces of the destination first (a no-op where the kernel has them off or the pages exist already).
void advise_huge(void *dst, size_t bytes) {
  const uintptr_t two_mb = (uintptr_t)2 << 20;
  const uintptr_t lo = (reinterpret_cast<uintptr_t>(dst) + two_mb - 1) & ~(two_mb - 1), hi = (reinterpret_cast<uintptr_t>(dst) + bytes) & ~(two_mb - 1);
  if (hi > lo) (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
}

// ---- pinned result arrays (smx_host_alloc / smx_host_free) -------------------------------------------------------------------
// The reference's faces return a FRESH host tensor per call (stft.ml:356-364: `analyse` allocates its result).  Fresh pageable memory
// is the slow half of a host call: every page of a 0.98 GB spectrogram faults and is zeroed by the kernel as the copying threads reach
// it, and the bytes cross memory twice (DMA into the staging ring, then memcpy).  A caller that takes its result array from this
// pool instead gets page-locked memory the DMA engine writes directly -- no staging, no faults -- and a released block is kept for
// the next result (at most kPinnedCacheMax bytes are held while unused).
constexpr size_t kPinnedCacheMax = (size_t)3 << 30;
struct PinnedPool {
  std::mutex mu;
  std::map<uintptr_t, size_t> live;                      // blocks handed out: base -> capacity
  std::vector<std::pair<void *, size_t>> cached;         // released blocks, oldest first
  size_t cached_bytes = 0;
};
PinnedPool &pinned_pool() {
  static PinnedPool *p = new PinnedPool;   // (never destroyed: a block may outlive the HIP runtime's own teardown order)
  return *p;
}

// Device -> page-locked host memory by the shader engines (the block is mapped into the device's address space).  Why not the DMA
// engine: an upload and a download that both go through hipMemcpyAsync overlap in a bare two-stream test (0.49 GB up + 0.98 GB
// down: 9.4 / 19.4 ms) but ran one after the other inside a pipelined call (12 / 25.5 ms = the sum of the two alone); with the
// download on a copy kernel only the upload needs a DMA engine and the two directions share the link: 12.6 / 18.9 ms
// (tools/probes/pcie_duplex_probe.hip, profiles/r07/pcie_duplex_probe.log).
__global__ void __launch_bounds__(256) copy_to_host_kernel(uint4 *dst, const uint4 *src, size_t n16, unsigned char *dst_tail, const unsigned char *src_tail, int tail) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
  if (blockIdx.x == 0 && (int)threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
hipError_t copy_to_host_by_kernel(void *dst, const void *src, size_t len, hipStream_t stream) {
  if ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) return hipMemcpyAsync(dst, src, len, hipMemcpyDeviceToHost, stream);
  const size_t n16 = len / 16;
  hipLaunchKernelGGL(copy_to_host_kernel, dim3(64), dim3(256), 0, stream, reinterpret_cast<uint4 *>(dst), reinterpret_cast<const uint4 *>(src), n16,
                     reinterpret_cast<unsigned char *>(dst) + n16 * 16, reinterpret_cast<const unsigned char *>(src) + n16 * 16, (int)(len - n16 * 16));
  return hipGetLastError();
}

// to_host: device -> host, !to_host: the reverse, with the host side page-locked: no staging, chunk by chunk so that a pipelined
// call's hooks see the same chunk indices as the staged form
void direct(void *dst, const void *src, size_t bytes, bool to_host, const Hooks *hooks) {
  StreamAndEvents se;
  const long chunks = (long)((bytes + kChunk - 1) / kChunk);
  try {
    for (long i = 0; i < chunks; ++i) {
      const size_t off = (size_t)i * kChunk, len = std::min(kChunk, bytes - off);
      if (hooks && hooks->before) hooks->before(i, se.stream);
      if (to_host) SMX_HIP_CHECK(copy_to_host_by_kernel((unsigned char *)dst + off, (const unsigned char *)src + off, len, se.stream));
      else SMX_HIP_CHECK(hipMemcpyAsync((unsigned char *)dst + off, (const unsigned char *)src + off, len, hipMemcpyHostToDevice, se.stream));
      if (hooks && hooks->enqueued) hooks->enqueued(i, se.stream);
    }
  } catch (...) {
    (void)hipStreamSynchronize(se.stream);   // nothing of this transfer is in flight once the caller unwinds (and frees what it touches)
    throw;
  }
  SMX_HIP_CHECK(hipStreamSynchronize(se.stream));
}

// to_host: device -> staging by DMA, staging -> user by the workers.  !to_host: the reverse.
void staged(void *dst, const void *src, size_t bytes, bool to_host, int workers = 0, const Hooks *hooks = nullptr) {
  if (host_is_pinned(to_host ? dst : src, bytes)) {
    direct(dst, src, bytes, to_host, hooks);
    return;
  }
  RingLease lease(to_host);
  if (!lease.ring) {   // no pinned memory to be had: the plain copy still works
    // (a pipelined call checks staging_available() before it starts its threads; this is the race where the memory went in between)
    if (hooks) throw Failure("pipelined transfer: no pinned staging memory");
    SMX_HIP_CHECK(hipMemcpy(dst, src, bytes, to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice));
    return;
  }
  Ring &g_ring = *lease.ring;
  StreamAndEvents se;
  const long chunks = (long)((bytes + kChunk - 1) / kChunk);
  const int t = workers > 0 ? workers : worker_count();
  auto len_of = [&](long i) { return std::min(kChunk, bytes - (size_t)i * kChunk); };
  std::atomic<long> ready{0};            // to_host: chunks whose DMA has landed; else: chunks whose buffer is free again
  std::vector<std::atomic<long>> done(chunks);   // workers finished with chunk i
  for (auto &d : done) d.store(0, std::memory_order_relaxed);
  std::atomic<bool> failed{false};
  auto work = [&](int w) {
    for (long i = 0; i < chunks; ++i) {
      wait_for(ready, to_host ? i + 1 : i - kRing + 1);
      if (failed.load()) return;
      size_t lo, hi;
      slice(len_of(i), w, t, lo, hi);
      unsigned char *stage = (unsigned char *)g_ring.buf[i % kRing];
      if (hi > lo) {
        if (to_host) std::memcpy((unsigned char *)dst + (size_t)i * kChunk + lo, stage + lo, hi - lo);
        else std::memcpy(stage + lo, (const unsigned char *)src + (size_t)i * kChunk + lo, hi - lo);
      }
      done[i].fetch_add(1, std::memory_order_release);
    }
  };
  // the calling thread only drives the DMA ring, so the ring never waits for a share of the copying
  std::vector<std::thread> pool;
  for (int w = 0; w < t; ++w) pool.emplace_back(work, w);
  hipError_t err = hipSuccess;
  std::exception_ptr hook_error;
  auto dma = [&](long i) {
    void *stage = g_ring.buf[i % kRing];
    auto hook = [&](const std::function<void(long, hipStream_t)> &f) {   // (a hook's failure ends the transfer like a HIP error: the workers are joined first)
      if (err != hipSuccess || !f) return;
      try {
        f(i, se.stream);
      } catch (...) {
        hook_error = std::current_exception();
        err = hipErrorUnknown;
      }
    };
    if (hooks) hook(hooks->before);
    if (err == hipSuccess)
      err = to_host ? hipMemcpyAsync(stage, (const unsigned char *)src + (size_t)i * kChunk, len_of(i), hipMemcpyDeviceToHost, se.stream)
                    : hipMemcpyAsync((unsigned char *)dst + (size_t)i * kChunk, stage, len_of(i), hipMemcpyHostToDevice, se.stream);
    if (err == hipSuccess) err = hipEventRecord(se.ev[i % kRing], se.stream);
    if (hooks) hook(hooks->enqueued);
  };
  auto workers_done_with = [&](long i) {
    int spins = 0;
    while (done[i].load(std::memory_order_acquire) < t) {
      if (++spins > 256) std::this_thread::sleep_for(std::chrono::microseconds(30));
      else if (spins > 64) std::this_thread::yield();
    }
  };
  if (to_host) {
    // `ready` = chunks whose DMA has landed in their staging buffer
    for (long i = 0; i < std::min<long>(kRing, chunks); ++i) dma(i);
    for (long i = 0; i < chunks && err == hipSuccess; ++i) {
      err = hipEventSynchronize(se.ev[i % kRing]);
      ready.store(i + 1, std::memory_order_release);
      if (i + kRing < chunks) {   // the buffer is refilled once every worker has emptied it
        workers_done_with(i);
        dma(i + kRing);
      }
    }
  } else {
    // `ready` = chunks whose DMA has left their staging buffer (the workers fill buffer i % kRing for chunk i once
    // chunk i - kRing has gone: ready >= i - kRing + 1)
    for (long i = 0; i < chunks && err == hipSuccess; ++i) {
      workers_done_with(i);
      dma(i);
      const long gone = i + 1 - kRing;   // the oldest chunk still in flight
      if (gone >= 0 && err == hipSuccess) {
        err = hipEventSynchronize(se.ev[gone % kRing]);
        ready.store(gone + 1, std::memory_order_release);
      }
    }
  }
  if (err != hipSuccess) {
    failed.store(true);
    ready.store(chunks + kRing, std::memory_order_release);
  }
  for (auto &th : pool) th.join();
  // on every path, also the failing ones: no DMA of this transfer is in flight when the ring goes back to the pool and the
  // caller unwinds (it frees the device arrays this transfer reads / writes)
  const hipError_t sync_err = hipStreamSynchronize(se.stream);
  if (err == hipSuccess) err = sync_err;
  if (hook_error) std::rethrow_exception(hook_error);
  SMX_HIP_CHECK(err);
}

}  // namespace

void *host_alloc(size_t bytes) {
  if (bytes == 0) bytes = 1;
  const size_t want = (bytes + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1);
  PinnedPool &pp = pinned_pool();
  {
    std::lock_guard<std::mutex> g(pp.mu);
    for (size_t i = pp.cached.size(); i-- > 0;)   // the most recently released first
      if (pp.cached[i].second >= want && pp.cached[i].second <= want + want / 4) {   // a released block of about this size
        void *p = pp.cached[i].first;
        pp.live[reinterpret_cast<uintptr_t>(p)] = pp.cached[i].second;
        pp.cached_bytes -= pp.cached[i].second;
        pp.cached.erase(pp.cached.begin() + (long)i);
        return p;
      }
  }
  void *p = nullptr;
  if (hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    std::vector<void *> drop;
    {   // make room: the cache goes first
      std::lock_guard<std::mutex> g(pp.mu);
      for (auto &c : pp.cached) drop.push_back(c.first);
      pp.cached.clear();
      pp.cached_bytes = 0;
    }
    for (void *d : drop) (void)hipHostFree(d);
    SMX_HIP_CHECK(hipHostMalloc(&p, want, hipHostMallocPortable));
  }
  std::lock_guard<std::mutex> g(pp.mu);
  pp.live[reinterpret_cast<uintptr_t>(p)] = want;
  return p;
}

void host_free(void *p) {
  if (!p) return;
  PinnedPool &pp = pinned_pool();
  std::vector<void *> drop;
  {
    std::lock_guard<std::mutex> g(pp.mu);
    auto it = pp.live.find(reinterpret_cast<uintptr_t>(p));
    if (it == pp.live.end()) throw Failure("smx_host_free: not a block of smx_host_alloc");
    pp.cached.emplace_back(p, it->second);
    pp.cached_bytes += it->second;
    pp.live.erase(it);
    while (pp.cached_bytes > kPinnedCacheMax && !pp.cached.empty()) {
      drop.push_back(pp.cached.front().first);
      pp.cached_bytes -= pp.cached.front().second;
      pp.cached.erase(pp.cached.begin());
    }
  }
  for (void *d : drop) (void)hipHostFree(d);
}

bool host_is_pinned(const void *p, size_t bytes) {
  PinnedPool &pp = pinned_pool();
  std::lock_guard<std::mutex> g(pp.mu);
  if (pp.live.empty()) return false;
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  auto it = pp.live.upper_bound(a);
  if (it == pp.live.begin()) return false;
  --it;
  return a >= it->first && a + bytes <= it->first + it->second;
}

// A host call cut into units of clips (the reference's leading axes are independent: stft.mli:214-218, so a unit's result is
// the slice of the whole call's, bit for bit): the upload of the units ahead, the kernels of one and the download of those
// behind run at the same time -- PCIe is full duplex, and serially a C2 power spectrogram spent 10 ms going up and 20 ms coming
// down around 0.6 ms of kernel (round 5).  Three host threads: one drives the upload (its own staging ring and copying
// threads), one the download (the same), the caller launches; they meet at HIP events -- a kernel waits for the DMA that
// completes its input, a download DMA for the kernel that completes its output.
//   launch(clip0, nclips, stream) enqueues the work of clips [clip0, clip0 + nclips) on `stream`, reading d_in and writing d_out
//   at the clips' own offsets.
void pipelined_host_call(const void *src, size_t in_clip_bytes, void *dst, size_t out_clip_bytes, int64_t clips, int64_t unit,
                         void *d_in, void *d_out, const std::function<void(int64_t, int64_t, hipStream_t)> &launch) {
  int dev = 0;
  SMX_HIP_CHECK(hipGetDevice(&dev));
  const size_t in_total = (size_t)clips * in_clip_bytes, out_total = (size_t)clips * out_clip_bytes;
  const bool dst_pinned = host_is_pinned(dst, out_total), src_pinned = host_is_pinned(src, in_total);
  if (!dst_pinned) advise_huge(dst, out_total);
  const long in_chunks = (long)((in_total + kChunk - 1) / kChunk);
  const int64_t units = (clips + unit - 1) / unit;
  std::vector<hipEvent_t> up_ev((size_t)in_chunks, nullptr), k_ev((size_t)units, nullptr);
  hipStream_t compute = nullptr;
  auto cleanup = [&] {
    for (auto e : up_ev) if (e) (void)hipEventDestroy(e);
    for (auto e : k_ev) if (e) (void)hipEventDestroy(e);
    if (compute) (void)hipStreamDestroy(compute);
  };
  try {
    for (auto &e : up_ev) SMX_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : k_ev) SMX_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    SMX_HIP_CHECK(hipStreamCreateWithFlags(&compute, hipStreamNonBlocking));
  } catch (...) {
    cleanup();
    throw;
  }
  // progress of the three stages, waited for WITHOUT spinning (the box's CPU quota is what bounds the copying threads; a spinning
  // waiter takes a core from them)
  std::mutex mu;
  std::condition_variable cv;
  long up_enqueued = 0, launched = 0;
  bool failed = false;
  std::exception_ptr err_up, err_down, err_main;
  auto publish = [&](long &v, long value) {
    { std::lock_guard<std::mutex> g(mu); v = value; }
    cv.notify_all();
  };
  auto fail = [&] {
    { std::lock_guard<std::mutex> g(mu); failed = true; }
    cv.notify_all();
  };
  auto wait_until = [&](const long &v, long target) {
    std::unique_lock<std::mutex> g(mu);
    cv.wait(g, [&] { return v >= target || failed; });
    if (failed) throw StageAborted("pipelined transfer: another stage failed");
  };
  Hooks hu, hd;
  hu.enqueued = [&](long i, hipStream_t s) {
    SMX_HIP_CHECK(hipEventRecord(up_ev[(size_t)i], s));
    publish(up_enqueued, i + 1);
  };
  hd.before = [&](long j, hipStream_t s) {   // the kernels that write output bytes [.., end) are on the compute stream
    const size_t end = std::min(out_total, (size_t)(j + 1) * kChunk);
    const int64_t last_clip = (int64_t)((end - 1) / out_clip_bytes), u = last_clip / unit;
    wait_until(launched, (long)u + 1);
    SMX_HIP_CHECK(hipStreamWaitEvent(s, k_ev[(size_t)u], 0));
  };
  // the copying threads of both directions together: as many as one serial transfer uses, shared by the bytes each side moves
  // (a page-locked side needs none: the DMA engine moves it)
  const int tw = worker_count();
  // (... and the upload alone is served by six: 4, 8 and 16 threads finish it at the same 13 ms, profiles/r07/host_path_pinned_threads.log)
  // (a synthesis uploads four times what it downloads: there the upload takes them all)
  const int t_up = dst_pinned ? (in_total > out_total ? tw : std::min(tw, 6)) : std::max(2, (int)((double)tw * (double)in_total / (double)(in_total + out_total) + 0.5));
  const int t_down = src_pinned ? tw : std::max(2, tw - t_up);
  static const bool trace = env_flag("SMX_HOST_TRACE") == 1;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_start = now();
  double t_up_done = 0, t_down_done = 0, t_launch_done = 0;
  std::thread up([&] {
    try {
      SMX_HIP_CHECK(hipSetDevice(dev));
      staged(d_in, src, in_total, false, t_up, &hu);
      t_up_done = now() - t_start;
    } catch (...) {
      err_up = std::current_exception();
      fail();
    }
  });
  std::thread down([&] {
    try {
      SMX_HIP_CHECK(hipSetDevice(dev));
      staged(dst, d_out, out_total, true, t_down, &hd);
      t_down_done = now() - t_start;
    } catch (...) {
      err_down = std::current_exception();
      fail();
    }
  });
  try {
    for (int64_t u = 0; u < units; ++u) {
      const int64_t clip0 = u * unit, nc = std::min<int64_t>(unit, clips - clip0);
      const long chunk = (long)(((size_t)(clip0 + nc) * in_clip_bytes - 1) / kChunk);   // the DMA that completes this unit's input
      wait_until(up_enqueued, chunk + 1);
      SMX_HIP_CHECK(hipStreamWaitEvent(compute, up_ev[(size_t)chunk], 0));
      launch(clip0, nc, compute);
      SMX_HIP_CHECK(hipEventRecord(k_ev[(size_t)u], compute));
      publish(launched, (long)u + 1);
    }
    t_launch_done = now() - t_start;
  } catch (...) {
    err_main = std::current_exception();
    fail();
  }
  up.join();
  down.join();
  if (trace)
    fprintf(stderr, "[smx] pipelined host call: upload done at %.2f ms (%d copying threads%s), last launch at %.2f, download done at %.2f (%d%s)\n", t_up_done, t_up,
            src_pinned ? ", page-locked source" : "", t_launch_done, t_down_done, t_down, dst_pinned ? ", page-locked destination" : "");
  (void)hipStreamSynchronize(compute);   // (the two transfer threads synchronise their own streams on every path: staged / direct)
  cleanup();
  // the root cause first: a stage that only noticed another one failing reports StageAborted
  auto aborted = [](const std::exception_ptr &e) {
    if (!e) return false;
    try { std::rethrow_exception(e); } catch (const StageAborted &) { return true; } catch (...) { return false; }
  };
  for (const std::exception_ptr *e : {&err_main, &err_up, &err_down})
    if (*e && !aborted(*e)) std::rethrow_exception(*e);
  for (const std::exception_ptr *e : {&err_main, &err_up, &err_down})
    if (*e) std::rethrow_exception(*e);
}

// Can a pipelined call stage its transfers on the current device?  (One ring per direction exists or can be allocated now.  Asked
// BEFORE the threads start: without page-locked memory the serial path's plain hipMemcpy still completes the call.)
bool staging_available() {
  for (int dir = 0; dir < 2; ++dir) {
    RingLease lease(dir == 1);
    if (!lease.ring) return false;
  }
  return true;
}

void set_transfer_share(int parts) { t_transfer_share = parts < 1 ? 1 : parts; }

void staging_peak(int *up, int *down, bool reset) {
  if (up) *up = g_staged_peak[0].load();
  if (down) *down = g_staged_peak[1].load();
  if (reset) { g_staged_peak[0].store(0); g_staged_peak[1].store(0); }
}

void copy_to_device(void *d_dst, const void *src, size_t bytes) {
  if (bytes == 0) return;
  if (bytes < kDirect || env_flag("SMX_COPY_PLAIN") == 1) {
    SMX_HIP_CHECK(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return;
  }
  staged(d_dst, src, bytes, false);
}

void copy_to_host(void *dst, const void *d_src, size_t bytes) {
  if (bytes == 0) return;
  if (host_is_pinned(dst, bytes)) {
    SMX_HIP_CHECK(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return;
  }
  advise_huge(dst, bytes);
  if (bytes < kDirect || env_flag("SMX_COPY_PLAIN") == 1) {
    SMX_HIP_CHECK(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return;
  }
  staged(dst, d_src, bytes, true);
}

}  // namespace smx
