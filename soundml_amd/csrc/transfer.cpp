// Host <-> device transfers of the host-pointer entry points (the drop-in boundary: nx hands over pageable host
// tensors, soundml/lib/stft.ml works on host memory throughout).  A plain hipMemcpy of pageable memory runs at
// 10-13 GB/s on this platform and most of a C2 call's 116 ms went there; large transfers go instead through a ring
// of pinned staging buffers, the DMA engine running asynchronously while a few host threads move the previous chunk
// between the staging buffer and the caller's (possibly never-touched, page-faulting) memory.
#include <atomic>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "smx_internal.hpp"

namespace smx {
namespace {

constexpr size_t kChunk = (size_t)16 << 20;   // bytes per staging buffer
constexpr int kRing = 4;                      // staging buffers in flight
constexpr size_t kDirect = (size_t)4 << 20;   // below this a plain hipMemcpy is as fast

struct Ring {
  void *buf[kRing] = {};
  std::mutex busy;   // one staged transfer at a time per process
  bool ensure() {
    if (buf[0]) return true;
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
Ring g_ring;

int worker_count() {
  static const int n = [] {
    if (env_flag("SMX_COPY_THREADS") >= 0) return std::max(1, std::min(64, (int)env_int("SMX_COPY_THREADS", 16)));
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(16u, hw / 2));
  }();
  return n;
}

struct StreamAndEvents {
  hipStream_t stream = nullptr;
  hipEvent_t ev[kRing] = {};
  StreamAndEvents() {
    SMX_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    for (auto &e : ev) SMX_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  ~StreamAndEvents() {
    for (auto &e : ev)
      if (e) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

inline void wait_for(const std::atomic<long> &v, long target) {
  int spins = 0;
  while (v.load(std::memory_order_acquire) < target)
    if (++spins > 64) std::this_thread::yield();
}

// The slice of chunk `i` that worker `w` of `t` moves (64-byte aligned cuts).
inline void slice(size_t len, int w, int t, size_t &lo, size_t &hi) {
  const size_t per = ((len + (size_t)t - 1) / (size_t)t + 63) & ~(size_t)63;
  lo = std::min(len, per * (size_t)w);
  hi = std::min(len, lo + per);
}

// to_host: device -> staging by DMA, staging -> user by the workers.  !to_host: the reverse.
void staged(void *dst, const void *src, size_t bytes, bool to_host) {
  std::lock_guard<std::mutex> lock(g_ring.busy);
  if (!g_ring.ensure()) {   // no pinned memory to be had: the plain copy still works
    SMX_HIP_CHECK(hipMemcpy(dst, src, bytes, to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice));
    return;
  }
  StreamAndEvents se;
  const long chunks = (long)((bytes + kChunk - 1) / kChunk);
  const int t = worker_count();
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
  auto dma = [&](long i) {
    void *stage = g_ring.buf[i % kRing];
    if (err == hipSuccess)
      err = to_host ? hipMemcpyAsync(stage, (const unsigned char *)src + (size_t)i * kChunk, len_of(i), hipMemcpyDeviceToHost, se.stream)
                    : hipMemcpyAsync((unsigned char *)dst + (size_t)i * kChunk, stage, len_of(i), hipMemcpyHostToDevice, se.stream);
    if (err == hipSuccess) err = hipEventRecord(se.ev[i % kRing], se.stream);
  };
  auto workers_done_with = [&](long i) {
    int spins = 0;
    while (done[i].load(std::memory_order_acquire) < t)
      if (++spins > 64) std::this_thread::yield();
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
  if (err == hipSuccess) err = hipStreamSynchronize(se.stream);
  SMX_HIP_CHECK(err);
}

}  // namespace

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
  if (bytes < kDirect || env_flag("SMX_COPY_PLAIN") == 1) {
    SMX_HIP_CHECK(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return;
  }
  staged(dst, d_src, bytes, true);
}

}  // namespace smx
