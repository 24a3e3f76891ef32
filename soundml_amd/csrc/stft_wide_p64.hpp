// stft2048_power_wide_kernel -- the power spectrogram at fft 2048 on the reference's own numerics (window, transform and
// |.|^p in float64: stft.ml:345-346, 356-364, 670-674), float32 audio in, float32 columns out (included by
// stft_generic.hip inside its anonymous namespace, after the kernels whose arithmetic it shares).
//
// The ARITHMETIC is stft_stockham_power16_kernel<11, float, false, 8, double, float>'s, call for call: the same three
// register passes (fftdev::stockham_pass: radix 16, 16, 4 over M = 1024 complex points, 16 per lane, a frame per wave),
// the same real-FFT post-pass expression, the same magnitude_pow -- every frame gets the same bits from either kernel
// (tests/test_gpu_wide_pipeline.py; SMX_WIDE_PIPELINE=0 selects the older one), so the range / slice / partition laws hold
// whichever of them a call takes.  What differs is the data movement, where that kernel lost its time (one tile of 8 frames
// per workgroup, nothing overlapping its first load and its last store, 32-byte store runs: 1.58 ms at C2; this one 1.11):
//  * persistent workgroups of 8 waves, one per CU, walking contiguous ranges of the (clip, tile) sequence; a tile is 16
//    consecutive frames of a clip = two rounds of a frame per wave; ONE copy of the frame code (a loop trip per frame: with
//    the rounds written out the hot loop is 60 KB of instructions, the whole instruction cache two CUs share);
//  * the two exchanges between the passes go through a wave-private 8 KB scratch ONE PLANE AT A TIME (real parts, then
//    imaginary parts), the exchange with the lane that holds Z[M - k] half a frame at a time: half the LDS of the complex
//    form, which is what leaves room for the whole tile of results beside the window and the pass twiddles;
//  * the post-pass twiddles (16 KB more) stay in L2: a frame's 16 values per lane are requested behind its second pass,
//    BEFORE the samples of the wave's next frame (32 registers, from HBM), so that waiting for the former leaves the
//    latter in flight; every per-lane LDS address is re-derived from an opaque lane index inside the loop (hoisted out of it,
//    a hundred of them spill, and a scratch reload waits for every outstanding memory operation of the wave);
//  * the flush reads 4 frames of a bin per lane (16-byte stores, 64-byte row runs), every LDS read conflict free (address =
//    1025 f + k: bank f + k), and sits in the MIDDLE of the next tile's first frame, the tile's next write at that frame's
//    END: two LDS counters (tiles filled / tiles drained), no workgroup barrier in the loop.  The two waves of a SIMD settle
//    half a frame apart: a wave of the late half finds the tile complete behind its first pass and flushes there.
// What bounds it (profiles/r06/NOTES.md section 4): float64 vector instructions take 4 cycles per wave (v_fma / v_add / v_mul_f64;
// the IEEE sqrt of magnitude_pow 69), ~5 900 cycles per frame, and the chip holds 2.20 GHz under the kernel (tools/clock_check.py):
// 0.61 ms of pure issue at C2; measured 1.11 ms, 59 % of that rate.
// LDS: 8 x 8,192 (scratch) + 65,664 (tile) + 16,384 (window) + 4,096 (pass twiddles) + 64 = 151,744 B.
#ifndef SMX_W64_EARLY
#define SMX_W64_EARLY 1
#endif
namespace wide64 {
using fftdev::cpx;
using V2 = double2;
constexpr int kN = 2048, kM = 1024, kFT = 16, kWaves = 8;
constexpr int kPitch = 1025;                                   // floats between the tile's frames (odd: bank = f + k)
constexpr int kScratchBytes = kWaves * kM * 8;                  // 65,536
constexpr int kOffTile = kScratchBytes;
constexpr int kOffWin = kOffTile + (kFT * kPitch * 4 + 127) / 128 * 128;
constexpr int kOffTwm = kOffWin + kM * 16;
constexpr int kOffCnt = kOffTwm + 256 * 16;
constexpr int kLds = kOffCnt + 64;
static_assert(kLds <= 160 * 1024, "LDS budget");

__device__ __forceinline__ void fence() { asm volatile("" ::: "memory"); }
// The counters: a wave's DS operations execute in order, so the add lands behind the LDS accesses the wave issued before it and a
// wave that has seen the count reads what those wrote; the compiler is held to the same order by the fences (a relaxed atomic
// alone orders nothing for it).
__device__ __forceinline__ void signal(unsigned *c, int lane) {
  fence();
  if (lane == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  fence();
}
__device__ __forceinline__ void wait_for(unsigned *c, unsigned target) {
  fence();
  while ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
    __builtin_amdgcn_s_sleep(2);
  fence();
}

struct Args {
  GenericArgs g;
  const double *window;     // 2048
  const V2 *tw_m, *tw_n;    // exp(-2 pi i j / 1024) (j < 256 read), exp(-2 pi i k / 2048) (k < 1024 read)
  int tiles_per_clip;
  int64_t total_tiles, blocks;
  int contiguous;   // 1: contiguous tile ranges per workgroup (rounds 4-5); 0: the workgroups side by side through the sequence
};

// the samples of frame p of the clip at x (lane tid: pairs 2 (tid + 64 m)), zero for a frame beyond the request
__device__ __forceinline__ void request_frame(const GenericArgs &a, const float *x, int64_t p, bool have, int tid_, float2 (&raw)[16]) {
  int tid = tid_;
  asm volatile("" : "+v"(tid));
#ifdef SMX_W64_NOLOAD
  if (a.power != 12345.0) have = false;
#endif
  if (!have) {
#pragma unroll
    for (int m = 0; m < 16; ++m) raw[m] = make_float2(0.f, 0.f);
    return;
  }
  const int64_t s0 = p * a.hop - a.left;
  if (s0 >= 0 && s0 + kN <= a.n) {
    const float *xs = x + s0 + 2 * tid;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const auto xv = *reinterpret_cast<const typename Pair<float>::type *>(xs + 128 * m);
      raw[m] = make_float2(xv.x, xv.y);
    }
  } else {   // a frame that touches a border of the signal: the padded signal's samples (stft.ml:300-338)
#pragma unroll 1
    for (int m = 0; m < 16; ++m) {
      const int64_t i = s0 + 2 * (tid + 64 * m);
      const float v0 = (float)fetch_sample<float>(x, a.n, i, a.pad, a.pad_value);
      const float v1 = (float)fetch_sample<float>(x, a.n, i + 1, a.pad, a.pad_value);
#pragma unroll
      for (int mm = 0; mm < 16; ++mm)
        if (mm == m) raw[mm] = make_float2(v0, v1);
    }
  }
}

// one exchange of the frame's 1024 complex values through the wave's scratch, a plane at a time: value j of the lane goes to
// position wpos(j), value j comes from position rpos(j) (positions in natural order; stored XOR-swizzled: fftdev::swz)
template <class WP, class RP>
__device__ __forceinline__ void exchange(cpx<double> (&r)[16], double *scr, WP wpos, RP rpos) {
  double t[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) scr[fftdev::swz(wpos(j))] = r[j].x;
  fence();
#pragma unroll
  for (int j = 0; j < 16; ++j) t[j] = scr[fftdev::swz(rpos(j))];
  fence();
#pragma unroll
  for (int j = 0; j < 16; ++j) scr[fftdev::swz(wpos(j))] = r[j].y;
  fence();
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    r[j].x = t[j];
    r[j].y = scr[fftdev::swz(rpos(j))];
  }
  fence();
}

#ifdef SMX_STAMPS
#define W64_STAMP(i) SMX_STAMP(i)
#else
#define W64_STAMP(i) do { } while (0)
#endif
struct Lds {
#ifdef SMX_STAMPS
  unsigned long long *stamp_sum, *stamp_prev_p;
#endif
  double *scr;          // this wave's scratch (1024 doubles)
  float *tile;          // results: bin k of frame f at 1025 f + k
  const V2 *win, *twm;  // tables in LDS: the window as pairs, the pass twiddles exp(-2 pi i j / 1024), j < 256
  const V2 *twn;        // the post-pass twiddles exp(-2 pi i k / 2048) stay in global memory (L2): 16 KB more do not fit
  unsigned *filled, *drained;
};

// passes 2 and 3, the exchange with the partner lane, post-pass and |.|^p of one frame whose first pass has run; results
// (bins tid + 64 m and, lane 0, the Nyquist bin) to dst[k]
template <int PMODE, class Req, class Mid, class Pre>
__device__ __forceinline__ void finish_frame(const GenericArgs &a, const Lds &l, cpx<double> (&r)[16], int tid_, float *dst, Req request_next, Mid mid, Pre before_results) {
  // (the lane index made opaque here: the compiler otherwise hoists ~100 loop-invariant LDS addresses out of the tile loop,
  // and they spill -- a scratch reload waits for every outstanding memory operation of the wave, the sample requests included)
  int tid = tid_;
  asm volatile("" : "+v"(tid));
#ifdef SMX_STAMPS
  unsigned long long *stamp_sum = l.stamp_sum;
  unsigned long long &stamp_prev = *l.stamp_prev_p;
#endif
  // |.|^p: the exponent as a constant for the two common powers (the general one drags a 200-instruction pow through 16 unrolled bins)
  const double power = PMODE == 2 ? 2.0 : PMODE == 1 ? 1.0 : a.power;
  using namespace fftdev;
  double *scr = l.scr;
  // pass 1's results to pass 2's positions (stockham_pass<1024, 16, 1>: writes 16 tid + j; <1024, 16, 16>: reads tid + 64 j)
  exchange(r, scr, [&](int j) { return 16 * tid + j; }, [&](int j) { return tid + 64 * j; });
  W64_STAMP(3);
  {
    const V2 w2[1] = {l.twm[(tid % 16) * 4]};
    stockham_pass<kM, 16, 16, false, false, true>(r, reinterpret_cast<V2 *>(scr), tid, l.twm, w2);
  }
  W64_STAMP(4);
  fence();
  // behind the pass that needs the most registers: the post-pass twiddles of this frame (L2), then the samples of the wave's
  // next frame (HBM) -- in that order, so that waiting for the former leaves the latter in flight
  V2 tw[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) tw[m] = l.twn[tid + 64 * m];
  request_next();
  fence();
  // (writes (tid / 16) 256 + tid % 16 + 16 j; pass 3 reads tid + 64 (i + 4 j) into r[4 i + j])
  {
    const int j0 = (tid / 16) * 256 + tid % 16;
    exchange(r, scr, [&](int j) { return j0 + 16 * j; }, [&](int e) { return tid + 64 * ((e >> 2) + 4 * (e & 3)); });
  }
  W64_STAMP(5);
  {
    V2 w3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w3[i] = l.twm[(tid + 64 * i) % 256];
    stockham_pass<kM, 4, 256, false, false, true>(r, reinterpret_cast<V2 *>(scr), tid, l.twm, w3);
  }
  W64_STAMP(6);
  mid();   // (the previous tile's flush, when this is the first frame of a tile and the wave has not flushed already)
  fence();
  // r[4 i + j] = Z[tid + 64 (i + 4 j)]: the lane's own bins k = tid + 64 m sit at r[4 (m & 3) + (m >> 2)].  Their partners
  // Z[(M - k) & (M - 1)] belong to lane 64 - tid, register 15 - m (lane 0: its own register 16 - m; bins 0 and 512 pair with
  // themselves).  They come through the scratch half a frame at a time, both parts together: the upper registers (m >= 8,
  // 512 complex values = the scratch's 8 KB) serve bins m < 8, then the lower ones serve bins m >= 8 -- 8 partner values
  // alive at a time instead of 16.
  float val[16];
  V2 *cs = reinterpret_cast<V2 *>(scr);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {   // the half that serves the other one: slot tid + 64 q
      const int ms = half == 0 ? 8 + q : q;
      const cpx<double> z = r[4 * (ms & 3) + (ms >> 2)];
      cs[tid + 64 * q] = make_double2(z.x, z.y);
    }
    fence();
    V2 pz[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) pz[q] = cs[((64 - tid) + 64 * (7 - q)) & 511];   // partner of bin tid + 64 (8 half + q)
    fence();
    W64_STAMP(7);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int m = 8 * half + q;
      const cpx<double> zk = r[4 * (m & 3) + (m >> 2)];
      cpx<double> zp = {pz[q].x, pz[q].y};
      if (q == 0 && tid == 0) zp = zk;   // Z[0] and Z[512]
      const double er = zk.x + zp.x, ei = zk.y - zp.y;
      const double dr = zk.x - zp.x, di = zk.y + zp.y;
      const V2 w = tw[m];
      val[m] = magnitude_pow<double, float>(er + (w.x * di + w.y * dr), ei - (w.x * dr - w.y * di), power);
    }
    W64_STAMP(8);
  }
  float nyq = 0.f;
  if (tid == 0) nyq = magnitude_pow<double, float>(2.0 * (r[0].x - r[0].y), 0.0, power);
  fence();
  before_results();
#pragma unroll
  for (int m = 0; m < 16; ++m) dst[tid + 64 * m] = val[m];
  if (tid == 0) dst[kM] = nyq;
  W64_STAMP(9);
}

__device__ __forceinline__ void store4(float *base /* wave-uniform */, unsigned byte_off, float a, float b, float c, float d) {
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  const f32x4 v = {a, b, c, d};
  asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(byte_off), "v"(v), "s"(base) : "memory");
}

// a finished tile out to memory: wave w takes bins [128 w, 128 w + 128) (wave 7 the Nyquist bin as well); a lane reads the four
// frames 4 g .. 4 g + 3 of one bin (g = lane & 3) and stores them as 16 bytes; the rows of an instruction are
// {0-3, 16-19} + 4 (lane >> 5) + 8 (q & 1) + 32 (q >> 1), which spreads a half-wave's reads over all 32 banks
__device__ __forceinline__ void flush_read(const Lds &l, int wave, int lane_, float (&v)[9][4]) {
  int lane = lane_;
  asm volatile("" : "+v"(lane));
  const int g = lane & 3, kb = (lane >> 2) & 7, h = lane >> 5;
  const int row0 = 128 * wave + (kb & 3) + 16 * (kb >> 2) + 4 * h;
  const float *base = l.tile + 4 * g * kPitch + row0;
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[q][i] = base[i * kPitch + 32 * (q >> 1) + 8 * (q & 1)];
  if (wave == 7) {   // the Nyquist bin (lanes with kb = 0, h = 0 store it)
    const float *nb = l.tile + 4 * g * kPitch + kM;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[8][i] = nb[i * kPitch];
  }
}
__device__ __forceinline__ void flush_store(const GenericArgs &a, float *obase, int nf, int wave, int lane_, const float (&v)[9][4]) {
  int lane = lane_;
  asm volatile("" : "+v"(lane));
#ifdef SMX_W64_NOSTORE
  if (a.power != 12345.0) return;
#endif
  const int g = lane & 3, kb = (lane >> 2) & 7, h = lane >> 5;
  const int row0 = 128 * wave + (kb & 3) + 16 * (kb >> 2) + 4 * h;
  if (nf == kFT) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned off = ((unsigned)(row0 + 32 * (q >> 1) + 8 * (q & 1)) * (unsigned)a.out_stride + 4u * g) * 4u;
      store4(obase, off, v[q][0], v[q][1], v[q][2], v[q][3]);
    }
    if (wave == 7 && kb == 0 && h == 0) store4(obase, ((unsigned)kM * (unsigned)a.out_stride + 4u * g) * 4u, v[8][0], v[8][1], v[8][2], v[8][3]);
    return;
  }
#pragma unroll
  for (int q = 0; q < 8; ++q)   // a clip's ragged last tile
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (4 * g + i < nf) obase[(int64_t)(row0 + 32 * (q >> 1) + 8 * (q & 1)) * a.out_stride + 4 * g + i] = v[q][i];
  if (wave == 7 && kb == 0 && h == 0)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (4 * g + i < nf) obase[(int64_t)kM * a.out_stride + 4 * g + i] = v[8][i];
}

template <int PMODE>
__global__ void __launch_bounds__(512) stft2048_power_wide_kernel(Args A) {
  using namespace fftdev;
  const GenericArgs &a = A.g;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  Lds l;
  l.scr = reinterpret_cast<double *>(smem) + wave * kM;
  l.tile = reinterpret_cast<float *>(smem + kOffTile);
  l.win = reinterpret_cast<const V2 *>(smem + kOffWin);
  l.twn = A.tw_n;
  l.twm = reinterpret_cast<const V2 *>(smem + kOffTwm);
  l.filled = reinterpret_cast<unsigned *>(smem + kOffCnt);
  l.drained = l.filled + 1;
  {
    V2 *win = reinterpret_cast<V2 *>(smem + kOffWin), *twm = reinterpret_cast<V2 *>(smem + kOffTwm);
    const V2 *gw = reinterpret_cast<const V2 *>(A.window);
    for (int e = threadIdx.x; e < kM; e += 512) win[e] = gw[e];
    for (int e = threadIdx.x; e < 256; e += 512) twm[e] = A.tw_m[e];
    if (threadIdx.x == 0) {
      *l.filled = 0u;
      *l.drained = 0u;
    }
  }
  // This workgroup's tiles of the flat (clip, tile) sequence.  Round 6: the workgroups walk the sequence SIDE BY SIDE -- tile
  // tau0 + it * blocks, the workgroups of an XCD (blockIdx.x % 8) on neighbouring tiles -- instead of contiguous ranges: the plain flush
  // writes a 64-byte run per row and tile, a 128-byte line of the result is complete after two consecutive tiles, and with contiguous
  // ranges those are ~70 us apart in one workgroup (34 MB of open lines against 32 MB of L2: WRITE_SIZE 1.35-1.38 x the output);
  // side by side an XCD writes 32 neighbouring tiles at once and its L2 assembles whole lines (what profiles/r08/NOTES.md section 9
  // found for the fft-4096 kernel).  A.contiguous (SMX_WIDE_CONTIGUOUS=1 in diagnostic builds): the ranges of rounds 4-5, A/B timing.
  const int64_t nb = A.blocks, vb = blockIdx.x;
  const int64_t per = A.total_tiles / nb, extra = A.total_tiles % nb;
  const int64_t xq = nb / 8, xr = nb % 8, xcd = vb % 8, xidx = vb / 8;
  const int64_t side0 = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xidx;
  const int64_t tau0 = A.contiguous ? vb * per + (vb < extra ? vb : extra) : side0;
  const int64_t tau_step = A.contiguous ? 1 : nb;
  const int ntiles = A.contiguous ? (int)(per + (vb < extra ? 1 : 0)) : (side0 < A.total_tiles ? (int)((A.total_tiles - side0 + nb - 1) / nb) : 0);
  const float *x0 = reinterpret_cast<const float *>(a.x);
  float *out0 = reinterpret_cast<float *>(a.out);
  auto tile_of = [&](int64_t tau, const float *&xc, float *&oc, int64_t &f0, int &nf) {
    const int64_t clip = tau / A.tiles_per_clip, t = tau - clip * A.tiles_per_clip;
    xc = x0 + clip * a.x_stride;
    f0 = t * kFT;
    oc = out0 + clip * a.bins * a.out_stride + a.out_offset + f0;
    const int64_t left = a.count - f0;
    nf = (int)(left < kFT ? left : kFT);
  };
  float2 raw[16];
  const float *xc = x0;
  float *oc = out0;
  int64_t f0 = 0;
  int nf = 0;
  if (ntiles > 0) {
    tile_of(tau0, xc, oc, f0, nf);
    request_frame(a, xc, a.p0 + f0 + wave, wave < nf, tid, raw);
  }
  __syncthreads();   // tables and counters in place: the only workgroup barrier
  float *pend_out = nullptr;
  int pend_nf = 0;
#ifdef SMX_STAMPS
  unsigned long long stamp_sum[kStampSlots] = {0}, stamp_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
  const unsigned long long clk_t0 = stamp_prev, clk_r0 = __builtin_amdgcn_s_memrealtime();
  l.stamp_sum = stamp_sum;
  l.stamp_prev_p = &stamp_prev;
#endif
  auto window_frame = [&](cpx<double> (&r)[16]) {   // r = x w / 2, as stft_stockham_power16_kernel forms it
    int tid = threadIdx.x & 63;
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const V2 wv = l.win[tid + 64 * m];
      r[m] = {(double)raw[m].x * wv.x * 0.5, (double)raw[m].y * wv.y * 0.5};
    }
  };
  // One loop trip = one frame of this wave (slot s: round s & 1 of tile s >> 1): the frame code exists once -- with the two
  // rounds written out the hot loop is 60 KB of instructions, the whole instruction cache two CUs share.
#pragma unroll 1
  for (int s = 0; s < 2 * ntiles; ++s) {
    const int rd = s & 1, it = s >> 1;
    const bool have = 8 * rd + wave < nf;
    cpx<double> r[16];
    W64_STAMP(0);
    if (have) {
      window_frame(r);
      W64_STAMP(1);
      stockham_pass<kM, 16, 1, false, false, true>(r, reinterpret_cast<V2 *>(l.scr), tid, l.twm);
    }
    W64_STAMP(2);
    // The previous tile goes out in the MIDDLE of this tile's first frame (behind its third pass): complete once every wave
    // has signalled, read our rows, release it, store.  The tile is written again at the END of this frame: a wave may run
    // half a frame ahead of or behind the others before either counter stops it (with the flush at the frame's start and the
    // second round's results in the scratch -- first form of this kernel -- the waves spent 7 000 of 41 000 cycles per tile
    // in these two waits).
    bool flush_due = rd == 0 && it > 0;
    auto flush_now = [&] {
      float v[9][4];
      flush_read(l, wave, tid, v);
      signal(l.drained, tid);
      flush_store(a, pend_out, pend_nf, wave, tid, v);
      flush_due = false;
      W64_STAMP(10);
    };
#if SMX_W64_EARLY
    // The two waves of a SIMD settle half a frame apart.  A wave of the LATE half finds the tile complete already here, behind
    // its first pass, and flushes at once -- half a frame before the early waves want to write the tile again; a wave of the
    // early half finds it incomplete and comes back behind its third pass, when the late ones have finished.
    if (flush_due && (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(l.filled, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= 8u * (unsigned)it) {
      fence();
      flush_now();
    }
#endif
    auto mid = [&] {
      if (flush_due) {
#ifdef SMX_STAMPS
        const unsigned long long w0 = __builtin_amdgcn_s_memtime();
#endif
        wait_for(l.filled, 8u * (unsigned)it);
#ifdef SMX_STAMPS
        stamp_sum[12] += __builtin_amdgcn_s_memtime() - w0;
#endif
        flush_now();
      }
    };
    // the frame of the next slot: round 2 of this tile, or round 1 of the next one
    const float *xq = xc;
    int64_t pq = a.p0 + f0 + 8 + wave;
    bool hq = 8 + wave < nf;
    const float *xn = xc;
    float *on = oc;
    int64_t f0n = 0;
    int nfn = 0;
    if (rd == 1) {
      hq = false;
      if (it + 1 < ntiles) {
        tile_of(tau0 + (int64_t)(it + 1) * tau_step, xn, on, f0n, nfn);
        xq = xn;
        pq = a.p0 + f0n + wave;
        hq = wave < nfn;
      }
    }
    auto request_next = [&] { request_frame(a, xq, pq, hq, tid, raw); };
    float *dst = l.tile + (8 * rd + wave) * kPitch;
    if (have) {
      finish_frame<PMODE>(a, l, r, tid, dst, request_next, mid, [&] {
        if (rd == 0 && it > 0) {   // every wave has read the previous tile out
#ifdef SMX_STAMPS
          const unsigned long long w1 = __builtin_amdgcn_s_memtime();
#endif
          wait_for(l.drained, 8u * (unsigned)it);
#ifdef SMX_STAMPS
          stamp_sum[13] += __builtin_amdgcn_s_memtime() - w1;
#endif
        }
      });
    } else {
      request_next();
      mid();
    }
    if (rd == 1) {
      signal(l.filled, tid);
      pend_out = oc;
      pend_nf = nf;
      xc = xn;
      oc = on;
      f0 = f0n;
      nf = nfn;
    }
    W64_STAMP(11);
  }
#ifdef SMX_STAMPS
  {
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    stamp_sum[20] = t1 - clk_t0;
    stamp_sum[21] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    if (tid == 0)
      for (int i = 0; i < kStampSlots; ++i) fftdev::g_stamp_sums[(blockIdx.x * 16 + wave) * kStampSlots + i] = stamp_sum[i];
  }
#endif
  if (ntiles > 0) {
    wait_for(l.filled, 8u * (unsigned)ntiles);
    float v[9][4];
    flush_read(l, wave, tid, v);
    flush_store(a, pend_out, pend_nf, wave, tid, v);
  }
}

// the launch: true = taken (fft 2048, float32 audio, power output, row offsets within 32 bits)
inline bool launch(const StftJob &job, const GenericArgs &g, const StftTables &t) {
  if (job.cfg->fft_size != kN || job.in_bytes != 4 || job.mode == OUT_COMPLEX) return false;
  if (!t.window_f64 || !t.fast_w_m_f64 || !t.twiddle_f64) return false;
  if (g.bins * g.out_stride * 4 >= (int64_t(1) << 32)) return false;
  if (g.count <= 0 || g.lead <= 0) return true;
  if (env_flag("SMX_WIDE_PIPELINE") == 0) return false;   // "0": the one-tile-per-workgroup kernel (A/B timing, bit-identical)
  Args A{};
  A.g = g;
  A.window = reinterpret_cast<const double *>(t.window_f64);
  A.tw_m = reinterpret_cast<const V2 *>(t.fast_w_m_f64);
  A.tw_n = reinterpret_cast<const V2 *>(t.twiddle_f64);
  const int64_t tiles = (g.count + kFT - 1) / kFT;
  if (tiles > 0x7fffffff) return false;
  A.tiles_per_clip = (int)tiles;
  A.total_tiles = g.lead * tiles;
  const int cu_count = device_cu_count();   // (per device, thread-safe: tables.cpp)
  A.blocks = A.total_tiles < cu_count ? A.total_tiles : cu_count;
  A.contiguous = diag_flag("SMX_WIDE_CONTIGUOUS") == 1 ? 1 : 0;
  auto kernel = g.power == 2.0 ? stft2048_power_wide_kernel<2> : g.power == 1.0 ? stft2048_power_wide_kernel<1> : stft2048_power_wide_kernel<0>;
  SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  SMX_LAUNCH(kernel, dim3((unsigned)A.blocks), dim3(512), kLds, job.stream, A);
  SMX_HIP_CHECK(hipGetLastError());
  return true;
}
}  // namespace wide64
