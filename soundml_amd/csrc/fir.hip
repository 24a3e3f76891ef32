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
#include <memory>

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
    // wave-split kernel (N = 32768: sixteen 1024-point sub-transforms, one per wave): tables in the order its lanes read
    float2 *h_split = nullptr;  // h_half[NS k' + r] at [1024 r + k'] (NS = M / 1024 sub-transforms: 16 or 8), h_half[M] at [M]
    float2 *w_split = nullptr;  // exp(-2 pi i (NS k' + r) / N) at [1024 r + k']
    float2 *tw_1k = nullptr;    // exp(-2 pi i j / 1024), j < 512
    // register-pipeline kernel (fir_ols_pk32_kernel): W_1024^(l k1) in the order its lanes read it -- rows m < 15 hold the pair
    // k1 = 2 m + 1, 2 m + 2 per lane l (a float4), then one row of k1 = 31 (as stft_fast_p32.hpp's twA4 / twA31)
    float2 *tw_a32 = nullptr;
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
  int64_t n, x_stride, y_stride;   // n: input samples per channel
  int64_t n_out, out_shift;         // convolution outputs [out_shift, out_shift + n_out) land at y[0 .. n_out)
  int64_t taps, nfft, valid;
  int log2n;
  int64_t pairs_per_channel;
  const float2 *h_nat;
  const float2 *tw;     // exp(-2 pi i j / N), j < N/2
  // half-size kernel
  const float2 *h_half; // H[k] / (4 M), k <= M
  const float2 *tw_m;   // exp(-2 pi i j / M), j < M/2
  const float2 *h_split, *w_split, *tw_1k;   // wave-split kernel
  const float2 *tw_a32;                      // register-pipeline kernel
  int64_t lead;         // samples of every circular result that are discarded (even, >= taps - 1)
  int64_t step;         // block advance = N - lead (even)
  int64_t blocks_per_channel;
  int64_t channels;
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
        const int64_t oa = out_a + idx - a.out_shift, ob = out_b + idx - a.out_shift;
        if (oa >= 0 && oa < a.n_out) y[oa] = r[i * RL + j].x;      // Re(conj(.)) =  Re
        if (ob >= 0 && ob < a.n_out) y[ob] = -r[i * RL + j].y;     // Im(conj(.)) = -Im
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
  const int64_t out0 = blk * a.step - a.out_shift;   // y index of the block's first kept sample
  int to = tid;
  asm volatile("" : "+v"(to));
  const bool whole = out0 >= 0 && out0 + a.step <= a.n_out;   // every kept sample of this block is wanted (block-uniform)
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
          if (o >= 0 && o < a.n_out) y[o] = re;
          if (o + 1 >= 0 && o + 1 < a.n_out) y[o + 1] = im;
        }
      }
    }
}

// ---- N = 32768: the half-size transform split over the workgroup's sixteen waves ---------------------------------
// M = 16384 = 16 x 1024.  One radix-16 pass over the whole block (element n = n' + 1024 q: thread n' holds q = 0..15,
// lane-contiguous 8-byte reads), then sixteen independent 1024-point transforms: bins k = 16 k' + r belong to sub-transform r,
//   X[16 k' + r] = sum_n' w_1024^(n' k') [ w_M^(n' r) sum_q x[n' + 1024 q] w_16^(q r) ].
// Sub-transform r is the work of WAVE r alone: its three passes exchange through the wave's own 8 KB of LDS with no workgroup
// barrier (DS operations of a wave complete in order), so the sixteen waves drift out of phase and one wave's LDS round trips
// overlap the others' butterflies.  The inverse mirrors it (decimation in time: the waves' sub-transforms first, then the
// twiddles and ONE radix-16 pass across waves whose outputs are lane-contiguous sample pairs again).  Five workgroup barriers per
// block where the pass-by-pass kernel has sixteen (it spends 42 % of its wave time waiting at them).  Measured on C4 (one box,
// alternating processes): pass-by-pass 0.107 ms, wave-split 0.104, wave-split + persistent workgroups that request the next
// block's samples before the last pass 0.098.  The pointwise stage
// takes pairs (k, M - k) = (r, k') and (16 - r, 1023 - k') from two waves' regions; its tables are stored in that order.
template <bool ALIGNED>
__global__ void __launch_bounds__(1024, 4) fir_ols_split_kernel(FirArgs a) {
  constexpr int M = 16384, T = 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2 *z = reinterpret_cast<float2 *>(smem);
  const int tid = threadIdx.x;
  // Persistent: one workgroup per CU (128 KB of LDS) walks blocks blockIdx.x, + gridDim.x, ...; the NEXT block's sixteen sample
  // pairs are requested before the current block's last pass and stores, so a block no longer opens with an exposed trip to HBM.
  auto load_block = [&](int64_t b, c32 (&v)[16]) {
    const int64_t channel = b / a.blocks_per_channel, blk = b % a.blocks_per_channel;
    const float *x = a.x + channel * a.x_stride;
    const int64_t base = blk * a.step - a.lead;      // first sample of the window (even)
    int tl = threadIdx.x;
    asm volatile("" : "+v"(tl));
    if (ALIGNED && base >= 0 && base + 2 * M <= a.n) {     // the whole window lies inside the stream (block-uniform)
      const float2 *src = reinterpret_cast<const float2 *>(x + base) + tl;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float2 t = src[T * q];
        v[q] = {t.x, t.y};
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t g = base + 2 * (int64_t)(tl + T * q);
        v[q].x = (g >= 0 && g < a.n) ? x[g] : 0.0f;
        v[q].y = (g + 1 >= 0 && g + 1 < a.n) ? x[g + 1] : 0.0f;
      }
    }
  };
  const int64_t total = a.channels * a.blocks_per_channel;
  c32 r[16], nxt[16];
  load_block(blockIdx.x, nxt);
  for (int64_t b = blockIdx.x; b < total; b += gridDim.x) {
#pragma unroll
    for (int q = 0; q < 16; ++q) r[q] = nxt[q];
    {   // the radix-16 pass across the block, then w_M^(n' r)
      const float2 w1 = a.tw_m[tid];
      fft16(r);
      c32 w[16];
      twiddle_powers<16>(c32{w1.x, w1.y}, w);
#pragma unroll
      for (int j = 1; j < 16; ++j) r[j] = cmul(r[j], w[j]);
      const int pos = swz(tid);
#pragma unroll
      for (int j = 0; j < 16; ++j) z[1024 * j + pos] = make_float2(r[j].x, r[j].y);
    }
    __syncthreads();
    int lane = tid & 63, wave = tid >> 6;
    asm volatile("" : "+v"(lane));
    float2 *zr = z + 1024 * wave;
    fft_passes<10, false, true, float, true>(r, zr, lane, a.tw_1k);
    // r[4 i + j] = bin k' = lane + 64 i + 256 j of this wave's sub-transform
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) zr[swz(lane + 64 * i + 256 * j)] = make_float2(r[4 * i + j].x, r[4 * i + j].y);
    __syncthreads();
    // pointwise stage (real post-pass, product with H, inverse pre-pass) over the pairs (k, M - k)
    auto pair = [&](int rr, int kq, bool self) {
      const int rp = (16 - rr) & 15, kp = rr ? 1023 - kq : ((1024 - kq) & 1023);
      float2 *pa = z + 1024 * rr + swz(kq), *pb = z + 1024 * rp + swz(kp);
      const float2 A = *pa, B = *pb;
      const float2 w = a.w_split[1024 * rr + kq];
      const float2 hk = a.h_split[1024 * rr + kq];
      const float2 hp = (rr == 0 && kq == 0) ? a.h_split[M] : a.h_split[1024 * rp + kp];
      const c32 E = {A.x + B.x, A.y - B.y}, D = {A.x - B.x, A.y + B.y};
      const c32 t = cmul(D, c32{w.x, w.y});
      const c32 Xk = {E.x + t.y, E.y - t.x};                  // E - i w D          (= 2 X[k])
      const c32 Xp = {E.x - t.y, -(E.y + t.x)};               // conj(E + i w D)    (= 2 X[M-k])
      const c32 Yk = cmul(Xk, c32{hk.x, hk.y}), Yp = cmul(Xp, c32{hp.x, hp.y});
      const c32 P = {Yk.x + Yp.x, Yk.y - Yp.y}, Q = {Yk.x - Yp.x, Yk.y + Yp.y};
      const c32 u = cmul(Q, c32{w.x, -w.y});                  // conj(w) Q
      // the inverse runs as conj(FFT(conj .)): store the conjugates of Z'[k] = P + i conj(w) Q, Z'[M-k] = conj(P - i conj(w) Q)
      *pa = make_float2(P.x - u.y, -(P.y + u.x));
      if (!self) *pb = make_float2(P.x + u.y, P.y - u.x);
    };
    int tp = tid;
    asm volatile("" : "+v"(tp));
#pragma unroll
    for (int m = 1; m < 8; ++m) {
      pair(m, tp, false);
      if (m == 4) __builtin_amdgcn_sched_barrier(0);
    }
    if (tp < 512) pair(8, tp, false);          // (8, k') with (8, 1023 - k')
    else pair(0, tp - 512, tp == 512);         // (0, k') with (0, 1024 - k'), k' < 512; k' = 0 pairs with itself and carries bin M
    if (tid == 0) pair(0, 512, true);          // k = M/2
    __syncthreads();
    int li = tid & 63;
    asm volatile("" : "+v"(li));
    float2 *zi = z + 1024 * (tid >> 6);
    fft_passes<10, false, true, float, true>(r, zi, li, a.tw_1k);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) zi[swz(li + 64 * i + 256 * j)] = make_float2(r[4 * i + j].x, r[4 * i + j].y);
    __syncthreads();
    int to = tid;
    asm volatile("" : "+v"(to));
    {
      const int pos = swz(to);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float2 v = z[1024 * j + pos];
        r[j] = {v.x, v.y};
      }
    }
    __syncthreads();   // every wave holds its points: the next block's first pass may overwrite the buffer
    if (b + gridDim.x < total) load_block(b + gridDim.x, nxt);   // in flight across the last pass and the stores (issuing it after the twiddles measured the same)
    {
      const float2 w1 = a.tw_m[to];
      c32 w[16];
      twiddle_powers<16>(c32{w1.x, w1.y}, w);
#pragma unroll
      for (int j = 1; j < 16; ++j) r[j] = cmul(r[j], w[j]);
      fft16(r);
    }
    const int64_t channel = b / a.blocks_per_channel, blk = b % a.blocks_per_channel;
    float *y = a.y + channel * a.y_stride;
    const int64_t out0 = blk * a.step - a.out_shift;   // y index of the block's first kept sample
    const bool whole = out0 >= 0 && out0 + a.step <= a.n_out;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int64_t s = 2 * (int64_t)(to + T * q) - a.lead;   // position inside the kept span (even)
      if (s >= 0 && s < a.step) {
        const int64_t o = out0 + s;
        const float re = r[q].x, im = -r[q].y;   // conj(FFT(conj .))
        if (ALIGNED && whole) {
          *reinterpret_cast<float2 *>(y + o) = make_float2(re, im);
        } else {
          if (o >= 0 && o < a.n_out) y[o] = re;
          if (o + 1 >= 0 && o + 1 < a.n_out) y[o + 1] = im;
        }
      }
    }
  }
}

// ---- N = 32768 on the frame pipeline's register form (round 6): fir_ols_pk32_kernel ---------------------------------------------
// What bounded fir_ols_split_kernel (profiles/r06/fir_channels.log, profiles/r07 PMC): 22-27 us per block and CU whatever the channel
// count -- ~3000 vector instructions per thread x 16 waves (a third of them address arithmetic of the XOR-swizzled Stockham passes)
// and nine to ten trips of the 128 KB block through LDS, one after the other because five workgroup barriers keep all sixteen waves
// in the same phase.  Here the same decomposition M = 16384 = 16 x 1024 runs on the STFT pipeline's register form (stft_fast_p32.hpp):
//   * a 1024-point sub-transform lives in a HALF-WAVE, 32 lanes x 32 points, 1024 = 32 x 32: radix-32 in registers, the twiddle
//     W_1024^(l k1) from an LDS table, ONE 32 x 32 transposition through the sub-transform's own LDS region (pitch-33 cells, conflict
//     free), radix-32 in registers -- one LDS exchange per sub-transform where the Stockham form has two, no swizzle arithmetic;
//   * every butterfly is packed float32 (the generated v_pk_* stages of stft_pk_fft.inc, one issue slot per complex add / half a
//     complex product), the twiddle powers of the cross-block passes and the pointwise stage too (pk_powers16, pk_ols_pairs1);
//   * 512 threads (8 waves, 32 points a thread, up to 256 registers) instead of 1024 x 16 points.
// Per block: radix-16 across the block (thread n' takes columns n' and n' + 512) -> [barrier] -> sub-transforms -> [barrier] -> pointwise
// pairs (k, M - k) -> [barrier] -> sub-transforms -> [barrier] -> radix-16 across the block, stores.  Same arithmetic as the split kernel up
// to the order of roundings inside a complex product (p32_cmul's form); parity is the FIR contract's (1e-5 sum|h|, tests).
using f2 = float __attribute__((ext_vector_type(2)));
#include "stft_pk_fft.inc"
__device__ __forceinline__ void pk_fft32_nat(f2 (&v)[32]) {   // 32-point forward DFT, natural order in and out (stft_fast_p32.hpp: pk_fft32)
  f2 e[16], o[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) { e[m] = v[2 * m]; o[m] = v[2 * m + 1]; }
  pk_fft16(e);
  pk_fft16(o);
  pk_fft32_combine0(v, e, o);
  pk_fft32_combine1(v, e, o);
}
constexpr size_t kPkTwBytes = 31 * 32 * sizeof(float2);

// the 1024-point forward transform of a half-wave, registers to registers: lane l holds points l + 32 j in v on entry and bins
// l + 32 q in t on return; `cells` = 1056 four-byte cells of LDS that belong to the half-wave for the duration
struct PkNoMid {
  __device__ __forceinline__ void operator()() const {}
};
// `mid()` runs once the transposition's reads are issued: v is dead there, so requests placed in it have their registers and the
// second radix-32 pass to arrive under
template <class Mid = PkNoMid>
__device__ __forceinline__ void pk_sub1024(f2 (&v)[32], f2 (&t)[32], float *cells, int l, const float4 *twA4, const float2 *twA31, const Mid &mid = Mid{}) {
  float4 tw[15];
#pragma unroll
  for (int m = 0; m < 15; ++m) tw[m] = twA4[32 * m + l];
  const float2 tw31 = twA31[l];
  __builtin_amdgcn_sched_barrier(0);
  pk_fft32_nat(v);   // y_l[k1] = sum_j a[l + 32 j] W_32^(j k1)
  __builtin_amdgcn_sched_barrier(0);
#define SMX_TWV(m) f2{tw[m].x, tw[m].y}, f2{tw[m].z, tw[m].w}
  pk_twiddle8(v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], SMX_TWV(0), SMX_TWV(1), SMX_TWV(2), SMX_TWV(3));
  pk_twiddle8(v[9], v[10], v[11], v[12], v[13], v[14], v[15], v[16], SMX_TWV(4), SMX_TWV(5), SMX_TWV(6), SMX_TWV(7));
  pk_twiddle8(v[17], v[18], v[19], v[20], v[21], v[22], v[23], v[24], SMX_TWV(8), SMX_TWV(9), SMX_TWV(10), SMX_TWV(11));
  pk_twiddle7(v[25], v[26], v[27], v[28], v[29], v[30], v[31], SMX_TWV(12), SMX_TWV(13), SMX_TWV(14), f2{tw31.x, tw31.y});
#undef SMX_TWV
  __builtin_amdgcn_sched_barrier(0);
  // lane l register k1 -> lane k1 register l: lane l writes register j to cell 33 l + j, lane k1 reads register l' from cell
  // 33 l' + k1 (bank = lane + register mod 32 on both sides: conflict free); the real parts, then the imaginary parts.  A wave's
  // LDS operations execute in order, so no wait separates the rounds.
  float *wc = cells + 33 * l, *rc = cells + l;
#pragma unroll
  for (int j = 0; j < 32; ++j) wc[j] = v[j].x;
#pragma unroll
  for (int i = 0; i < 32; ++i) t[i].x = rc[33 * i];
#pragma unroll
  for (int j = 0; j < 32; ++j) wc[j] = v[j].y;
#pragma unroll
  for (int i = 0; i < 32; ++i) t[i].y = rc[33 * i];
  __builtin_amdgcn_sched_barrier(0);
  mid();
  __builtin_amdgcn_sched_barrier(0);
  pk_fft32_nat(t);   // lane k1, register q: bin k1 + 32 q
  __builtin_amdgcn_sched_barrier(0);
}

// A WAVE holds a sub-transform and its partner in the pointwise stage: half 0 sub-transform r, half 1 sub-transform 16 - r (waves
// 1 .. 7; wave 0: sub-transforms 0 and 8, which pair inside themselves).  Bin k = 16 k' + r pairs with M - k = 16 (1023 - k') + (16 - r):
// lane l, register q of one half against lane 31 - l, register 31 - q of the other -- so forward sub-transform, pointwise stage and
// inverse sub-transform are ONE chain inside the wave, through the wave's own LDS cells and with no workgroup barrier: the eight
// waves drift apart and one wave's LDS round trips run under the others' butterflies (fir_ols_split_kernel held all sixteen waves
// in one phase with a barrier either side of the pointwise stage and sent the block through LDS twice more).
// Every pair is formed ONCE, by the member with q < 16: exchange 1 hands it the partner's registers 16 .. 31, exchange 2 returns
// the partner's results.  Sub-transform 0 pairs (0, k') with (0, 1024 - k'): lane (32 - l) mod 32, and lane 0 pairs inside itself one
// register further (k' = 32 q with 32 (32 - q)); k' = 0 (whose partner in the product is bin M) and k' = 512 pair with themselves.
// NS = 16: N = 32768 (M = 16 x 1024), 512 threads, 139 KB of LDS, one workgroup per CU.  NS = 8: N = 16384 (M = 8 x 1024: filters of
// up to 4096 taps at 75 % kept, up to 8192 at 50 %), 256 threads with FOUR columns each, 73.5 KB of LDS: TWO workgroups per CU, whose
// barrier-separated phases interleave -- what one block of 128 KB cannot have.
template <int NS> struct PkRadix;
template <> struct PkRadix<16> {
  static __device__ __forceinline__ void dft(f2 (&v)[16]) { pk_fft16(v); }
  static __device__ __forceinline__ void twiddle(f2 (&v)[16], f2 w1) {
    f2 w[16];
    w[1] = w1;
    pk_powers16(w);
    pk_twiddle8(v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8]);
    pk_twiddle7(v[9], v[10], v[11], v[12], v[13], v[14], v[15], w[9], w[10], w[11], w[12], w[13], w[14], w[15]);
  }
};
template <> struct PkRadix<8> {
  static __device__ __forceinline__ void dft(f2 (&v)[8]) { pk_fft8(v); }
  static __device__ __forceinline__ void twiddle(f2 (&v)[8], f2 w1) {
    f2 w[16];
    w[1] = w1;
    pk_powers8(w);
    pk_twiddle7(v[1], v[2], v[3], v[4], v[5], v[6], v[7], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
  }
};
template <int NS> constexpr size_t pk_lds_bytes() { return (size_t)NS * 1024 * sizeof(float2) + kPkTwBytes; }

template <bool ALIGNED, int NS>
__global__ void __launch_bounds__(32 * NS) fir_ols_pk32_kernel(FirArgs a) {
  constexpr int M = 1024 * NS, T = 32 * NS, CPT = 1024 / T;   // threads, columns per thread (2 / 4): CPT x NS = 32 points a thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f2 *z = reinterpret_cast<f2 *>(smem);   // [16][1024]: sub-transform r at z + 1024 r
  const float4 *twA4 = reinterpret_cast<const float4 *>(smem + (size_t)M * sizeof(float2));
  const float2 *twA31 = reinterpret_cast<const float2 *>(smem + (size_t)M * sizeof(float2) + 15 * 32 * sizeof(float4));
  const int tid = threadIdx.x;
  for (int i = tid; i < 31 * 32; i += T) reinterpret_cast<float2 *>(smem + (size_t)M * sizeof(float2))[i] = a.tw_a32[i];
  // columns tid + T c (c < CPT) of the block: element n = column + 1024 q, q < NS
  auto load_block = [&](int64_t b, f2 (&va)[CPT][NS]) {
    const int64_t channel = b / a.blocks_per_channel, blk = b % a.blocks_per_channel;
    const float *x = a.x + channel * a.x_stride;
    const int64_t base = blk * a.step - a.lead;      // first sample of the window (even)
    int tl = threadIdx.x;
    asm volatile("" : "+v"(tl));
    if (ALIGNED && base >= 0 && base + 2 * M <= a.n) {     // the whole window lies inside the stream (block-uniform)
      // (a wave-uniform base and ONE 32-bit lane offset: the sixteen row offsets are immediates)
      const char *src = reinterpret_cast<const char *>(x + base);
      const unsigned off = 8u * (unsigned)tl;
#pragma unroll
      for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
          const float2 p = *reinterpret_cast<const float2 *>(src + (size_t)(off + 8192u * (unsigned)q + 8u * (unsigned)(T * c)));
          va[c][q] = f2{p.x, p.y};
        }
    } else {
#pragma unroll
      for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
          const int64_t g = base + 2 * (int64_t)(tl + T * c + 1024 * q);
          va[c][q] = f2{(g >= 0 && g < a.n) ? x[g] : 0.0f, (g + 1 >= 0 && g + 1 < a.n) ? x[g + 1] : 0.0f};
        }
    }
  };
  const int64_t total = a.channels * a.blocks_per_channel;
#ifdef SMX_STAMPS
  unsigned long long stamp_sum[kStampSlots] = {0}, stamp_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
  const unsigned long long clk_t0 = stamp_prev, clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  f2 w1[CPT];   // W_M^(column): the same for every block of this thread
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const float2 w = a.tw_m[tid + T * c];
    w1[c] = f2{w.x, w.y};
  }
  f2 na[CPT][NS];
  load_block(blockIdx.x, na);
  __syncthreads();   // the twiddle table
  for (int64_t b = blockIdx.x; b < total; b += gridDim.x) {
    {   // radix-NS across the block, then W_M^(n' r): u_r[n'] for sub-transform r
      int tw_ = tid;
      asm volatile("" : "+v"(tw_));
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        PkRadix<NS>::dft(na[c]);
        PkRadix<NS>::twiddle(na[c], w1[c]);
#pragma unroll
        for (int r = 0; r < NS; ++r) z[1024 * r + tw_ + T * c] = na[c][r];
      }
    }
    SMX_STAMP(0);
    __syncthreads();
    SMX_STAMP(1);
#ifdef SMX_STAMPS
    ++stamp_sum[22];
#endif
    {
      int ls = tid;
      asm volatile("" : "+v"(ls));
      const int l = ls & 31, h = (ls >> 5) & 1, wave = ls >> 6;
      const int rs = wave ? (h ? NS - wave : wave) : (NS / 2) * h;   // this half-wave's sub-transform
      f2 *zr = z + 1024 * rs;
      f2 v[32], t[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) v[j] = zr[l + 32 * j];
      // the pointwise stage's tables for this lane's pairs q < 16, bin (rs, kq = l + 32 q): w_k, H_k and H_(M-k) (the same for every
      // block of this thread, but 48 values do not stay in registers across the sub-transforms): requested inside the forward
      // sub-transform, behind its transposition, where the first pass's registers are free
      // (a wave-uniform base and a 32-bit unsigned lane offset per request: no 64-bit address pair per value)
      auto tab = [](const float2 *base, unsigned idx) { return *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(base) + (size_t)(idx * 8u)); };
      const unsigned rp = ((unsigned)NS - (unsigned)rs) & (unsigned)(NS - 1), ul = (unsigned)l, urs = (unsigned)rs;
      float2 pw[16], phk[16], php[16], w5, h5;
      auto request_tables = [&]() {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const unsigned kq = ul + 32u * (unsigned)q;
          pw[q] = tab(a.w_split, 1024u * urs + kq);
          phk[q] = tab(a.h_split, 1024u * urs + kq);
          php[q] = tab(a.h_split, (urs == 0u && kq == 0u) ? (unsigned)M : 1024u * rp + (urs ? 1023u - kq : 1024u - kq));
        }
      };
      auto request_partner_tables = [&]() {   // (H_(M-k) requested here instead, behind exchange 1's writes, measured the same: profiles/r08/ab_fir_php.log)
        w5 = tab(a.w_split, 512u);   // bin M/2 = (0, 512): lane 0 of sub-transform 0, with itself
        h5 = tab(a.h_split, 512u);
      };
      pk_sub1024(v, t, reinterpret_cast<float *>(zr), l, twA4, twA31, request_tables);
      SMX_STAMP(2);
      // exchange 1: rows of 17 eight-byte cells per lane (rows two banks apart: a half-wave's 32 accesses spread evenly over the banks);
      // a lane parks registers 16 .. 31 in cells 0 .. 15 of its row and register 0 in cell 16 (where lane 0 of sub-transform 0 finds
      // the partner of bin 0: itself); it reads, for q < 16, register 31 - q of its partner lane: cell 15 - q of that row -- lane 0 of
      // sub-transform 0: its OWN register 32 - q, cell 16 - q of its own row
      // the rows live in the half-wave's own region (32 x 17 x 8 = 4352 of its 8192 bytes; the transposition's cells before and after
      // use the same bytes -- a wave's LDS operations execute in order).  Partner: waves 1 .. 7 the other half's lane 31 - l; wave 0
      // the same half's lane 31 - l (sub-transform 8) or (32 - l) mod 32 (sub-transform 0)
      const int partner_lane = wave ? 31 - l : (h ? 31 - l : (32 - l) & 31);
      f2 *own_row = zr + 17 * l;
      f2 *partner_row = (wave ? z + 1024 * (NS - rs) : zr) + 17 * partner_lane + ((wave == 0 && h == 0 && l == 0) ? 1 : 0);
#pragma unroll
      for (int q = 16; q < 32; ++q) own_row[q - 16] = t[q];
      own_row[16] = t[0];
      request_partner_tables();
      f2 pb[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) pb[q] = partner_row[15 - q];
      // the pairs
      f2 bq[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        bq[q] = pb[q];
        pk_ols_pairs1(t[q], bq[q], f2{pw[q].x, pw[q].y}, f2{phk[q].x, phk[q].y}, f2{php[q].x, php[q].y});
      }
      f2 mid = t[16], mid_b = t[16];
      pk_ols_pairs1(mid, mid_b, f2{w5.x, w5.y}, f2{h5.x, h5.y}, f2{h5.x, h5.y});
      // exchange 2: the partner members' results go back: cell q of the own row holds the result for the partner's register 31 - q
#pragma unroll
      for (int q = 0; q < 16; ++q) own_row[q] = bq[q];
#pragma unroll
      for (int Q = 16; Q < 32; ++Q) t[Q] = partner_row[31 - Q];
      if (wave == 0 && h == 0 && l == 0) t[16] = mid;   // bin M/2 pairs with itself (the read above fetched cell 16 of the own row: stale)
      SMX_STAMP(3);
      // inverse sub-transform (as conj(FFT(conj .)): the pairs stored conjugates), its points back into the region for the last pass
      pk_sub1024(t, v, reinterpret_cast<float *>(zr), l, twA4, twA31);
#pragma unroll
      for (int q = 0; q < 32; ++q) zr[l + 32 * q] = v[q];
    }
    // the next block's samples are requested HERE, before the barrier: the sub-transforms' registers are free, and the requests have
    // the barrier, the last pass's LDS reads, its barrier and its arithmetic to arrive under (behind the second barrier, as the split
    // kernel has them, a block opened with ~1 us of exposed HBM latency: eight waves do not cover it)
    if (b + gridDim.x < total) load_block(b + gridDim.x, na);
    SMX_STAMP(4);
    __syncthreads();
    SMX_STAMP(5);
    int to = tid;
    asm volatile("" : "+v"(to));
    f2 va[CPT][NS];
#pragma unroll
    for (int c = 0; c < CPT; ++c)
#pragma unroll
      for (int r = 0; r < NS; ++r) va[c][r] = z[1024 * r + to + T * c];
    __syncthreads();   // every thread holds its columns: the next block's first pass may overwrite the buffer
    SMX_STAMP(6);
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      PkRadix<NS>::twiddle(va[c], w1[c]);
      PkRadix<NS>::dft(va[c]);
    }
    SMX_STAMP(7);
    const int64_t channel = b / a.blocks_per_channel, blk = b % a.blocks_per_channel;
    float *y = a.y + channel * a.y_stride;
    const int64_t out0 = blk * a.step - a.out_shift;   // y index of the block's first kept sample
    const bool whole = out0 >= 0 && out0 + a.step <= a.n_out;
    if (ALIGNED && whole) {   // (block-uniform) every kept sample of the block is wanted: 32-bit positions, one unsigned compare per value
      char *dst = reinterpret_cast<char *>(y + out0);
      const int s0 = 2 * to - (int)a.lead;
      const unsigned span = (unsigned)a.step;
#pragma unroll
      for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
          const int sp = s0 + 2 * T * c + 2048 * q;   // position inside the kept span (even)
          const f2 v = va[c][q];
          if ((unsigned)sp < span) *reinterpret_cast<float2 *>(dst + (size_t)(4u * (unsigned)sp)) = make_float2(v.x, -v.y);   // conj(FFT(conj .))
        }
    } else {
#pragma unroll
      for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
          const int64_t sp = 2 * (int64_t)(to + T * c + 1024 * q) - a.lead;
          if (sp >= 0 && sp < a.step) {
            const int64_t o = out0 + sp;
            const f2 v = va[c][q];
            if (o >= 0 && o < a.n_out) y[o] = v.x;
            if (o + 1 >= 0 && o + 1 < a.n_out) y[o + 1] = -v.y;
          }
        }
    }
    SMX_STAMP(8);
  }
#ifdef SMX_STAMPS
  stamp_sum[20] = __builtin_amdgcn_s_memtime() - clk_t0;
  stamp_sum[21] = __builtin_amdgcn_s_memrealtime() - clk_r0;
  if ((tid & 63) == 0 && blockIdx.x < 4096)
    for (int i = 0; i < kStampSlots; ++i) g_stamp_sums[(blockIdx.x * 16 + (tid >> 6)) * kStampSlots + i] = stamp_sum[i];
#endif
}

// ---- short filters (taps <= 128: pre-emphasis, DC blockers, smoothers, short anti-alias filters): the sum itself ----------------
// An FFT block of 1024 samples costs these the same 0.077 - 0.10 ms as a 256-tap filter (C4-shaped input: 0.30 of the roof).
// Direct form: a workgroup stages 2048 + taps - 1 input samples in LDS (zeros outside the stream); every thread forms EIGHT
// CONSECUTIVE outputs from a 16-value register window that slides down the taps eight at a time (two 16-byte LDS reads per 64
// multiply-adds), taps in ascending order, float32 fused multiply-adds (h rounded to float32, as the FFT path rounds H).
//   y[c][i] = sum_k h[k] x[c][out_shift + i - k]
struct FirDirectArgs {
  const float *x;
  float *y;
  int64_t n, x_stride, y_stride, n_out, out_shift;
  int taps;
  float h[128];        // zero beyond taps (the loop runs over whole groups of eight)
};
__global__ void __launch_bounds__(256) fir_direct_kernel(FirDirectArgs a) {
  constexpr int PER = 8, TILE = 256 * PER;
  __shared__ __attribute__((aligned(16))) float xs[TILE + 128 + 16];
  const int64_t c = blockIdx.y, o0 = (int64_t)blockIdx.x * TILE;      // first output of the tile
  const float *x = a.x + c * a.x_stride;
  const int groups = (a.taps + 7) >> 3, halo = 8 * groups;            // xs[0] is input sample out_shift + o0 - halo
  const int64_t base = a.out_shift + o0 - halo;
  for (int i = threadIdx.x; i < TILE + halo; i += 256) {
    const int64_t g = base + i;
    xs[i] = (g >= 0 && g < a.n) ? x[g] : 0.0f;
  }
  __syncthreads();
  // thread t: outputs o0 + 8 t + j, j < 8.  x[out_j - k] sits at xs[8 t + j - k + halo].  Window for taps 8 g .. 8 g + 7:
  // W[i] = xs[8 t + halo - 8 g - 8 + i], i < 16, and x[out_j - (8 g + kk)] = W[j - kk + 8].
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  const float *p = xs + 8 * threadIdx.x + halo;
  float acc[PER], w[16];
#pragma unroll
  for (int j = 0; j < PER; ++j) acc[j] = 0.0f;
  {
    const f32x4 u0 = *reinterpret_cast<const f32x4 *>(p), u1 = *reinterpret_cast<const f32x4 *>(p + 4);
    w[8] = u0[0]; w[9] = u0[1]; w[10] = u0[2]; w[11] = u0[3]; w[12] = u1[0]; w[13] = u1[1]; w[14] = u1[2]; w[15] = u1[3];
  }
  for (int g = 0; g < groups; ++g) {
    const f32x4 u0 = *reinterpret_cast<const f32x4 *>(p - 8 * g - 8), u1 = *reinterpret_cast<const f32x4 *>(p - 8 * g - 4);
    w[0] = u0[0]; w[1] = u0[1]; w[2] = u0[2]; w[3] = u0[3]; w[4] = u1[0]; w[5] = u1[1]; w[6] = u1[2]; w[7] = u1[3];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const float hk = a.h[8 * g + kk];                               // uniform: scalar load from the kernel arguments
#pragma unroll
      for (int j = 0; j < PER; ++j) acc[j] = __builtin_fmaf(hk, w[j - kk + 8], acc[j]);
    }
#pragma unroll
    for (int i = 15; i >= 8; --i) w[i] = w[i - 8];                    // the window slides down by eight taps
  }
  float *y = a.y + c * a.y_stride;
  const int64_t o = o0 + 8 * (int64_t)threadIdx.x;
  if (o + 8 <= a.n_out && ((reinterpret_cast<uintptr_t>(y + o) & 15) == 0)) {
    *reinterpret_cast<f32x4 *>(y + o) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *reinterpret_cast<f32x4 *>(y + o + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
  } else {
#pragma unroll
    for (int j = 0; j < PER; ++j)
      if (o + j < a.n_out) y[o + j] = acc[j];
  }
}

// ---- resample stages ("Resample OLS stages -- true rate conversion on the FIR kernel") ----------
// xu[c][q L] = x[c][q], zeros between: the interpolated-rate input of a xL stage
__global__ void __launch_bounds__(256) zero_stuff_kernel(const float *x, int64_t n, int64_t x_stride, int l, float *xu,
                                                         int64_t u_stride) {
  const int64_t c = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n * l; i += (int64_t)gridDim.x * 256)
    xu[c * u_stride + i] = (i % l == 0) ? x[c * x_stride + i / l] : 0.0f;
}
// y[c][i] = v[c][i M]: the kept phase of a /M stage
__global__ void __launch_bounds__(256) decimate_kernel(const float *v, int64_t v_stride, int m, float *y, int64_t n_out,
                                                       int64_t y_stride) {
  const int64_t c = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * 256)
    y[c * y_stride + i] = v[c * v_stride + i * m];
}

// ---- a pure xL or /M stage in its polyphase form, block by block in the frequency domain ----------------------------
// The stage is y[i] = sum_t proto[t] xu[i M + K L - t], xu = x zero-stuffed by L.  Split t = p + j F (F = L or M):
//   xL:  y[i L + p] = sum_j h_p[j] x[i + K - j],                 h_p[j] = proto[p + j L]   (2 K + 1 taps per phase)
//   /M:  y[i] = sum_p sum_j h_p[j] x_p[i - j], x_p[q] = x[q M + K - p], h_p[j] = proto[p + j M]   (2 K / M + 1 taps)
// i.e. L filters of the input at the input rate, or the sum of M filters of the input's phases at the output rate: the
// same arithmetic as the reference's spectral shortcut (one spectrum replicated L times against the prototype's on the fine
// grid / folded M times, resample_stubs.c:329-372) regrouped so that every transform is a power of two at the LOW rate --
// 1/L (1/M) of the flops of filtering the zero-stuffed signal, and no interpolated-rate scratch.
// Overlap-save blocks of N points at the low rate, two real blocks per complex transform (the packed pair of
// fir_ols_kernel): block pair P owns low-rate outputs [2 P V, 2 P V + 2 V), V = N - taps + 1, on a grid fixed to the stream's
// first sample -- so a streaming caller that runs pair P whenever its inputs are in computes the very values the offline
// call does (the partition law of Resample.Kernel, resample.mli:296-317).
struct PolyArgs {
  const float *x;            // x[s - x0] is sample s of the stream for x_lo <= s < x_hi (x_lo >= x0: the buffer starts at x0); zero elsewhere
  float *y;                  // y[o - o0] for outputs o0 <= o < o0 + n_out
  int64_t x0, x_lo, x_hi, x_stride, y_stride, o0, n_out;
  int64_t pair0, pairs;      // block pairs [pair0, pair0 + pairs) of every channel
  int l, m, k, taps, valid;
  const float2 *h;           // [phases][N] spectra of the phase filters, natural order, 1/N folded in
  const float2 *tw;          // exp(-2 pi i j / N), j < N/2
};

template <int LOG2N, bool UP>
__global__ void __launch_bounds__((1 << LOG2N) / 16) resample_poly_kernel(PolyArgs a) {
  constexpr int N = 1 << LOG2N, T = N / 16;
  constexpr int RL = LastPass<LOG2N>::R, NSL = LastPass<LOG2N>::NS, GL = 16 / RL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2 *z = reinterpret_cast<float2 *>(smem);
  float2 *zf = z + N;   // the pair's spectrum (xL) / the sum over the phases (/M): every thread touches its own 16 cells only
  const int tid = threadIdx.x;
  const int64_t channel = blockIdx.x / a.pairs;
  const int64_t pair = a.pair0 + blockIdx.x % a.pairs;
  const int64_t i_a = 2 * pair * a.valid;   // first low-rate output of block a; block b: + valid
  const int skip = a.taps - 1;
  const int F = UP ? a.l : a.m;
  // 32-bit arithmetic relative to two origins, sample q0 of the stream (the first one block a's window may touch) and block
  // a's first output; offset e is inside the stream (the call's outputs) <=> lo <= e < hi.  Loads and stores address
  // base + unsigned offset from the first valid element: one code path for interior and border blocks, no 64-bit lanes.
  auto clamp32 = [](int64_t v) { return (int)(v < -(1 << 30) ? -(1 << 30) : (v > (1 << 30) ? (1 << 30) : v)); };
  const int64_t q0 = UP ? i_a + a.k - skip : (i_a - skip) * a.m + a.k - (a.m - 1);
  const int lo = clamp32(a.x_lo - q0), hi = clamp32(a.x_hi - q0);
  const bool some = hi > lo;                                                // block-uniform: the pair sees a sample at all
  const float *xb = a.x + channel * a.x_stride + (q0 - a.x0) + (some ? lo : 0);   // the first valid sample
  const unsigned xn = some ? (unsigned)(hi - lo) : 0u;
  auto sample = [&](int e) {
    const unsigned u = (unsigned)(e - lo);
    const float v = xb[u < xn ? u : 0u];
    return u < xn ? v : 0.0f;
  };
  const int64_t o_a = (UP ? i_a * a.l : i_a) - a.o0;
  const int olo = clamp32(-o_a), ohi = clamp32(a.n_out - o_a);
  float *yb = a.y + channel * a.y_stride + o_a + olo;
  const unsigned yn = ohi > olo ? (unsigned)(ohi - olo) : 0u;
  auto emit = [&](int e, float v) {
    const unsigned u = (unsigned)(e - olo);
    if (u < yn) yb[u] = v;
  };
  const int ob = UP ? a.valid * a.l : a.valid;                              // block b's outputs start ob after block a's
  c32 r[16];
  if constexpr (UP) {
    if (some) {
#pragma unroll
      for (int m = 0; m < 16; ++m) r[m] = {sample(tid + T * m), sample(a.valid + tid + T * m)};
    } else {
#pragma unroll
      for (int m = 0; m < 16; ++m) r[m] = {0.0f, 0.0f};
    }
    fft_passes<LOG2N, true>(r, z, tid, a.tw);
#pragma unroll
    for (int i = 0; i < GL; ++i)
#pragma unroll
      for (int j = 0; j < RL; ++j) zf[swz(out_index<RL, NSL, T>(tid, i, j))] = make_float2(r[i * RL + j].x, r[i * RL + j].y);
#pragma unroll 1
    for (int p = 0; p < F; ++p) {
      __builtin_amdgcn_sched_barrier(0);
      const float2 *hp = a.h + p * N;
#pragma unroll
      for (int i = 0; i < GL; ++i)
#pragma unroll
        for (int j = 0; j < RL; ++j) {   // Y = Z H_p; the inverse runs as conj(FFT(conj Y))
          const int idx = out_index<RL, NSL, T>(tid, i, j);
          const float2 hv = hp[idx], zv = zf[swz(idx)];
          const c32 yv = cmul(c32{zv.x, zv.y}, c32{hv.x, hv.y});
          z[swz(idx)] = make_float2(yv.x, -yv.y);
        }
      __builtin_amdgcn_sched_barrier(0);
      fft_passes<LOG2N, false>(r, z, tid, a.tw);
#pragma unroll
      for (int i = 0; i < GL; ++i)
#pragma unroll
        for (int j = 0; j < RL; ++j) {
          const int idx = out_index<RL, NSL, T>(tid, i, j) - skip;
          if (idx >= 0) {
            emit(idx * F + p, r[i * RL + j].x);
            emit(idx * F + p + ob, -r[i * RL + j].y);
          }
        }
    }
  } else {
    const int bo = a.valid * F;                                             // block b's window starts bo samples after block a's
#pragma unroll 1
    for (int p = 0; p < F; ++p) {
      const int ph = F - 1 - p;                                             // x_p[q] = x[q M + K - p]: offset M - 1 - p from q0
      if (some) {
#pragma unroll
        for (int m = 0; m < 16; ++m) r[m] = {sample((tid + T * m) * F + ph), sample((tid + T * m) * F + ph + bo)};
      } else {
#pragma unroll
        for (int m = 0; m < 16; ++m) r[m] = {0.0f, 0.0f};
      }
      fft_passes<LOG2N, true>(r, z, tid, a.tw);
      __builtin_amdgcn_sched_barrier(0);   // (the table reads below stay below: hoisted above the transform they cost 60 registers)
      const float2 *hp = a.h + p * N;
      float2 *dst = p == F - 1 ? z : zf;   // the last phase leaves the sum where the inverse reads it
#pragma unroll
      for (int i = 0; i < GL; ++i)
#pragma unroll
        for (int j = 0; j < RL; ++j) {
          const int idx = out_index<RL, NSL, T>(tid, i, j);
          const float2 hv = hp[idx];
          const c32 yv = cmul(r[i * RL + j], c32{hv.x, hv.y});
          // the sum over the phases, conjugated for the inverse (conj(FFT(conj Y)))
          float2 acc = make_float2(0.0f, 0.0f);
          if (p > 0) acc = zf[swz(idx)];
          dst[swz(idx)] = make_float2(acc.x + yv.x, acc.y - yv.y);
        }
    }
    fft_passes<LOG2N, false>(r, z, tid, a.tw);
#pragma unroll
    for (int i = 0; i < GL; ++i)
#pragma unroll
      for (int j = 0; j < RL; ++j) {
        const int e = out_index<RL, NSL, T>(tid, i, j) - skip;
        if (e >= 0) {
          emit(e, r[i * RL + j].x);
          emit(e + ob, -r[i * RL + j].y);
        }
      }
  }
}

// The block identity of the reference's overlap-save executor, `soundml_resample_shape_run`
// (resample_stubs.c:329-372), operation for operation in float64: one thread per output bin.  Every product is the
// plain four-multiply form with each operation rounded on its own (see cx_mul_exact for what keeps hipcc
// from contracting them into fused multiply-adds), and the /M fold adds its terms in ascending fold order, so the result is bit for
// bit what the C stub computes.
struct cx128 { double re, im; };
__device__ __forceinline__ cx128 cx_mul_exact(cx128 a, cx128 b) {
  // The library is built with -ffp-contract=fast, under which the backend fuses a product into the following add whatever
  // the source says (HIP's __dmul_rn / __dadd_rn are plain operators, `#pragma clang fp contract(off)` does not survive
  // inlining): the empty asm makes each rounded product opaque, so it cannot become half of a fused multiply-add.
  cx128 r;
  double p0 = a.re * b.re, p1 = a.im * b.im, p2 = a.re * b.im, p3 = a.im * b.re;
  asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
  r.re = p0 - p1;
  r.im = p2 + p3;
  return r;
}
__global__ void __launch_bounds__(256) resample_shape_kernel(const cx128 *x, const cx128 *h, cx128 *y, int64_t lines,
                                                             int64_t n, int64_t sl, int64_t sm, int64_t w) {
  const int64_t bins = n / 2 + 1, obins = w / 2 + 1, half = n / 2;
  const int64_t total = lines * obins;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t line = e / obins, k = e - line * obins;
    const cx128 *xs = x + line * bins;
    cx128 out;
    if (sl > 1) {                       // xL: bin k reads full-grid bin k mod N of the block's spectrum
      const int64_t j = k % n;
      cx128 z = j <= half ? xs[j] : xs[n - j];
      if (j > half) z.im = -z.im;
      out = cx_mul_exact(z, h[k]);
    } else if (sm > 1) {                // /M: product on the half grid, alias fold in ascending order
      int64_t j = k;
      out = cx_mul_exact(xs[k], h[k]);
      for (int64_t r = 1; r < sm; ++r) {
        j += w;
        cx128 p;
        if (j <= half) {
          p = cx_mul_exact(xs[j], h[j]);
        } else {
          p = cx_mul_exact(xs[n - j], h[n - j]);
          p.im = -p.im;
        }
        out.re = out.re + p.re;
        out.im = out.im + p.im;
      }
    } else {
      out = cx_mul_exact(xs[k], h[k]);
    }
    y[e] = out;
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
    if (log2n == 15 || log2n == 14) {   // the split kernels' tables (M = NS x 1024 points, NS = 16 / 8 sub-transforms), in the order their lanes read them
      const int64_t ns = M / 1024;
      std::vector<float2> hs((size_t)M + 1), ws((size_t)M), t1k(512);
      for (int64_t k = 0; k < M; ++k) {
        const size_t at = (size_t)((k % ns) * 1024 + k / ns);
        hs[at] = hh[(size_t)k];
        const double ang = -2.0 * M_PI * (double)k / (double)N;
        ws[at] = make_float2((float)std::cos(ang), (float)std::sin(ang));
      }
      hs[(size_t)M] = hh[(size_t)M];
      for (int j = 0; j < 512; ++j) {
        const double ang = -2.0 * M_PI * (double)j / 1024.0;
        t1k[(size_t)j] = make_float2((float)std::cos(ang), (float)std::sin(ang));
      }
      SMX_HIP_CHECK(hipMalloc((void **)&t.h_split, hs.size() * sizeof(float2)));
      SMX_HIP_CHECK(hipMemcpy(t.h_split, hs.data(), hs.size() * sizeof(float2), hipMemcpyHostToDevice));
      SMX_HIP_CHECK(hipMalloc((void **)&t.w_split, ws.size() * sizeof(float2)));
      SMX_HIP_CHECK(hipMemcpy(t.w_split, ws.data(), ws.size() * sizeof(float2), hipMemcpyHostToDevice));
      SMX_HIP_CHECK(hipMalloc((void **)&t.tw_1k, t1k.size() * sizeof(float2)));
      SMX_HIP_CHECK(hipMemcpy(t.tw_1k, t1k.data(), t1k.size() * sizeof(float2), hipMemcpyHostToDevice));
      std::vector<float2> ta(31 * 32);
      auto w1k = [](int e) {
        const double ang = -2.0 * M_PI * (double)(e % 1024) / 1024.0;
        return make_float2((float)std::cos(ang), (float)std::sin(ang));
      };
      for (int m = 0; m < 15; ++m)
        for (int l = 0; l < 32; ++l) {
          ta[(size_t)(2 * (32 * m + l))] = w1k(l * (2 * m + 1));
          ta[(size_t)(2 * (32 * m + l) + 1)] = w1k(l * (2 * m + 2));
        }
      for (int l = 0; l < 32; ++l) ta[(size_t)(2 * 15 * 32 + l)] = w1k(l * 31);
      SMX_HIP_CHECK(hipMalloc((void **)&t.tw_a32, ta.size() * sizeof(float2)));
      SMX_HIP_CHECK(hipMemcpy(t.tw_a32, ta.data(), ta.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
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
    (void)hipFree(kv.second.h_split);
    (void)hipFree(kv.second.w_split);
    (void)hipFree(kv.second.tw_1k);
    (void)hipFree(kv.second.tw_a32);
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

}  // namespace

namespace smx {
// y[c][i] = (h * x[c])[out_shift + i], i in [0, n_out): the convolution of n input samples (zeros outside), any window of it
void fir_apply_window_dev(const smx_fir_plan &p, const float *d_x, int64_t channels, int64_t n, int64_t x_stride,
                          float *d_y, int64_t y_stride, int64_t n_out, int64_t out_shift, hipStream_t stream) {
  if (channels < 0 || n < 0 || n_out < 0 || out_shift < 0) throw Failure("fir_apply: negative extent");
  if (channels == 0 || n_out == 0) return;
  if (x_stride < n || y_stride < n_out) throw Failure("fir_apply: stride smaller than the signal length");
  if (!d_x || !d_y) throw Failure("fir_apply: null device pointer");
  const smx_fir_plan::Tables &t = p.tables();
  FirArgs a{};
  a.x = d_x;
  a.y = d_y;
  a.n = n;
  a.n_out = n_out;
  a.out_shift = out_shift;
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
  a.h_split = t.h_split;
  a.w_split = t.w_split;
  a.tw_1k = t.tw_1k;
  a.tw_a32 = t.tw_a32;
  static const bool direct_off = diag_flag("SMX_FIR_DIRECT") == 0;   // A/B timing: FFT blocks for short filters too
  static const int64_t direct_max = (int64_t)diag_int("SMX_FIR_DIRECT_MAX", 80);
  if (p.taps <= direct_max && p.taps <= 128 && !direct_off && channels <= 65535) {   // measured crossover with the FFT blocks: see DESIGN 4.5
    FirDirectArgs da{};
    da.x = d_x;
    da.y = d_y;
    da.n = n;
    da.x_stride = x_stride;
    da.y_stride = y_stride;
    da.n_out = n_out;
    da.out_shift = out_shift;
    da.taps = (int)p.taps;
    for (int64_t k = 0; k < p.taps; ++k) da.h[k] = (float)p.h[(size_t)k];
    const int64_t tiles = (n_out + 2047) / 2048;
    if (tiles > 0x7fffffff) throw Failure("fir_apply: too many blocks for one launch");
    SMX_LAUNCH(fir_direct_kernel, dim3((unsigned)tiles, (unsigned)channels), dim3(256), 0, stream, da);
    SMX_HIP_CHECK(hipGetLastError());
    return;
  }
  static const bool packed_env = diag_flag("SMX_FIR_PACKED") == 1;
  const bool packed = packed_env && p.log2n <= 14;
  if (!packed) {   // one real block per workgroup, half-size transform
    a.lead = (p.taps - 1 + 1) & ~int64_t(1);                 // even, >= taps - 1
    a.step = p.nfft - a.lead;
    a.blocks_per_channel = (out_shift + n_out + a.step - 1) / a.step;
    const int64_t grid = channels * a.blocks_per_channel;
    if (grid > 0x7fffffff) throw Failure("fir_apply: too many blocks for one launch");
    const size_t lds = (size_t)(p.nfft / 2) * sizeof(float2);
    const bool aligned = x_stride % 2 == 0 && y_stride % 2 == 0 && out_shift % 2 == 0 &&
                         reinterpret_cast<uintptr_t>(d_x) % 8 == 0 && reinterpret_cast<uintptr_t>(d_y) % 8 == 0;
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
      case 14: {
        // N = 16384 stays on the Stockham kernel: the register pipeline with eight sub-transforms and TWO workgroups per CU (73.5 KB of LDS
        // each) measured 4 % behind it (8192 taps 0.1316 against 0.1286 ms, 4096 taps 0.1008 / 0.0969, 3000 taps 0.0922 / 0.0882:
        // profiles/r08/fir_time_n16384.log) -- two independent workgroups did not cover what bounds a wave there, its own chain of LDS round
        // trips at two waves per SIMD.  SMX_FIR_PK=1 in diagnostic builds selects it (same results to rounding; A/B timing).
        const bool stockham = diag_flag("SMX_FIR_PK") != 1;
        if (stockham) aligned ? launch(fir_ols_real_kernel<13, true>, 512) : launch(fir_ols_real_kernel<13, false>, 512);
        else {   // persistent, two workgroups per CU
          a.channels = channels;
          const int64_t slots = 2 * (int64_t)device_cu_count();
          const unsigned g = (unsigned)(grid < slots ? grid : slots);
          auto launch_p = [&](auto kernel) {
            SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pk_lds_bytes<8>()));
            SMX_LAUNCH(kernel, dim3(g), dim3(256), pk_lds_bytes<8>(), stream, a);
          };
          aligned ? launch_p(fir_ols_pk32_kernel<true, 8>) : launch_p(fir_ols_pk32_kernel<false, 8>);
        }
        break;
      }
      case 15: {
        static const bool pass_by_pass = diag_flag("SMX_FIR_SPLIT") == 0;   // A/B timing
        const bool split16 = diag_flag("SMX_FIR_PK") == 0;                 // A/B timing (read per launch): round 2-5's wave-split kernel
        if (pass_by_pass) aligned ? launch(fir_ols_real_kernel<14, true>, 1024) : launch(fir_ols_real_kernel<14, false>, 1024);
        else {
          a.channels = channels;
          const int64_t cus = device_cu_count();
          const unsigned g = (unsigned)(grid < cus ? grid : cus);   // persistent: one workgroup per CU walks the blocks
          auto launch_p = [&](auto kernel, int threads, size_t bytes) {
            SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            SMX_LAUNCH(kernel, dim3(g), dim3(threads), bytes, stream, a);
          };
          if (split16) aligned ? launch_p(fir_ols_split_kernel<true>, 1024, lds) : launch_p(fir_ols_split_kernel<false>, 1024, lds);
          else aligned ? launch_p(fir_ols_pk32_kernel<true, 16>, 512, pk_lds_bytes<16>()) : launch_p(fir_ols_pk32_kernel<false, 16>, 512, pk_lds_bytes<16>());   // the register pipeline (round 6)
        }
        break;
      }
      default: throw Failure("fir_apply: unsupported block size");
    }
    SMX_HIP_CHECK(hipGetLastError());
    return;
  }
  const int64_t blocks = (out_shift + n_out + p.valid - 1) / p.valid;
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
}  // namespace smx

namespace {
void fir_apply_dev(const smx_fir_plan &p, const float *d_x, int64_t channels, int64_t n, int64_t x_stride,
                   float *d_y, int64_t y_stride, hipStream_t stream) {
  if (!d_x || !d_y) {
    if (channels > 0 && n > 0) throw Failure("fir_apply: null device pointer");
  }
  fir_apply_window_dev(p, d_x, channels, n, x_stride, d_y, y_stride, n, 0, stream);
}
}  // namespace

#ifdef SMX_STAMPS
extern "C" int smx_debug_read_stamps_fir(unsigned long long *out, int count) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(smx::fftdev::g_stamp_sums), sizeof(unsigned long long) * (size_t)count);
}
#endif

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
    static const bool big = diag_flag("SMX_FIR_BIG") != 0;
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


/* ---- Resample stages ------------------------------------------------------------------------------------------- */
}  // extern "C"

struct smx_resample_stage {
  int64_t l = 1, m = 1, k = 0;
  smx_fir_plan *fir = nullptr;      // a general L / M stage (and 1 / 1): one block convolution at the interpolated rate
  // a pure xL or /M stage (the overlap-save eligible ones, resample.ml:279-300): polyphase blocks (resample_poly_kernel)
  bool poly = false;
  int phases = 0, log2n = 0;
  int64_t taps = 0, nfft = 0, valid = 0;   // per phase: taps, block length N, kept outputs V = N - taps + 1
  std::vector<double> proto;
  struct Tables {
    float2 *h = nullptr;    // [phases][N]
    float2 *tw = nullptr;
  };
  const Tables &tables() const;
  ~smx_resample_stage() {
    delete fir;
    for (auto &kv : tables_) {
      (void)hipFree(kv.second.h);
      (void)hipFree(kv.second.tw);
    }
  }

 private:
  mutable std::mutex mutex_;
  mutable std::map<int, Tables> tables_;
};

// Resample.Kernel (resample.mli:270-319) of ONE overlap-save stage: all channels in one state, the unconsumed input on the
// device, block pairs executed as their inputs complete (always whole pairs on the stream's grid: the partition law)
struct smx_resample_kernel {
  const smx_resample_stage *stage = nullptr;   // borrowed: outlives the kernel
  int64_t channels = 0, max_block = 0;
  int64_t fed = 0, emitted = 0, pairs_done = 0;
  int64_t carry_from = 0, carry_cap = 0;       // carry[c][i] = sample carry_from + i of channel c, i < fed - carry_from
  float *carry = nullptr;
  bool drained = false;
  int device = 0;
  ~smx_resample_kernel() {
    if (carry) (void)hipFree(carry);
  }
};

// in-place radix-2 DFT of 2^bits complex points in float64 (natural order in and out): the phase filters' spectra
static void dft_pow2_host(std::vector<double> &re, std::vector<double> &im, int bits) {
  const size_t n = (size_t)1 << bits;
  for (size_t i = 0; i < n; ++i) {
    const size_t j = smx::brev_host((unsigned)i, bits);
    if (i < j) {
      std::swap(re[i], re[j]);
      std::swap(im[i], im[j]);
    }
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const size_t half = len >> 1;
    for (size_t j = 0; j < half; ++j) {
      const double ang = -2.0 * M_PI * (double)j / (double)len, wr = std::cos(ang), wi = std::sin(ang);
      for (size_t i0 = j; i0 < n; i0 += len) {
        const size_t i1 = i0 + half;
        const double tr = re[i1] * wr - im[i1] * wi, ti = re[i1] * wi + im[i1] * wr;
        re[i1] = re[i0] - tr;
        im[i1] = im[i0] - ti;
        re[i0] += tr;
        im[i0] += ti;
      }
    }
  }
}

const smx_resample_stage::Tables &smx_resample_stage::tables() const {
  smx::init_device_pool();
  int device = 0;
  SMX_HIP_CHECK(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(mutex_);
  auto it = tables_.find(device);
  if (it != tables_.end()) return it->second;
  const int64_t N = nfft, f = phases;
  std::vector<float2> hb((size_t)(f * N)), tw((size_t)(N / 2));
  std::vector<double> re((size_t)N), im((size_t)N);
  for (int64_t p = 0; p < f; ++p) {   // h_p[j] = proto[p + j F], zero beyond the prototype; H_p / N in natural order
    std::fill(re.begin(), re.end(), 0.0);
    std::fill(im.begin(), im.end(), 0.0);
    for (int64_t j = 0; j < taps; ++j) {
      const int64_t t = p + j * f;
      if (t < (int64_t)proto.size()) re[(size_t)j] = proto[(size_t)t];
    }
    dft_pow2_host(re, im, log2n);
    for (int64_t i = 0; i < N; ++i)
      hb[(size_t)(p * N + i)] = make_float2((float)(re[(size_t)i] / (double)N), (float)(im[(size_t)i] / (double)N));
  }
  for (int64_t j = 0; j < N / 2; ++j) {
    const double ang = -2.0 * M_PI * (double)j / (double)N;
    tw[(size_t)j] = make_float2((float)std::cos(ang), (float)std::sin(ang));
  }
  Tables t;
  SMX_HIP_CHECK(hipMalloc((void **)&t.h, hb.size() * sizeof(float2)));
  SMX_HIP_CHECK(hipMemcpy(t.h, hb.data(), hb.size() * sizeof(float2), hipMemcpyHostToDevice));
  SMX_HIP_CHECK(hipMalloc((void **)&t.tw, tw.size() * sizeof(float2)));
  SMX_HIP_CHECK(hipMemcpy(t.tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
  return tables_.emplace(device, t).first->second;
}

namespace {
// resample.ml:279-300
bool ols_geom_host(int64_t rate, int64_t l, int64_t m, int64_t k, int64_t &n, int64_t &b, int64_t &delta) {
  const int64_t f_div = l == 1 ? m : 1;
  const int64_t target = std::max<int64_t>(64, 10 * k);
  n = f_div % 3 == 0 ? 3 : 1;
  while (n < target) n *= 2;
  if (n * 1000 > 130 * rate) return false;          // ols_ceiling_ms
  b = (n - 2 * k) / f_div * f_div;
  delta = (f_div - (3 * k % f_div)) % f_div;
  return b >= 1;
}

void shape_check(int64_t lines, int64_t n, int64_t sl, int64_t sm, int64_t &w) {   // resample_stubs.c:383-389
  if (lines < 0 || n < 2 || (n % 2) != 0 || sl < 1 || sm < 1 || (sl > 1 && sm > 1))
    throw Failure("soundml_resample_shape: invalid geometry");
  w = sl > 1 ? n * sl : (sm > 1 ? n / sm : n);
  if (w < 2 || (sm > 1 && (n % sm) != 0)) throw Failure("soundml_resample_shape: invalid geometry");
}

void shape_dev(const double *d_x, const double *d_h, double *d_y, int64_t lines, int64_t n, int64_t sl, int64_t sm,
               hipStream_t stream) {
  int64_t w = 0;
  shape_check(lines, n, sl, sm, w);
  const int64_t total = lines * (w / 2 + 1);
  if (total == 0) return;
  if (!d_x || !d_h || !d_y) throw Failure("soundml_resample_shape: null device pointer");
  const int64_t blocks = std::min<int64_t>((total + 255) / 256, 4096);
  SMX_LAUNCH(smx::resample_shape_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const smx::cx128 *>(d_x),
             reinterpret_cast<const smx::cx128 *>(d_h), reinterpret_cast<smx::cx128 *>(d_y), lines, n, sl, sm, w);
  SMX_HIP_CHECK(hipGetLastError());
}

// Device scratch that is released on every way out of a function (a throwing launch or check included): stream-ordered
// (the retained pool) when a stream is given, plain hipMalloc / hipFree otherwise.
struct DeviceScratch {
  void *p = nullptr;
  hipStream_t stream = nullptr;
  bool pooled = false;
  DeviceScratch() = default;
  DeviceScratch(const DeviceScratch &) = delete;
  DeviceScratch &operator=(const DeviceScratch &) = delete;
  void alloc_async(size_t bytes, hipStream_t st) {
    stream = st;
    pooled = true;
    SMX_HIP_CHECK(smx::pool_malloc_async(&p, bytes, st));
  }
  void alloc(size_t bytes) { SMX_HIP_CHECK(hipMalloc(&p, bytes)); }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
  ~DeviceScratch() {
    if (!p) return;
    if (pooled) (void)hipFreeAsync(p, stream);
    else (void)hipFree(p);
  }
};

int64_t stage_out_length(const smx_resample_stage &s, int64_t n) { return (n * s.l + s.m - 1) / s.m; }   // ceil(n L / M)

// blocks of the polyphase form: pairs [pair0, pair0 + pairs) of every channel; x[s - x0] = sample s for x_lo <= s < x_hi
void poly_run(const smx_resample_stage &s, const float *d_x, int64_t x0, int64_t x_lo, int64_t x_hi, int64_t x_stride,
              int64_t channels, float *d_y, int64_t y_stride, int64_t o0, int64_t n_out, int64_t pair0, int64_t pairs,
              hipStream_t stream) {
  if (channels <= 0 || pairs <= 0 || n_out <= 0) return;
  const smx_resample_stage::Tables &t = s.tables();
  smx::PolyArgs a{};
  a.x = d_x;
  a.y = d_y;
  a.x0 = x0;
  a.x_lo = x_lo;
  a.x_hi = x_hi;
  a.x_stride = x_stride;
  a.y_stride = y_stride;
  a.o0 = o0;
  a.n_out = n_out;
  a.pair0 = pair0;
  a.pairs = pairs;
  a.l = (int)s.l;
  a.m = (int)s.m;
  a.k = (int)s.k;
  a.taps = (int)s.taps;
  a.valid = (int)s.valid;
  a.h = t.h;
  a.tw = t.tw;
  const int64_t grid = channels * pairs;
  if (grid > 0x7fffffff) throw Failure("resample_stage: too many blocks for one launch");
  const size_t lds = 2 * (size_t)s.nfft * sizeof(float2);
  const bool up = s.l > 1;
  auto launch = [&](auto kernel, int threads) {
    SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SMX_LAUNCH(kernel, dim3((unsigned)grid), dim3(threads), lds, stream, a);
  };
  switch (s.log2n) {
    case 10: up ? launch(smx::resample_poly_kernel<10, true>, 64) : launch(smx::resample_poly_kernel<10, false>, 64); break;
    case 11: up ? launch(smx::resample_poly_kernel<11, true>, 128) : launch(smx::resample_poly_kernel<11, false>, 128); break;
    case 12: up ? launch(smx::resample_poly_kernel<12, true>, 256) : launch(smx::resample_poly_kernel<12, false>, 256); break;
    case 13: up ? launch(smx::resample_poly_kernel<13, true>, 512) : launch(smx::resample_poly_kernel<13, false>, 512); break;
    default: throw Failure("resample_stage: unsupported block length");
  }
  SMX_HIP_CHECK(hipGetLastError());
}
// block pairs whose low-rate outputs cover the first `count` ones
int64_t poly_pairs_for(const smx_resample_stage &s, int64_t count) { return (count + 2 * s.valid - 1) / (2 * s.valid); }

// ---- Resample.Kernel of one stage ------------------------------------------------------------------------------------------
// pairs whose every input lies below `fed`: xL: last input (2 P + 2) V - 1 + K; /M: ((2 P + 2) V - 1) M + K
int64_t kernel_pairs_ready(const smx_resample_stage &s, int64_t fed) {
  if (s.l > 1) return fed >= s.k ? (fed - s.k) / (2 * s.valid) : 0;
  return fed - 1 - s.k >= 0 ? (((fed - 1 - s.k) / s.m) + 1) / (2 * s.valid) : 0;
}
// first input pair P reads (never negative: what lies before the stream is zero)
int64_t kernel_first_input(const smx_resample_stage &s, int64_t pair) {
  const int64_t skip = s.taps - 1;
  const int64_t q = s.l > 1 ? 2 * pair * s.valid + s.k - skip : (2 * pair * s.valid - skip) * s.m + s.k - (s.m - 1);
  return q > 0 ? q : 0;
}
int64_t kernel_low_rate_count(const smx_resample_stage &s, int64_t total_out) { return s.l > 1 ? (total_out + s.l - 1) / s.l : total_out; }

void copy_rows(float *dst, int64_t dst_stride, const float *src, int64_t src_stride, int64_t cols, int64_t rows, hipStream_t stream) {
  if (cols <= 0 || rows <= 0) return;
  SMX_HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)dst_stride * sizeof(float), src, (size_t)src_stride * sizeof(float),
                                 (size_t)cols * sizeof(float), (size_t)rows, hipMemcpyDeviceToDevice, stream));
}

void kernel_check(const smx_resample_kernel *k) {
  if (!k || !k->stage) throw Failure("resample_kernel: null kernel");
  int device = 0;
  SMX_HIP_CHECK(hipGetDevice(&device));
  if (device != k->device) throw Failure("resample_kernel: the kernel was prepared on another device");
}

int64_t kernel_step_dev(smx_resample_kernel &k, const float *d_x, int64_t n, int64_t x_stride, float *d_y, int64_t y_stride,
                        hipStream_t stream) {
  const smx_resample_stage &s = *k.stage;
  if (k.drained)
    throw InvalidArgument("resample_kernel_step: cannot feed a kernel drained by flush (reset it before a new signal)");
  if (n < 0) throw Failure("resample_kernel_step: negative extent");
  if (n > k.max_block)
    throw InvalidArgument(format("resample_kernel_step: cannot feed a chunk of %lld samples to a kernel prepared for at most %lld",
                                 (long long)n, (long long)k.max_block));
  if (n == 0) return 0;
  if (!d_x || x_stride < n) throw Failure("resample_kernel_step: null chunk or stride smaller than the chunk");
  const int64_t pend = k.fed - k.carry_from, alen = pend + n;
  const int64_t ready = kernel_pairs_ready(s, k.fed + n);
  if (ready == k.pairs_done) {   // no block completes: the samples only extend the carry (resample.ml:1468-1484)
    if (alen > k.carry_cap) throw Failure("resample_kernel_step: carry overflow");
    copy_rows(k.carry + pend, k.carry_cap, d_x, x_stride, n, k.channels, stream);
    k.fed += n;
    return 0;
  }
  const int64_t a_stride = (alen + 1) & ~int64_t(1);
  DeviceScratch av;   // [carry ++ chunk], as ols_run assembles it (resample.ml:1487-1507)
  av.alloc_async((size_t)k.channels * (size_t)a_stride * sizeof(float), stream);
  copy_rows(av.as<float>(), a_stride, k.carry, k.carry_cap, pend, k.channels, stream);
  copy_rows(av.as<float>() + pend, a_stride, d_x, x_stride, n, k.channels, stream);
  k.fed += n;
  const int64_t o_end = ready * 2 * s.valid * (s.l > 1 ? s.l : 1), n_out = o_end - k.emitted;
  if (!d_y || y_stride < n_out) throw Failure("resample_kernel_step: null output or stride smaller than the emitted run");
  poly_run(s, av.as<float>(), k.carry_from, k.carry_from, k.fed, a_stride, k.channels, d_y, y_stride, k.emitted, n_out, k.pairs_done,
           ready - k.pairs_done, stream);
  const int64_t keep = kernel_first_input(s, ready), left = k.fed - keep;
  if (left > k.carry_cap) throw Failure("resample_kernel_step: carry overflow");
  copy_rows(k.carry, k.carry_cap, av.as<float>() + (keep - k.carry_from), a_stride, left, k.channels, stream);
  k.carry_from = keep;
  k.pairs_done = ready;
  k.emitted = o_end;
  return n_out;
}

int64_t kernel_flush_pending(const smx_resample_kernel &k) {
  return k.drained ? 0 : stage_out_length(*k.stage, k.fed) - k.emitted;
}

int64_t kernel_flush_dev(smx_resample_kernel &k, float *d_y, int64_t y_stride, hipStream_t stream) {
  const smx_resample_stage &s = *k.stage;
  if (k.drained) return 0;   // draining consumed the tail: a second flush has nothing (resample.mli:313-317)
  const int64_t total = stage_out_length(s, k.fed), n_out = total - k.emitted;
  k.drained = true;
  if (n_out <= 0) return 0;
  if (!d_y || y_stride < n_out) throw Failure("resample_kernel_flush: null output or stride smaller than the tail");
  const int64_t fin = poly_pairs_for(s, kernel_low_rate_count(s, total));   // virtual silence past the stream's end (resample.ml:1745-1755)
  poly_run(s, k.carry, k.carry_from, k.carry_from, k.fed, k.carry_cap, k.channels, d_y, y_stride, k.emitted, n_out, k.pairs_done,
           fin - k.pairs_done, stream);
  k.pairs_done = fin;
  k.emitted = total;
  return n_out;
}

// y[c][i] = sum_t proto[t] xu[c][i M + K L - t], xu = x zero-stuffed by L (resample.ml:1318-1326): the stage's
// definition, run as ONE block convolution at the interpolated rate on the FIR kernel, windowed at the group delay
void stage_apply_dev(const smx_resample_stage &s, const float *d_x, int64_t channels, int64_t n, int64_t x_stride,
                     float *d_y, int64_t y_stride, hipStream_t stream) {
  if (channels < 0 || n < 0) throw Failure("resample_stage: negative extent");
  const int64_t n_out = stage_out_length(s, n);
  if (channels == 0 || n_out == 0) return;
  if (x_stride < n || y_stride < n_out) throw Failure("resample_stage: stride smaller than the signal length");
  if (!d_x || !d_y) throw Failure("resample_stage: null device pointer");
  if (channels > 65535)   // the zero-stuffing and decimation launches put the channel in grid.y
    throw Failure("resample_stage: more than 65535 channels in one call");
  smx::init_device_pool();
  if (s.poly) {   // a pure xL or /M stage: low-rate outputs i < n (xL: y[i L + p]) or i < n_out (/M)
    poly_run(s, d_x, 0, 0, n, x_stride, channels, d_y, y_stride, 0, n_out, 0, poly_pairs_for(s, s.l > 1 ? n : n_out), stream);
    return;
  }
  const float *xin = d_x;
  int64_t n_in = n, in_stride = x_stride;
  DeviceScratch xu, v;   // freed on the stream whichever way this function is left
  if (s.l > 1) {
    n_in = n * s.l;
    in_stride = (n_in + 1) & ~int64_t(1);
    xu.alloc_async((size_t)channels * (size_t)in_stride * sizeof(float), stream);
    const int64_t gx = std::min<int64_t>((n_in + 255) / 256, 2048);
    SMX_LAUNCH(smx::zero_stuff_kernel, dim3((unsigned)gx, (unsigned)channels), dim3(256), 0, stream, d_x, n, x_stride, (int)s.l,
               xu.as<float>(), in_stride);
    SMX_HIP_CHECK(hipGetLastError());
    xin = xu.as<float>();
  }
  const int64_t shift = s.k * s.l;
  if (s.m == 1) {
    smx::fir_apply_window_dev(*s.fir, xin, channels, n_in, in_stride, d_y, y_stride, n_out, shift, stream);
  } else {
    const int64_t nv = (n_out - 1) * s.m + 1, v_stride = (nv + 1) & ~int64_t(1);
    v.alloc_async((size_t)channels * (size_t)v_stride * sizeof(float), stream);
    smx::fir_apply_window_dev(*s.fir, xin, channels, n_in, in_stride, v.as<float>(), v_stride, nv, shift, stream);
    const int64_t gy = std::min<int64_t>((n_out + 255) / 256, 2048);
    SMX_LAUNCH(smx::decimate_kernel, dim3((unsigned)gy, (unsigned)channels), dim3(256), 0, stream, v.as<float>(), v_stride, (int)s.m, d_y,
               n_out, y_stride);
  }
  SMX_HIP_CHECK(hipGetLastError());
}

void require_hip_device() {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count < 1)
    throw Failure("soundml_amd: no HIP device is available (this library has no CPU fallback)");
}
}  // namespace

extern "C" {

int smx_resample_ols_geom(int64_t rate, int64_t l, int64_t m, int64_t k, int64_t *n, int64_t *b, int64_t *delta,
                          int *eligible) {
  return guarded_fir([&] {
    if (rate < 1 || l < 1 || m < 1 || k < 0) throw Failure("ols_geom: invalid stage");
    int64_t nn = 0, bb = 0, dd = 0;
    const bool ok = ols_geom_host(rate, l, m, k, nn, bb, dd);
    if (eligible) *eligible = ok ? 1 : 0;
    if (n) *n = ok ? nn : 0;
    if (b) *b = ok ? bb : 0;
    if (delta) *delta = ok ? dd : 0;
  });
}

int smx_resample_prototype(int64_t l, int64_t k, double fc, double beta, double *h) {
  return guarded_fir([&] {   // resample.ml:145-163 `design_prototype`: right half evaluated, left half mirrored, sum = L
    if (l < 1 || k < 0) throw Failure("design_prototype: invalid stage");
    if (!h) throw Failure("design_prototype: null output");
    const int64_t mid = k * l, n = 2 * mid + 1;
    const double i0_beta = bessel_i0(beta);
    for (int64_t i = mid; i < n; ++i) {
      const double z = (double)(i - mid);
      const double sv = i == mid ? fc : std::sin(M_PI * fc * z) / (M_PI * z);
      const double r = mid > 0 ? z / (double)mid : 0.0;
      const double v = sv * (bessel_i0(beta * std::sqrt(1.0 - r * r)) / i0_beta);
      h[i] = v;
      h[n - 1 - i] = v;
    }
    double sum = 0.0;
    for (int64_t i = 0; i < n; ++i) sum += h[i];
    const double gain = (double)l / sum;
    for (int64_t i = 0; i < n; ++i) h[i] = h[i] * gain;
  });
}

int smx_resample_shape_c128_dev(const double *d_x, const double *d_h, double *d_y, int64_t lines, int64_t n, int64_t sl,
                                int64_t sm, void *stream) {
  return guarded_fir([&] { shape_dev(d_x, d_h, d_y, lines, n, sl, sm, (hipStream_t)stream); });
}

int smx_resample_shape_c128(const double *x, const double *h, double *y, int64_t lines, int64_t n, int64_t sl, int64_t sm) {
  return guarded_fir([&] {
    int64_t w = 0;
    shape_check(lines, n, sl, sm, w);
    if (lines == 0) return;
    if (!x || !h || !y) throw Failure("soundml_resample_shape: null pointer");
    require_hip_device();
    const size_t xb = (size_t)lines * (size_t)(n / 2 + 1) * 16, hb = (size_t)(sl > 1 ? w / 2 + 1 : n / 2 + 1) * 16,
                 yb = (size_t)lines * (size_t)(w / 2 + 1) * 16;
    DeviceScratch dx, dh, dy;   // a failed second or third allocation releases the earlier ones
    dx.alloc(xb);
    dh.alloc(hb);
    dy.alloc(yb);
    SMX_HIP_CHECK(hipMemcpy(dx.p, x, xb, hipMemcpyHostToDevice));
    SMX_HIP_CHECK(hipMemcpy(dh.p, h, hb, hipMemcpyHostToDevice));
    shape_dev(dx.as<double>(), dh.as<double>(), dy.as<double>(), lines, n, sl, sm, nullptr);
    SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
    SMX_HIP_CHECK(hipMemcpy(y, dy.p, yb, hipMemcpyDeviceToHost));
  });
}

int smx_resample_stage_create(const double *proto, int64_t l, int64_t m, int64_t k, smx_resample_stage **out) {
  return guarded_fir([&] {
    if (!out) throw Failure("resample_stage_create: null output handle");
    if (l < 1 || m < 1 || k < 0)
      throw InvalidArgument(format("resample_stage_create: cannot build a x%lld / %lld stage of group delay %lld "
                                   "(factors must be at least 1, the delay non-negative)", (long long)l, (long long)m, (long long)k));
    const int64_t taps = 2 * k * l + 1;
    if (taps > 16384)
      throw InvalidArgument(format("resample_stage_create: cannot run a %lld-tap prototype (this device path holds at "
                                   "most 16384 taps per 32768-sample block)", (long long)taps));
    if (!proto) throw Failure("resample_stage_create: null prototype");
    const bool pure = (l > 1) != (m > 1);
    const int64_t f = l > 1 ? l : m, taps_p = l > 1 ? 2 * k + 1 : (2 * k) / f + 1;
    if (pure && taps_p <= 4096 && diag_flag("SMX_RESAMPLE_POLY") != 0) {   // polyphase blocks: at least 4 x taps per block (75 % of a block kept), 1024 .. 8192 points
      auto st = std::make_unique<smx_resample_stage>();
      st->l = l; st->m = m; st->k = k;
      st->poly = true;
      st->phases = (int)f;
      st->taps = taps_p;
      int64_t n = 1024;
      while (n < 4 * taps_p && n < 8192) n *= 2;
      while (n < 2 * taps_p) n *= 2;
      st->nfft = n;
      st->valid = n - taps_p + 1;
      while ((int64_t(1) << st->log2n) < n) ++st->log2n;
      st->proto.assign(proto, proto + taps);
      *out = st.release();
      return;
    }
    smx_fir_plan *fir = nullptr;
    if (smx_fir_plan_create(proto, taps, &fir) != SMX_OK) throw Failure(smx_last_error());
    auto *s = new smx_resample_stage();
    s->l = l; s->m = m; s->k = k; s->fir = fir;
    *out = s;
  });
}
void smx_resample_stage_destroy(smx_resample_stage *s) { delete s; }
int64_t smx_resample_stage_out_length(const smx_resample_stage *s, int64_t n) { return s && n >= 0 ? stage_out_length(*s, n) : -1; }

int smx_resample_stage_apply_f32_dev(const smx_resample_stage *s, const float *d_x, int64_t channels, int64_t n,
                                     int64_t x_stride, float *d_y, int64_t y_stride, void *stream) {
  return guarded_fir([&] {
    if (!s) throw Failure("resample_stage: null stage");
    stage_apply_dev(*s, d_x, channels, n, x_stride, d_y, y_stride, (hipStream_t)stream);
  });
}

int smx_resample_stage_apply_f32(const smx_resample_stage *s, const float *x, int64_t channels, int64_t n, float *y) {
  return guarded_fir([&] {
    if (!s) throw Failure("resample_stage: null stage");
    if (channels < 0 || n < 0) throw Failure("resample_stage: negative extent");
    const int64_t n_out = stage_out_length(*s, n);
    if (channels == 0 || n_out == 0) return;
    if (!x || !y) throw Failure("resample_stage: null pointer");
    require_hip_device();
    float *dx = nullptr, *dy = nullptr;
    SMX_HIP_CHECK(hipMalloc((void **)&dx, (size_t)channels * (size_t)n * sizeof(float)));
    if (hipMalloc((void **)&dy, (size_t)channels * (size_t)n_out * sizeof(float)) != hipSuccess) {
      (void)hipFree(dx);
      throw Failure("resample_stage: device allocation failed");
    }
    try {
      SMX_HIP_CHECK(hipMemcpy(dx, x, (size_t)channels * (size_t)n * sizeof(float), hipMemcpyHostToDevice));
      stage_apply_dev(*s, dx, channels, n, n, dy, n_out, nullptr);
      SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
      SMX_HIP_CHECK(hipMemcpy(y, dy, (size_t)channels * (size_t)n_out * sizeof(float), hipMemcpyDeviceToHost));
    } catch (...) {
      (void)hipFree(dx); (void)hipFree(dy);
      throw;
    }
    (void)hipFree(dx); (void)hipFree(dy);
  });
}

/* ---- Resample.Kernel of one overlap-save stage (resample.mli:270-319) ---- */
int smx_resample_kernel_prepare(const smx_resample_stage *s, int64_t channels, int64_t max_block, smx_resample_kernel **out) {
  return guarded_fir([&] {
    if (!out) throw Failure("resample_kernel_prepare: null output handle");
    if (!s) throw Failure("resample_kernel_prepare: null stage");
    if (channels < 1 || max_block < 1)
      throw InvalidArgument(format("resample_kernel_prepare: cannot prepare a kernel for %lld channels and chunks of %lld samples "
                                   "(both must be at least 1)", (long long)channels, (long long)max_block));
    if (!s->poly)
      throw InvalidArgument(format("resample_kernel_prepare: cannot stream a x%lld / %lld stage block by block (only a pure xL or /M "
                                   "stage is overlap-save eligible)", (long long)s->l, (long long)s->m));
    require_hip_device();
    auto k = std::make_unique<smx_resample_kernel>();
    k->stage = s;
    k->channels = channels;
    k->max_block = max_block;
    SMX_HIP_CHECK(hipGetDevice(&k->device));
    // what a step may have to hold: the inputs of one unfinished pair (two windows at the input rate) plus a chunk
    const int64_t f = s->m > 1 ? s->m : 1;
    k->carry_cap = ((2 * s->nfft + 2) * f + s->k + max_block + 17) & ~int64_t(1);
    SMX_HIP_CHECK(hipMalloc((void **)&k->carry, (size_t)channels * (size_t)k->carry_cap * sizeof(float)));
    *out = k.release();
  });
}
void smx_resample_kernel_destroy(smx_resample_kernel *k) { delete k; }
int smx_resample_kernel_reset(smx_resample_kernel *k) {
  return guarded_fir([&] {
    if (!k) throw Failure("resample_kernel: null kernel");
    k->fed = k->emitted = k->pairs_done = k->carry_from = 0;
    k->drained = false;
  });
}
/* most samples per channel one step of n input samples can emit (burst emission: whole block pairs) */
int64_t smx_resample_kernel_out_bound(const smx_resample_kernel *k, int64_t n) {
  if (!k || !k->stage || n < 0) return -1;
  const smx_resample_stage &s = *k->stage;
  return s.l > 1 ? (n + 2 * s.valid) * s.l : n / s.m + 2 * s.valid + 1;
}
/* samples per channel the next flush emits */
int64_t smx_resample_kernel_pending(const smx_resample_kernel *k) { return k && k->stage ? kernel_flush_pending(*k) : -1; }

int smx_resample_kernel_step_f32_dev(smx_resample_kernel *k, const float *d_x, int64_t n, int64_t x_stride, float *d_y,
                                     int64_t y_stride, int64_t *n_out, void *stream) {
  return guarded_fir([&] {
    kernel_check(k);
    const int64_t got = kernel_step_dev(*k, d_x, n, x_stride, d_y, y_stride, (hipStream_t)stream);
    if (n_out) *n_out = got;
  });
}
int smx_resample_kernel_flush_f32_dev(smx_resample_kernel *k, float *d_y, int64_t y_stride, int64_t *n_out, void *stream) {
  return guarded_fir([&] {
    kernel_check(k);
    const int64_t got = kernel_flush_dev(*k, d_y, y_stride, (hipStream_t)stream);
    if (n_out) *n_out = got;
  });
}
/* host chunks [channels; n] (row stride x_stride) -> y [channels; *n_out] (row stride y_stride >= out_bound(n)) */
int smx_resample_kernel_step_f32(smx_resample_kernel *k, const float *x, int64_t n, int64_t x_stride, float *y, int64_t y_stride,
                                 int64_t *n_out) {
  return guarded_fir([&] {
    kernel_check(k);
    if (n_out) *n_out = 0;
    if (n > 0 && (!x || x_stride < n)) throw Failure("resample_kernel_step: null chunk or stride smaller than the chunk");
    if (n <= 0 || n > k->max_block || k->drained) {   // the checks and their messages live in one place
      (void)kernel_step_dev(*k, nullptr, n, x_stride, nullptr, 0, nullptr);
      return;
    }
    const int64_t bound = smx_resample_kernel_out_bound(k, n);
    DeviceScratch dx, dy;
    dx.alloc((size_t)k->channels * (size_t)n * sizeof(float));
    dy.alloc((size_t)k->channels * (size_t)bound * sizeof(float));
    SMX_HIP_CHECK(hipMemcpy2D(dx.p, (size_t)n * sizeof(float), x, (size_t)x_stride * sizeof(float), (size_t)n * sizeof(float),
                              (size_t)k->channels, hipMemcpyHostToDevice));
    const int64_t got = kernel_step_dev(*k, dx.as<float>(), n, n, dy.as<float>(), bound, nullptr);
    SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
    if (got > 0) {
      if (!y || y_stride < got) throw Failure("resample_kernel_step: null output or stride smaller than the emitted run");
      SMX_HIP_CHECK(hipMemcpy2D(y, (size_t)y_stride * sizeof(float), dy.p, (size_t)bound * sizeof(float), (size_t)got * sizeof(float),
                                (size_t)k->channels, hipMemcpyDeviceToHost));
    }
    if (n_out) *n_out = got;
  });
}
int smx_resample_kernel_flush_f32(smx_resample_kernel *k, float *y, int64_t y_stride, int64_t *n_out) {
  return guarded_fir([&] {
    kernel_check(k);
    if (n_out) *n_out = 0;
    const int64_t pending = kernel_flush_pending(*k);
    if (pending <= 0) {
      (void)kernel_flush_dev(*k, nullptr, 0, nullptr);
      return;
    }
    if (!y || y_stride < pending) throw Failure("resample_kernel_flush: null output or stride smaller than the tail");
    DeviceScratch dy;
    dy.alloc((size_t)k->channels * (size_t)pending * sizeof(float));
    const int64_t got = kernel_flush_dev(*k, dy.as<float>(), pending, nullptr);
    SMX_HIP_CHECK(hipStreamSynchronize(nullptr));
    SMX_HIP_CHECK(hipMemcpy2D(y, (size_t)y_stride * sizeof(float), dy.p, (size_t)pending * sizeof(float), (size_t)got * sizeof(float),
                              (size_t)k->channels, hipMemcpyDeviceToHost));
    if (n_out) *n_out = got;
  });
}


}  // extern "C"
