// istft2048_pipe_kernel -- Stft.invert at fft 2048 / hop 512 on the forward kernel's 32-lane frame pipeline run backwards
// (included by istft.hip inside namespace smx::<anon>; round 5).  Replaces, for complex64 spectra and the float32 interior,
// the reference's overlap-add synthesis: stft.ml:900-939 (invert), 806-831 (overlap_add's summation order), 836-889 (envelope).
//
// istft2048_kernel (rounds 1-4, below in istft.hip) is one tile per workgroup with three workgroup barriers and nothing in
// flight while it stages its 131 KB of spectra: 16.2 us per tile of 13 hops, 4.8 of them the staging's memory round trip, 3.7
// the barriers (profiles/r06/NOTES.md).  Here:
//   * PERSISTENT workgroups of 8 waves walk consecutive 16-frame tiles of a clip; every frame is inverted once and the three
//     frames that reach into the next tile are carried (no frame inverted twice: 16 hops per tile instead of 13);
//   * the NEXT tile's spectra are requested a whole tile ahead into 66 registers per thread (coalesced: 16 lanes a 128-byte row
//     piece), so the memory round trip runs under the current tile;
//   * a frame lives in 32 lanes with 32 points per lane (two frames per wave, one instruction stream), exactly as in
//     stft2048_power32_kernel: the generated packed-float32 radix-32 stages (stft_pk_fft.inc), one transposition through LDS,
//     no cross-lane arithmetic.  The inverse is conj(FFT_M(conj Z')): the SAME forward stages between a pre-pass
//     Z'[k] = E + i conj(w_k) D  (E = Z[k] + conj Z[M-k], D = Z[k] - conj Z[M-k]; 1/2 and 1/M live in the synthesis window)
//     and the window product; the layout "lane a, register b <-> index a + 32 b" is the same on both sides of the transform;
//   * the phases of a tile meet at three monotonic LDS counters instead of workgroup barriers, and the envelope's periodic part
//     is two float64 reciprocals per thread, in registers.
// LDS: 16 frame slots of 1025 complex cells (8200 B: the staged bins 0..1024 of a frame, then its transposition scratch, then
// its 2048 windowed samples) = 131,200 B + synthesis window 8,192 + W_M^(l k1) 7,936 + exp(-2 pi i k / N), k < 1024, 8,192
// + counters = 155,584 B.
//
// A tile (it = 0, 1, ... of this workgroup):
//   S1  wait drained(it - 1)           every wave has gathered the previous tile: the slots are free
//   S2  the staged registers -> slots (element (row, frame) to cell `row` of slot `frame`: 8-byte stores, conflict free
//       because a slot is 2050 floats: neighbouring frames sit two banks apart); signal staged(it)
//   S3  request the next tile's spectra (33 coalesced 8-byte loads per thread)
//   S4  wait staged(it)
//   S5  per frame (lane k1, register q <-> k = k1 + 32 q): Z[k] and Z[M - k] from the frame's slot, pre-pass, radix-32,
//       twiddle, transposition (cells 33 l + j of the frame's own slot, one plane at a time), radix-32, window product,
//       samples 2 n, 2 n + 1 (n = n1 + 32 q) to cell n of the slot; signal filled(it)
//   S6  wait filled(it)
//   S7  overlap-add: thread t owns sample pair u = t & 255 of the hops of parity t >> 8; a hop sums its <= 4 frames in the
//       reference's order (frame index descending), the previous tile's last three frames from 4 carried register pairs;
//       envelope (float64 product, a division on the clip's first and last hops); 8-byte stores, 512 contiguous bytes per
//       wave and hop; signal drained(it).
// A frame gets the same bits wherever it sits and a position sums the same values in the same order whatever the tiling,
// so the streaming synthesis (capi.cpp, which runs this kernel over [history ++ chunk]) totals Stft.invert bit for bit.
using f2 = float __attribute__((ext_vector_type(2)));
#include "stft_pk_fft.inc"

constexpr int kIpM = 1024, kIpFT = 16;
constexpr int kIpSlot = 2050;                                              // floats per frame slot (1025 complex cells)
constexpr size_t kIpSlotsBytes = (size_t)kIpFT * kIpSlot * sizeof(float);  // 131,200
constexpr size_t kIpSwinBytes = 1024 * sizeof(float2);                     // synthesis window pairs
constexpr size_t kIpTwABytes = 31 * 32 * sizeof(float2);                   // W_M^(l k1), k1 = 1..31
constexpr size_t kIpTwNBytes = 1024 * sizeof(float2);                      // exp(-2 pi i k / N), k < 1024
constexpr size_t kIpLds = kIpSlotsBytes + kIpSwinBytes + kIpTwABytes + kIpTwNBytes + 64;
static_assert(kIpLds <= 160 * 1024, "LDS budget");

struct PipeArgs {
  SynArgs s;
  int64_t total_tiles;             // lead * tiles_per_clip (tiles of 16 hops)
  int64_t range_base, range_extra; // total_tiles / blocks and the remainder
  int blocks;
  int aligned_out;                 // every clip's output pairs are 8-byte aligned (out, out_len and left even)
  // FM (Griffin-Lim's own spectra, capi.cpp): z, prev and mag FRAME-MAJOR -- [clip][frame][bin] in rows of fm_pitch elements,
  // fm_clip elements per clip (stft2048_complex_fm_kernel writes z / prev that way)
  int fm_pitch;
  int64_t fm_clip;
};

struct IpLds {
  float *slots;
  float4 *swin4;    // [m][l] = synthesis-window pairs of points n = l + 32 (2 m), l + 32 (2 m + 1)         m < 16
  float4 *twA4;     // [m][l] = W_M^(l k1) for k1 = 2 m + 1, 2 m + 2 (m < 15), then one float2 row of k1 = 31
  float2 *twA31;
  float4 *twN4;     // [m][l] = exp(-2 pi i k / N) for k = l + 32 (2 m), l + 32 (2 m + 1)                    m < 16
  unsigned *staged, *filled, *drained;
};
__device__ __forceinline__ IpLds ip_carve(unsigned char *smem) {
  IpLds l;
  l.slots = reinterpret_cast<float *>(smem);
  l.swin4 = reinterpret_cast<float4 *>(smem + kIpSlotsBytes);
  l.twA4 = reinterpret_cast<float4 *>(smem + kIpSlotsBytes + kIpSwinBytes);
  l.twA31 = reinterpret_cast<float2 *>(smem + kIpSlotsBytes + kIpSwinBytes + 15 * 32 * sizeof(float4));
  l.twN4 = reinterpret_cast<float4 *>(smem + kIpSlotsBytes + kIpSwinBytes + kIpTwABytes);
  unsigned *c = reinterpret_cast<unsigned *>(smem + kIpSlotsBytes + kIpSwinBytes + kIpTwABytes + kIpTwNBytes);
  l.staged = c;
  l.filled = c + 4;
  l.drained = c + 8;
  return l;
}

// counters: relaxed signals (one wave's LDS operations execute in order, so a signal needs no wait for the wave's earlier LDS
// accesses), acquire polls; compiler fences on both sides (a relaxed atomic alone orders nothing for the compiler)
__device__ __forceinline__ void ip_signal(unsigned *c, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void ip_wait(unsigned *c, unsigned target) {
  asm volatile("" ::: "memory");
  while ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
    __builtin_amdgcn_s_sleep(2);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ int ip_opaque(int v) {   // (see opaque32 in stft_fast_p32.hpp: invariant addresses are re-derived, not hoisted and spilled)
  asm volatile("" : "+v"(v));
  return v;
}
#define IP_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifdef SMX_STAMPS   // diagnostic builds (make STAMPS=1, tools/stamps_istft.py): shader-clock sums per phase and wave
#define IP_STAMP(i) do { unsigned long long t__; IP_FENCE(); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory"); IP_FENCE(); \
    ip_sum[i] += t__ - ip_prev; ip_prev = t__; } while (0)
#else
#define IP_STAMP(i) do { } while (0)
#endif
#ifndef IP_GL_BULK
#define IP_GL_BULK 1   // 0: the factors applied element by element behind the wait for the slots (A/B builds: 139 against 75 ms for 32 Griffin-Lim iterations at C2 -- 33 dependent float64 chains one after the other instead of interleaved; same bits)
#endif

// Two frames (one per lane-half): the frame's staged spectrum in `slot` -> its 2048 windowed samples in `slot`.
__device__ __forceinline__ void ip_frame(const IpLds &lds, float *slot, int lane) {
#pragma clang fp contract(off)
  const int l = ip_opaque(lane & 31);
  f2 v[32];
  {
    // pre-pass: register q <-> k = l + 32 q; Z[k] from cell k, Z[M - k] from cell M - k = (32 - l) + 32 (31 - q) (lane 0, q = 0:
    // cell 1024, the Nyquist bin; bin 512 pairs with itself).  conj(Z') = conj(E + i conj(w) D)
    const float2 *own = reinterpret_cast<const float2 *>(slot) + l;
    const float2 *par = reinterpret_cast<const float2 *>(slot) + (32 - l);
    const float keep = l == 0 ? 0.0f : 1.0f;   // the imaginary parts of the DC and Nyquist bins do not take part
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float2 z[8], p[8];
      float4 w[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = 8 * c + i;
        z[i] = own[32 * q];
        p[i] = par[32 * (31 - q)];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) w[i] = lds.twN4[32 * (4 * c + i) + l];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = 8 * c + i;
        float zi = z[i].y, pi = p[i].y;
        if (q == 0) { zi *= keep; pi *= keep; }
        const float wx = (i & 1) ? w[i >> 1].z : w[i >> 1].x, wy = (i & 1) ? w[i >> 1].w : w[i >> 1].y;
        const float er = z[i].x + p[i].x, ei = zi - pi;           // E = Z[k] + conj Z[M-k]
        const float dr = z[i].x - p[i].x, di = zi + pi;           // D = Z[k] - conj Z[M-k]
        // i conj(w) D = -(w.x di - w.y dr) + i (w.x dr + w.y di)
        const float tr = er - __builtin_fmaf(wx, di, -(wy * dr));
        const float ti = ei + __builtin_fmaf(wx, dr, wy * di);
        v[q] = f2{tr, -ti};
      }
    }
  }
  IP_FENCE();
  // A: radix 32 over q, twiddle W_M^(l k1)
  f2 t[32];
  {
    f2 e[16], o[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) { e[m] = v[2 * m]; o[m] = v[2 * m + 1]; }
    float4 tw[15];
#pragma unroll
    for (int m = 0; m < 15; ++m) tw[m] = lds.twA4[32 * m + l];
    const float2 tw31 = lds.twA31[l];
    IP_FENCE();
    pk_fft16(e);
    pk_fft16(o);
    pk_fft32_combine0(v, e, o);
    pk_fft32_combine1(v, e, o);
    IP_FENCE();
    // X: lane l register k1 -> lane k1 register l through the frame's slot, one plane at a time: lane l writes register j to
    // cell 33 l + j (floats), lane k1 reads register l' from cell 33 l' + k1 -- conflict free on both sides
    float *const wc = slot + 33 * l;
    const float *const rc = slot + l;
    auto put2 = [&](int j) { wc[j] = v[j].x; wc[j + 1] = v[j + 1].x; };
#define IP_TWV(m) f2{tw[m].x, tw[m].y}, f2{tw[m].z, tw[m].w}
    pk_twiddle8(v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], IP_TWV(0), IP_TWV(1), IP_TWV(2), IP_TWV(3));
#pragma unroll
    for (int j = 0; j < 8; j += 2) put2(j);
    IP_FENCE();
    pk_twiddle8(v[9], v[10], v[11], v[12], v[13], v[14], v[15], v[16], IP_TWV(4), IP_TWV(5), IP_TWV(6), IP_TWV(7));
#pragma unroll
    for (int j = 8; j < 16; j += 2) put2(j);
    IP_FENCE();
    pk_twiddle8(v[17], v[18], v[19], v[20], v[21], v[22], v[23], v[24], IP_TWV(8), IP_TWV(9), IP_TWV(10), IP_TWV(11));
#pragma unroll
    for (int j = 16; j < 24; j += 2) put2(j);
    IP_FENCE();
    pk_twiddle7(v[25], v[26], v[27], v[28], v[29], v[30], v[31], IP_TWV(12), IP_TWV(13), IP_TWV(14), f2{tw31.x, tw31.y});
#pragma unroll
    for (int j = 24; j < 32; j += 2) put2(j);
#undef IP_TWV
    IP_FENCE();
#pragma unroll
    for (int i = 0; i < 32; ++i) t[i].x = rc[33 * i];
#pragma unroll
    for (int j = 0; j < 32; ++j) wc[j] = v[j].y;
#pragma unroll
    for (int i = 0; i < 32; ++i) t[i].y = rc[33 * i];
  }
  IP_FENCE();
  // B: radix 32 over l: lane n1, register q holds FFT(conj Z')[n1 + 32 q]
  {
    f2 e[16], o[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) { e[m] = t[2 * m]; o[m] = t[2 * m + 1]; }
    float4 sw[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) sw[m] = lds.swin4[32 * m + l];
    IP_FENCE();
    pk_fft16(e);
    pk_fft16(o);
    pk_fft32_combine0(t, e, o);
    pk_fft32_combine1(t, e, o);
    IP_FENCE();
    // x[2 n] + i x[2 n + 1] = conj(.) / M, windowed: the signs and 1 / (2 M) are in the table.  Cell n of the slot.
    float2 *const out = reinterpret_cast<float2 *>(slot) + l;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const float wx = (q & 1) ? sw[q >> 1].z : sw[q >> 1].x, wy = (q & 1) ? sw[q >> 1].w : sw[q >> 1].y;
      out[32 * q] = make_float2(t[q].x * wx, t[q].y * wy);
    }
  }
}

// GL: Griffin-Lim's factors are taken (SynArgs::mag / unit / prev); the plain kernel holds no registers for them
// FM: the spectra (and factors) are frame-major: a thread stages 33 bins of ONE frame (32 lanes a contiguous 256-byte piece of the
// frame's row) -- and a wave stages exactly the two frames it inverts, so it does not wait for the other waves' staging
template <bool GL, bool FM = false>
__global__ void __launch_bounds__(512) istft2048_pipe_kernel(PipeArgs pa) {
#pragma clang fp contract(off)
  const SynArgs &a = pa.s;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IpLds lds = ip_carve(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- tables (one round trip) and this workgroup's range of the flat (clip, tile) sequence ----
  {
    const int row = tid >> 5, l = tid & 31;
    const float2 s0 = a.synth_window[l + 32 * (2 * row)], s1 = a.synth_window[l + 32 * (2 * row + 1)];
    const float2 n0 = a.w_n[l + 32 * (2 * row)], n1 = a.w_n[l + 32 * (2 * row + 1)];
    const int m = tid < 15 * 32 ? row : 0;
    const float2 a0 = a.w_m[l * (2 * m + 1)], a1 = a.w_m[l * (2 * m + 2)], a31 = a.w_m[l * 31];
    lds.swin4[tid] = make_float4(s0.x, s0.y, s1.x, s1.y);
    lds.twN4[tid] = make_float4(n0.x, n0.y, n1.x, n1.y);
    if (tid < 15 * 32) lds.twA4[tid] = make_float4(a0.x, a0.y, a1.x, a1.y);
    if (tid < 32) lds.twA31[tid] = a31;
    if (tid == 0) { *lds.staged = 0u; *lds.filled = 0u; *lds.drained = 0u; }
  }
  // the envelope's periodic part at this thread's two positions (Q mod 512 = 2 u, 2 u + 1): reciprocals, so that the overlap-add
  // multiplies (one float64 product per sample where a division costs ~35 float64 operations: as istft2048_kernel)
  const int u = tid & 255, par = tid >> 8;
  const double renv0 = 1.0 / a.env_period[(int)((2 * u + a.env_q0) & 511)];
  const double renv1 = 1.0 / a.env_period[(int)((2 * u + 1 + a.env_q0) & 511)];
  int64_t tau0, tau1;
  {
    const unsigned nb = (unsigned)pa.blocks, q = nb / 8u, r = nb % 8u, xcd = blockIdx.x % 8u, idx = blockIdx.x / 8u;
    const unsigned vb = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;   // blocks of an XCD take neighbouring ranges
    tau0 = (int64_t)vb * pa.range_base + ((int64_t)vb < pa.range_extra ? (int64_t)vb : pa.range_extra);
    tau1 = tau0 + pa.range_base + ((int64_t)vb < pa.range_extra ? 1 : 0);
  }
  const int tpc = a.tiles_per_clip;
  int64_t clip;
  int ft;
  if (pa.total_tiles < (int64_t(1) << 31)) {
    const unsigned c0 = (unsigned)tau0 / (unsigned)tpc;
    clip = c0;
    ft = (int)((unsigned)tau0 - c0 * (unsigned)tpc);
  } else {
    clip = tau0 / tpc;
    ft = (int)(tau0 - clip * tpc);
  }
  int todo = (int)(tau1 - tau0);             // tiles whose output this workgroup stores
  bool store = true;
  if (todo > 0 && ft > 0) {                  // the range begins inside a clip: the tile before it is run for its carry only
    --ft;
    ++todo;
    store = false;
  }
  const size_t zclip = (size_t)(kIpM + 1) * (size_t)a.frames;   // complex values per clip
  // this thread's elements of a tile: (row, frame) = ((tid >> 4) + 32 i, tid & 15), i < 33 (i = 32: row 1024, threads 0..15)
  // (FM: (row, frame) = ((tid & 31) + 32 i, tid >> 5))
  const int sf = FM ? tid >> 5 : tid & 15, srow = FM ? tid & 31 : tid >> 4;
  float2 raw[33];
  auto request = [&](int64_t cl, int t) {
    const int64_t p = (int64_t)kIpFT * t + sf;
    const bool ok = p < a.count;
    if constexpr (FM) {
      const char *src = reinterpret_cast<const char *>(a.z + (size_t)cl * (size_t)pa.fm_clip);   // (wave-uniform)
      const unsigned off = (unsigned)(((size_t)(ok ? p : 0) * (size_t)pa.fm_pitch + (size_t)srow) * 8u);
#pragma unroll
      for (int i = 0; i < 33; ++i) {
        float2 vv = make_float2(0.f, 0.f);
        if (ok && (i < 32 || srow == 0)) vv = *reinterpret_cast<const float2 *>(src + off + 256u * (unsigned)i);
        raw[i] = vv;
      }
      return;
    }
    const float2 *src = a.z + (size_t)cl * zclip;   // (wave-uniform)
    const unsigned off = (unsigned)(((size_t)srow * (size_t)a.frames + (size_t)(ok ? p : 0)) * 8u);   // < 2^32: the launcher checks
    const size_t pitch = (size_t)a.frames * 32u * 8u;
#pragma unroll
    for (int i = 0; i < 33; ++i) {
      float2 vv = make_float2(0.f, 0.f);
      if (ok && (i < 32 || srow == 0)) vv = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(src) + (size_t)i * pitch + off);
      raw[i] = vv;
    }
  };
#pragma unroll
  for (int i = 0; i < 33; ++i) raw[i] = make_float2(0.f, 0.f);
  if (todo > 0) request(clip, ft);
  // Griffin-Lim's factors of a tile (see the loop's top): c_(k-1) and the magnitudes at this thread's elements
  float2 pv[33];
  float mg[33];
  auto request_factors = [&](int64_t cl, int t) {   // (a wave-uniform base and a 32-bit offset per thread, as request())
    const int64_t p = (int64_t)kIpFT * t + sf;
    const bool ok = p < a.count;
    if constexpr (FM) {
      const unsigned eoff = (unsigned)((size_t)(ok ? p : 0) * (size_t)pa.fm_pitch + (size_t)srow);
      const bool has_prev = a.unit && a.prev;
      const char *pb = reinterpret_cast<const char *>(a.prev + (size_t)cl * (size_t)pa.fm_clip);
      const char *mb = reinterpret_cast<const char *>(a.mag + (size_t)cl * (size_t)pa.fm_clip);
#pragma unroll
      for (int i = 0; i < 33; ++i) {
        pv[i] = make_float2(0.f, 0.f);
        if (has_prev && ok && (i < 32 || srow == 0)) pv[i] = *reinterpret_cast<const float2 *>(pb + 8u * eoff + 256u * (unsigned)i);
      }
#pragma unroll
      for (int i = 0; i < 33; ++i) {
        mg[i] = 0.f;
        if (a.mag && ok && (i < 32 || srow == 0)) mg[i] = *reinterpret_cast<const float *>(mb + 4u * eoff + 128u * (unsigned)i);
      }
      return;
    }
    const unsigned eoff = (unsigned)((size_t)srow * (size_t)a.frames + (size_t)(ok ? p : 0));   // elements
    const size_t pitch = (size_t)a.frames * 32u;
    const bool has_prev = a.unit && a.prev;
    const char *pb = reinterpret_cast<const char *>(a.prev + (size_t)cl * zclip);
    const char *mb = reinterpret_cast<const char *>(a.mag + (size_t)cl * zclip);
#pragma unroll
    for (int i = 0; i < 33; ++i) {
      pv[i] = make_float2(0.f, 0.f);
      if (has_prev && ok && (i < 32 || srow == 0)) pv[i] = *reinterpret_cast<const float2 *>(pb + (size_t)i * pitch * 8u + 8u * eoff);
    }
#pragma unroll
    for (int i = 0; i < 33; ++i) {
      mg[i] = 0.f;
      if (a.mag && ok && (i < 32 || srow == 0)) mg[i] = *reinterpret_cast<const float *>(mb + (size_t)i * pitch * 4u + 4u * eoff);
    }
  };
#pragma unroll
  for (int i = 0; i < 33; ++i) { pv[i] = make_float2(0.f, 0.f); mg[i] = 0.f; }
  f2 carry[4];   // parity 0: frames 15 / 14 / 13 of the previous tile at segments 1 / 2 / 3, then frame 15 at segment 3; parity 1: frame 15 at segment 2, frame 14 at segment 3
#pragma unroll
  for (int i = 0; i < 4; ++i) carry[i] = f2{0.f, 0.f};
  __syncthreads();   // tables and counters
#ifdef SMX_STAMPS
  unsigned long long ip_sum[fftdev::kStampSlots] = {0}, ip_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ip_prev)::"memory");
  const unsigned long long ip_t0 = ip_prev, ip_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  // S6 / S7 of a tile (clip g_clip, tile g_ft of it; g_store: its output is this workgroup's): wait for every frame's samples, the
  // overlap-add, "drained", envelope and stores.  Called at the TOP of the next tile's iteration (and once behind the loop), so that
  // Griffin-Lim's factors of that next tile, requested just before, are in flight under it -- requested anywhere earlier and carried
  // over the loop edge hipcc parks them in scratch load by load.
  auto overlap_add = [&](int64_t g_clip, int g_ft, bool g_store, unsigned filled_target) {
    ip_wait(lds.filled, filled_target);
    IP_STAMP(7);   // wait: every frame's samples in
    {
      constexpr int SP = kIpSlot / 2;   // cells per slot
      const float2 *s2 = reinterpret_cast<const float2 *>(lds.slots) + ip_opaque(u) + ip_opaque(par) * SP;   // hop h = 2 i + par: frame h's cell u
      // hop h of parity par: frames h, h - 1, h - 2, h - 3 at segments 0, 1, 2, 3, summed in that order (frame index descending,
      // stft.ml:806-831); the frames before this tile from the carried pairs
      f2 acc[8];
      auto cell = [&](int rel, int d) { const float2 x = s2[rel * SP + 256 * d]; return f2{x.x, x.y}; };   // frame 2 i + par + rel
      if (par == 0) {
        acc[0] = ((cell(0, 0) + carry[0]) + carry[1]) + carry[2];                 // hop 0: frames 0, -1, -2, -3
        acc[1] = ((cell(2, 0) + cell(1, 1)) + cell(0, 2)) + carry[3];             // hop 2: frames 2, 1, 0, -1
      } else {
        acc[0] = ((cell(0, 0) + cell(-1, 1)) + carry[0]) + carry[1];              // hop 1: frames 1, 0, -1, -2
        acc[1] = ((cell(2, 0) + cell(1, 1)) + cell(0, 2)) + cell(-1, 3);          // hop 3: frames 3, 2, 1, 0
      }
      IP_FENCE();   // (in batches: with Griffin-Lim's 99 prefetched registers beside the 66 staged ones, 32 cells in flight spill)
#pragma unroll
      for (int i = 2; i < 8; ++i) {
        acc[i] = ((cell(2 * i, 0) + cell(2 * i - 1, 1)) + cell(2 * i - 2, 2)) + cell(2 * i - 3, 3);
        if (GL && (i & 1)) IP_FENCE();
      }
      // what the next tile needs of frames 13, 14, 15 (parity 0: frame 15 at segment 1, 14 at 2, 13 at 3, then 15 at 3;
      // parity 1: frame 15 at segment 2, 14 at 3)
      {
        const float2 *t2 = reinterpret_cast<const float2 *>(lds.slots) + ip_opaque(u);
        if (par == 0) {
          const float2 c0 = t2[15 * SP + 256], c1 = t2[14 * SP + 512], c2 = t2[13 * SP + 768], c3 = t2[15 * SP + 768];
          carry[0] = f2{c0.x, c0.y}; carry[1] = f2{c1.x, c1.y}; carry[2] = f2{c2.x, c2.y}; carry[3] = f2{c3.x, c3.y};
        } else {
          const float2 c0 = t2[15 * SP + 512], c1 = t2[14 * SP + 768];
          carry[0] = f2{c0.x, c0.y}; carry[1] = f2{c1.x, c1.y};
        }
      }
      ip_signal(lds.drained, lane);   // (behind this wave's reads in LDS order)
      IP_STAMP(8);   // overlap-add reads and sums
      if (g_store) {
        float *out = a.out + (size_t)g_clip * (size_t)a.out_len;
        const int64_t q_tile = (int64_t)512 * kIpFT * g_ft;
        const int64_t e_tile = q_tile + a.env_q0, m_tile = q_tile - a.left;
        // (wave-uniform) every position of the tile is an interior one: inside the synthesis' span, the envelope's periodic part
        // and the output
        const bool inner = pa.aligned_out && e_tile >= a.head && e_tile + 512 * kIpFT <= a.stop && q_tile + 512 * kIpFT <= a.span &&
                           m_tile >= 0 && m_tile + 512 * kIpFT <= a.out_len;
        if (inner) {
          float2 *o2 = reinterpret_cast<float2 *>(out + m_tile) + 256 * par + u;
#pragma unroll
          for (int i = 0; i < 8; ++i) o2[512 * i] = make_float2((float)((double)acc[i].x * renv0), (float)((double)acc[i].y * renv1));
        } else {
          // a sample that the output holds: 0 past the synthesis' span, else the sum over the envelope (its periodic part as a
          // product with the reciprocal, the clip's first and last hops as a division: as istft2048_kernel)
          auto value = [&](float sum, int64_t q, double renv) {
            if (q >= a.span) return 0.f;
            const int64_t E = q + a.env_q0;
            if (E >= a.head && E < a.stop) return (float)((double)sum * renv);
            return (float)((double)sum / (E < a.head ? a.env_head[E] : a.env_tail[E - a.stop]));
          };
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int64_t Q = q_tile + 512 * (2 * i + par) + 2 * u, mo = Q - a.left;
            if (mo >= 0 && mo < a.out_len) out[mo] = value(acc[i].x, Q, renv0);
            if (mo + 1 >= 0 && mo + 1 < a.out_len) out[mo + 1] = value(acc[i].y, Q + 1, renv1);
          }
        }
      }
    }
  };
  int64_t p_clip = clip;
  int p_ft = ft;
  bool p_store = false;
  for (int it = 0; it < todo; ++it) {
    IP_STAMP(0);   // loop edge
    // S1 / S2: the slots are free; the staged registers go in.
    // Griffin-Lim's phase update is folded in here (stft.ml:1003-1012, as istft2048_kernel and gl_update_kernel form it): with
    // `unit`, z is the rebuilt spectrum c_k and what is inverted is mag * unit(c_k - beta c_(k-1)), unit(e) = e / (|e| + min_float).
    // The tile's factors are requested here, before the wait for the slots.
    // (Requested a phase earlier -- behind the previous tile's frames or its overlap-add -- and carried over the loop edge, hipcc
    // parks the 99 values in scratch as they arrive, one wait per load: 352 bytes of scratch against none.  The synthesis of a
    // Griffin-Lim iteration reads 4.9 GB for 0.5 GB written: it sits on the memory system either way.)
    if (GL) request_factors(clip, ft);
    if (it > 0) {   // the previous tile's overlap-add
      overlap_add(p_clip, p_ft, p_store, 8u * (unsigned)it);
      if (clip != p_clip) {   // a new clip: nothing reaches into it
#pragma unroll
        for (int i = 0; i < 4; ++i) carry[i] = f2{0.f, 0.f};
      }
    }
#if IP_GL_BULK
    if constexpr (GL) {   // the factors applied to all 33 elements at once (33 independent float64 chains for the scheduler), before the wait
      if (a.unit) {
        if (a.prev) {
#pragma unroll
          for (int i = 0; i < 33; ++i) { raw[i].x -= a.beta * pv[i].x; raw[i].y -= a.beta * pv[i].y; }
        }
#pragma unroll
        for (int i = 0; i < 33; ++i) {
          const float m = (float)hypot((double)raw[i].x, (double)raw[i].y) + FLT_MIN;
          raw[i].x /= m;
          raw[i].y /= m;
        }
      }
      if (a.mag) {
#pragma unroll
        for (int i = 0; i < 33; ++i) { raw[i].x *= mg[i]; raw[i].y *= mg[i]; }
      }
    }
#endif
    IP_STAMP(1);   // Griffin-Lim's factors
    ip_wait(lds.drained, 8u * (unsigned)it);
    IP_STAMP(2);   // wait: slots free
    {
      float2 *cell = reinterpret_cast<float2 *>(lds.slots + ip_opaque(sf) * kIpSlot) + ip_opaque(srow);
      auto staged_value = [&](int i) {
        float2 e = raw[i];
        if constexpr (GL && !IP_GL_BULK) {
          if (a.unit) {   // (wave-uniform)
            if (a.prev) {
              e.x -= a.beta * pv[i].x;
              e.y -= a.beta * pv[i].y;
            }
            const float m = (float)hypot((double)e.x, (double)e.y) + FLT_MIN;
            e.x /= m;
            e.y /= m;
          }
          if (a.mag) {
            e.x *= mg[i];
            e.y *= mg[i];
          }
        }
        return e;
      };
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        cell[32 * i] = staged_value(i);
        if (GL && (i & 3) == 3) IP_FENCE();
      }
      if (srow == 0) cell[1024] = staged_value(32);
    }
    ip_signal(lds.staged, lane);
    // S3: the next tile (of this clip, or the first one of the next clip)
    int ftn = ft + 1;
    int64_t clipn = clip;
    if (ftn == tpc) { ftn = 0; ++clipn; }
    IP_STAMP(3);   // staging stores (arrival of the staged registers included)
    // (The 33 requests hold the wave at their issue for ~8 000 of a tile's 23 700 cycles: the memory side accepts a CU's 264
    // requests of 512 bytes no faster than it streams them.  Spread in threes between the frame's stages they cost the same
    // ~500 cycles per group there and 28 bytes of scratch: profiles/r07/stamps_istft_*.log.  One run it stays.)
    if (it + 1 < todo) request(clipn, ftn);
    IP_FENCE();
    IP_STAMP(4);   // next tile's requests issued
    // S4 / S5
    if constexpr (!FM) ip_wait(lds.staged, 8u * (unsigned)(it + 1));   // (FM: the wave staged its own two frames; its LDS accesses complete in order)
    IP_STAMP(5);   // wait: every element staged
    ip_frame(lds, lds.slots + (2 * wave + (ip_opaque(lane) >> 5)) * kIpSlot, lane);
    ip_signal(lds.filled, lane);
    IP_FENCE();
    IP_STAMP(6);   // the two frames
    p_clip = clip;
    p_ft = ft;
    p_store = store;
    store = true;
    ft = ftn;
    clip = clipn;
  }
  if (todo > 0) overlap_add(p_clip, p_ft, p_store, 8u * (unsigned)todo);
#ifdef SMX_STAMPS
  ip_sum[20] = __builtin_amdgcn_s_memtime() - ip_t0;
  ip_sum[21] = __builtin_amdgcn_s_memrealtime() - ip_r0;
  ip_sum[22] = (unsigned long long)todo;
  if (lane == 0 && blockIdx.x < 4096)
    for (int i = 0; i < fftdev::kStampSlots; ++i) fftdev::g_stamp_sums[(blockIdx.x * 16 + wave) * fftdev::kStampSlots + i] = ip_sum[i];
#endif
}
