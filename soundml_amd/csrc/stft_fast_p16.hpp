// stft1024_power16_kernel -- the power spectrogram at fft 1024 on the register frame pipeline of stft_fast_p32.hpp, with a
// frame in 16 lanes (included by stft_fast.hip after stft_fast_p32.hpp, inside its anonymous namespace).  Replaces the
// reference's hot call for Stft.power_spectrum at BASELINE C1's geometry (fft 1024 / hop 256), stft.ml:356-364 + 670-691.
//
// M = N/2 = 512 = 16 x 32 complex points, z[n] = x[2n] + i x[2n+1]:
//   A. n = l + 16 j  : radix-32 over j in registers (fft32 of the 32-lane pipeline)      -> y_l[k1], twiddle W_M^(l k1)
//   X. 16 x 32 transposition through the frame's own column of the output tile: lane l register k1 -> cell l + 17 k1 ->
//      lane k1 mod 16, registers (k1 div 16, l); one plane at a time, base(lane) + immediate on both sides
//   B. two radix-16 over l in registers (k1 = lam, lam + 16)                               -> lane lam, register u: Z[lam + 16 u]
//   P. real-FFT post-pass, one slot per pair (k, M - k): lane lam owns slots s = 0..15 (k = lam + 16 s, its registers 0..15
//      against registers 31..16 of lane 16 - lam, fetched through the cells); lanes 0 and 8 pair inside themselves, slot
//      k = 0 yields X[0] and the Nyquist bin, bin M/2 = 256 (lane 0, register 16) is one extra product.
// A wave carries FOUR frames (16 lanes each), a workgroup is 8 waves = 32 frames = one tile [544 rows = bins][32 frames + pad],
// 33 floats per row.  The frames of a wave sit in columns w, w + 16, w + 8, w + 24 (lane quarters 0..3): the two frames of a
// 32-lane half are 16 columns = 16 banks apart, which keeps every transposition and exchange access conflict free at a
// cell pitch of 17 rows (bank = row + column mod 32).  Per frame the instruction stream is the 32-lane kernel's with the
// second radix-32 replaced by two radix-16, i.e. half the work per point; every function rounds operation by operation
// (contraction off), so a frame gets the same bits wherever it sits.
// LDS: 2 x 71,808 (tiles of 544 rows x 33) + 4,096 (window) + 3,968 (W_M^(l k1)) + 2,048 (post-pass twiddles) = 153,728 B;
// the counters sit in the pad column (rows 528..531 of buffer 0).

constexpr int kN16 = 1024, kM16 = 512, kBins16 = 513;
constexpr int kFT16 = 32;                          // frames per tile
constexpr int kTS16 = kFT16 + 1;                   // floats per tile row
constexpr int kRows16 = 544;
constexpr int kTile16Floats = kRows16 * kTS16;
constexpr size_t kTile16Bytes = (size_t)kTile16Floats * sizeof(float);     // 71,808
constexpr size_t kWin16Bytes = 16 * 16 * sizeof(float4);                    // window pairs of points l + 16 (2m), l + 16 (2m + 1)
constexpr size_t kTwA16Bytes = 15 * 16 * sizeof(float4) + 16 * sizeof(float2);
constexpr size_t kTwP16Bytes = 8 * 16 * sizeof(float4);
constexpr size_t kFast16Lds = 2 * kTile16Bytes + kWin16Bytes + kTwA16Bytes + kTwP16Bytes;
static_assert(kFast16Lds <= 160 * 1024, "LDS budget");
constexpr int kCellPitch16 = 17 * kTS16;           // floats between cells c and c + 17
constexpr int kRowPitch16 = 16 * kTS16;            // floats between rows r and r + 16

struct Lds16 {
  float *tiles;
  float4 *win4, *twA4, *twP4;
  float2 *twA31;
  unsigned *filled, *drained;   // [2] each, kTS16 floats apart
};
__device__ __forceinline__ Lds16 carve_lds16(unsigned char *smem) {
  Lds16 l;
  l.tiles = reinterpret_cast<float *>(smem);
  l.win4 = reinterpret_cast<float4 *>(smem + 2 * kTile16Bytes);
  l.twA4 = reinterpret_cast<float4 *>(smem + 2 * kTile16Bytes + kWin16Bytes);
  l.twA31 = reinterpret_cast<float2 *>(smem + 2 * kTile16Bytes + kWin16Bytes + 15 * 16 * sizeof(float4));
  l.twP4 = reinterpret_cast<float4 *>(smem + 2 * kTile16Bytes + kWin16Bytes + kTwA16Bytes);
  l.filled = reinterpret_cast<unsigned *>(l.tiles + 528 * kTS16 + kFT16);
  l.drained = reinterpret_cast<unsigned *>(l.tiles + 530 * kTS16 + kFT16);
  return l;
}

// the column (= frame of the tile) of a lane quarter of wave w
__device__ __forceinline__ int column16(int wave, int quarter) { return wave + 16 * (quarter & 1) + 8 * (quarter >> 1); }

struct Lane16 {
  int l, col;
  int own;        // cell l of the frame's column: transposition / exchange writes (cell l + 17 j), results of bins l + 16 s
  int rd;         // cell 17 l: transposition reads (cell i + 17 (l + 16 a))
  int xr;         // exchange reads: cell p + 17 (15 - s) of slot s, p = 16 - l (l = 0: 17, i.e. its own register 32 - s)
  int rm;         // results of bins M - k: row (16 - l) + 16 (31 - s)  (l = 0: 16 (32 - s); s = 0 is row 512 = Nyquist)
  int self;       // lane 0: row 256 (bin M/2); other lanes: a cell they overwrite afterwards
  const float4 *win_l, *twA_l, *twP_l;
  const float2 *twA31_l;
};
__device__ __forceinline__ Lane16 setup_lane16(const Lds16 &lds, int lane, int wave) {
  Lane16 L;
  L.l = lane & 15;
  L.col = column16(wave, lane >> 4);
  L.own = L.l * kTS16 + L.col;
  L.rd = 17 * L.l * kTS16 + L.col;
  L.xr = (L.l == 0 ? 17 : 16 - L.l) * kTS16 + L.col;
  L.rm = ((L.l == 0 ? 16 : 16 - L.l) + 16 * 16) * kTS16 + L.col;
  L.self = (L.l == 0 ? 256 : L.l) * kTS16 + L.col;
  L.win_l = lds.win4 + L.l;
  L.twA_l = lds.twA4 + L.l;
  L.twA31_l = lds.twA31 + L.l;
  L.twP_l = lds.twP4 + L.l;
  return L;
}

__device__ __forceinline__ void fill_tables16(const FastArgs &a, const Lds16 &lds, int tid, int nthreads) {
  const float2 *hw = reinterpret_cast<const float2 *>(a.hwin);
  for (int e = tid; e < 16 * 16; e += nthreads) {
    const int m = e >> 4, l = e & 15;
    const float2 w0 = hw[l + 16 * (2 * m)], w1 = hw[l + 16 * (2 * m + 1)];
    lds.win4[e] = make_float4(w0.x, w0.y, w1.x, w1.y);
  }
  for (int e = tid; e < 15 * 16; e += nthreads) {
    const int m = e >> 4, l = e & 15;
    const float2 w0 = a.w_m[l * (2 * m + 1)], w1 = a.w_m[l * (2 * m + 2)];
    lds.twA4[e] = make_float4(w0.x, w0.y, w1.x, w1.y);
  }
  for (int e = tid; e < 16; e += nthreads) lds.twA31[e] = a.w_m[e * 31];
  for (int e = tid; e < 8 * 16; e += nthreads) {
    const int m = e >> 4, l = e & 15;
    const float2 w0 = a.w_n[l + 16 * (2 * m)], w1 = a.w_n[l + 16 * (2 * m + 1)];
    lds.twP4[e] = make_float4(w0.x, w0.y, w1.x, w1.y);
  }
  if (tid < 2) { lds.filled[tid * kTS16] = 0u; lds.drained[tid * kTS16] = 0u; }
}

// Four frames (one per lane quarter): raw samples -> window -> FFT(512 complex) -> post-pass -> |X|^p in the frames' columns
// of `tile`; the hooks of `mid` are those of frame32_to_tile.
template <int PMODE, class Mid>
__device__ __forceinline__ void frame16_to_tile(const FastArgs &a, const Lane16 &L, float2 (&raw)[32], float *tile, const Mid &mid) {
#pragma clang fp contract(off)
  c32 v[32], t[32];
#pragma unroll
  for (int m0 = 0; m0 < 16; m0 += 8) {
    float4 win[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) win[m] = L.win_l[16 * (m0 + m)];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      v[2 * (m0 + m)] = {raw[2 * (m0 + m)].x * win[m].x, raw[2 * (m0 + m)].y * win[m].y};
      v[2 * (m0 + m) + 1] = {raw[2 * (m0 + m) + 1].x * win[m].z, raw[2 * (m0 + m) + 1].y * win[m].w};
    }
    SMX_FENCE();
  }
  SMX_FENCE();
  {   // A: radix-32 over j, then twiddle W_M^(l k1), the first plane of the transposition written between the products
    float4 tw[15];
#pragma unroll
    for (int m = 0; m < 15; ++m) tw[m] = L.twA_l[16 * m];
    const float2 tw31 = L.twA31_l[0];
    fft32(v, [&] { SMX_FENCE(); mid.early(); SMX_FENCE(); });
    SMX_FENCE();
    mid.before_cells();
    float *const wr = tile + opaque32(L.own);
    float *const wr_hi = wr + 16 * kCellPitch16;
    wr[0] = v[0].x;
#pragma unroll
    for (int m = 0; m < 15; ++m) {
      v[2 * m + 1] = p32_cmul(v[2 * m + 1], tw[m].x, tw[m].y);
      v[2 * m + 2] = p32_cmul(v[2 * m + 2], tw[m].z, tw[m].w);
      (2 * m + 1 < 16 ? wr : wr_hi)[kCellPitch16 * ((2 * m + 1) & 15)] = v[2 * m + 1].x;
      (2 * m + 2 < 16 ? wr : wr_hi)[kCellPitch16 * ((2 * m + 2) & 15)] = v[2 * m + 2].x;
      if ((m & 1) == 1) SMX_FENCE();
    }
    v[31] = p32_cmul(v[31], tw31.x, tw31.y);
    wr_hi[kCellPitch16 * 15] = v[31].x;
  }
  SMX_FENCE();
  float *const wr = tile + opaque32(L.own);
  float *const wr_hi = wr + 16 * kCellPitch16;   // (ds offsets are 16 bits)
  const float *const rd = tile + opaque32(L.rd);
  // X: lane lam takes V[i][lam + 16 a], i < 16, a < 2: cell i + 17 (lam + 16 a); t[16 a + i]
#pragma unroll
  for (int i = 0; i < 32; ++i) t[i].x = rd[kTS16 * (i & 15) + 16 * kCellPitch16 * (i >> 4)];
#pragma unroll
  for (int j = 0; j < 32; ++j) (j < 16 ? wr : wr_hi)[kCellPitch16 * (j & 15)] = v[j].y;
#pragma unroll
  for (int i = 0; i < 32; ++i) t[i].y = rd[kTS16 * (i & 15) + 16 * kCellPitch16 * (i >> 4)];
  SMX_FENCE();
  mid.after_transposition_issue();
  SMX_FENCE();
  // B: two radix-16 over l; register u = a + 2 q holds Z[lam + 16 u]
  {
    c32 e[16], o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { e[i] = t[i]; o[i] = t[16 + i]; }
    p32_fft16(e);
    p32_fft16(o);
#pragma unroll
    for (int q = 0; q < 16; ++q) { t[2 * q] = e[q]; t[2 * q + 1] = o[q]; }
  }
  SMX_FENCE();
  // P: partners through the cells (as frame32_to_tile, 16 lanes)
  float px[16], py[16];
  const float *const xr = tile + opaque32(L.xr);
#pragma unroll
  for (int q = 16; q < 32; ++q) wr[kCellPitch16 * (q - 16)] = t[q].x;
  wr_hi[0] = t[0].x;
#pragma unroll
  for (int s = 0; s < 16; ++s) px[s] = xr[kCellPitch16 * (15 - s)];
#pragma unroll
  for (int q = 16; q < 32; ++q) wr[kCellPitch16 * (q - 16)] = t[q].y;
  wr_hi[0] = t[0].y;
#pragma unroll
  for (int s = 0; s < 16; ++s) py[s] = xr[kCellPitch16 * (15 - s)];
  float4 tw[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) tw[m] = L.twP_l[16 * m];
  SMX_FENCE();
  mid.after_exchange_issue();
  SMX_FENCE();
  auto power_of = [&](float re, float im) { return power_from_square<PMODE>(__builtin_fmaf(re, re, im * im), a); };
  {   // bin M/2 (lane 0, register 16): X = 2 conj(Z)
    const float zx = t[16].x + t[16].x, zy = t[16].y + t[16].y;
    tile[opaque32(L.self)] = power_of(zx, zy);
  }
  float *const rk = wr;                        // row l + 16 s
  float *const rm = tile + opaque32(L.rm);     // row (16 - l) + 16 (31 - s) = rm base + 16 (15 - s)
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const float wx = (s & 1) ? tw[s >> 1].z : tw[s >> 1].x, wy = (s & 1) ? tw[s >> 1].w : tw[s >> 1].y;
    const c32 e = {t[s].x + px[s], t[s].y - py[s]};
    const c32 d = {t[s].x - px[s], t[s].y + py[s]};
    const float tr = __builtin_fmaf(wx, d.y, wy * d.x);
    const float ti = __builtin_fmaf(wy, d.y, -(wx * d.x));
    rk[kRowPitch16 * s] = power_of(e.x + tr, e.y + ti);
    rm[kRowPitch16 * (15 - s)] = power_of(e.x - tr, e.y - ti);
    if (s == SMX_P32_STORE_AT || s == SMX_P32_LOAD_AT) { SMX_FENCE(); mid.postpass_at(s); SMX_FENCE(); }
  }
}

// raw samples of the lane's frame: z[n] = (x[2n], x[2n+1]), n = l + 16 j; `src` is the frame's first sample (per lane)
template <bool ALIGNED>
__device__ __forceinline__ void load_frame16(const float *src, int l, float2 (&raw)[32]) {
  if constexpr (ALIGNED) {
    const float2 *p = reinterpret_cast<const float2 *>(src) + l;
#pragma unroll
    for (int j = 0; j < 32; ++j) raw[j] = p[16 * j];
  } else {
    const float *p = src + 2 * l;
#pragma unroll
    for (int j = 0; j < 32; ++j) raw[j] = make_float2(p[32 * j], p[32 * j + 1]);
  }
}

// A wave's share of a finished tile: 8 parts of 8 rows (bins) x 4 frames per lane -> out[clip][bin][f0 + 4 g ..]; a row is
// one 128-byte run of 8 lanes.  Rows 4 h + r' (r' < 4) per 32-lane half keep the LDS reads conflict free.
struct Flush16 {
  int src0;         // float offset in the tile of part 0: row0 * 33 + 4 g, row0 = 8 wave + (lane >> 3)
  unsigned goff0;   // byte offset of out[row0][4 g] from the tile's origin
  int g;
};
struct FlushRegs16 {
  float v[8][4];
  float nyq[4];     // bin 512 (wave 0, lanes 0..7)
};
__device__ __forceinline__ void flush16_read(const float *tile, const Flush16 &fl, FlushRegs16 &r) {
  const float *src0 = tile + opaque32(fl.src0);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const float *src = src0 + 64 * it * kTS16;
    r.v[it][0] = src[0]; r.v[it][1] = src[1]; r.v[it][2] = src[2]; r.v[it][3] = src[3];
  }
  const float *ny = tile + kM16 * kTS16 + 4 * fl.g;   // row 512 = Nyquist bin (every wave reads it, wave 0 lanes 0..7 store it)
  r.nyq[0] = ny[0]; r.nyq[1] = ny[1]; r.nyq[2] = ny[2]; r.nyq[3] = ny[3];
}
__device__ __forceinline__ void flush16_store(const FastArgs &a, const Flush16 &fl, float *obase, int frames_left, int wave, int lane,
                                              const FlushRegs16 &r) {
  const unsigned pitch = (unsigned)a.out_stride * 4u;
  const unsigned goff0 = opaque32(fl.goff0);
  const int fleft = frames_left - 4 * fl.g;
  auto put = [&](unsigned goff, const float (&v)[4]) {
    if (fleft >= 4) {
      store4_unaligned(obase, goff, v[0], v[1], v[2], v[3]);
    } else {
      float *dst = reinterpret_cast<float *>(reinterpret_cast<char *>(obase) + goff);
      if (fleft > 0) dst[0] = v[0];
      if (fleft > 1) dst[1] = v[1];
      if (fleft > 2) dst[2] = v[2];
    }
  };
  if (frames_left >= kFT16) {   // wave-uniform: a whole tile, no masks
#pragma unroll
    for (int it = 0; it < 8; ++it) store4_unaligned(obase, goff0 + (unsigned)(64 * it) * pitch, r.v[it][0], r.v[it][1], r.v[it][2], r.v[it][3]);
  } else {
#pragma unroll
    for (int it = 0; it < 8; ++it) put(goff0 + (unsigned)(64 * it) * pitch, r.v[it]);
  }
  if (wave == 0 && lane < 8) put((unsigned)kM16 * pitch + 16u * (unsigned)fl.g, r.nyq);
}

template <bool ALIGNED>
struct PowerMid16 {
  const FastArgs &a;
  const Lds16 &lds;
  const Flush16 &fl;
  FlushRegs16 &fr;
  float2 (&raw)[32];
  const float *src;      // the next frames' samples (per lane)
  float *pend_out;       // output origin and frames of the previous tile
  int pend_left;
  int lane, wave, b, it;
  unsigned &pk_drained, &pk_filled;
  __device__ __forceinline__ void early() const { pk_drained = peek32(lds.drained + b * kTS16); }
  __device__ __forceinline__ void before_cells() const {
    lds_wait32(lds.drained + b * kTS16, 8u * ((unsigned)it >> 1), pk_drained);
  }
  __device__ __forceinline__ void after_transposition_issue() const {
    if (it > 0) pk_filled = peek32(lds.filled + (b ^ 1) * kTS16);
  }
  __device__ __forceinline__ void after_exchange_issue() const {
    if (it > 0) {
      lds_wait32(lds.filled + (b ^ 1) * kTS16, 8u * (((unsigned)(it - 1) >> 1) + 1), pk_filled);
      flush16_read(lds.tiles + (b ^ 1) * kTile16Floats, fl, fr);
      lds_signal32(lds.drained + (b ^ 1) * kTS16, lane);   // "read out" as soon as the reads are issued (in-order LDS)
    }
  }
  __device__ __forceinline__ void postpass_at(int s) const {
    const bool same = SMX_P32_STORE_AT == SMX_P32_LOAD_AT;
    if (s == SMX_P32_LOAD_AT && same && SMX_P32_LOADS_FIRST) { load_frame16<ALIGNED>(src, lane & 15, raw); SMX_FENCE(); }
    if (s == SMX_P32_STORE_AT && it > 0) flush16_store(a, fl, pend_out, pend_left, wave, lane, fr);
    SMX_FENCE();
    if (s == SMX_P32_LOAD_AT && !(same && SMX_P32_LOADS_FIRST)) load_frame16<ALIGNED>(src, lane & 15, raw);
  }
};

template <bool ALIGNED, int PMODE, bool STRIP>
__global__ void __launch_bounds__(512) stft1024_power16_kernel(FastArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Lds16 lds = carve_lds16(smem);
  const Lane16 L = setup_lane16(lds, lane, wave);
  fill_tables16(a, lds, tid, 512);
  TileWalk tw;
  tw.init(a, a.out + a.out_offset, kBins16 * a.out_stride);
  const int ntiles = tw.ntiles > 0 ? tw.ntiles : 0;

  // first sample of this lane's frame in tile t of the clip at xc (a lane quarter without a frame re-reads the tile's first
  // frame and its results are never stored)
  auto frame_ptr = [&](const float *xc, int t) {
    const int64_t f0 = (int64_t)t * kFT16;
    const int avail = (int)(a.count - f0 < kFT16 ? a.count - f0 : kFT16) - 1;   // last frame of the tile that exists (wave-uniform)
    const int fi = L.col;
    const int64_t p = a.p0 + f0 + (fi <= avail ? fi : 0);
    if (a.fold_frames && (p < a.border_i0 || p >= a.border_i1)) {
      const int64_t clip = (xc - a.x) / a.x_stride;
      return p < a.border_i0 ? a.strip_l + clip * a.strip_l_stride + (p - a.p0) * a.hop
                             : a.strip_r + clip * a.strip_r_stride + (p - a.border_i1) * a.hop;
    }
    return xc + (p * a.hop - a.left);
  };

  float2 raw[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) raw[j] = make_float2(0.f, 0.f);
  if (ntiles > 0) load_frame16<ALIGNED>(frame_ptr(tw.xclip, tw.ft), L.l, raw);
  __syncthreads();   // tables and zeroed counters visible: the only workgroup barrier of the main loop
  float *pend_out = nullptr;
  int pend_left = 0;
  Flush16 fl;
  {
    fl.g = lane & 7;
    const int row0 = 8 * wave + (lane >> 3);
    fl.src0 = row0 * kTS16 + 4 * fl.g;
    fl.goff0 = ((unsigned)row0 * (unsigned)a.out_stride + 4u * fl.g) * 4u;
  }
  FlushRegs16 fr;
  unsigned pk_drained = 0, pk_filled = 0;
  for (int it = 0; it < ntiles; ++it) {   // tile `it` of this workgroup lives in buffer it & 1
    const int b = it & 1;
    int ftnext;
    const float *xnext;
    float *onext;
    tw.peek(a, ftnext, xnext, onext);
    const bool more = it + 1 < ntiles;
    const float *src = frame_ptr(more ? xnext : tw.xclip, more ? ftnext : tw.ft);
    const PowerMid16<ALIGNED> mid{a, lds, fl, fr, raw, src, pend_out, pend_left, lane, wave, b, it, pk_drained, pk_filled};
    frame16_to_tile<PMODE>(a, L, raw, lds.tiles + b * kTile16Floats, mid);
    lds_signal32(lds.filled + b * kTS16, lane);
    pend_out = tw.oclip + tw.ft * kFT16;   // wave-uniform
    const int64_t left = a.count - (int64_t)tw.ft * kFT16;
    pend_left = left < kFT16 ? (int)left : kFT16;
    tw.xclip = xnext;
    tw.oclip = onext;
    tw.ft = ftnext;
  }
  if (ntiles > 0) {   // the last tile of this workgroup
    const int b = (ntiles - 1) & 1;
    lds_wait(lds.filled + b * kTS16, 8u * (((unsigned)(ntiles - 1) >> 1) + 1));
    flush16_read(lds.tiles + b * kTile16Floats, fl, fr);
    flush16_store(a, fl, pend_out, pend_left, wave, lane, fr);
  }

  // Border frames (the few per clip whose window reaches past either end of the signal): same frame code on samples fetched
  // through the padding rule, 32 (clip, frame) pairs per tile, results scattered to their places.
  if (a.border_left + a.border_right > 0) {
    const int per = a.border_left + a.border_right;
    const int64_t lead = a.total_tiles / a.tiles_per_clip;
    const int64_t total = lead * per;
    auto locate = [&](int64_t beta, int64_t &clip, int64_t &p) {
      clip = beta / per;
      const int r = (int)(beta % per);
      p = r < a.border_left ? a.border_p0 + r : a.border_i1 + (r - a.border_left);
    };
    float *bt_tile = lds.tiles;
    for (int64_t bt = blockIdx.x; bt * kFT16 < total; bt += gridDim.x) {
      __syncthreads();   // the buffer is free: every wave is past its last flush / the previous border tile
      {
        int64_t beta = bt * kFT16 + L.col;
        if (beta >= total) beta = bt * kFT16;   // a quarter without a pair repeats the tile's first one (never stored)
        int64_t clip, p;
        locate(beta, clip, p);
        const float *xs = a.x + clip * a.x_stride;
        const int s0 = (int)(p * a.hop - a.left);
        float2 braw[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          const int s = s0 + 2 * (L.l + 16 * j);
          braw[j] = make_float2(fetch_padded(xs, (int)a.n, s, a.pad, a.pad_value), fetch_padded(xs, (int)a.n, s + 1, a.pad, a.pad_value));
        }
        frame16_to_tile<PMODE>(a, L, braw, bt_tile, NoMid32{});
      }
      __syncthreads();
      for (int e = tid; e < kBins16 * kFT16; e += 512) {
        const int k = e / kFT16, f = e % kFT16;
        const int64_t bf = bt * kFT16 + f;
        if (bf < total) {
          int64_t clip, p;
          locate(bf, clip, p);
          a.out[(clip * kBins16 + k) * a.out_stride + a.border_out_offset + (p - a.border_p0)] = bt_tile[k * kTS16 + f];
        }
      }
    }
  }
}
