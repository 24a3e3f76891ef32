// stft2048_power32_kernel -- the power spectrogram at fft 2048 on a 32-lane frame pipeline (included by stft_fast.hip,
// inside its anonymous namespace).  Replaces the reference's hot call for Stft.power_spectrum, stft.ml:356-364 + 670-691.
//
// Why a second pipeline (round 3, profiles/r05/issue_probe.log): on gfx950 a SIMD retires one plain VOP2 vector instruction
// per 2.07 cycles but a DPP form takes 4.24, a v_permlane*_swap 8.1 and an instruction with an SGPR operand 4.06; the
// 64-lane pipeline of stft2048_power_kernel spends a third of its vector time in exactly those (in-wave transposes, the
// radix-4 stage across the quad, selects).  Here a frame lives in 32 lanes with 32 complex points per lane, so that
// M = 1024 = 32 x 32 needs NO cross-lane arithmetic at all:
//   A. n = l + 32 j  : radix-32 over j in registers (two radix-16 + one combining pass)      -> y_l[k1], twiddle W_M^(l k1)
//   X. 32 x 32 transposition through the frame's own column of the output tile in LDS (the cells are free until the
//      frame's results are written): one plane at a time, cell l + 33 j, i.e. base(lane) + immediate on both sides and
//      every access bank-conflict free (a pitch of 33 cells needs 1056 rows per column instead of 1024)
//   B. radix-32 over l in registers                                                         -> lane k1, register q: Z[k1 + 32 q]
//   P. real-FFT post-pass, one slot per PAIR (k, M - k): E = Z[k] + conj Z[M-k], D = Z[k] - conj Z[M-k], T = -i w_k D,
//      X[k] = E + T, X[M-k] = conj(E - T) -- 16 instructions for two bins where the per-bin form takes 20.  Lane k1
//      owns slots q = 0..15 (its registers 0..15 against registers 31..16 of lane 32 - k1, fetched through the tile
//      cells); k1 = 0 and 16 pair inside themselves by the same address rule, the slot k = 0 yields X[0] and the
//      Nyquist bin X[M] = conj(E - T), and bin M/2 is one extra product.
// A wave carries TWO frames (lanes 0-31 / 32-63, identical instruction stream, so any frame gets the same bits wherever
// it sits); a workgroup is 8 waves = 16 consecutive frames of one clip = one tile, two tile buffers, one persistent
// workgroup per CU.  Per frame and lane-half: ~650 plain vector instructions (845 incl. 128 DPP / 32 swaps / 32 selects
// before) and the waves need 2 per SIMD (<= 256 registers) instead of 4.
// Tile rows are bins (row 1024 = Nyquist), 17 floats apart; the flush, the counters and the tile order are those of
// stft2048_power_kernel.  LDS: 2 x 71,808 (tiles of 1056 rows) + 8,192 (window) + 7,936 (W_M^(l k1)) + 4,096 (post-pass
// twiddles) = 163,840 B; the four synchronisation counters sit in unused cells of the pad column.

constexpr int kRows32 = 1056;
constexpr int kTile32Floats = kRows32 * kTileStride;
constexpr size_t kTile32Bytes = (size_t)kTile32Floats * sizeof(float);     // 71,808
constexpr size_t kTwA32Bytes = 31 * 32 * sizeof(float2);                    // W_M^(l k1), k1 = 1..31   [k1-1][l]
constexpr size_t kTwP32Bytes = 512 * sizeof(float2);                        // exp(-2 pi i k / N), k < 512
constexpr size_t kFast32Lds = 2 * kTile32Bytes + kWinBytes + kTwA32Bytes + kTwP32Bytes;
static_assert(kFast32Lds <= 160 * 1024, "LDS budget");
constexpr int kCellPitch32 = 33 * kTileStride;     // floats between cells c and c + 33
constexpr int kRowPitch32 = 32 * kTileStride;      // floats between rows r and r + 32

struct Lds32 {
  float *tiles;
  float2 *win, *twA, *twP;
  unsigned *filled, *drained;   // [2] each, kTileStride floats apart (pad cells of rows 1040..1043 of buffer 0)
};
__device__ __forceinline__ Lds32 carve_lds32(unsigned char *smem) {
  Lds32 l;
  l.tiles = reinterpret_cast<float *>(smem);
  l.win = reinterpret_cast<float2 *>(smem + 2 * kTile32Bytes);
  l.twA = reinterpret_cast<float2 *>(smem + 2 * kTile32Bytes + kWinBytes);
  l.twP = reinterpret_cast<float2 *>(smem + 2 * kTile32Bytes + kWinBytes + kTwA32Bytes);
  l.filled = reinterpret_cast<unsigned *>(l.tiles + 1040 * kTileStride + kFT);
  l.drained = reinterpret_cast<unsigned *>(l.tiles + 1042 * kTileStride + kFT);
  return l;
}

// per-lane constants: offsets (floats) into a tile buffer
struct Lane32 {
  int l, h;
  int own;        // cell l of the frame's column: transposition / exchange writes (cell l + 33 j), results of bins l + 32 s
  int rd;         // cell 33 l: transposition reads (cell i + 33 l)
  int xr;         // exchange reads: cell p + 33 (15 - s) of slot s, p = 32 - l (l = 0: 33, i.e. its own register 32 - s)
  int rm;         // results of bins M - k: row (32 - l) + 32 (31 - s)  (l = 0: 32 (32 - s); s = 0 is row 1024 = Nyquist)
  int self;       // lane 0: row 512 (bin M/2); other lanes: a cell they overwrite afterwards
  const float2 *win_l, *twA_l, *twP_l;
};
__device__ __forceinline__ Lane32 setup_lane32(const Lds32 &lds, int lane, int wave) {
  Lane32 L;
  L.l = lane & 31;
  L.h = lane >> 5;
  const int col = 2 * wave + L.h;
  L.own = L.l * kTileStride + col;
  L.rd = 33 * L.l * kTileStride + col;
  L.xr = (L.l == 0 ? 33 : 32 - L.l) * kTileStride + col;
  L.rm = ((L.l == 0 ? 32 : 32 - L.l) + 32 * 16) * kTileStride + col;
  L.self = (L.l == 0 ? 512 : L.l) * kTileStride + col;
  L.win_l = lds.win + L.l;
  L.twA_l = lds.twA + L.l - 32;   // row k1 - 1
  L.twP_l = lds.twP + L.l;
  return L;
}

// The arithmetic of this pipeline is written out operation by operation (explicit fused multiply-adds, contraction off
// inside these functions): the kernel is instantiated several times (aligned / unaligned samples, strips, the border
// epilogue) and every copy must round identically -- a frame has ONE value wherever and however it is computed
// (range tiling, streaming partition and batch-slice laws: stft_grid.ml:32-73,180-205, stft_law.ml:79-164).
// a * (cx + i cy): 2 products + 2 fused multiply-adds
__device__ __forceinline__ c32 p32_cmul(c32 a, float cx, float cy) {
#pragma clang fp contract(off)
  return {__builtin_fmaf(a.y, -cy, a.x * cx), __builtin_fmaf(a.y, cx, a.x * cy)};
}
__device__ __forceinline__ void p32_fft4(c32 &a, c32 &b, c32 &c, c32 &d) {
#pragma clang fp contract(off)
  const c32 t0 = {a.x + c.x, a.y + c.y}, t1 = {a.x - c.x, a.y - c.y};
  const c32 t2 = {b.x + d.x, b.y + d.y}, t3 = {b.y - d.y, d.x - b.x};   // -i (b - d)
  a = {t0.x + t2.x, t0.y + t2.y};
  b = {t1.x + t3.x, t1.y + t3.y};
  c = {t0.x - t2.x, t0.y - t2.y};
  d = {t1.x - t3.x, t1.y - t3.y};
}
// 16-point forward DFT, natural order in and out (4 x 4)
__device__ __forceinline__ void p32_fft16(c32 (&v)[16]) {
#pragma clang fp contract(off)
  constexpr float c1 = (float)0.92387953251128674, s1 = (float)0.38268343236508977;
  constexpr float hh = (float)0.70710678118654752;
#pragma unroll
  for (int n0 = 0; n0 < 4; ++n0) p32_fft4(v[n0], v[4 + n0], v[8 + n0], v[12 + n0]);
  v[5] = p32_cmul(v[5], c1, -s1);     // W16^1
  v[6] = p32_cmul(v[6], hh, -hh);     // W16^2
  v[7] = p32_cmul(v[7], s1, -c1);     // W16^3
  v[9] = p32_cmul(v[9], hh, -hh);     // W16^2
  v[10] = {v[10].y, -v[10].x};        // W16^4
  v[11] = p32_cmul(v[11], -hh, -hh);  // W16^6
  v[13] = p32_cmul(v[13], s1, -c1);   // W16^3
  v[14] = p32_cmul(v[14], -hh, -hh);  // W16^6
  v[15] = p32_cmul(v[15], -c1, s1);   // W16^9
#pragma unroll
  for (int k0 = 0; k0 < 4; ++k0) p32_fft4(v[4 * k0], v[4 * k0 + 1], v[4 * k0 + 2], v[4 * k0 + 3]);
  c32 t[16];
#pragma unroll
  for (int k0 = 0; k0 < 4; ++k0)
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) t[k0 + 4 * k1] = v[4 * k0 + k1];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = t[i];
}
// 32-point forward DFT, natural order in and out, in registers: two 16-point DFTs (even / odd inputs) and one
// combining pass X[k] = E[k] + W32^k O[k], X[k + 16] = E[k] - W32^k O[k]
__device__ __forceinline__ void fft32(c32 (&v)[32]) {
#pragma clang fp contract(off)
  constexpr float c1 = (float)0.98078528040323043, s1 = (float)0.19509032201612825;
  constexpr float c2 = (float)0.92387953251128674, s2 = (float)0.38268343236508977;
  constexpr float c3 = (float)0.83146961230254524, s3 = (float)0.55557023301960218;
  constexpr float hh = (float)0.70710678118654752;
  c32 e[16], o[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) { e[m] = v[2 * m]; o[m] = v[2 * m + 1]; }
  p32_fft16(e);
  p32_fft16(o);
  o[1] = p32_cmul(o[1], c1, -s1);
  o[2] = p32_cmul(o[2], c2, -s2);
  o[3] = p32_cmul(o[3], c3, -s3);
  o[4] = p32_cmul(o[4], hh, -hh);
  o[5] = p32_cmul(o[5], s3, -c3);
  o[6] = p32_cmul(o[6], s2, -c2);
  o[7] = p32_cmul(o[7], s1, -c1);
  o[8] = {o[8].y, -o[8].x};
  o[9] = p32_cmul(o[9], -s1, -c1);
  o[10] = p32_cmul(o[10], -s2, -c2);
  o[11] = p32_cmul(o[11], -s3, -c3);
  o[12] = p32_cmul(o[12], -hh, -hh);
  o[13] = p32_cmul(o[13], -c3, -s3);
  o[14] = p32_cmul(o[14], -c2, -s2);
  o[15] = p32_cmul(o[15], -c1, -s1);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const c32 a = e[k], b = o[k];
    v[k] = {a.x + b.x, a.y + b.y};
    v[k + 16] = {a.x - b.x, a.y - b.y};
  }
}

struct NoMid32 {
  __device__ __forceinline__ void before_cells() const {}
  __device__ __forceinline__ void after_transposition() const {}
  __device__ __forceinline__ void after_stage_b() const {}
  template <int I> __device__ __forceinline__ void stamp() const {}
};

// Two frames (one per lane-half): raw samples (registers) -> window -> FFT(1024 complex) -> post-pass -> |X|^p in the
// frames' columns of `tile`.  mid.before_cells() is called before the first access to the columns (the power kernel
// waits there until the buffer has been read out), mid.after_transposition() once the raw-sample registers and the
// first half of the pipeline are dead (the next frames' loads go there), mid.after_stage_b() between the second
// radix-32 and the post-pass (the previous tile's stores).
template <bool SQUARE, class Mid>
__device__ __forceinline__ void frame32_to_tile(const FastArgs &a, const Lane32 &L, float2 (&raw)[32], float *tile,
                                                const Mid &mid) {
#pragma clang fp contract(off)
  c32 v[32], t[32];
  {
    float2 win[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) win[j] = L.win_l[32 * j];
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = {raw[j].x * win[j].x, raw[j].y * win[j].y};
  }
  mid.template stamp<1>();
  SMX_FENCE();
  // A: radix-32 over j, then twiddle W_M^(l k1)
  {
    float2 tw[32];
#pragma unroll
    for (int k = 1; k < 32; ++k) tw[k] = L.twA_l[32 * k];
    fft32(v);
#pragma unroll
    for (int k = 1; k < 32; ++k) v[k] = p32_cmul(v[k], tw[k].x, tw[k].y);
  }
  SMX_FENCE();
  // X: lane l register k1 -> lane k1 register l through the frame's column, real parts then imaginary parts.
  // One wave's LDS operations execute in order, so no wait separates the rounds.
  mid.template stamp<2>();
  mid.before_cells();
  mid.template stamp<3>();
  float *const wr = tile + L.own;
  float *const wr_hi = wr + 16 * kCellPitch32;   // (ds offsets are 16 bits: 31 x 2244 bytes does not fit)
  const float *const rd = tile + L.rd;
#pragma unroll
  for (int j = 0; j < 32; ++j) (j < 16 ? wr : wr_hi)[kCellPitch32 * (j & 15)] = v[j].x;
#pragma unroll
  for (int i = 0; i < 32; ++i) t[i].x = rd[kTileStride * i];
#pragma unroll
  for (int j = 0; j < 32; ++j) (j < 16 ? wr : wr_hi)[kCellPitch32 * (j & 15)] = v[j].y;
#pragma unroll
  for (int i = 0; i < 32; ++i) t[i].y = rd[kTileStride * i];
  mid.template stamp<4>();
  mid.after_transposition();
  mid.template stamp<5>();
  SMX_FENCE();
  // B: radix-32 over l
  fft32(t);
  SMX_FENCE();
  mid.template stamp<6>();
  mid.after_stage_b();
  mid.template stamp<7>();
  // P: partners through the cells.  Every lane parks registers 16..31 (cell l + 33 (q - 16)) and reads, for slot s,
  // register 31 - s of lane 32 - l (lanes 0 and 16: their own; lane 0: register 32 - s, and itself for s = 0).
  float px[16], py[16];
  const float *const xr = tile + L.xr;
#pragma unroll
  for (int q = 16; q < 32; ++q) wr[kCellPitch32 * (q - 16)] = t[q].x;
#pragma unroll
  for (int s = 0; s < 16; ++s) px[s] = xr[kCellPitch32 * (15 - s)];
#pragma unroll
  for (int q = 16; q < 32; ++q) wr[kCellPitch32 * (q - 16)] = t[q].y;
#pragma unroll
  for (int s = 0; s < 16; ++s) py[s] = xr[kCellPitch32 * (15 - s)];
  float2 tw[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) tw[s] = L.twP_l[32 * s];
  if (L.l == 0) { px[0] = t[0].x; py[0] = t[0].y; }   // bin 0 pairs with itself: X[0] and the Nyquist bin
  mid.template stamp<8>();
  SMX_FENCE();
  auto power_of = [&](float re, float im) {
    float pw = __builtin_fmaf(re, re, im * im);
    if constexpr (!SQUARE) pw = a.pmode == 1 ? sqrtf(pw) : __powf(pw, a.half_power);
    return pw;
  };
  {   // bin M/2 (lane 0, register 16): X = 2 conj(Z)
    const float zx = t[16].x + t[16].x, zy = t[16].y + t[16].y;
    tile[L.self] = power_of(zx, zy);
  }
  float *const rk = wr;                 // row l + 32 s
  float *const rm = tile + L.rm;        // row (32 - l) + 32 (31 - s) = rm base + 32 (15 - s)
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const c32 e = {t[s].x + px[s], t[s].y - py[s]};
    const c32 d = {t[s].x - px[s], t[s].y + py[s]};
    // T = -i w D
    const float tr = __builtin_fmaf(tw[s].x, d.y, tw[s].y * d.x);
    const float ti = __builtin_fmaf(tw[s].y, d.y, -(tw[s].x * d.x));
    rk[kRowPitch32 * s] = power_of(e.x + tr, e.y + ti);
    rm[kRowPitch32 * (15 - s)] = power_of(e.x - tr, e.y - ti);
  }
  mid.template stamp<9>();
}

// raw samples of the lane's frame: z[n] = (x[2n], x[2n+1]), n = l + 32 j; `src` is the frame's first sample (per lane:
// the two halves of a wave read different frames)
template <bool ALIGNED>
__device__ __forceinline__ void load_frame32(const float *src, int l, float2 (&raw)[32]) {
  if constexpr (ALIGNED) {
    const float2 *p = reinterpret_cast<const float2 *>(src) + l;
    long hi_off = 512;
    asm volatile("" : "+s"(hi_off));   // keeps ONE second base (13-bit immediates reach 16 x 256 bytes)
    const float2 *p_hi = p + hi_off;
#pragma unroll
    for (int j = 0; j < 32; ++j) raw[j] = j < 16 ? p[32 * j] : p_hi[32 * (j - 16)];
  } else {
    const float *p = src + 2 * l;
    long hi_off = 1024;
    asm volatile("" : "+s"(hi_off));
    const float *p_hi = p + hi_off;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const float *b = j < 16 ? p : p_hi;
      const int e = 64 * (j & 15);
      raw[j] = make_float2(b[e], b[e + 1]);
    }
  }
}

// One eighth of a wave's share of a finished tile: 16 rows (bins) x 4 frames per lane -> out[clip][bin][f0 + 4 g ..].
// Rows {0-3, 16-19} + 4 h per half-wave keep the LDS reads conflict free; a 4-lane group stores one 64-byte run.
struct Flush32 {
  int row0;         // tile row (= bin) of part 0
  unsigned goff0;   // byte offset of out[bin0][4 g] from the tile's origin
  int g;
};
__device__ __forceinline__ void flush32_part(const FastArgs &a, const float *tile, int it, const Flush32 &fl, float *obase,
                                             int frames_left, int wave, int lane) {
  const int drow = 32 * (it >> 1) + 8 * (it & 1);
  const float *src = tile + (fl.row0 + drow) * kTileStride + 4 * fl.g;
  const float v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
  const unsigned goff = fl.goff0 + (unsigned)drow * (unsigned)a.out_stride * 4u;
  const int fleft = frames_left - 4 * fl.g;
#ifdef SMX_DIAG
  if (a.abl_nostore == 1) {   // timing-only ablation: keep the LDS reads alive, drop the HBM stores
    asm volatile("" ::"v"(v0), "v"(v1), "v"(v2), "v"(v3));
    return;
  }
#endif
  if (fleft >= 4) {
    store4_unaligned(obase, goff, v0, v1, v2, v3);
  } else {
    float *dst = reinterpret_cast<float *>(reinterpret_cast<char *>(obase) + goff);
    if (fleft > 0) dst[0] = v0;
    if (fleft > 1) dst[1] = v1;
    if (fleft > 2) dst[2] = v2;
  }
  if (it == 7 && wave == 0 && lane < 16) {   // bin 1024 = row 1024
    if (lane < frames_left) obase[(int64_t)kM * a.out_stride + lane] = tile[kM * kTileStride + lane];
  }
}

#ifndef SMX_P32_PREFETCH_AT
#define SMX_P32_PREFETCH_AT 1   // 1: the next frames' loads are issued after the transposition, 2: after the post-pass
#endif

// what the power kernel does between the stages of a frame pair (see frame32_to_tile)
template <bool ALIGNED, class Flush>
struct PowerMid32 {
  const Lds32 &lds;
  const Flush &flush;
  float2 (&raw)[32];
  const float *src;
  int l, b, it;
#ifdef SMX_STAMPS
  unsigned long long *stamp_sum, *stamp_prev_p;
  template <int I> __device__ __forceinline__ void stamp() const {
    unsigned long long &stamp_prev = *stamp_prev_p;
    SMX_STAMP(I);
  }
#else
  template <int I> __device__ __forceinline__ void stamp() const {}
#endif
  __device__ __forceinline__ void before_cells() const {
    // buffer b last held tile it - 2, the (it >> 1)-th tile written there
    lds_wait(lds.drained + b * kTileStride, 8u * ((unsigned)it >> 1));
  }
  __device__ __forceinline__ void after_transposition() const {
    if constexpr (SMX_P32_PREFETCH_AT == 1) load_frame32<ALIGNED>(src, l, raw);
  }
  __device__ __forceinline__ void after_stage_b() const {
    if (it > 0) flush(b ^ 1, ((unsigned)(it - 1) >> 1) + 1);   // tile it - 1
  }
};

template <bool ALIGNED, bool SQUARE, bool STRIP>
__global__ void __launch_bounds__(512) stft2048_power32_kernel(FastArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Lds32 lds = carve_lds32(smem);
  const Lane32 L = setup_lane32(lds, lane, wave);
  // tables (once per workgroup)
  lds.win[tid] = reinterpret_cast<const float2 *>(a.hwin)[tid];
  lds.win[tid + 512] = reinterpret_cast<const float2 *>(a.hwin)[tid + 512];
  for (int e = tid; e < 31 * 32; e += 512) lds.twA[e] = a.w_m[(e & 31) * ((e >> 5) + 1)];
  lds.twP[tid] = a.w_n[tid];
  if (tid < 2) { lds.filled[tid * kTileStride] = 0u; lds.drained[tid * kTileStride] = 0u; }
  TileWalk tw;
  tw.init(a, a.out + a.out_offset, kBins * a.out_stride);
  const int ntiles = tw.ntiles > 0 ? tw.ntiles : 0;
  constexpr unsigned kWaves = 8;

  // first sample of this lane's frame in tile t of the clip at xc (a lane-half without a frame re-reads the tile's
  // first frame and its results are never stored)
  auto frame_ptr = [&](const float *xc, int t) {
    const int64_t f0 = (int64_t)t * kFT;
    const int64_t f = f0 + 2 * wave + L.h;
    const int64_t p = a.p0 + (f < a.count ? f : f0);
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
  if (ntiles > 0) load_frame32<ALIGNED>(frame_ptr(tw.xclip, tw.ft), L.l, raw);
  __syncthreads();   // tables and zeroed counters visible: the only workgroup barrier of the main loop
  float *pend_out = nullptr;
  int pend_left = 0;
  Flush32 fl;
  {
    const int hsel = lane >> 5, jj = (lane & 31) >> 2;
    fl.g = lane & 3;
    fl.row0 = 128 * wave + (jj & 3) + 16 * (jj >> 2) + 4 * hsel;
    fl.goff0 = ((unsigned)fl.row0 * (unsigned)a.out_stride + 4u * fl.g) * 4u;
  }
#ifdef SMX_STAMPS
  unsigned long long stamp_sum[kStampSlots] = {0}, stamp_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
  const unsigned long long clk_t0 = stamp_prev, clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  // this wave's share of the tile in buffer b (the `fills`-th tile written there), once every column is in
  auto flush_tile = [&](int b, unsigned fills) {
    lds_wait(lds.filled + b * kTileStride, kWaves * fills);
    SMX_STAMP(11);
    const float *ptile = lds.tiles + b * kTile32Floats;
#pragma unroll
    for (int part = 0; part < 8; ++part) flush32_part(a, ptile, part, fl, pend_out, pend_left, wave, lane);
    lds_signal(lds.drained + b * kTileStride, lane);
  };

  for (int it = 0; it < ntiles; ++it) {   // tile `it` of this workgroup lives in buffer it & 1
    const int b = it & 1;
    int ftnext;
    const float *xnext;
    float *onext;
    tw.peek(a, ftnext, xnext, onext);
    const bool more = it + 1 < ntiles;
    const float *src = frame_ptr(more ? xnext : tw.xclip, more ? ftnext : tw.ft);
    const bool have = (int64_t)tw.ft * kFT + 2 * wave < a.count;   // wave-uniform: at least the first half has a frame
#ifdef SMX_STAMPS
    const PowerMid32<ALIGNED, decltype(flush_tile)> mid{lds, flush_tile, raw, src, L.l, b, it, stamp_sum, &stamp_prev};
#else
    const PowerMid32<ALIGNED, decltype(flush_tile)> mid{lds, flush_tile, raw, src, L.l, b, it};
#endif
    mid.template stamp<0>();
    if (have) {
      frame32_to_tile<SQUARE>(a, L, raw, lds.tiles + b * kTile32Floats, mid);
    } else {
      mid.before_cells();
      mid.after_transposition();
      mid.after_stage_b();
    }
    lds_signal(lds.filled + b * kTileStride, lane);
    mid.template stamp<10>();
    if constexpr (SMX_P32_PREFETCH_AT != 1) load_frame32<ALIGNED>(src, L.l, raw);
    pend_out = tw.oclip + tw.ft * kFT;   // wave-uniform
    const int64_t left = a.count - (int64_t)tw.ft * kFT;
    pend_left = left < kFT ? (int)left : kFT;
    tw.xclip = xnext;
    tw.oclip = onext;
    tw.ft = ftnext;
  }
  if (ntiles > 0) flush_tile((ntiles - 1) & 1, ((unsigned)(ntiles - 1) >> 1) + 1);   // the last tile of this workgroup
#ifdef SMX_STAMPS
  stamp_sum[20] = __builtin_amdgcn_s_memtime() - clk_t0;
  stamp_sum[21] = __builtin_amdgcn_s_memrealtime() - clk_r0;
  if (lane == 0 && blockIdx.x < 4096)
    for (int i = 0; i < kStampSlots; ++i) g_stamp_sums[(blockIdx.x * 16 + wave) * kStampSlots + i] = stamp_sum[i];
#endif

  // Border frames (the few per clip whose window reaches past either end of the signal): same frame code on samples
  // fetched through the padding rule, 16 (clip, frame) pairs per tile, results scattered to their places.
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
    for (int64_t bt = blockIdx.x; bt * kFT < total; bt += gridDim.x) {
      __syncthreads();   // the buffer is free: every wave is past its last flush / the previous border tile
      if (bt * kFT + 2 * wave < total) {   // wave-uniform
        int64_t beta = bt * kFT + 2 * wave + L.h;
        if (beta >= total) beta = bt * kFT + 2 * wave;   // a half without a pair repeats the first one (never stored)
        int64_t clip, p;
        locate(beta, clip, p);
        const float *xs = a.x + clip * a.x_stride;
        const int s0 = (int)(p * a.hop - a.left);
        float2 braw[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          const int s = s0 + 2 * (L.l + 32 * j);
          braw[j] = make_float2(fetch_padded(xs, (int)a.n, s, a.pad, a.pad_value),
                                fetch_padded(xs, (int)a.n, s + 1, a.pad, a.pad_value));
        }
        frame32_to_tile<SQUARE>(a, L, braw, bt_tile, NoMid32{});
      }
      __syncthreads();
      for (int e = tid; e < kBins * kFT; e += 512) {
        const int k = e / kFT, f = e % kFT;
        const int64_t bf = bt * kFT + f;
        if (bf < total) {
          int64_t clip, p;
          locate(bf, clip, p);
          a.out[(clip * kBins + k) * a.out_stride + a.border_out_offset + (p - a.border_p0)] = bt_tile[k * kTileStride + f];
        }
      }
    }
  }
}

// ---- the same pipeline at four waves per SIMD: stft2048_power32g_kernel ---------------------------------------------
// At two waves per SIMD a wave's own instruction stream bounds the tile (one instruction of any kind per ~4.2 cycles and
// wave, every LDS round trip exposed: profiles/r05/stamps32*.log -- 15.5 k cycles per tile, the vector pipe 43 % busy).
// Here the frame pair lives in at most 128 registers (the transposition and both radix-32 passes work in place, tables are
// streamed a few rows at a time, samples are loaded when needed instead of a frame ahead), so a workgroup holds 16 waves:
// two GROUPS of 8, each with ONE tile buffer and its own tile sequence.  A group computes its 16 frames, meets
// (counter `filled`), stores the tile, meets again (`drained`) before the columns are reused; while one group stores or
// waits the other one computes, and every SIMD always has four waves to pick from.
// LDS pointers with their address space in the type, and a way to pin a table read behind a value: hipcc otherwise
// hoists every table read of the frame to its top (and spills them: the frame pair must fit 128 registers here)
typedef __attribute__((address_space(3))) const float2 lds_cf2;
typedef __attribute__((address_space(3))) const float lds_cf;
typedef __attribute__((address_space(3))) float lds_f;
__device__ __forceinline__ lds_cf2 *pin_after(lds_cf2 *p, float dep) {
  asm volatile("" : "+v"(p) : "v"(dep));
  return p;
}
// All 32 complex registers pass through three empty asm statements: nothing computed from them can be scheduled above,
// nothing that feeds them below -- the phases of the frame stay apart (hipcc otherwise overlaps them until the frame
// pair no longer fits its 128 registers).
#define SMX_RB10(V, B) "+v"((V)[B].x), "+v"((V)[B].y), "+v"((V)[B + 1].x), "+v"((V)[B + 1].y), "+v"((V)[B + 2].x), "+v"((V)[B + 2].y), \
    "+v"((V)[B + 3].x), "+v"((V)[B + 3].y), "+v"((V)[B + 4].x), "+v"((V)[B + 4].y), "+v"((V)[B + 5].x), "+v"((V)[B + 5].y),             \
    "+v"((V)[B + 6].x), "+v"((V)[B + 6].y), "+v"((V)[B + 7].x), "+v"((V)[B + 7].y), "+v"((V)[B + 8].x), "+v"((V)[B + 8].y),             \
    "+v"((V)[B + 9].x), "+v"((V)[B + 9].y)
__device__ __forceinline__ void reg_barrier32(c32 (&v)[32]) {
  asm volatile("" : SMX_RB10(v, 0));
  asm volatile("" : SMX_RB10(v, 10));
  asm volatile("" : SMX_RB10(v, 20), "+v"(v[30].x), "+v"(v[30].y), "+v"(v[31].x), "+v"(v[31].y));
}
// a table pointer that becomes usable only once eight complex values exist
__device__ __forceinline__ lds_cf2 *pin_after8(lds_cf2 *p, const c32 *w) {
  asm volatile("" : "+v"(p) : "v"(w[0].x), "v"(w[0].y), "v"(w[1].x), "v"(w[1].y), "v"(w[2].x), "v"(w[2].y), "v"(w[3].x), "v"(w[3].y),
               "v"(w[4].x), "v"(w[4].y), "v"(w[5].x), "v"(w[5].y), "v"(w[6].x), "v"(w[6].y), "v"(w[7].x), "v"(w[7].y));
  return p;
}
struct NoStamp32 {
  template <int I> __device__ __forceinline__ void stamp() const {}
};
#ifdef SMX_STAMPS
struct Stamp32 {
  unsigned long long *stamp_sum, *stamp_prev_p;
  template <int I> __device__ __forceinline__ void stamp() const {
    unsigned long long &stamp_prev = *stamp_prev_p;
    SMX_STAMP(I);
  }
};
#endif
template <bool SQUARE, class St>
__device__ __forceinline__ void frame32g_to_tile(const FastArgs &a, const Lane32 &L, c32 (&v)[32], float *tile,
                                                 unsigned *drained, unsigned drained_target, const St &st) {
#pragma clang fp contract(off)
  // window, eight points at a time
  {
    lds_cf2 *wp = (lds_cf2 *)L.win_l;
#pragma unroll
    for (int j0 = 0; j0 < 32; j0 += 8) {
      float2 win[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { win[j].x = wp[32 * (j0 + j)].x; win[j].y = wp[32 * (j0 + j)].y; }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j0 + j] = {v[j0 + j].x * win[j].x, v[j0 + j].y * win[j].y};
      wp = pin_after8(wp, &v[j0]);
    }
  }
  reg_barrier32(v);
  st.template stamp<1>();
  // A: radix-32 over j, then twiddle W_M^(l k1)
  fft32(v);
  reg_barrier32(v);
  {
    lds_cf2 *tp = pin_after((lds_cf2 *)L.twA_l, v[31].y);
#pragma unroll
    for (int k0 = 0; k0 < 32; k0 += 8) {
      float2 tw[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int kk = k0 + k == 0 ? 1 : k0 + k;
        tw[k].x = tp[32 * kk].x;
        tw[k].y = tp[32 * kk].y;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k0 + k > 0) v[k0 + k] = p32_cmul(v[k0 + k], tw[k].x, tw[k].y);
      tp = pin_after8(tp, &v[k0]);
    }
  }
  reg_barrier32(v);
  st.template stamp<2>();
  // X: lane l register k1 -> lane k1 register l through the frame's column, in place, real parts then imaginary parts
  lds_wait(drained, drained_target);   // the group has stored the tile that was in this buffer
  st.template stamp<3>();
  lds_f *const wr = (lds_f *)(tile + L.own);
  lds_f *const wr_hi = wr + 16 * kCellPitch32;
  lds_cf *const rd = (lds_cf *)(tile + L.rd);
#pragma unroll
  for (int j = 0; j < 32; ++j) (j < 16 ? wr : wr_hi)[kCellPitch32 * (j & 15)] = v[j].x;
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i].x = rd[kTileStride * i];
#pragma unroll
  for (int j = 0; j < 32; ++j) (j < 16 ? wr : wr_hi)[kCellPitch32 * (j & 15)] = v[j].y;
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i].y = rd[kTileStride * i];
  SMX_FENCE();
  // B: radix-32 over l
  fft32(v);
  SMX_FENCE();
  // P: registers 16..31 go to the partner lane through the cells and are dead afterwards (bin M/2 first)
  auto power_of = [&](float re, float im) {
    float pw = __builtin_fmaf(re, re, im * im);
    if constexpr (!SQUARE) pw = a.pmode == 1 ? sqrtf(pw) : __powf(pw, a.half_power);
    return pw;
  };
  const float self_pw = power_of(v[16].x + v[16].x, v[16].y + v[16].y);   // bin M/2 (lane 0): X = 2 conj(Z)
  lds_cf *const xr = (lds_cf *)(tile + L.xr);
  float px[16], py[16];
#pragma unroll
  for (int q = 16; q < 32; ++q) wr[kCellPitch32 * (q - 16)] = v[q].x;
#pragma unroll
  for (int s = 0; s < 16; ++s) px[s] = xr[kCellPitch32 * (15 - s)];
#pragma unroll
  for (int q = 16; q < 32; ++q) wr[kCellPitch32 * (q - 16)] = v[q].y;
#pragma unroll
  for (int s = 0; s < 16; ++s) py[s] = xr[kCellPitch32 * (15 - s)];
  if (L.l == 0) { px[0] = v[0].x; py[0] = v[0].y; }   // bin 0 pairs with itself: X[0] and the Nyquist bin
  st.template stamp<6>();
  SMX_FENCE();
  *(lds_f *)(tile + L.self) = self_pw;
  lds_f *const rk = wr;                         // row l + 32 s
  lds_f *const rm = (lds_f *)(tile + L.rm);     // row (32 - l) + 32 (31 - s) = rm base + 32 (15 - s)
  lds_cf2 *pp = pin_after((lds_cf2 *)L.twP_l, py[15]);
#pragma unroll
  for (int s0 = 0; s0 < 16; s0 += 4) {
    float2 tw[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { tw[s].x = pp[32 * (s0 + s)].x; tw[s].y = pp[32 * (s0 + s)].y; }
    float last = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = s0 + u;
      const c32 e = {v[s].x + px[s], v[s].y - py[s]};
      const c32 d = {v[s].x - px[s], v[s].y + py[s]};
      const float tr = __builtin_fmaf(tw[u].x, d.y, tw[u].y * d.x);
      const float ti = __builtin_fmaf(tw[u].y, d.y, -(tw[u].x * d.x));
      rk[kRowPitch32 * s] = power_of(e.x + tr, e.y + ti);
      rm[kRowPitch32 * (15 - s)] = last = power_of(e.x - tr, e.y - ti);
    }
    pp = pin_after(pp, last);
  }
  st.template stamp<7>();
}

template <bool ALIGNED>
__device__ __forceinline__ void load_frame32c(const float *src, int l, c32 (&v)[32]) {
  float2 raw[32];
  load_frame32<ALIGNED>(src, l, raw);
#pragma unroll
  for (int j = 0; j < 32; ++j) v[j] = {raw[j].x, raw[j].y};
}

template <bool ALIGNED, bool SQUARE, bool STRIP>
__global__ void __launch_bounds__(1024) stft2048_power32g_kernel(FastArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave16 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave16 >> 3, wave = wave16 & 7;   // group, wave inside the group
  const Lds32 lds = carve_lds32(smem);
  const Lane32 L = setup_lane32(lds, lane, wave);
  lds.win[tid] = reinterpret_cast<const float2 *>(a.hwin)[tid];
  if (tid < 31 * 32) lds.twA[tid] = a.w_m[(tid & 31) * ((tid >> 5) + 1)];
  if (tid < 512) lds.twP[tid] = a.w_n[tid];
  if (tid < 2) { lds.filled[tid * kTileStride] = 0u; lds.drained[tid * kTileStride] = 0u; }
  float *const tile = lds.tiles + g * kTile32Floats;
  unsigned *const c_filled = lds.filled + g * kTileStride, *const c_drained = lds.drained + g * kTileStride;
  // this group's tiles: the groups are 2 x blocks virtual workgroups, those of one XCD side by side (TileWalk::init)
  TileWalk tw;
  {
    int64_t tau0;
    int step;
    const int64_t nb = a.blocks, xcd = blockIdx.x % 8, idx = 2 * (blockIdx.x / 8) + g;
    const int64_t q = nb / 8, r = nb % 8;
    if (a.interleave == 2 || a.interleave == 0) {
      tau0 = 2 * (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
      step = (int)(2 * nb);
      tw.ntiles = tau0 < a.total_tiles ? (int)((a.total_tiles - tau0 + 2 * nb - 1) / (2 * nb)) : 0;
    } else {
      const int64_t nx = 2 * ((nb - xcd + 7) / 8);
      const int64_t nxcd = nb < 8 ? nb : 8;
      const int64_t x0 = a.total_tiles * xcd / nxcd, x1 = a.total_tiles * (xcd + 1) / nxcd;
      tau0 = x0 + idx;
      step = (int)nx;
      tw.ntiles = tau0 < x1 ? (int)((x1 - tau0 + nx - 1) / nx) : 0;
    }
    if (tw.ntiles < 0) tw.ntiles = 0;
    tw.ft = (int)(tau0 % a.tiles_per_clip);
    tw.x_step = a.x_stride;
    tw.o_step = kBins * a.out_stride;
    tw.xclip = a.x + (tau0 / a.tiles_per_clip) * tw.x_step;
    tw.oclip = a.out + a.out_offset + (tau0 / a.tiles_per_clip) * tw.o_step;
    tw.step_clips = step / a.tiles_per_clip;
    tw.step_tiles = step % a.tiles_per_clip;
  }
  const int ntiles = tw.ntiles;
  Flush32 fl;
  {
    const int hsel = lane >> 5, jj = (lane & 31) >> 2;
    fl.g = lane & 3;
    fl.row0 = 128 * wave + (jj & 3) + 16 * (jj >> 2) + 4 * hsel;
    fl.goff0 = ((unsigned)fl.row0 * (unsigned)a.out_stride + 4u * fl.g) * 4u;
  }
  __syncthreads();   // tables and zeroed counters visible: the only workgroup barrier of the main loop
#ifdef SMX_STAMPS
  unsigned long long stamp_sum[kStampSlots] = {0}, stamp_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
  const unsigned long long clk_t0 = stamp_prev, clk_r0 = __builtin_amdgcn_s_memrealtime();
  const Stamp32 st{stamp_sum, &stamp_prev};
#else
  const NoStamp32 st{};
#endif

  for (int it = 0; it < ntiles; ++it) {
    const int64_t f0 = (int64_t)tw.ft * kFT;
    const bool have = f0 + 2 * wave < a.count;   // wave-uniform: at least the first half has a frame
    if (have) {
      const int64_t f = f0 + 2 * wave + L.h;
      const int64_t p = a.p0 + (f < a.count ? f : f0 + 2 * wave);
      const float *src = tw.xclip + (p * a.hop - a.left);
      if (a.fold_frames && (p < a.border_i0 || p >= a.border_i1)) {
        const int64_t clip = (tw.xclip - a.x) / a.x_stride;
        src = p < a.border_i0 ? a.strip_l + clip * a.strip_l_stride + (p - a.p0) * a.hop
                              : a.strip_r + clip * a.strip_r_stride + (p - a.border_i1) * a.hop;
      }
      c32 v[32];
      load_frame32c<ALIGNED>(src, L.l, v);
      st.template stamp<0>();
      frame32g_to_tile<SQUARE>(a, L, v, tile, c_drained, 8u * (unsigned)it, st);
    }
    lds_signal(c_filled, lane);
    lds_wait(c_filled, 8u * (unsigned)(it + 1));   // every column of the tile is in
    st.template stamp<8>();
    {
      float *obase = tw.oclip + f0;
      const int64_t left = a.count - f0;
      const int frames_left = left < kFT ? (int)left : kFT;
#pragma unroll
      for (int part = 0; part < 8; ++part) flush32_part(a, tile, part, fl, obase, frames_left, wave, lane);
    }
    st.template stamp<9>();
    lds_signal(c_drained, lane);
    int ftnext;
    const float *xnext;
    float *onext;
    tw.peek(a, ftnext, xnext, onext);
    tw.xclip = xnext;
    tw.oclip = onext;
    tw.ft = ftnext;
    st.template stamp<10>();
  }
#ifdef SMX_STAMPS
  stamp_sum[20] = __builtin_amdgcn_s_memtime() - clk_t0;
  stamp_sum[21] = __builtin_amdgcn_s_memrealtime() - clk_r0;
  if (lane == 0 && blockIdx.x < 4096)
    for (int i = 0; i < kStampSlots; ++i) g_stamp_sums[(blockIdx.x * 16 + wave16) * kStampSlots + i] = stamp_sum[i];
#endif

  // Border frames: same frame code on samples fetched through the padding rule, 32 (clip, frame) pairs per step
  // (16 per group), results scattered to their places.
  if (a.border_left + a.border_right > 0) {
    const int per = a.border_left + a.border_right;
    const int64_t lead = a.total_tiles / a.tiles_per_clip;
    const int64_t total = lead * per;
    auto locate = [&](int64_t beta, int64_t &clip, int64_t &p) {
      clip = beta / per;
      const int r = (int)(beta % per);
      p = r < a.border_left ? a.border_p0 + r : a.border_i1 + (r - a.border_left);
    };
    for (int64_t bt = blockIdx.x; bt * 32 < total; bt += gridDim.x) {
      __syncthreads();   // both buffers are free
      const int64_t base = bt * 32 + 16 * g;
      if (base + 2 * wave < total) {   // wave-uniform
        int64_t beta = base + 2 * wave + L.h;
        if (beta >= total) beta = base + 2 * wave;   // a half without a pair repeats the first one (never stored)
        int64_t clip, p;
        locate(beta, clip, p);
        const float *xs = a.x + clip * a.x_stride;
        const int s0 = (int)(p * a.hop - a.left);
        c32 v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          const int s = s0 + 2 * (L.l + 32 * j);
          v[j] = {fetch_padded(xs, (int)a.n, s, a.pad, a.pad_value), fetch_padded(xs, (int)a.n, s + 1, a.pad, a.pad_value)};
        }
        frame32g_to_tile<SQUARE>(a, L, v, tile, c_drained, 0u, NoStamp32{});
      }
      __syncthreads();
      for (int e = tid; e < 2 * kBins * kFT; e += 1024) {
        const int gg = e / (kBins * kFT), ee = e % (kBins * kFT);
        const int k = ee / kFT, f = ee % kFT;
        const int64_t bf = bt * 32 + 16 * gg + f;
        if (bf < total) {
          int64_t clip, p;
          locate(bf, clip, p);
          a.out[(clip * kBins + k) * a.out_stride + a.border_out_offset + (p - a.border_p0)] =
              lds.tiles[gg * kTile32Floats + k * kTileStride + f];
        }
      }
    }
  }
}
