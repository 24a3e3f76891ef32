// stft4096_power64_kernel -- the power spectrogram at fft 4096 on the register frame pipeline, a frame in a WHOLE WAVE (round 6;
// included by stft_fast.hip inside its anonymous namespace, after stft_fast_p32.hpp).  Replaces the reference's hot call for
// Stft.power_spectrum at that size (stft.ml:356-364 + 670-691; callers: cqt.ml:648, hpss.ml:490-492).  Before: the Stockham kernel,
// one 8-frame tile per workgroup, eleven barrier-separated passes: 0.15-0.16 of the HBM roof.
//
// A frame of N = 4096 samples is M = 2048 complex points z[m] = (x[2m], x[2m+1]) x window.  Decimation in time over the two halves
// of the wave: half h holds the subsequence z[2 n' + h], n' < 1024, lane l its points n' = l + 32 j (so a wave's load instruction j
// covers 512 contiguous bytes), and transforms it exactly as stft_fast_p32.hpp transforms a 2048-sample frame: 1024 = 32 x 32,
// radix-32 in registers, W_1024^(l k1), one 32 x 32 transposition through the frame's own column of the output tile, radix-32 in
// registers -> E[k'] (half 0) / O[k'] (half 1), lane k1, register q: k' = k1 + 32 q.  Then
//   Z[k'] = E[k'] + W_2048^k' O[k']   (half 0),      Z[1024 + k'] = E[k'] - W_2048^k' O[k']   (half 1):
// each half multiplies its own values by F = 1 / W_2048^k' (a table per half: uniform code), the halves exchange them through the
// column's cells, and Z = +-A + B is one packed fused multiply-add per value with a per-half sign.  The real-FFT post-pass pairs bin
// k with 2048 - k: lane l, register q of one half against lane (32 - l) mod 32, register 31 - q of the OTHER half (lanes 0 pair
// with each other one register further; bins 0 and 1024 pair with themselves): every pair is formed once, by its q < 16 member,
// with the generated blocks of the 32-lane pipeline (pk_post_power*), and its two values |X_k|^p, |X_(2048-k)|^p go into the
// frame's column of the tile.
// A workgroup is 8 waves = 8 consecutive frames of one clip = one tile of 2049 rows x 8 frames (pitch 9 floats; 2112 rows: the two
// halves' 1056 transposition cells each); with 2 x 76 KB of tile and 48 KB of tables the LDS holds ONE tile, so the tile is
// single-buffered the way stft2048_complex32_kernel is: a wave reads its share of the previous tile out in the middle of the next
// frame's first radix-32 (its stores run under the rest of the frame) and nobody writes a cell before every wave has done so.
// Persistent workgroups walking the tile sequence side by side (TileWalk's interleaved orders: see the kernel), the next frame's samples
// requested at the end of the current one.
// LDS: 76,032 (tile) + 16,384 (window) + 7,936 (W_1024^(l k1)) + 16,384 (F per half) + 8,192 (post-pass twiddles) = 124,928 B.
// The arithmetic is written out operation by operation (packed stages, explicit fused multiply-adds): a frame has ONE value wherever
// and however it is computed (range tiling, slices, streaming partitions: stft_grid.ml:32-73,180-205, stft_law.ml:79-164).

constexpr int k64N = 4096, k64M = 2048, k64Bins = 2049;
constexpr int k64FT = 8;                       // frames per tile: one per wave
constexpr int k64TS = k64FT + 1;               // floats per tile row (pad column 8)
constexpr int k64Rows = 2112;                  // two cell blocks of 1056 rows (>= 2049 bins)
constexpr int k64TileFloats = k64Rows * k64TS;
constexpr size_t k64TileBytes = (size_t)k64TileFloats * sizeof(float);      // 76,032
constexpr size_t k64WinBytes = 16 * 64 * sizeof(float4);                    // window pairs of points j = 2 r, 2 r + 1 per lane' = 2 l + h
constexpr size_t k64TwABytes = 31 * 32 * sizeof(float2);                    // W_1024^(l k1)
constexpr size_t k64TwCBytes = 2 * 16 * 32 * sizeof(float4);                // F[h][q pair][l]: 1 (half 0), W_2048^(l + 32 q) (half 1)
constexpr size_t k64TwPBytes = 2 * 8 * 32 * sizeof(float4);                 // exp(-2 pi i k / 4096), k = l + 32 q + 1024 h, q < 16
constexpr size_t k64Lds = k64TileBytes + k64WinBytes + k64TwABytes + k64TwCBytes + k64TwPBytes;
static_assert(k64Lds <= 160 * 1024, "LDS budget");

struct Lds64 {
  float *tile;
  float4 *win4, *twA4, *twC4, *twP4;
  float2 *twA31;
  unsigned *filled, *drained;   // pad cells of rows 2050 / 2052
};
__device__ __forceinline__ Lds64 carve_lds64(unsigned char *smem) {
  Lds64 l;
  l.tile = reinterpret_cast<float *>(smem);
  unsigned char *p = smem + k64TileBytes;
  l.win4 = reinterpret_cast<float4 *>(p);
  p += k64WinBytes;
  l.twA4 = reinterpret_cast<float4 *>(p);
  l.twA31 = reinterpret_cast<float2 *>(p + 15 * 32 * sizeof(float4));
  p += k64TwABytes;
  l.twC4 = reinterpret_cast<float4 *>(p);
  p += k64TwCBytes;
  l.twP4 = reinterpret_cast<float4 *>(p);
  l.filled = reinterpret_cast<unsigned *>(l.tile + 2050 * k64TS + k64FT);
  l.drained = reinterpret_cast<unsigned *>(l.tile + 2052 * k64TS + k64FT);
  return l;
}

// the workgroup's tables (512 threads); the caller synchronises before they are read
__device__ __forceinline__ void fill_tables64(const FastArgs &a, const Lds64 &lds, int tid) {
  const float2 *hw = reinterpret_cast<const float2 *>(a.hwin);   // (0.5 w[2 m], 0.5 w[2 m + 1]), m < 2048
  for (int i = tid; i < 16 * 64; i += 512) {   // win4[r][lane'], lane' = 2 l + h: points m = lane' + 128 r and + 64
    const int r = i >> 6, lp = i & 63;
    const float2 w0 = hw[lp + 128 * r], w1 = hw[lp + 128 * r + 64];
    lds.win4[i] = make_float4(w0.x, w0.y, w1.x, w1.y);
  }
  for (int i = tid; i < 15 * 32; i += 512) {   // twA4[m][l] = W_1024^(l (2 m + 1)), W_1024^(l (2 m + 2)); a.w_m = exp(-2 pi i j / 2048)
    const int m = i >> 5, l = i & 31;
    const float2 w0 = a.w_m[(2 * l * (2 * m + 1)) & 2047], w1 = a.w_m[(2 * l * (2 * m + 2)) & 2047];
    lds.twA4[i] = make_float4(w0.x, w0.y, w1.x, w1.y);
  }
  if (tid < 32) lds.twA31[tid] = a.w_m[(2 * tid * 31) & 2047];
  for (int i = tid; i < 2 * 16 * 32; i += 512) {   // twC4[h][m][l]: q = 2 m, 2 m + 1
    const int h = i >> 9, m = (i >> 5) & 15, l = i & 31;
    const float2 w0 = a.w_m[l + 64 * m], w1 = a.w_m[l + 64 * m + 32];
    lds.twC4[i] = h ? make_float4(w0.x, w0.y, w1.x, w1.y) : make_float4(1.0f, 0.0f, 1.0f, 0.0f);
  }
  for (int i = tid; i < 2 * 8 * 32; i += 512) {   // twP4[h][m][l]: q = 2 m, 2 m + 1 (q < 16), k = l + 32 q + 1024 h
    const int h = i >> 8, m = (i >> 5) & 7, l = i & 31;
    const float2 w0 = a.w_n[l + 64 * m + 1024 * h], w1 = a.w_n[l + 64 * m + 32 + 1024 * h];
    lds.twP4[i] = make_float4(w0.x, w0.y, w1.x, w1.y);
  }
  if (tid < 2) { (tid ? lds.drained : lds.filled)[0] = 0u; }
}

// raw samples of the lane's subsequence: z[2 (l + 32 j) + h], i.e. float2 index lp + 64 j with lp = 2 l + h
template <bool ALIGNED>
__device__ __forceinline__ void load_frame64(const float *src, int lp, float2 (&raw)[32]) {
  // (four bases, 8 x 512 bytes apart: the instruction's immediate reaches 4095 bytes; the scalar offsets are opaque so that they stay scalars)
  long o1 = 4096, o2 = 8192, o3 = 12288;
  asm volatile("" : "+s"(o1), "+s"(o2), "+s"(o3));
  if constexpr (ALIGNED) {
    const char *p0 = reinterpret_cast<const char *>(reinterpret_cast<const float2 *>(src) + lp);
    const char *pb[4] = {p0, p0 + o1, p0 + o2, p0 + o3};
#pragma unroll
    for (int j = 0; j < 32; ++j) raw[j] = *reinterpret_cast<const float2 *>(pb[j >> 3] + 512 * (j & 7));
  } else {
    const char *p0 = reinterpret_cast<const char *>(src + 2 * lp);
    const char *pb[4] = {p0, p0 + o1, p0 + o2, p0 + o3};
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const float *b = reinterpret_cast<const float *>(pb[j >> 3] + 512 * (j & 7));
      raw[j] = make_float2(b[0], b[1]);
    }
  }
}

// A wave's share of a finished tile: rows 256 wave .. + 255 as 16 parts of 16 rows x 4 frame pairs.  Per part a half-wave takes
// rows {0, 1, 8, 9, 16, 17, 24, 25} (+ 2 for the upper half) of a 32-row group: with the row pitch of 9 floats the 32 lanes of a
// half-wave then read 32 different banks; part p: group p >> 1, + 4 rows for odd p.
struct Flush64 {
  int src0;          // float offset of (row0, frame 2 g) in the tile
  unsigned goff0;    // byte offset of out[row0][2 g] from the tile's origin
  int g;             // frames 2 g, 2 g + 1
};
__device__ __forceinline__ Flush64 setup_flush64(const FastArgs &a, int lane, int wave) {
  Flush64 fl;
  const int half = lane >> 5, ridx = (lane & 31) >> 2;   // ridx 0..7 -> row {0, 1, 8, 9, 16, 17, 24, 25}
  fl.g = lane & 3;
  const int row0 = 256 * wave + (ridx & 1) + 8 * (ridx >> 1) + 2 * half;
  fl.src0 = row0 * k64TS + 2 * fl.g;
  fl.goff0 = ((unsigned)row0 * (unsigned)a.out_stride + 2u * fl.g) * 4u;
  return fl;
}
struct Flush64Regs {
  float2 v[16];
  float nyq;
};
__device__ __forceinline__ constexpr int flush64_rows(int p) { return 32 * (p >> 1) + 4 * (p & 1); }
__device__ __forceinline__ void flush64_read(const float *tile, const Flush64 &fl, int lane, Flush64Regs &r) {
  const float *src0 = tile + opaque32(fl.src0);
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const float *src = src0 + flush64_rows(p) * k64TS;
    r.v[p] = make_float2(src[0], src[1]);
  }
  r.nyq = tile[2048 * k64TS + (lane & 7)];   // row 2048 (every wave reads it, wave 0 stores it)
}
template <bool EVEN>
__device__ __forceinline__ void flush64_store(const FastArgs &a, const Flush64 &fl, float *obase, int frames_left, int wave, int lane,
                                              const Flush64Regs &r) {
  const unsigned pitch = (unsigned)a.out_stride * 4u, goff0 = opaque32(fl.goff0);
  const int fleft = frames_left - 2 * fl.g;
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    float *dst = reinterpret_cast<float *>(reinterpret_cast<char *>(obase) + (goff0 + (unsigned)flush64_rows(p) * pitch));
    if (fleft >= 2) {
      if constexpr (EVEN) *reinterpret_cast<float2 *>(dst) = r.v[p];   // (8-byte aligned: even pitch and origin)
      else { dst[0] = r.v[p].x; dst[1] = r.v[p].y; }
    } else if (fleft == 1) {
      dst[0] = r.v[p].x;
    }
  }
  if (wave == 0 && lane < 8 && lane < frames_left) obase[(int64_t)2048 * a.out_stride + lane] = r.nyq;
}

// One frame: raw samples (registers) -> |X|^p in the frame's column `col` of the tile.  `early()` runs between the two 16-point
// transforms of the first radix-32 (the previous tile is read out there), `before_cells()` before the column's first write.
template <int PMODE, class Mid>
__device__ __forceinline__ void frame64_to_tile(const FastArgs &a, const Lds64 &lds, int lane, int col, float2 (&raw)[32], const Mid &mid) {
#pragma clang fp contract(off)
  int lv = lane;
  asm volatile("" : "+v"(lv));
  const int l = lv & 31, h = lv >> 5;
  f2 v[32], t[32];
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the samples were requested a frame ago
  {   // window
    const float4 *wq = lds.win4 + (2 * l + h);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 w = wq[64 * r];
      v[2 * r] = f2{raw[2 * r].x, raw[2 * r].y} * f2{w.x, w.y};
      v[2 * r + 1] = f2{raw[2 * r + 1].x, raw[2 * r + 1].y} * f2{w.z, w.w};
    }
  }
  SMX_FENCE();
  {
    float4 tw[15];
#pragma unroll
    for (int m = 0; m < 15; ++m) tw[m] = lds.twA4[32 * m + l];
    const float2 tw31 = lds.twA31[l];
    f2 e[16], o[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) { e[m] = v[2 * m]; o[m] = v[2 * m + 1]; }
    SMX_FENCE();
    pk_fft16(e);
    SMX_FENCE(); mid.early(); SMX_FENCE();
    pk_fft16(o);
    pk_fft32_combine0(v, e, o);
    pk_fft32_combine1(v, e, o);
    SMX_FENCE();
#define SMX_TWV(m) f2{tw[m].x, tw[m].y}, f2{tw[m].z, tw[m].w}
    pk_twiddle8(v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], SMX_TWV(0), SMX_TWV(1), SMX_TWV(2), SMX_TWV(3));
    pk_twiddle8(v[9], v[10], v[11], v[12], v[13], v[14], v[15], v[16], SMX_TWV(4), SMX_TWV(5), SMX_TWV(6), SMX_TWV(7));
    pk_twiddle8(v[17], v[18], v[19], v[20], v[21], v[22], v[23], v[24], SMX_TWV(8), SMX_TWV(9), SMX_TWV(10), SMX_TWV(11));
    pk_twiddle7(v[25], v[26], v[27], v[28], v[29], v[30], v[31], SMX_TWV(12), SMX_TWV(13), SMX_TWV(14), f2{tw31.x, tw31.y});
#undef SMX_TWV
  }
  SMX_FENCE();
  mid.before_cells();
  // cells of this half: rows 1056 h + c of the frame's column (pitch 9 floats: a half-wave's 32 cells 33 l + j / l + 33 i lie in 32 banks)
  float *cb = lds.tile + (1056 * h) * k64TS + col;
  float *ob = lds.tile + (1056 * (1 - h)) * k64TS + col;   // the other half's cells
  {   // transposition: lane l register k1 -> lane k1 register l; real parts, then imaginary parts (a wave's LDS operations run in order)
    float *wc = cb + 33 * k64TS * l, *rc = cb + k64TS * l;
    float *rc_hi = rc + 16 * 33 * k64TS;   // (ds offsets are 16 bits)
#pragma unroll
    for (int j = 0; j < 32; ++j) wc[k64TS * j] = v[j].x;
#pragma unroll
    for (int i = 0; i < 32; ++i) t[i].x = (i < 16 ? rc : rc_hi)[33 * k64TS * (i & 15)];
#pragma unroll
    for (int j = 0; j < 32; ++j) wc[k64TS * j] = v[j].y;
#pragma unroll
    for (int i = 0; i < 32; ++i) t[i].y = (i < 16 ? rc : rc_hi)[33 * k64TS * (i & 15)];
  }
  SMX_FENCE();
  {   // second radix-32: E (half 0) / O (half 1) at k' = l + 32 q
    f2 e[16], o[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) { e[m] = t[2 * m]; o[m] = t[2 * m + 1]; }
    pk_fft16(e);
    pk_fft16(o);
    pk_fft32_combine0(t, e, o);
    pk_fft32_combine1(t, e, o);
  }
  SMX_FENCE();
  {   // A = F x own (F = 1 / W_2048^k' per half), exchanged between the halves: cell 33 l + q of the own block, read from the other's
    float4 fc[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) fc[m] = lds.twC4[512 * h + 32 * m + l];
#define SMX_FC(m) f2{fc[m].x, fc[m].y}, f2{fc[m].z, fc[m].w}
    pk_twiddle8(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], SMX_FC(0), SMX_FC(1), SMX_FC(2), SMX_FC(3));
    pk_twiddle8(t[8], t[9], t[10], t[11], t[12], t[13], t[14], t[15], SMX_FC(4), SMX_FC(5), SMX_FC(6), SMX_FC(7));
    pk_twiddle8(t[16], t[17], t[18], t[19], t[20], t[21], t[22], t[23], SMX_FC(8), SMX_FC(9), SMX_FC(10), SMX_FC(11));
    pk_twiddle8(t[24], t[25], t[26], t[27], t[28], t[29], t[30], t[31], SMX_FC(12), SMX_FC(13), SMX_FC(14), SMX_FC(15));
#undef SMX_FC
    float *wx = cb + 33 * k64TS * l, *rx = ob + 33 * k64TS * l;
#pragma unroll
    for (int q = 0; q < 32; ++q) wx[k64TS * q] = t[q].x;
#pragma unroll
    for (int q = 0; q < 32; ++q) v[q].x = rx[k64TS * q];
#pragma unroll
    for (int q = 0; q < 32; ++q) wx[k64TS * q] = t[q].y;
#pragma unroll
    for (int q = 0; q < 32; ++q) v[q].y = rx[k64TS * q];
    // Z = sign x A + B: half 0 Z[k'] = E + W O (own + other), half 1 Z[1024 + k'] = E - W O (other - own)
    const float sg = h ? -1.0f : 1.0f;
    const f2 sgn = {sg, sg};
#pragma unroll
    for (int q = 0; q < 32; ++q) t[q] = __builtin_elementwise_fma(t[q], sgn, v[q]);
  }
  SMX_FENCE();
  // post-pass pairs (k, 2048 - k): the q < 16 member forms the pair.  Registers 16 .. 31 are parked in cells 33 l + (q - 16) of the own
  // block, register 0 in cell 33 l + 16; the partner's register 31 - q is cell 33 pl + (15 - q) of the OTHER block, pl = (32 - l) mod 32
  // (lanes 0: one cell further -- register 32 - q; q = 0 pairs with itself: bins 0 / 2048 in half 0, bin 1024 in half 1)
  // Bin 512 (half 0, lane 0, register 16) pairs with bin 1536 (half 1, lane 0, register 16): both members have q = 16, so the pair
  // is formed as a seventeenth slot by every lane (uniform code) and stored by lane 0 of half 0.
  f2 pp[16], px;
  {
    float *wx = cb + 33 * k64TS * l;
    const int pl = (32 - l) & 31;
    const float *rx = ob + (33 * pl + (l == 0 ? 1 : 0)) * k64TS;
#pragma unroll
    for (int q = 16; q < 32; ++q) wx[k64TS * (q - 16)] = t[q].x;
#pragma unroll
    for (int s = 0; s < 16; ++s) pp[s].x = rx[k64TS * (15 - s)];
    px.x = ob[0];   // the other half's lane 0, register 16 (cell 0 of its block)
#pragma unroll
    for (int q = 16; q < 32; ++q) wx[k64TS * (q - 16)] = t[q].y;
#pragma unroll
    for (int s = 0; s < 16; ++s) pp[s].y = rx[k64TS * (15 - s)];
    px.y = ob[0];
    if (l == 0) pp[0] = t[0];   // the lane-0 rule above reads cell 16 for q = 0: nobody's; bins 0 and 1024 pair with themselves
  }
  const float2 wx512 = a.w_n[512];
  float4 tw[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) tw[m] = lds.twP4[256 * h + 32 * m + l];
  SMX_FENCE();
  // rows: bin k = l + 32 s + 1024 h, bin 2048 - k
  float *rk = lds.tile + (l + 1024 * h) * k64TS + col;            // + 32 s rows
  float *rm = lds.tile + (2048 - l - 1024 * h) * k64TS + col;     // - 32 s rows
  auto wtw = [&](int s) { return (s & 1) ? f2{tw[s >> 1].z, tw[s >> 1].w} : f2{tw[s >> 1].x, tw[s >> 1].y}; };
  auto put = [&](int s, f2 pw) {
    rk[32 * k64TS * s] = power_from_square<PMODE>(pw.x, a);
    rm[-32 * k64TS * s] = power_from_square<PMODE>(pw.y, a);
  };
#define SMX_PA(s) t[s], pp[s], wtw(s)
  f2 r[5];
  pk_post_power2(SMX_PA(0), SMX_PA(1), r[0], r[1]);
  put(0, r[0]); put(1, r[1]);
  pk_post_power4(SMX_PA(2), SMX_PA(3), SMX_PA(4), SMX_PA(5), r[0], r[1], r[2], r[3]);
#pragma unroll
  for (int i = 0; i < 4; ++i) put(2 + i, r[i]);
  SMX_FENCE(); mid.load_next(); SMX_FENCE();
  pk_post_power5(SMX_PA(6), SMX_PA(7), SMX_PA(8), SMX_PA(9), SMX_PA(10), r[0], r[1], r[2], r[3], r[4]);
#pragma unroll
  for (int i = 0; i < 5; ++i) put(6 + i, r[i]);
  pk_post_power5(SMX_PA(11), SMX_PA(12), SMX_PA(13), SMX_PA(14), t[16], px, f2{wx512.x, wx512.y}, r[0], r[1], r[2], r[3], r[4]);
#pragma unroll
  for (int i = 0; i < 4; ++i) put(11 + i, r[i]);
  if (lv == 0) {   // (lane 0 of half 0: bins 512 and 1536)
    lds.tile[512 * k64TS + col] = power_from_square<PMODE>(r[4].x, a);
    lds.tile[1536 * k64TS + col] = power_from_square<PMODE>(r[4].y, a);
  }
  pk_post_power2(SMX_PA(15), SMX_PA(15), r[0], r[1]);   // (slot 15; a block of one slot does not exist: formed twice, stored once)
  put(15, r[0]);
#undef SMX_PA
}

template <bool ALIGNED, bool EVEN>
struct PowerMid64 {
  const FastArgs &a;
  const Lds64 &lds;
  const Flush64 &fl;
  float2 (&raw)[32];
  const float *src;
  float *pend_out;
  int pend_left;
  int lane, wave, it;
  __device__ __forceinline__ void early() const {   // between the two 16-point transforms of the first radix-32
    if (it > 0) {
      lds_wait(lds.filled, 8u * (unsigned)it);       // every wave's column of the previous tile is in
      Flush64Regs fr;
      flush64_read(lds.tile, fl, lane, fr);
      lds_signal32(lds.drained, lane);               // behind this wave's reads in LDS order
      flush64_store<EVEN>(a, fl, pend_out, pend_left, wave, lane, fr);
    }
  }
  __device__ __forceinline__ void before_cells() const {
    if (it > 0) lds_wait(lds.drained, 8u * (unsigned)it);   // every wave has read the previous tile out
  }
  __device__ __forceinline__ void load_next() const { load_frame64<ALIGNED>(src, 2 * (lane & 31) + (lane >> 5), raw); }
};

// EVEN: even row pitch and origin (8-byte stores of a frame pair).  Tile order: the workgroups walk the flat (clip, tile) sequence SIDE BY
// SIDE (TileWalk's interleaved orders), not contiguous ranges: a tile's row run is only 32 bytes, and a 128-byte line of the result is
// complete after FOUR consecutive tiles -- with contiguous ranges those are 60 us apart in one workgroup and 67 MB of lines stand open
// chip-wide against 32 MB of L2 (0.74 ms; aligned-sector carries helped 5 %); side by side the 32 workgroups of an XCD write 32
// neighbouring tiles at once, 1 KB per row, and its L2 assembles whole lines: 0.54 ms (profiles/r08/ab_p64_interleave.log).
template <bool ALIGNED, int PMODE, bool EVEN>
__global__ void __launch_bounds__(512) stft4096_power64_kernel(FastArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Lds64 lds = carve_lds64(smem);
  fill_tables64(a, lds, tid);
  TileWalk tw;
  tw.init(a, a.out + a.out_offset, (int64_t)k64Bins * a.out_stride);
  const int ntiles = tw.ntiles > 0 ? tw.ntiles : 0;
  // first sample of this wave's frame in tile t of the clip at xc (a wave without a frame re-reads the tile's first frame; its column is
  // never stored); fold_frames == 1: a frame that touches a border of the signal comes from a gathered, already padded strip
  auto frame_ptr = [&](const float *xc, int t) {
    const int64_t f0 = (int64_t)t * k64FT;
    const int avail = (int)(a.count - f0 < k64FT ? a.count - f0 : k64FT) - 1;
    const int64_t p = a.p0 + f0 + (wave <= avail ? wave : 0);
    if (a.fold_frames == 1 && (p < a.border_i0 || p >= a.border_i1)) {
      const int64_t clip = (xc - a.x) / a.x_stride;
      return p < a.border_i0 ? a.strip_l + clip * a.strip_l_stride + (p - a.p0) * a.hop
                             : a.strip_r + clip * a.strip_r_stride + (p - a.border_i1) * a.hop;
    }
    return xc + (p * a.hop - a.left);
  };
  float2 raw[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) raw[j] = make_float2(0.f, 0.f);
  if (ntiles > 0) load_frame64<ALIGNED>(frame_ptr(tw.xclip, tw.ft), 2 * (lane & 31) + (lane >> 5), raw);
  __syncthreads();   // tables and zeroed counters
  const Flush64 fl = setup_flush64(a, lane, wave);
  float *pend_out = nullptr;
  int pend_left = 0;
  for (int it = 0; it < ntiles; ++it) {
    int ftnext;
    const float *xnext;
    float *onext;
    tw.peek(a, ftnext, xnext, onext);
    const bool more = it + 1 < ntiles;
    const float *src = frame_ptr(more ? xnext : tw.xclip, more ? ftnext : tw.ft);
    const PowerMid64<ALIGNED, EVEN> mid{a, lds, fl, raw, src, pend_out, pend_left, lane, wave, it};
    frame64_to_tile<PMODE>(a, lds, lane, wave, raw, mid);
    lds_signal32(lds.filled, lane);
    pend_out = tw.oclip + tw.ft * k64FT;
    const int64_t left = a.count - (int64_t)tw.ft * k64FT;
    pend_left = left < k64FT ? (int)left : k64FT;
    tw.xclip = xnext;
    tw.oclip = onext;
    tw.ft = ftnext;
  }
  if (ntiles > 0) {
    lds_wait(lds.filled, 8u * (unsigned)ntiles);
    Flush64Regs fr;
    flush64_read(lds.tile, fl, lane, fr);
    flush64_store<EVEN>(a, fl, pend_out, pend_left, wave, lane, fr);
  }
}
