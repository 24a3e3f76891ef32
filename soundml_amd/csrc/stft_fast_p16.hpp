// stft_power_lanes_kernel<LL> -- the power spectrogram at fft 1024 (LL = 16), fft 512 (LL = 8) and fft 256 (LL = 4) on the register frame pipeline
// of stft_fast_p32.hpp with a frame in LL lanes (included by stft_fast.hip after stft_fast_p32.hpp, inside its anonymous
// namespace).  Replaces the reference's hot call for Stft.power_spectrum at BASELINE C1's geometry (fft 1024 / hop 256) and
// at fft 512, stft.ml:356-364 + 670-691.
//
// M = N/2 = 32 LL complex points, z[n] = x[2n] + i x[2n+1]:
//   A. n = l + LL j  : radix-32 over j in registers (fft32 of the 32-lane pipeline)       -> y_l[k1], twiddle W_M^(l k1)
//   X. LL x 32 transposition through the frame's own column of the output tile: lane l register k1 -> cell l + (LL+1) k1 ->
//      lane k1 mod LL, registers (k1 div LL, l); one plane at a time, base(lane) + immediate on both sides
//   B. 32 / LL radix-LL transforms over l in registers (k1 = lam + LL a)                    -> lane lam, register u: Z[lam + LL u]
//   P. real-FFT post-pass, one slot per pair (k, M - k): lane lam owns slots s = 0..15 (k = lam + LL s, its registers 0..15
//      against registers 31..16 of lane LL - lam, fetched through the cells); lanes 0 and LL/2 pair inside themselves, slot
//      k = 0 yields X[0] and the Nyquist bin, bin M/2 (lane 0, register 16) is one extra product.
// A wave carries 64 / LL frames, a workgroup is 8 waves = one tile [rows = bins][FT frames + pad], FT = 32 (64), FT + 1 floats
// per row (= 1 mod 32: bank = row + column).  The frames of a 32-lane half sit LL columns apart, which keeps every
// transposition and exchange access conflict free at a cell pitch of LL + 1 rows.  Per frame the instruction stream is the
// 32-lane kernel's with the second radix-32 replaced by radix-LL transforms; every function rounds operation by operation
// (contraction off), so a frame gets the same bits wherever it sits.
// LDS (LL = 16): 2 x 71,808 (tiles of 544 rows x 33) + 4,096 (window) + 3,968 (W_M^(l k1)) + 2,048 (post-pass twiddles);
// (LL = 8): 2 x 74,880 (288 rows x 65) + half the tables; (LL = 4): 2 x 68,112 (132 rows x 129: the transposition in two passes of 16
// cells, see PL) + a quarter of the tables.  The counters sit in the pad column.

constexpr int kN16 = 1024, kN8 = 512, kN4 = 256;
template <int LL>
struct PL {
  static constexpr int N = 64 * LL, M = 32 * LL, Bins = M + 1;
  static constexpr int FT = 8 * (64 / LL);               // frames per tile
  static constexpr int TS = FT + 1;                      // floats per tile row
  static constexpr int CP = LL + 1;                      // rows between cells c and c + 1 of a lane
  // LL = 4 (fft 256, round 5): the transposition runs in two passes of 16 cells (registers 0..15, then 16..31 through the same
  // cells: a wave's LDS accesses complete in order), so that the cells stay inside the tile's 129 rows -- with 32 cells of pitch
  // 5 a buffer would be 160 rows x 129 floats and two of them more than the CU's LDS.
  static constexpr int CellsPerPass = LL == 4 ? 16 : 32;
  static constexpr int Rows = LL == 4 ? 132 : ((LL - 1 + CP * 31 + 1 + 15) / 16) * 16;
  static constexpr int TileFloats = Rows * TS;
  static constexpr size_t TileBytes = (size_t)TileFloats * sizeof(float);
  static constexpr size_t WinBytes = 16 * LL * sizeof(float4);
  static constexpr size_t TwABytes = 15 * LL * sizeof(float4) + LL * sizeof(float2);
  static constexpr size_t TwPBytes = 8 * LL * sizeof(float4);
  static constexpr size_t Lds = 2 * TileBytes + WinBytes + TwABytes + TwPBytes;
  static constexpr int CellPitch = CP * TS;              // floats between cells c and c + CP
  static constexpr int RowPitch = LL * TS;               // floats between rows r and r + LL
  static constexpr int CounterRow = LL == 4 ? 0 : M + 16;   // (the pad column of four rows)
  static_assert(Lds <= 160 * 1024, "LDS budget");
  static_assert(Rows > CounterRow + 3 && Rows >= Bins && Rows > LL - 1 + CP * (CellsPerPass == 32 ? 31 : 16), "counter rows, cells");
};
template <int LL>
struct LdsL {
  float *tiles;
  // tables, read two complex values (16 bytes) per lane and instruction:
  //   win4[m][l] = window pairs of points l + LL (2m), l + LL (2m + 1)                     m < 16
  //   twA4[m][l] = W_M^(l k1) for k1 = 2m + 1, 2m + 2 (m < 15), then one row of k1 = 31
  //   twP4[m][l] = exp(-2 pi i k / N) for k = l + LL (2m), l + LL (2m + 1)                 m < 8
  float4 *win4, *twA4, *twP4;
  float2 *twA31;
  unsigned *filled, *drained;   // [2] each, TS floats apart
};
template <int LL>
__device__ __forceinline__ LdsL<LL> carve_ldsL(unsigned char *smem) {
  using P = PL<LL>;
  LdsL<LL> l;
  l.tiles = reinterpret_cast<float *>(smem);
  l.win4 = reinterpret_cast<float4 *>(smem + 2 * P::TileBytes);
  l.twA4 = reinterpret_cast<float4 *>(smem + 2 * P::TileBytes + P::WinBytes);
  l.twA31 = reinterpret_cast<float2 *>(smem + 2 * P::TileBytes + P::WinBytes + 15 * LL * sizeof(float4));
  l.twP4 = reinterpret_cast<float4 *>(smem + 2 * P::TileBytes + P::WinBytes + P::TwABytes);
  l.filled = reinterpret_cast<unsigned *>(l.tiles + P::CounterRow * P::TS + P::FT);
  l.drained = reinterpret_cast<unsigned *>(l.tiles + (P::CounterRow + 2) * P::TS + P::FT);
  return l;
}

// the column (= frame of the tile) of lane `lane` of wave w: the 32 / LL frames of a 32-lane half sit LL columns apart
// (LL = 16: w + 8 h + 16 r; LL = 8: w + 8 r + 32 h; h = lane half, r = frame inside the half)
template <int LL>
__device__ __forceinline__ int columnL(int wave, int lane) {
  const int h = lane >> 5, r = (lane & 31) / LL;
  if constexpr (LL == 4) return 4 * r + (wave & 3) + 32 * (2 * h + (wave >> 2));   // (8 frames a half, 4 columns apart)
  return wave + 8 * (LL == 16 ? h + 2 * r : r + 4 * h);
}

struct LaneL {
  int l, col;
  int own;        // cell l of the frame's column: transposition / exchange writes (cell l + CP j), results of bins l + LL s
  int rd;         // cell CP l: transposition reads (cell i + CP (l + LL a))
  int xr;         // exchange reads: cell p + CP (15 - s) of slot s, p = LL - l (l = 0: CP, i.e. its own register 32 - s)
  int rm;         // results of bins M - k: row (LL - l) + LL (31 - s)  (l = 0: LL (32 - s); s = 0 is row M = Nyquist)
  int self;       // lane 0: row M/2; other lanes: a cell they overwrite afterwards
  const float4 *win_l, *twA_l, *twP_l;
  const float2 *twA31_l;
};
template <int LL>
__device__ __forceinline__ LaneL setup_laneL(const LdsL<LL> &lds, int lane, int wave) {
  using P = PL<LL>;
  LaneL L;
  L.l = lane & (LL - 1);
  L.col = columnL<LL>(wave, lane);
  L.own = L.l * P::TS + L.col;
  L.rd = P::CP * L.l * P::TS + L.col;
  L.xr = (L.l == 0 ? P::CP : LL - L.l) * P::TS + L.col;
  L.rm = ((L.l == 0 ? LL : LL - L.l) + LL * 16) * P::TS + L.col;
  L.self = (L.l == 0 ? P::M / 2 : L.l) * P::TS + L.col;
  L.win_l = lds.win4 + L.l;
  L.twA_l = lds.twA4 + L.l;
  L.twA31_l = lds.twA31 + L.l;
  L.twP_l = lds.twP4 + L.l;
  return L;
}

template <int LL>
__device__ __forceinline__ void fill_tablesL(const FastArgs &a, const LdsL<LL> &lds, int tid, int nthreads) {
  using P = PL<LL>;
  const float2 *hw = reinterpret_cast<const float2 *>(a.hwin);
  for (int e = tid; e < 16 * LL; e += nthreads) {
    const int m = e / LL, l = e % LL;
    const float2 w0 = hw[l + LL * (2 * m)], w1 = hw[l + LL * (2 * m + 1)];
    lds.win4[e] = make_float4(w0.x, w0.y, w1.x, w1.y);
  }
  for (int e = tid; e < 15 * LL; e += nthreads) {
    const int m = e / LL, l = e % LL;
    const float2 w0 = a.w_m[l * (2 * m + 1)], w1 = a.w_m[l * (2 * m + 2)];
    lds.twA4[e] = make_float4(w0.x, w0.y, w1.x, w1.y);
  }
  for (int e = tid; e < LL; e += nthreads) lds.twA31[e] = a.w_m[e * 31];
  for (int e = tid; e < 8 * LL; e += nthreads) {
    const int m = e / LL, l = e % LL;
    const float2 w0 = a.w_n[l + LL * (2 * m)], w1 = a.w_n[l + LL * (2 * m + 1)];
    lds.twP4[e] = make_float4(w0.x, w0.y, w1.x, w1.y);
  }
  if (tid < 2) { lds.filled[tid * P::TS] = 0u; lds.drained[tid * P::TS] = 0u; }
}

// 64 / LL frames (one per LL lanes): raw samples -> window -> FFT(M complex) -> post-pass -> |X|^p in the frames' columns of
// `tile`; the hooks of `mid` are those of frame32_to_tile.
template <int LL, int PMODE, class Mid, bool CPLX = false>
__device__ __forceinline__ void frameL_to_tile(const FastArgs &a, const LaneL &L, float2 (&raw)[32], float *tile, const Mid &mid) {
#pragma clang fp contract(off)
  // (the arithmetic on packed pairs, one generated inline-assembly statement per stage: stft_pk_fft.inc, as frame32_to_tile)
  using P = PL<LL>;
  f2 v[32], t[32];
#pragma unroll
  for (int m0 = 0; m0 < 16; m0 += 8) {
    float4 win[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) win[m] = L.win_l[LL * (m0 + m)];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      v[2 * (m0 + m)] = f2{raw[2 * (m0 + m)].x, raw[2 * (m0 + m)].y} * f2{win[m].x, win[m].y};
      v[2 * (m0 + m) + 1] = f2{raw[2 * (m0 + m) + 1].x, raw[2 * (m0 + m) + 1].y} * f2{win[m].z, win[m].w};
    }
    SMX_FENCE();
  }
  SMX_FENCE();
  {   // A: radix-32 over j, then twiddle W_M^(l k1), the first plane of the transposition written between the products
    float4 tw[15];
#pragma unroll
    for (int m = 0; m < 15; ++m) tw[m] = L.twA_l[LL * m];
    const float2 tw31 = L.twA31_l[0];
    pk_fft32(v, [&] { SMX_FENCE(); mid.early(); SMX_FENCE(); });
    SMX_FENCE();
    mid.before_cells();
    float *const wr = tile + opaque32(L.own);
    float *const wr_hi = wr + 16 * P::CellPitch;
    auto put = [&](int j) { if (LL != 4 || j < 16) (j < 16 ? wr : wr_hi)[P::CellPitch * (j & 15)] = v[j].x; };   // (LL = 4: registers 16..31 in the second pass)
#define SMX_TWV(m) f2{tw[m].x, tw[m].y}, f2{tw[m].z, tw[m].w}
    put(0);
    pk_twiddle8(v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], SMX_TWV(0), SMX_TWV(1), SMX_TWV(2), SMX_TWV(3));
#pragma unroll
    for (int j = 1; j <= 8; ++j) put(j);
    SMX_FENCE();
    pk_twiddle8(v[9], v[10], v[11], v[12], v[13], v[14], v[15], v[16], SMX_TWV(4), SMX_TWV(5), SMX_TWV(6), SMX_TWV(7));
#pragma unroll
    for (int j = 9; j <= 16; ++j) put(j);
    SMX_FENCE();
    pk_twiddle8(v[17], v[18], v[19], v[20], v[21], v[22], v[23], v[24], SMX_TWV(8), SMX_TWV(9), SMX_TWV(10), SMX_TWV(11));
#pragma unroll
    for (int j = 17; j <= 24; ++j) put(j);
    SMX_FENCE();
    pk_twiddle7(v[25], v[26], v[27], v[28], v[29], v[30], v[31], SMX_TWV(12), SMX_TWV(13), SMX_TWV(14), f2{tw31.x, tw31.y});
#pragma unroll
    for (int j = 25; j <= 31; ++j) put(j);
#undef SMX_TWV
  }
  SMX_FENCE();
  float *const wr = tile + opaque32(L.own);
  float *const wr_hi = wr + 16 * P::CellPitch;   // (ds offsets are 16 bits)
  const float *const rd = tile + opaque32(L.rd);
  // X: lane lam takes V[i][lam + LL a], i < LL, a < 32 / LL: cell i + CP (lam + LL a); t[LL a + i]
  if constexpr (LL == 4) {   // two passes of 16 cells: k1 = lam + 4 a < 16 (a < 4) is register j = k1 of lane i, k1 >= 16 its register 16 + ...
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i].x = rd[P::TS * (i % 4) + 4 * P::CellPitch * (i / 4)];
#pragma unroll
    for (int j = 16; j < 32; ++j) wr[P::CellPitch * (j - 16)] = v[j].x;
#pragma unroll
    for (int i = 16; i < 32; ++i) t[i].x = rd[P::TS * (i % 4) + 4 * P::CellPitch * (i / 4 - 4)];
#pragma unroll
    for (int j = 0; j < 16; ++j) wr[P::CellPitch * j] = v[j].y;
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i].y = rd[P::TS * (i % 4) + 4 * P::CellPitch * (i / 4)];
#pragma unroll
    for (int j = 16; j < 32; ++j) wr[P::CellPitch * (j - 16)] = v[j].y;
#pragma unroll
    for (int i = 16; i < 32; ++i) t[i].y = rd[P::TS * (i % 4) + 4 * P::CellPitch * (i / 4 - 4)];
  } else {
#pragma unroll
    for (int i = 0; i < 32; ++i) t[i].x = rd[P::TS * (i % LL) + LL * P::CellPitch * (i / LL)];
#pragma unroll
    for (int j = 0; j < 32; ++j) (j < 16 ? wr : wr_hi)[P::CellPitch * (j & 15)] = v[j].y;
#pragma unroll
    for (int i = 0; i < 32; ++i) t[i].y = rd[P::TS * (i % LL) + LL * P::CellPitch * (i / LL)];
  }
  SMX_FENCE();
  mid.after_transposition_issue();
  SMX_FENCE();
  // B: 32 / LL radix-LL transforms over l; register u = a + (32 / LL) q holds Z[lam + LL u]
  if constexpr (LL == 16) {
    f2 e[16], o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { e[i] = t[i]; o[i] = t[16 + i]; }
    pk_fft16(e);
    pk_fft16(o);
#pragma unroll
    for (int q = 0; q < 16; ++q) { t[2 * q] = e[q]; t[2 * q + 1] = o[q]; }
  } else if constexpr (LL == 4) {   // eight 4-point transforms: t[4 a + i] -> register a + 8 q
    f2 g[8][4];
#pragma unroll
    for (int aa = 0; aa < 8; ++aa)
#pragma unroll
      for (int i = 0; i < 4; ++i) g[aa][i] = t[4 * aa + i];
#pragma unroll
    for (int aa = 0; aa < 8; aa += 2) pk_fft4x2(g[aa], g[aa + 1]);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int aa = 0; aa < 8; ++aa) t[aa + 8 * q] = g[aa][q];
  } else {
    f2 g[4][8];
#pragma unroll
    for (int aa = 0; aa < 4; ++aa) {
#pragma unroll
      for (int i = 0; i < 8; ++i) g[aa][i] = t[8 * aa + i];
      pk_fft8(g[aa]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int aa = 0; aa < 4; ++aa) t[aa + 4 * q] = g[aa][q];
  }
  SMX_FENCE();
  // P: partners through the cells (as frame32_to_tile, LL lanes)
  f2 pp[16];
  const float *const xr = tile + opaque32(L.xr);
#pragma unroll
  for (int q = 16; q < 32; ++q) wr[P::CellPitch * (q - 16)] = t[q].x;
  wr_hi[0] = t[0].x;
#pragma unroll
  for (int s = 0; s < 16; ++s) pp[s].x = xr[P::CellPitch * (15 - s)];
#pragma unroll
  for (int q = 16; q < 32; ++q) wr[P::CellPitch * (q - 16)] = t[q].y;
  wr_hi[0] = t[0].y;
#pragma unroll
  for (int s = 0; s < 16; ++s) pp[s].y = xr[P::CellPitch * (15 - s)];
  float4 tw[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) tw[m] = L.twP_l[LL * m];
  SMX_FENCE();
  mid.after_exchange_issue();
  SMX_FENCE();
  // CPLX: the spectrum itself, real parts in `tile`, imaginary parts in the plane after it (the other tile buffer)
  {   // bin M/2 (lane 0, register 16): X = 2 conj(Z)
    const f2 z = t[16] + t[16];
    if constexpr (CPLX) {
      tile[opaque32(L.self)] = z.x;
      tile[P::TileFloats + opaque32(L.self)] = -z.y;
    } else {
      tile[opaque32(L.self)] = power_from_square<PMODE>(__builtin_fmaf(z.x, z.x, z.y * z.y), a);
    }
  }
  float *const rk = wr;                        // row l + LL s
  float *const rm = tile + opaque32(L.rm);     // row (LL - l) + LL (31 - s) = rm base + LL (15 - s)
  auto wtw = [&](int s) { return (s & 1) ? f2{tw[s >> 1].z, tw[s >> 1].w} : f2{tw[s >> 1].x, tw[s >> 1].y}; };
  auto put = [&](int s, f2 re, f2 im) {
    if constexpr (CPLX) {
      rk[P::RowPitch * s] = re.x;
      rk[P::TileFloats + P::RowPitch * s] = im.x;
      rm[P::RowPitch * (15 - s)] = re.y;
      rm[P::TileFloats + P::RowPitch * (15 - s)] = im.y;
    } else {
      rk[P::RowPitch * s] = power_from_square<PMODE>(re.x, a);
      rm[P::RowPitch * (15 - s)] = power_from_square<PMODE>(re.y, a);
    }
  };
#define SMX_PA(s) t[s], pp[s], wtw(s)
  f2 r[6], q[6];
  if constexpr (CPLX) pk_post_cplx2(SMX_PA(0), SMX_PA(1), r[0], q[0], r[1], q[1]);
  else pk_post_power2(SMX_PA(0), SMX_PA(1), r[0], r[1]);
#pragma unroll
  for (int i = 0; i < 2; ++i) put(i, r[i], q[i]);
  SMX_FENCE(); mid.postpass_at(1); SMX_FENCE();
  if constexpr (CPLX) pk_post_cplx4(SMX_PA(2), SMX_PA(3), SMX_PA(4), SMX_PA(5), r[0], q[0], r[1], q[1], r[2], q[2], r[3], q[3]);
  else pk_post_power4(SMX_PA(2), SMX_PA(3), SMX_PA(4), SMX_PA(5), r[0], r[1], r[2], r[3]);
#pragma unroll
  for (int i = 0; i < 4; ++i) put(2 + i, r[i], q[i]);
  SMX_FENCE(); mid.postpass_at(5); SMX_FENCE();
  if constexpr (CPLX) pk_post_cplx5(SMX_PA(6), SMX_PA(7), SMX_PA(8), SMX_PA(9), SMX_PA(10), r[0], q[0], r[1], q[1], r[2], q[2], r[3], q[3], r[4], q[4]);
  else pk_post_power5(SMX_PA(6), SMX_PA(7), SMX_PA(8), SMX_PA(9), SMX_PA(10), r[0], r[1], r[2], r[3], r[4]);
#pragma unroll
  for (int i = 0; i < 5; ++i) put(6 + i, r[i], q[i]);
  if constexpr (CPLX) pk_post_cplx5(SMX_PA(11), SMX_PA(12), SMX_PA(13), SMX_PA(14), SMX_PA(15), r[0], q[0], r[1], q[1], r[2], q[2], r[3], q[3], r[4], q[4]);
  else pk_post_power5(SMX_PA(11), SMX_PA(12), SMX_PA(13), SMX_PA(14), SMX_PA(15), r[0], r[1], r[2], r[3], r[4]);
#pragma unroll
  for (int i = 0; i < 5; ++i) put(11 + i, r[i], q[i]);
#undef SMX_PA
  SMX_FENCE(); mid.postpass_at(15); SMX_FENCE();
}

// raw samples of the lane's frame: z[n] = (x[2n], x[2n+1]), n = l + LL j; `src` is the frame's first sample (per lane)
template <int LL, bool ALIGNED>
__device__ __forceinline__ void load_frameL(const float *src, int l, float2 (&raw)[32]) {
  if constexpr (ALIGNED) {
    const float2 *p = reinterpret_cast<const float2 *>(src) + l;
#pragma unroll
    for (int j = 0; j < 32; ++j) raw[j] = p[LL * j];
  } else {
    const float *p = src + 2 * l;
#pragma unroll
    for (int j = 0; j < 32; ++j) raw[j] = make_float2(p[2 * LL * j], p[2 * LL * j + 1]);
  }
}

// the same for a tile with frames that reach past either end of the signal (see load_frame32_padded in stft_fast_p32.hpp: round 5)
template <int LL>
__device__ __forceinline__ void load_frameL_padded(const FastArgs &a, const float *xc /* wave-uniform */, int s0, int l, float2 (&raw)[32]) {
  asm volatile("" : "+v"(s0));
  const int n = (int)a.n, top = 2 * (n - 1);
  const bool refl = a.pad == SMX_PAD_REFLECT;
  auto at = [&](int s) {
    const int m = s < 0 ? -s : s;
    const int ir = m < top - m ? m : top - m;
    const int ie = s < 0 ? 0 : (s < n ? s : n - 1);
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(xc) + 4u * (unsigned)(refl ? ir : ie));
  };
  const int s00 = s0 + 2 * l;
#pragma unroll
  for (int j = 0; j < 32; ++j) raw[j] = make_float2(at(s00 + 2 * LL * j), at(s00 + 2 * LL * j + 1));
  if (a.pad != SMX_PAD_REFLECT && a.pad != SMX_PAD_EDGE) {   // constant padding (wave-uniform)
    const float pv = a.pad_value;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int s = s00 + 2 * LL * j;
      if ((unsigned)s >= (unsigned)n) raw[j].x = pv;
      if ((unsigned)(s + 1) >= (unsigned)n) raw[j].y = pv;
    }
  }
}

// A wave's share of a finished tile: 8 parts of (LL = 16: 8 rows; LL = 8: 4 rows of one 32-frame half of the tile per lane
// half) x 4 frames per lane -> out[clip][bin][f0 + cb + 4 g ..]; 8 lanes store one 128-byte run.  Rows r' (r' < 4) and
// frame groups g < 8 per 32-lane half keep the LDS reads conflict free.
// (LL = 4: FT = 128, a part is 4 rows x 64 frames, parts 2 i / 2 i + 1 are the two 64-frame halves of rows 32 i + 4 wave ...)
struct FlushL {
  int src0;         // float offset in the tile of part 0: row0 * TS + cb + 4 g
  unsigned goff0;   // byte offset of out[row0][cb + 4 g] from the tile's origin
  int f0;           // cb + 4 g: the lane's first frame (LL = 4: of the even parts; the odd ones 64 further)
  int nyq_f0;       // the lane's first frame of the Nyquist row
  bool nyq;         // this lane stores 4 frames of the Nyquist row
};
template <int LL> __device__ __forceinline__ constexpr int flush_rw() { return LL == 16 ? 8 : 4; }                  // rows per wave and part
template <int LL> __device__ __forceinline__ constexpr int flush_part_row(int it) { return LL == 4 ? 32 * (it >> 1) : 8 * flush_rw<LL>() * it; }
template <int LL> __device__ __forceinline__ constexpr int flush_part_col(int it) { return LL == 4 ? 64 * (it & 1) : 0; }
struct FlushRegsL {
  float v[8][4];
  float nyq[4];
};
template <int LL>
__device__ __forceinline__ FlushL setup_flushL(const FastArgs &a, int lane, int wave) {
  using P = PL<LL>;
  constexpr int RW = flush_rw<LL>();
  FlushL fl;
  const int g = lane & 7, cb = LL == 16 ? 0 : 32 * (lane >> 5);
  const int row0 = RW * wave + ((lane >> 3) & (RW - 1));
  fl.f0 = cb + 4 * g;
  fl.src0 = row0 * P::TS + fl.f0;
  fl.goff0 = ((unsigned)row0 * (unsigned)a.out_stride + (unsigned)fl.f0) * 4u;
  fl.nyq = (LL == 4 ? wave < 2 : wave == 0) && (lane & 31) < 8 && (LL != 16 || lane < 32);
  fl.nyq_f0 = fl.f0 + (LL == 4 ? 64 * (wave & 1) : 0);
  return fl;
}
template <int LL>
__device__ __forceinline__ void flushL_read(const float *tile, const FlushL &fl, FlushRegsL &r) {
  using P = PL<LL>;
  const float *src0 = tile + opaque32(fl.src0);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const float *src = src0 + flush_part_row<LL>(it) * P::TS + flush_part_col<LL>(it);
    r.v[it][0] = src[0]; r.v[it][1] = src[1]; r.v[it][2] = src[2]; r.v[it][3] = src[3];
  }
  const float *ny = tile + P::M * P::TS + opaque32(fl.nyq_f0);   // the Nyquist row (every lane reads, FT / 4 lanes of wave 0 (LL = 4: waves 0, 1) store)
  r.nyq[0] = ny[0]; r.nyq[1] = ny[1]; r.nyq[2] = ny[2]; r.nyq[3] = ny[3];
}
template <int LL>
__device__ __forceinline__ void flushL_store(const FastArgs &a, const FlushL &fl, float *obase, int frames_left, const FlushRegsL &r) {
  using P = PL<LL>;
  const unsigned pitch = (unsigned)a.out_stride * 4u;
  const unsigned goff0 = opaque32(fl.goff0);
  auto put = [&](unsigned goff, const float (&v)[4], int fleft) {
    if (fleft >= 4) {
      store4_unaligned(obase, goff, v[0], v[1], v[2], v[3]);
    } else {
      float *dst = reinterpret_cast<float *>(reinterpret_cast<char *>(obase) + goff);
      if (fleft > 0) dst[0] = v[0];
      if (fleft > 1) dst[1] = v[1];
      if (fleft > 2) dst[2] = v[2];
    }
  };
  if (frames_left >= P::FT) {   // wave-uniform: a whole tile, no masks
#pragma unroll
    for (int it = 0; it < 8; ++it)
      store4_unaligned(obase, goff0 + (unsigned)flush_part_row<LL>(it) * pitch + 4u * (unsigned)flush_part_col<LL>(it), r.v[it][0], r.v[it][1], r.v[it][2], r.v[it][3]);
  } else {
#pragma unroll
    for (int it = 0; it < 8; ++it)
      put(goff0 + (unsigned)flush_part_row<LL>(it) * pitch + 4u * (unsigned)flush_part_col<LL>(it), r.v[it], frames_left - fl.f0 - flush_part_col<LL>(it));
  }
  if (fl.nyq) put((unsigned)P::M * pitch + 4u * (unsigned)fl.nyq_f0, r.nyq, frames_left - fl.nyq_f0);
}

// ---- the flush in whole aligned 128-byte lines (round 5, late; LL = 16 / 8: as skewg32_* of stft_fast_p32.hpp, a frame per lane) ----
// A tile's FT frames of a row straddle the row's 128-byte lines wherever the row does not start on one (C1: rows of 1723 floats), and
// runs that straddle lines stream at 1.7-2.4 TB/s against 5.3 for whole lines (profiles/r06/store_shape_probe.log; the counters put the
// plain flush of fft 1024 at 1.23x its bytes).  Here a lane holds ONE frame of a row: the FT - phi frames that complete lines are stored at
// once, the trailing phi frames (phi = the row's offset into its line, the same for every tile of the row: FT is a multiple of 32) wait in
// registers for the next tile -- every store instruction writes whole lines: lane c stores either its frame of this tile or the carried
// frame of the previous one, each at its own natural address.  Rows a lane serves are 32 apart, so that they share phi; the 32 residues
// are NP passes of 8 x RPI (RPI = rows per instruction = 64 / FT): NP x PARTS = 32 carried registers.  Needs consecutive tiles of a
// clip on one workgroup (contiguous ranges).  Same values as the plain flush (tests: SMX_POWER_SKEW=0).
template <int LL>
struct SkL {
  static constexpr int FT = PL<LL>::FT, RPI = 64 / FT, NP = 4 / RPI, PARTS = PL<LL>::M / 32;
  static_assert(LL == 16 || LL == 8, "a frame per lane, 64 lanes = 1 or 2 rows");
  static_assert(NP * PARTS == 32, "carried registers");
};
template <int LL>
struct SkewL {
  unsigned goff[SkL<LL>::NP];             // from FT floats BEFORE a tile's origin: byte offset of out[row][col], plus 4 FT for the lanes that store this tile's frame
  unsigned long long sel[SkL<LL>::NP];    // the lanes whose frame of the current tile completes a line
};
struct SkewLRegs {
  float cur[32];
  float nyq;
};
template <int LL>
__device__ __forceinline__ void skewL_lane(int lane, int wave, int &row0, int &col) {   // pass q: rows row0 + 8 RPI q + 32 p
  asm volatile("" : "+v"(lane));
  col = lane & (SkL<LL>::FT - 1);
  row0 = SkL<LL>::RPI * wave + lane / SkL<LL>::FT;
}
template <int LL>
__device__ __forceinline__ void skewL_clip(const FastArgs &a, SkewL<LL> &sk, const float *oclip, int lane, int wave) {
  using K = SkL<LL>;
  int row0, col;
  skewL_lane<LL>(lane, wave, row0, col);
#pragma unroll
  for (int q = 0; q < K::NP; ++q) {
    const int row = row0 + 8 * K::RPI * q;
    const unsigned phi = (unsigned)((reinterpret_cast<uintptr_t>(oclip) >> 2) + (uintptr_t)row * (uintptr_t)a.out_stride) & 31u;
    const bool now = (unsigned)col < (unsigned)K::FT - phi;
    sk.sel[q] = __ballot(now);
    sk.goff[q] = ((unsigned)row * (unsigned)a.out_stride + (unsigned)col) * 4u + (now ? 4u * K::FT : 0u);
  }
}
template <int LL>
__device__ __forceinline__ void skewL_read(const float *tile, int lane, int wave, SkewLRegs &r) {
  using K = SkL<LL>;
  using P = PL<LL>;
  int row0, col;
  skewL_lane<LL>(lane, wave, row0, col);
  const float *src = tile + row0 * P::TS + col;
#pragma unroll
  for (int q = 0; q < K::NP; ++q)
#pragma unroll
    for (int pp = 0; pp < K::PARTS; ++pp) r.cur[q * K::PARTS + pp] = src[(8 * K::RPI * q + 32 * pp) * P::TS];
  r.nyq = tile[P::M * P::TS + col];   // the Nyquist row (every wave reads it, wave 0 stores it)
}
template <int LL>
__device__ __forceinline__ void skewL_store(const FastArgs &a, const SkewL<LL> &sk, float *obase, int frames_left, bool fresh, bool closing, int wave,
                                            int lane, const SkewLRegs &r, float (&carry)[32]) {
  using K = SkL<LL>;
  using P = PL<LL>;
  const unsigned pitch = (unsigned)a.out_stride * 4u;
  int row0, col;
  skewL_lane<LL>(lane, wave, row0, col);
  if (frames_left >= K::FT && !fresh && !closing) {   // wave-uniform: whole lines for every row and part
#pragma unroll
    for (int q = 0; q < K::NP; ++q) {
      const unsigned g = opaque32(sk.goff[q]);
#pragma unroll
      for (int pp = 0; pp < K::PARTS; ++pp) {
        const int i = q * K::PARTS + pp;
        store1_at(obase - K::FT, g + 32u * (unsigned)pp * pitch, select_lanes(carry[i], r.cur[i], sk.sel[q]));
        carry[i] = r.cur[i];
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < K::NP; ++q) {
      const bool now = (sk.sel[q] >> lane) & 1;
      const unsigned gl = ((unsigned)(row0 + 8 * K::RPI * q) * (unsigned)a.out_stride + (unsigned)col) * 4u;
#pragma unroll
      for (int pp = 0; pp < K::PARTS; ++pp) {
        const int i = q * K::PARTS + pp;
        float *dst = reinterpret_cast<float *>(reinterpret_cast<char *>(obase) + gl + 32u * (unsigned)pp * pitch);
        if ((now || closing) && col < frames_left) dst[0] = r.cur[i];
        if (!now && !fresh) dst[-K::FT] = carry[i];
        carry[i] = r.cur[i];
      }
    }
  }
  if (wave == 0 && lane < K::FT && lane < frames_left) obase[(int64_t)P::M * a.out_stride + lane] = r.nyq;
}

template <int LL, bool ALIGNED, bool SKEW = false>
struct PowerMidL {
  using P = PL<LL>;
  const FastArgs &a;
  const LdsL<LL> &lds;
  const FlushL &fl;
  FlushRegsL &fr;
  const SkewL<(LL >= 8 ? LL : 8)> &sk;   // SKEW: the flush in whole aligned lines
  SkewLRegs &sr;
  float (&carry)[32];
  bool pend_fresh, pend_closing;
  float2 (&raw)[32];
  const float *src;      // the next frames' samples (per lane)
  const float *src_clip; // ... their clip and whether their tile holds a frame that reaches past the signal (see PowerMid32)
  bool src_border;
  float *pend_out;       // output origin and frames of the previous tile
  int pend_left;
  int lane, wave, b, it;
  unsigned &pk_drained, &pk_filled;
  __device__ __forceinline__ void early() const { pk_drained = peek32(lds.drained + b * P::TS); }
  __device__ __forceinline__ void before_cells() const {
    lds_wait32(lds.drained + b * P::TS, 8u * ((unsigned)it >> 1), pk_drained);
  }
  __device__ __forceinline__ void after_transposition_issue() const {
    if (it > 0) pk_filled = peek32(lds.filled + (b ^ 1) * P::TS);
  }
  __device__ __forceinline__ void after_exchange_issue() const {
    if (it > 0) {
      lds_wait32(lds.filled + (b ^ 1) * P::TS, 8u * (((unsigned)(it - 1) >> 1) + 1), pk_filled);
      if constexpr (SKEW) skewL_read<LL>(lds.tiles + (b ^ 1) * P::TileFloats, lane, wave, sr);
      else flushL_read<LL>(lds.tiles + (b ^ 1) * P::TileFloats, fl, fr);
      lds_signal32(lds.drained + (b ^ 1) * P::TS, lane);   // "read out" as soon as the reads are issued (in-order LDS)
    }
  }
  __device__ __forceinline__ void postpass_at(int s) const {
    const bool same = SMX_P32_STORE_AT == SMX_P32_LOAD_AT;
    if (s == SMX_P32_LOAD_AT && same && SMX_P32_LOADS_FIRST) { load_frameL<LL, ALIGNED>(src_border ? src_clip : src, lane & (LL - 1), raw); SMX_FENCE(); }
    if (s == SMX_P32_STORE_AT && it > 0) {
      if constexpr (SKEW) skewL_store<LL>(a, sk, pend_out, pend_left, pend_fresh, pend_closing, wave, lane, sr, carry);
      else flushL_store<LL>(a, fl, pend_out, pend_left, fr);
    }
    SMX_FENCE();
    if (s == SMX_P32_LOAD_AT && !(same && SMX_P32_LOADS_FIRST)) load_frameL<LL, ALIGNED>(src_border ? src_clip : src, lane & (LL - 1), raw);
    if (s == 15 && src_border) load_frameL_padded<LL>(a, src_clip, (int)(src - src_clip), lane & (LL - 1), raw);   // (wave-uniform; see PowerMid32::load_next)
  }
};

template <int LL, bool ALIGNED, int PMODE, bool STRIP, bool SKEW = false>
__global__ void __launch_bounds__(512) stft_power_lanes_kernel(FastArgs a) {
  using P = PL<LL>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const LdsL<LL> lds = carve_ldsL<LL>(smem);
  const LaneL L = setup_laneL<LL>(lds, lane, wave);
  fill_tablesL<LL>(a, lds, tid, 512);
  TileWalk tw;
  tw.init(a, a.out + a.out_offset, P::Bins * a.out_stride);
  const int ntiles = tw.ntiles > 0 ? tw.ntiles : 0;

  // first sample of this lane's frame in tile t of the clip at xc (lanes without a frame re-read the tile's first frame and
  // their results are never stored)
  auto frame_ptr = [&](const float *xc, int t) {
    const int64_t f0 = (int64_t)t * P::FT;
    const int avail = (int)(a.count - f0 < P::FT ? a.count - f0 : P::FT) - 1;   // last frame of the tile that exists (wave-uniform)
    const int fi = L.col;
    const int64_t p = a.p0 + f0 + (fi <= avail ? fi : 0);
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
  auto tile_border = [&](int t) {   // fold_frames == 2 (as stft2048_power32_kernel): the tile takes load_frameL_padded
    const int64_t q0 = a.p0 + (int64_t)t * P::FT;
    return a.fold_frames == 2 && (q0 < a.border_i0 || q0 + P::FT > a.border_i1);
  };
  if (ntiles > 0) {
    const float *src0 = frame_ptr(tw.xclip, tw.ft);
    if (tile_border(tw.ft)) load_frameL_padded<LL>(a, tw.xclip, (int)(src0 - tw.xclip), L.l, raw);
    else load_frameL<LL, ALIGNED>(src0, L.l, raw);
  }
  __syncthreads();   // tables and zeroed counters visible: the only workgroup barrier of the main loop
  float *pend_out = nullptr;
  int pend_left = 0;
  const FlushL fl = setup_flushL<LL>(a, lane, wave);
  FlushRegsL fr;
  SkewL<(LL >= 8 ? LL : 8)> sk{};
  SkewLRegs sr;
  float carry[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) carry[i] = 0.f;
  bool pend_fresh = true, pend_closing = false;
  const float *pend_oclip = nullptr;
  unsigned pk_drained = 0, pk_filled = 0;
  for (int it = 0; it < ntiles; ++it) {   // tile `it` of this workgroup lives in buffer it & 1
    const int b = it & 1;
    int ftnext;
    const float *xnext;
    float *onext;
    tw.peek(a, ftnext, xnext, onext);
    const bool more = it + 1 < ntiles;
    const float *src_clip = more ? xnext : tw.xclip;
    const float *src = frame_ptr(src_clip, more ? ftnext : tw.ft);
    const bool src_border = tile_border(more ? ftnext : tw.ft);
    if constexpr (SKEW) {
      if (it > 0 && pend_fresh) skewL_clip<LL>(a, sk, pend_oclip, lane, wave);   // (wave-uniform) the pending tile begins a clip or this workgroup's range
    }
    const PowerMidL<LL, ALIGNED, SKEW> mid{a, lds, fl, fr, sk, sr, carry, pend_fresh, pend_closing, raw, src, src_clip, src_border, pend_out, pend_left, lane, wave, b, it, pk_drained, pk_filled};
    frameL_to_tile<LL, PMODE>(a, L, raw, lds.tiles + b * P::TileFloats, mid);
    lds_signal32(lds.filled + b * P::TS, lane);
    pend_out = tw.oclip + tw.ft * P::FT;   // wave-uniform
    const int64_t left = a.count - (int64_t)tw.ft * P::FT;
    pend_left = left < P::FT ? (int)left : P::FT;
    pend_oclip = tw.oclip;
    pend_fresh = it == 0 || tw.ft == 0;
    pend_closing = tw.ft == a.tiles_per_clip - 1;
    tw.xclip = xnext;
    tw.oclip = onext;
    tw.ft = ftnext;
  }
  if (ntiles > 0) {   // the last tile of this workgroup
    const int b = (ntiles - 1) & 1;
    lds_wait(lds.filled + b * P::TS, 8u * (((unsigned)(ntiles - 1) >> 1) + 1));
    if constexpr (SKEW) {
      if (pend_fresh) skewL_clip<LL>(a, sk, pend_oclip, lane, wave);
      skewL_read<LL>(lds.tiles + b * P::TileFloats, lane, wave, sr);
      skewL_store<LL>(a, sk, pend_out, pend_left, pend_fresh, true, wave, lane, sr, carry);
    } else {
      flushL_read<LL>(lds.tiles + b * P::TileFloats, fl, fr);
      flushL_store<LL>(a, fl, pend_out, pend_left, fr);
    }
  }

  // Border frames (the few per clip whose window reaches past either end of the signal): same frame code on samples fetched
  // through the padding rule, FT (clip, frame) pairs per tile, results scattered to their places.
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
    for (int64_t bt = (int64_t)gridDim.x - 1 - tw.uid; bt * P::FT < total; bt += gridDim.x) {   // from the last workgroup down: idle ones first
      __syncthreads();   // the buffer is free: every wave is past its last flush / the previous border tile
      {
        int64_t beta = bt * P::FT + L.col;
        if (beta >= total) beta = bt * P::FT;   // lanes without a pair repeat the tile's first one (never stored)
        int64_t clip, p;
        locate(beta, clip, p);
        const float *xs = a.x + clip * a.x_stride;
        const int s0 = (int)(p * a.hop - a.left);
        float2 braw[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          const int s = s0 + 2 * (L.l + LL * j);
          braw[j] = make_float2(fetch_padded(xs, (int)a.n, s, a.pad, a.pad_value), fetch_padded(xs, (int)a.n, s + 1, a.pad, a.pad_value));
        }
        frameL_to_tile<LL, PMODE>(a, L, braw, bt_tile, NoMid32{});
      }
      __syncthreads();
      for (int e = tid; e < P::Bins * P::FT; e += 512) {
        const int k = e / P::FT, f = e % P::FT;
        const int64_t bf = bt * P::FT + f;
        if (bf < total) {
          int64_t clip, p;
          locate(bf, clip, p);
          a.out[(clip * P::Bins + k) * a.out_stride + a.border_out_offset + (p - a.border_p0)] = bt_tile[k * P::TS + f];
        }
      }
    }
  }
}

// ---- the fused audio -> mel spectrogram on the same pipeline (fft 1024: LL = 16) -------------------------------------------
// stft_fast_mel32.hpp's scheme on the LL-lane tiles: where the power kernel reads the previous tile out, every wave multiplies
// its filterbank items into it, one group of 16 frame columns after the other (the MFMA's N = 16).
#ifndef SMX_MEL_LANES_MULTI
#define SMX_MEL_LANES_MULTI 1   // A operands once per chunk for all column groups (fft 1024: 0.572 -> 0.523 ms, fft 512: 0.671 -> 0.560)
#endif
constexpr int kMel4rChunksL = 4;   // resident A operands of the lanes kernels: 32 registers (their bands are 2 - 4 times shorter)
template <int LL, bool ALIGNED, bool FOUR>
struct MelMidL {
  using P = PL<LL>;
  const FastArgs &a;
  const Mel32Args &m;
  const float (&areg)[8 * kMel4rChunksL];
  const Mel4rSlots &slots;
  int iv;
  const LdsL<LL> &lds;
  float2 (&raw)[32];
  const float *src;
  const float *src_clip;
  bool src_border;
  float *pend_out;
  int pend_left;
  int lane, wave, b, it;
  unsigned &pk_drained, &pk_filled;
  __device__ __forceinline__ void early() const { pk_drained = peek32(lds.drained + b * P::TS); }
  __device__ __forceinline__ void before_cells() const {
    lds_wait32(lds.drained + b * P::TS, 8u * ((unsigned)it >> 1), pk_drained);
  }
  __device__ __forceinline__ void after_transposition_issue() const {
    if (it > 0) pk_filled = peek32(lds.filled + (b ^ 1) * P::TS);
  }
  __device__ __forceinline__ void after_exchange_issue() const {
    if (it > 0) {
      lds_wait32(lds.filled + (b ^ 1) * P::TS, 8u * (((unsigned)(it - 1) >> 1) + 1), pk_filled);
      if constexpr (FOUR) {
        mel4r_items<P::TS, P::FT / 16, kMel4rChunksL, (LL == 16 ? 8 : 4), true>(m, iv, areg, lds.tiles + (b ^ 1) * P::TileFloats, pend_out, pend_left, lane, slots);
        lds_signal32(lds.drained + (b ^ 1) * P::TS, lane);
        return;
      }
#if SMX_MEL_LANES_MULTI
      mel32_items_multi<P::TS, P::FT / 16, (LL == 16 ? 4 : 2)>(m, iv, lds.tiles + (b ^ 1) * P::TileFloats, pend_out, pend_left, lane);
#else
#pragma unroll 1
      for (int cg = 0; cg < P::FT / 16; ++cg)
        mel32_items<P::TS>(m, iv, lds.tiles + (b ^ 1) * P::TileFloats + 16 * cg, pend_out + 16 * cg, pend_left - 16 * cg, lane);
#endif
      lds_signal32(lds.drained + (b ^ 1) * P::TS, lane);
    }
  }
  __device__ __forceinline__ void postpass_at(int s) const {
    if (s == SMX_P32_LOAD_AT) load_frameL<LL, ALIGNED>(src_border ? src_clip : src, lane & (LL - 1), raw);
    if (s == 15 && src_border) load_frameL_padded<LL>(a, src_clip, (int)(src - src_clip), lane & (LL - 1), raw);
  }
};

template <int LL, bool ALIGNED, int PMODE, bool FOUR = false>
__global__ void __launch_bounds__(512) stft_mel_lanes_kernel(FastArgs a, Mel32Args m) {
  using P = PL<LL>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const LdsL<LL> lds = carve_ldsL<LL>(smem);
  const LaneL L = setup_laneL<LL>(lds, lane, wave);
  fill_tablesL<LL>(a, lds, tid, 512);
  TileWalk tw;
  tw.init(a, m.out + m.out_offset, (int64_t)m.n_mels * m.out_stride);
  const int ntiles = tw.ntiles > 0 ? tw.ntiles : 0;
  auto frame_ptr = [&](const float *xc, int t) {   // as stft_power_lanes_kernel
    const int64_t f0 = (int64_t)t * P::FT;
    const int avail = (int)(a.count - f0 < P::FT ? a.count - f0 : P::FT) - 1;
    const int fi = L.col;
    const int64_t p = a.p0 + f0 + (fi <= avail ? fi : 0);
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
  auto tile_border = [&](int t) {   // fold_frames == 2 (as stft2048_power32_kernel): the tile takes load_frameL_padded
    const int64_t q0 = a.p0 + (int64_t)t * P::FT;
    return a.fold_frames == 2 && (q0 < a.border_i0 || q0 + P::FT > a.border_i1);
  };
  if (ntiles > 0) {
    const float *src0 = frame_ptr(tw.xclip, tw.ft);
    if (tile_border(tw.ft)) load_frameL_padded<LL>(a, tw.xclip, (int)(src0 - tw.xclip), L.l, raw);
    else load_frameL<LL, ALIGNED>(src0, L.l, raw);
  }
  __syncthreads();
  float *pend_out = nullptr;
  int pend_left = 0;
  unsigned pk_drained = 0, pk_filled = 0;
  const int iv = reinterpret_cast<const int *>(m.items + wave * kMel32MaxItems)[lane];   // this wave's items (8 x 8 ints)
  float areg[8 * kMel4rChunksL];
  if constexpr (FOUR) {   // the wave's A operands of the banded product: [wave][step][lane], loaded once
#pragma unroll
    for (int q = 0; q < 8 * kMel4rChunksL; ++q) areg[q] = m.w[(wave * 8 * kMel4rChunksL + q) * 64 + lane];
  }
  Mel4rSlots slots{};
  if constexpr (FOUR) slots = mel4r_slots<P::TS, kMel4rChunksL, (LL == 16 ? 8 : 4)>(m, iv, lane);
  for (int it = 0; it < ntiles; ++it) {
    const int b = it & 1;
    int ftnext;
    const float *xnext;
    float *onext;
    tw.peek(a, ftnext, xnext, onext);
    const bool more = it + 1 < ntiles;
    const float *src_clip = more ? xnext : tw.xclip;
    const float *src = frame_ptr(src_clip, more ? ftnext : tw.ft);
    const bool src_border = tile_border(more ? ftnext : tw.ft);
    const MelMidL<LL, ALIGNED, FOUR> mid{a, m, areg, slots, iv, lds, raw, src, src_clip, src_border, pend_out, pend_left, lane, wave, b, it, pk_drained, pk_filled};
    frameL_to_tile<LL, PMODE>(a, L, raw, lds.tiles + b * P::TileFloats, mid);
    lds_signal32(lds.filled + b * P::TS, lane);
    pend_out = tw.oclip + tw.ft * P::FT;
    const int64_t left = a.count - (int64_t)tw.ft * P::FT;
    pend_left = left < P::FT ? (int)left : P::FT;
    tw.xclip = xnext;
    tw.oclip = onext;
    tw.ft = ftnext;
  }
  if (ntiles > 0) {
    const int b = (ntiles - 1) & 1;
    lds_wait(lds.filled + b * P::TS, 8u * (((unsigned)(ntiles - 1) >> 1) + 1));
    if constexpr (FOUR) {
      mel4r_items<P::TS, P::FT / 16, kMel4rChunksL, (LL == 16 ? 8 : 4), true>(m, iv, areg, lds.tiles + b * P::TileFloats, pend_out, pend_left, lane, slots);
      return;
    }
#if SMX_MEL_LANES_MULTI
    mel32_items_multi<P::TS, P::FT / 16, (LL == 16 ? 4 : 2)>(m, iv, lds.tiles + b * P::TileFloats, pend_out, pend_left, lane);
#else
#pragma unroll 1
    for (int cg = 0; cg < P::FT / 16; ++cg)
      mel32_items<P::TS>(m, iv, lds.tiles + b * P::TileFloats + 16 * cg, pend_out + 16 * cg, pend_left - 16 * cg, lane);
#endif
  }
}

// ---- Stft.transform at fft 1024 / 512 on the same frame code: as stft2048_complex32_kernel (the spectrum's planes fill both
// tile buffers; the previous tile is read out in the middle of the next frame's first radix-32) ------------------------------
struct CplxFlushL {
  int src0;          // float offset of (row0, frame 2 g) in a plane
  unsigned goff0;    // byte offset of out[row0][2 g] from the tile's origin (complex64)
  int g;             // frames 2 g, 2 g + 1
};
template <int LL>
__device__ __forceinline__ CplxFlushL setup_cplx_flushL(const FastArgs &a, int lane, int wave) {
  using P = PL<LL>;
  constexpr int LPR = P::FT / 2, RPI = 64 / LPR;   // lanes per row, rows per store instruction (4 / 2)
  CplxFlushL fl;
  fl.g = lane % LPR;
  const int row0 = RPI * 16 * wave + lane / LPR;
  fl.src0 = row0 * P::TS + 2 * fl.g;
  fl.goff0 = ((unsigned)row0 * (unsigned)a.out_stride + 2u * fl.g) * 8u;
  return fl;
}
template <int LL>
__device__ __forceinline__ void cplx_flushL(const FastArgs &a, const float *re, const CplxFlushL &fl, float *obase, int frames_left,
                                            int wave, int lane) {
  using P = PL<LL>;
  constexpr int LPR = P::FT / 2, RPI = 64 / LPR;
  const float *pr0 = re + opaque32(fl.src0);
  const int fleft = frames_left - 2 * fl.g;
  const unsigned pitch = (unsigned)a.out_stride * 8u, goff0 = opaque32(fl.goff0);
  auto put = [&](const float *pr, unsigned goff) {
    const float *pi = pr + P::TileFloats;
    const float r0 = pr[0], r1 = pr[1], i0 = pi[0], i1 = pi[1];
    if (fleft >= 2) {
      store4_unaligned(obase, goff, r0, i0, r1, i1);
    } else if (fleft == 1) {
      float *dst = reinterpret_cast<float *>(reinterpret_cast<char *>(obase) + goff);
      dst[0] = r0;
      dst[1] = i0;
    }
  };
#pragma unroll
  for (int i = 0; i < 16; ++i) put(pr0 + RPI * i * P::TS, goff0 + (unsigned)(RPI * i) * pitch);   // rows RPI (16 wave + i) + lane / LPR
  if (wave == 0 && lane < LPR && fleft >= 1) {   // bin M: real
    float *dst = reinterpret_cast<float *>(reinterpret_cast<char *>(obase) + (unsigned)P::M * pitch + 16u * (unsigned)fl.g);
    const float *pr = re + P::M * P::TS + 2 * fl.g;
    dst[0] = pr[0];
    dst[1] = 0.0f;
    if (fleft >= 2) {
      dst[2] = pr[1];
      dst[3] = 0.0f;
    }
  }
}
template <int LL, bool ALIGNED>
struct CplxMidL {
  using P = PL<LL>;
  const FastArgs &a;
  const LdsL<LL> &lds;
  const CplxFlushL &fl;
  float2 (&raw)[32];
  const float *src;
  const float *src_clip;
  bool src_border;
  float *pend_out;
  int pend_left;
  int lane, wave, it;
  __device__ __forceinline__ void early() const {
    if (it > 0) {
      lds_wait(lds.filled, 8u * (unsigned)it);
      cplx_flushL<LL>(a, lds.tiles, fl, pend_out, pend_left, wave, lane);
      lds_signal32(lds.drained, lane);
    }
  }
  __device__ __forceinline__ void before_cells() const {
    if (it > 0) lds_wait(lds.drained, 8u * (unsigned)it);
  }
  __device__ __forceinline__ void after_transposition_issue() const {}
  __device__ __forceinline__ void after_exchange_issue() const {}
  __device__ __forceinline__ void postpass_at(int s) const {
    if (s == SMX_P32_LOAD_AT) load_frameL<LL, ALIGNED>(src_border ? src_clip : src, lane & (LL - 1), raw);
    if (s == 15 && src_border) load_frameL_padded<LL>(a, src_clip, (int)(src - src_clip), lane & (LL - 1), raw);
  }
};

template <int LL, bool ALIGNED>
__global__ void __launch_bounds__(512) stft_complex_lanes_kernel(FastArgs a) {
  using P = PL<LL>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const LdsL<LL> lds = carve_ldsL<LL>(smem);
  const LaneL L = setup_laneL<LL>(lds, lane, wave);
  fill_tablesL<LL>(a, lds, tid, 512);
  TileWalk tw;
  tw.init(a, a.out + 2 * a.out_offset, 2 * (int64_t)P::Bins * a.out_stride);
  const int ntiles = tw.ntiles > 0 ? tw.ntiles : 0;
  auto frame_ptr = [&](const float *xc, int t) {   // as stft_power_lanes_kernel
    const int64_t f0 = (int64_t)t * P::FT;
    const int avail = (int)(a.count - f0 < P::FT ? a.count - f0 : P::FT) - 1;
    const int fi = L.col;
    const int64_t p = a.p0 + f0 + (fi <= avail ? fi : 0);
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
  auto tile_border = [&](int t) {   // fold_frames == 2 (as stft2048_power32_kernel): the tile takes load_frameL_padded
    const int64_t q0 = a.p0 + (int64_t)t * P::FT;
    return a.fold_frames == 2 && (q0 < a.border_i0 || q0 + P::FT > a.border_i1);
  };
  if (ntiles > 0) {
    const float *src0 = frame_ptr(tw.xclip, tw.ft);
    if (tile_border(tw.ft)) load_frameL_padded<LL>(a, tw.xclip, (int)(src0 - tw.xclip), L.l, raw);
    else load_frameL<LL, ALIGNED>(src0, L.l, raw);
  }
  __syncthreads();
  const CplxFlushL fl = setup_cplx_flushL<LL>(a, lane, wave);
  float *pend_out = nullptr;
  int pend_left = 0;
  for (int it = 0; it < ntiles; ++it) {
    int ftnext;
    const float *xnext;
    float *onext;
    tw.peek(a, ftnext, xnext, onext);
    const bool more = it + 1 < ntiles;
    const float *src_clip = more ? xnext : tw.xclip;
    const float *src = frame_ptr(src_clip, more ? ftnext : tw.ft);
    const bool src_border = tile_border(more ? ftnext : tw.ft);
    const CplxMidL<LL, ALIGNED> mid{a, lds, fl, raw, src, src_clip, src_border, pend_out, pend_left, lane, wave, it};
    frameL_to_tile<LL, 2, CplxMidL<LL, ALIGNED>, true>(a, L, raw, lds.tiles, mid);
    lds_signal32(lds.filled, lane);
    pend_out = tw.oclip + 2 * tw.ft * P::FT;
    const int64_t left = a.count - (int64_t)tw.ft * P::FT;
    pend_left = left < P::FT ? (int)left : P::FT;
    tw.xclip = xnext;
    tw.oclip = onext;
    tw.ft = ftnext;
  }
  if (ntiles > 0) {
    lds_wait(lds.filled, 8u * (unsigned)ntiles);
    cplx_flushL<LL>(a, lds.tiles, fl, pend_out, pend_left, wave, lane);
  }
}
