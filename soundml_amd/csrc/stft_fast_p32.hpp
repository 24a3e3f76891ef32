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
//      these accesses bank-conflict free (bank = 17 l + const mod 32 within a half-wave; a pitch of 33 cells needs 1056 rows
//      per column instead of 1024).  The one conflict of the pipeline is in P's exchange reads: lane 0 looks for its own
//      register one cell column further than the rule of the other lanes (cell 33, bank of lane 31's cell 1), a 2-way
//      conflict on those 32 reads per frame = the 8 % of LDS cycles SQ_LDS_BANK_CONFLICT shows; removing it needs a select
//      per parked register (32 per frame), which costs more vector time than the 64 LDS cycles it frees.
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
  // tables, read two complex values (16 bytes) per lane and instruction:
  //   win4[m][l] = window pairs of points l + 32 j, l + 32 (j + 2); j = 4 m (rows 0..7, the even points), j = 4 (m - 8) + 1 (rows 8..15)
  //   twA4[m][l] = W_M^(l k1) for k1 = 2m + 1, 2m + 2 (m < 15), then one row of k1 = 31    (15 x 32 float4 + 32 float2)
  //   twP4[m][l] = exp(-2 pi i k / N) for k = l + 32 (2m), l + 32 (2m + 1)                 m < 8
  float4 *win4, *twA4, *twP4;
  float2 *twA31;
  unsigned *filled, *drained;   // [2] each, kTileStride floats apart (pad cells of rows 1040..1043 of buffer 0)
};
__device__ __forceinline__ Lds32 carve_lds32(unsigned char *smem) {
  Lds32 l;
  l.tiles = reinterpret_cast<float *>(smem);
  l.win4 = reinterpret_cast<float4 *>(smem + 2 * kTile32Bytes);
  l.twA4 = reinterpret_cast<float4 *>(smem + 2 * kTile32Bytes + kWinBytes);
  l.twA31 = reinterpret_cast<float2 *>(smem + 2 * kTile32Bytes + kWinBytes + 15 * 32 * sizeof(float4));
  l.twP4 = reinterpret_cast<float4 *>(smem + 2 * kTile32Bytes + kWinBytes + kTwA32Bytes);
  l.filled = reinterpret_cast<unsigned *>(l.tiles + 1040 * kTileStride + kFT);
  l.drained = reinterpret_cast<unsigned *>(l.tiles + 1042 * kTileStride + kFT);
  return l;
}

// Per-lane constants.  Only the lane's indices are kept in registers; the offsets into a tile buffer and the table addresses
// are re-derived where they are used (a few integer instructions per tile): ten loop-invariant address registers beside the
// frame's 64, the samples' 64 and the carried output pairs spill, and every scratch reload waits for ALL of the wave's
// outstanding memory operations.
struct Lane32 {
  int l, h, col;
  const float4 *win0, *twA0, *twP0;   // the tables (wave-uniform)
  const float2 *twA310;
  // offsets (floats) into a tile buffer:
  __device__ __forceinline__ int li() const { int v = l; asm volatile("" : "+v"(v)); return v; }
  __device__ __forceinline__ int own() const { return li() * kTileStride + col; }                 // cell l of the frame's column: transposition / exchange writes (cell l + 33 j), results of bins l + 32 s
  __device__ __forceinline__ int rd() const { return 33 * kTileStride * li() + col; }             // cell 33 l: transposition reads (cell i + 33 l)
  __device__ __forceinline__ int xr() const {                                                     // exchange reads: cell p + 33 (15 - s) of slot s, p = 32 - l (l = 0: 33, i.e. its own register 32 - s)
    const int v = li();
    return (33 - v - (v < 1 ? v : 1)) * kTileStride + col;
  }
  __device__ __forceinline__ int xr2() const {                                                    // exchange reads, cells 33 p + (15 - s) of slot s, p = 32 - l (l = 0: its own register 32 - s in cell 16 - s, i.e. base cell 1)
    const int v = li();
    return (v < 1 ? 1 : 33 * (32 - v)) * kTileStride + col;
  }
  __device__ __forceinline__ int rm() const { return (32 + 32 * 16 - li()) * kTileStride + col; }   // results of bins M - k: row (32 - l) + 32 (31 - s)  (l = 0: 32 (32 - s); s = 0 is row 1024 = Nyquist)
  __device__ __forceinline__ int self() const {                                                   // lane 0: row 512 (bin M/2); other lanes: a cell they overwrite afterwards
    const int v = li();
    return (v + 512 * (1 - (v < 1 ? v : 1))) * kTileStride + col;
  }
  __device__ __forceinline__ const float4 *win_l() const { return win0 + li(); }
  __device__ __forceinline__ const float4 *twA_l() const { return twA0 + li(); }
  __device__ __forceinline__ const float4 *twP_l() const { return twP0 + li(); }
  __device__ __forceinline__ const float2 *twA31_l() const { return twA310 + li(); }
};
__device__ __forceinline__ Lane32 setup_lane32(const Lds32 &lds, int lane, int wave) {
  Lane32 L;
  L.l = lane & 31;
  L.h = lane >> 5;
  L.col = 2 * wave + L.h;
  L.win0 = lds.win4;
  L.twA0 = lds.twA4;
  L.twA310 = lds.twA31;
  L.twP0 = lds.twP4;
  return L;
}

// A lane's loop-invariant value that must be re-derived where it is used: hipcc otherwise hoists every address built from it
// (one per tile buffer and store part) out of the tile loop, runs out of registers and parks them in scratch memory --
// and every reload waits for ALL of the wave's outstanding memory operations.
__device__ __forceinline__ int opaque32(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ unsigned opaque32(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}

// Counter traffic of this kernel.  The LDS executes one wave's operations in order, so a signal needs no wait for the
// wave's earlier LDS accesses (a release would drain them: a ~1000-cycle round trip while eight waves keep the queue
// full), and a counter can be READ long before it is needed: peek32() issues the read, the value is looked at a stage
// later, and only a wave that finds it short falls back to polling.
__device__ __forceinline__ void lds_signal32(unsigned *c, int lane) {
  if (lane == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned peek32(unsigned *c) {
  return __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_wait32(unsigned *c, unsigned target, unsigned peeked) {
  if ((unsigned)__builtin_amdgcn_readfirstlane((int)peeked) >= target) return;
  lds_wait(c, target);
}

// two values into cells OFF0 and OFF1 dwords beyond the LDS byte address `at`: 6 cycles of the LDS store path where two
// ds_write_b32 take 8 (MI355X_MICROARCH.md, LDS: 2 cycles per source dword).  Invisible to hipcc's lgkmcnt bookkeeping, which
// only makes its counted waits stricter (the LDS executes a wave's operations in order).
template <int OFF0, int OFF1>
__device__ __forceinline__ void lds_write2_b32(unsigned at, float x, float y) {
  static_assert(OFF0 >= 0 && OFF0 < 256 && OFF1 >= 0 && OFF1 < 256, "ds_write2_b32 offsets are 8 bits of dwords");
  asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" : : "v"(at), "v"(x), "v"(y), "n"(OFF0), "n"(OFF1) : "memory");
}

// The workgroup's tables, 512 threads: tables32_request() asks for a thread's share (registers), tables32_commit() puts it into
// LDS; the caller synchronises before the tables are read.  In two steps so that the requests are ONE round trip that the tile
// walk's divisions and the first frames' sample requests overlap (round 5: as four loops of load -> LDS store the fill was four
// dependent round trips, 5.8 us of a C2 launch -- profiles/r07/timeline_before.log).
struct Tables32Regs {
  float2 w0, w1, a0, a1, a31, p0, p1;
};
__device__ __forceinline__ void tables32_request(const FastArgs &a, int tid, Tables32Regs &r) {
  const float2 *hw = reinterpret_cast<const float2 *>(a.hwin);
  {   // win4: rows 0..7: the even points j = 4 m, 4 m + 2; rows 8..15: the odd points j = 4 m + 1, 4 m + 3
    const int row = tid >> 5, l = tid & 31, j0 = 4 * (row & 7) + (row >> 3);
    r.w0 = hw[l + 32 * j0];
    r.w1 = hw[l + 32 * (j0 + 2)];
  }
  {   // twA4 (the threads past its 15 rows re-read row 0; not stored)
    const int m = tid < 15 * 32 ? tid >> 5 : 0, l = tid & 31;
    r.a0 = a.w_m[l * (2 * m + 1)];
    r.a1 = a.w_m[l * (2 * m + 2)];
  }
  r.a31 = a.w_m[(tid & 31) * 31];
  {   // twP4 (8 rows)
    const int m = (tid >> 5) & 7, l = tid & 31;
    r.p0 = a.w_n[l + 32 * (2 * m)];
    r.p1 = a.w_n[l + 32 * (2 * m + 1)];
  }
}
__device__ __forceinline__ void tables32_commit(const Lds32 &lds, int tid, const Tables32Regs &r) {
  lds.win4[tid] = make_float4(r.w0.x, r.w0.y, r.w1.x, r.w1.y);
  if (tid < 15 * 32) lds.twA4[tid] = make_float4(r.a0.x, r.a0.y, r.a1.x, r.a1.y);
  if (tid < 32) lds.twA31[tid] = r.a31;
  if (tid < 8 * 32) lds.twP4[tid] = make_float4(r.p0.x, r.p0.y, r.p1.x, r.p1.y);
  if (tid < 2) { lds.filled[tid * kTileStride] = 0u; lds.drained[tid * kTileStride] = 0u; }
}
__device__ __forceinline__ void fill_tables32(const FastArgs &a, const Lds32 &lds, int tid, int /* 512 */) {
  Tables32Regs r;
  tables32_request(a, tid, r);
  tables32_commit(lds, tid, r);
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
struct NoHalf32 {
  __device__ __forceinline__ void operator()() const {}
};
// `half()` is called between the two 16-point transforms (the power kernel reads a counter there)
template <class Half = NoHalf32>
__device__ __forceinline__ void fft32(c32 (&v)[32], const Half &half = Half{}) {
#pragma clang fp contract(off)
  constexpr float c1 = (float)0.98078528040323043, s1 = (float)0.19509032201612825;
  constexpr float c2 = (float)0.92387953251128674, s2 = (float)0.38268343236508977;
  constexpr float c3 = (float)0.83146961230254524, s3 = (float)0.55557023301960218;
  constexpr float hh = (float)0.70710678118654752;
  c32 e[16], o[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) { e[m] = v[2 * m]; o[m] = v[2 * m + 1]; }
  p32_fft16(e);
  half();
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

// ---- the same arithmetic on packed pairs (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32; round 4) -------------------------------
// A complex value is an aligned register pair (re, im).  One wave issues one instruction per ~4.2 cycles whatever it is
// (profiles/r05/issue_probe.log) and two waves per SIMD leave the vector pipe idle more than half the time, so what bounds
// the frame pipeline is the length of a wave's own instruction stream: a packed instruction does two of the operations
// above in one issue slot.  Every packed operation is the IEEE operation of its two halves (v_pk_fma_f32 is a fused
// multiply-add per half), the operations and their order are those of p32_cmul / p32_fft4 / p32_fft16 / fft32 above, and a
// product by -i is folded into the operand selects of the instruction that consumes it: the results are the same bits.
// The stages are generated as one inline-assembly statement each (tools/gen/gen_pk_fft.py says why).
using f2 = float __attribute__((ext_vector_type(2)));
#include "stft_pk_fft.inc"
template <class Half = NoHalf32>
__device__ __forceinline__ void pk_fft32(f2 (&v)[32], const Half &half = Half{}) {
  f2 e[16], o[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) { e[m] = v[2 * m]; o[m] = v[2 * m + 1]; }
  pk_fft16(e);
  half();
  pk_fft16(o);
  pk_fft32_combine0(v, e, o);
  pk_fft32_combine1(v, e, o);
}

struct NoMid32 {
  __device__ __forceinline__ void early() const {}
  __device__ __forceinline__ void before_cells() const {}
  __device__ __forceinline__ void after_transposition_issue() const {}
  __device__ __forceinline__ void after_exchange_issue() const {}
  __device__ __forceinline__ void postpass_at(int) const {}
  template <int I> __device__ __forceinline__ void stamp() const {}
};

// Two frames (one per lane-half): raw samples (registers) -> window -> FFT(1024 complex) -> post-pass -> |X|^p in the
// frames' columns of `tile`.  The caller's `mid` is given the three places where the wave would otherwise only wait for
// the LDS (a round trip takes ~1000 cycles while eight waves keep its queue full):
//   before_cells()               before the first access to the columns (the power kernel waits until the buffer is free);
//   after_transposition_issue()  the transposition's reads are in flight;
//   after_exchange_issue()       the partner reads are in flight: the previous tile's LDS reads are issued behind them;
//   postpass_at(s)               slot s of the post-pass is done: the previous tile's stores go out behind slot 1 and the next
//                                frames' loads behind slot 5, each as ONE run of instructions (measured, profiles/r05: stores
//                                and loads together behind slot 7 +2 %; loads before stores no better; one store and two loads
//                                behind every slot +18 % -- a memory instruction between arithmetic holds the wave at its issue
//                                every time, a run of them once).
// (The power kernel needs the previous tile complete only at the third point and its buffer free only at the first one:
// the two waves of a SIMD run half a frame apart by themselves -- one computes while the other waits for the LDS -- and
// every wait that asks for less slack than that turns the counters into a barrier: 0.56 -> 0.58 ms when the tile's reads
// sat behind the transposition, profiles/r05.)
#ifndef SMX_P32_HAVE_BRANCH
#define SMX_P32_HAVE_BRANCH 0
#endif
#ifndef SMX_P32_STORE_AT
#define SMX_P32_STORE_AT 1
#endif
#ifndef SMX_P32_LOAD_AT
#define SMX_P32_LOAD_AT 5
#endif
#ifndef SMX_P32_LOADS_FIRST
#define SMX_P32_LOADS_FIRST 0
#endif
#ifndef SMX_P32_WAIT0
#define SMX_P32_WAIT0 1
#endif
#ifndef SMX_P32_CELLS
#define SMX_P32_CELLS 1   // 1: cell rule 33 l + j (ds_write2_b32 on the store side, round 5); 0: rounds 3-4's rule l + 33 j (A/B builds)
#endif
// The window rows of the odd points and the twiddles are requested before the first 16-point transform (which needs the even
// points only) and arrive under it.  (Requesting the even rows a frame pair ahead as well -- 32 registers across the loop
// edge -- changed nothing in the power kernel and cost the fused mel kernel 40 %: profiles/r06/ab_mel_winpre2.log.)
// TWFIRST: twiddle rows requested before the first 16-point transform (the rest behind it): 15 where the registers allow
// (their late arrival sits on the critical path of a kernel whose LDS queue is long), fewer in the 12-wave kernels whose
// frame waves must fit 168 registers.
// CPLX: 0 |X|^p into the tile; 1 the spectrum into the tile's two planes; 2 the spectrum handed to mid.emit() / mid.emit_self() slot
// by slot (Griffin-Lim's frame-major spectra: nothing of a frame stays in LDS, so waves never meet)
template <int PMODE, class Mid, int CPLX = 0, int TWFIRST = 15>
__device__ __forceinline__ void frame32_to_tile(const FastArgs &a, const Lane32 &L, float2 (&raw)[32], float *tile,
                                                const Mid &mid) {
#pragma clang fp contract(off)
  f2 v[32], t[32];
#if SMX_P32_WAIT0
  // the samples were requested most of a tile ago: one wait for all of them instead of one per product (a wait is an issue slot)
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
#endif
  f2 e[16], o[16];
  float4 winE[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) winE[m] = L.win_l()[32 * m];
#pragma unroll
  for (int m = 0; m < 8; ++m) {   // even points j = 4 m, 4 m + 2 -> e[2 m], e[2 m + 1]
    e[2 * m] = f2{raw[4 * m].x, raw[4 * m].y} * f2{winE[m].x, winE[m].y};
    e[2 * m + 1] = f2{raw[4 * m + 2].x, raw[4 * m + 2].y} * f2{winE[m].z, winE[m].w};
  }
  SMX_FENCE();
  float4 winO[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) winO[m] = L.win_l()[32 * (8 + m)];
  mid.template stamp<1>();
  SMX_FENCE();
  // A: radix-32 over j, then twiddle W_M^(l k1) (requested only after the first transform -- 650 cycles of a wave's own issue
  // before they are used -- the second half of the rows cost the fused mel kernel 8 %)
  {
    float4 tw[15];
#pragma unroll
    for (int m = 0; m < TWFIRST; ++m) tw[m] = L.twA_l()[32 * m];
    float2 tw31;
    if constexpr (TWFIRST == 15) tw31 = L.twA31_l()[0];
    SMX_FENCE();
    pk_fft16(e);
    SMX_FENCE(); mid.early(); SMX_FENCE();
    if constexpr (TWFIRST < 15) {
#pragma unroll
      for (int m = TWFIRST; m < 15; ++m) tw[m] = L.twA_l()[32 * m];
      tw31 = L.twA31_l()[0];
      SMX_FENCE();
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {   // odd points j = 4 m + 1, 4 m + 3 -> o[2 m], o[2 m + 1]
      o[2 * m] = f2{raw[4 * m + 1].x, raw[4 * m + 1].y} * f2{winO[m].x, winO[m].y};
      o[2 * m + 1] = f2{raw[4 * m + 3].x, raw[4 * m + 3].y} * f2{winO[m].z, winO[m].w};
    }
    SMX_FENCE();
    pk_fft16(o);
    pk_fft32_combine0(v, e, o);
    pk_fft32_combine1(v, e, o);
    SMX_FENCE();
    // X: lane l register k1 -> lane k1 register l through the frame's column, real parts then imaginary parts.
    // One wave's LDS operations execute in order, so no wait separates the rounds.
    mid.template stamp<2>();
    mid.before_cells();
    mid.template stamp<3>();
#if SMX_P32_CELLS
    // (round 5: lane l writes register j to cell 33 l + j and lane k1 reads register l' from cell 33 l' + k1 -- the transpose of
    // rounds 3-4's cell rule, conflict free on both sides as that one.  Consecutive registers now sit 68 bytes apart, within the
    // reach of ds_write2_b32: 6 cycles of the LDS store path for two values where two ds_write_b32 take 8
    // (profiles/r07/lds_pattern_probe.log: the transposition's 64 writes + 64 reads 3.0 -> 2.6 cycles per value and CU).)
    float *const wc = tile + L.rd();
    // the first plane is written while the twiddle products are formed (shorter LDS bursts)
    auto put2 = [&](int j) { wc[kTileStride * j] = v[j].x; wc[kTileStride * (j + 1)] = v[j + 1].x; };
#define SMX_TWV(m) f2{tw[m].x, tw[m].y}, f2{tw[m].z, tw[m].w}
    pk_twiddle8(v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], SMX_TWV(0), SMX_TWV(1), SMX_TWV(2), SMX_TWV(3));
#pragma unroll
    for (int j = 0; j < 8; j += 2) put2(j);
    SMX_FENCE();
    pk_twiddle8(v[9], v[10], v[11], v[12], v[13], v[14], v[15], v[16], SMX_TWV(4), SMX_TWV(5), SMX_TWV(6), SMX_TWV(7));
#pragma unroll
    for (int j = 8; j < 16; j += 2) put2(j);
    SMX_FENCE();
    pk_twiddle8(v[17], v[18], v[19], v[20], v[21], v[22], v[23], v[24], SMX_TWV(8), SMX_TWV(9), SMX_TWV(10), SMX_TWV(11));
#pragma unroll
    for (int j = 16; j < 24; j += 2) put2(j);
    SMX_FENCE();
    pk_twiddle7(v[25], v[26], v[27], v[28], v[29], v[30], v[31], SMX_TWV(12), SMX_TWV(13), SMX_TWV(14), f2{tw31.x, tw31.y});
#pragma unroll
    for (int j = 24; j < 32; j += 2) put2(j);
#else
    float *const wr = tile + L.own();
    float *const wr_hi = wr + 16 * kCellPitch32;
    // the first plane is written while the twiddle products are formed (shorter LDS bursts)
    auto put = [&](int j) { (j < 16 ? wr : wr_hi)[kCellPitch32 * (j & 15)] = v[j].x; };
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
#endif
#undef SMX_TWV
  }
  SMX_FENCE();
#if SMX_P32_CELLS
  float *const wr = tile + L.own();
  float *const wr_hi = wr + 16 * kCellPitch32;   // (ds offsets are 16 bits: 31 x 2244 bytes does not fit)
  float *const wc = tile + L.rd();
#pragma unroll
  for (int i = 0; i < 32; ++i) t[i].x = (i < 16 ? wr : wr_hi)[kCellPitch32 * (i & 15)];
#pragma unroll
  for (int j = 0; j < 32; ++j) wc[kTileStride * j] = v[j].y;
#pragma unroll
  for (int i = 0; i < 32; ++i) t[i].y = (i < 16 ? wr : wr_hi)[kCellPitch32 * (i & 15)];
#else
  float *const wr = tile + L.own();
  float *const wr_hi = wr + 16 * kCellPitch32;   // (ds offsets are 16 bits: 31 x 2244 bytes does not fit)
  const float *const rd = tile + L.rd();
#pragma unroll
  for (int i = 0; i < 32; ++i) t[i].x = rd[kTileStride * i];
#pragma unroll
  for (int j = 0; j < 32; ++j) (j < 16 ? wr : wr_hi)[kCellPitch32 * (j & 15)] = v[j].y;
#pragma unroll
  for (int i = 0; i < 32; ++i) t[i].y = rd[kTileStride * i];
#endif
  SMX_FENCE();
  mid.after_transposition_issue();
  mid.template stamp<4>();
  SMX_FENCE();
  // B: radix-32 over l
  pk_fft32(t);
  SMX_FENCE();
  mid.template stamp<5>();
  // P: partners through the cells.  Every lane parks registers 16..31 (cell l + 33 (q - 16)) and reads, for slot s,
  // register 31 - s of lane 32 - l (lanes 0 and 16: their own; lane 0: register 32 - s, and itself for s = 0).
#if SMX_P32_CELLS
  f2 pp[16];
  const float *const xr = tile + L.xr2();
  // (register 0 goes to cell 33 l + 16 as well: that is where lane 0 looks for the partner of bin 0 -- itself; slot 0
  // of that lane yields X[0] and the Nyquist bin.  A select instead would be two v_cndmask_b32 on vcc, 19 cycles each.)
  // Cells 33 l + (q - 16): consecutive registers 68 bytes apart (ds_write2_b32, as the transposition).
  // (written as ds_write2_b32 by hand: hipcc pairs the transposition's stores by itself and leaves these seventeen single)
  const unsigned wc_lds = (unsigned)reinterpret_cast<uintptr_t>(wc);
#define SMX_PARK2(k, m) lds_write2_b32<2 * kTileStride * (k), 2 * kTileStride * (k) + kTileStride>(wc_lds, t[16 + 2 * (k)].m, t[17 + 2 * (k)].m)
  SMX_PARK2(0, x); SMX_PARK2(1, x); SMX_PARK2(2, x); SMX_PARK2(3, x); SMX_PARK2(4, x); SMX_PARK2(5, x); SMX_PARK2(6, x); SMX_PARK2(7, x);
  wc[kTileStride * 16] = t[0].x;
#pragma unroll
  for (int s = 0; s < 16; ++s) pp[s].x = xr[kTileStride * (15 - s)];
  SMX_PARK2(0, y); SMX_PARK2(1, y); SMX_PARK2(2, y); SMX_PARK2(3, y); SMX_PARK2(4, y); SMX_PARK2(5, y); SMX_PARK2(6, y); SMX_PARK2(7, y);
#undef SMX_PARK2
  wc[kTileStride * 16] = t[0].y;
#pragma unroll
  for (int s = 0; s < 16; ++s) pp[s].y = xr[kTileStride * (15 - s)];
#else
  f2 pp[16];
  const float *const xr = tile + L.xr();
  // (register 0 goes to cell l + 33 x 16 as well: that is where lane 0 looks for the partner of bin 0 -- itself; slot 0
  // of that lane yields X[0] and the Nyquist bin.  A select instead would be two v_cndmask_b32 on vcc, 19 cycles each.)
#pragma unroll
  for (int q = 16; q < 32; ++q) wr[kCellPitch32 * (q - 16)] = t[q].x;
  wr_hi[0] = t[0].x;
#pragma unroll
  for (int s = 0; s < 16; ++s) pp[s].x = xr[kCellPitch32 * (15 - s)];
#pragma unroll
  for (int q = 16; q < 32; ++q) wr[kCellPitch32 * (q - 16)] = t[q].y;
  wr_hi[0] = t[0].y;
#pragma unroll
  for (int s = 0; s < 16; ++s) pp[s].y = xr[kCellPitch32 * (15 - s)];
#endif
  float4 tw[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) tw[m] = L.twP_l()[32 * m];
  SMX_FENCE();
  mid.after_exchange_issue();
  mid.template stamp<6>();
  SMX_FENCE();
  // CPLX: the spectrum itself, real parts in `tile`, imaginary parts in the plane after it (the other tile buffer)
  {   // bin M/2 (lane 0, register 16): X = 2 conj(Z)
    const f2 z = t[16] + t[16];
    if constexpr (CPLX == 2) {
      mid.emit_self(z.x, -z.y);
    } else if constexpr (CPLX == 1) {
      tile[L.self()] = z.x;
      tile[kTile32Floats + L.self()] = -z.y;
    } else {
      tile[L.self()] = power_from_square<PMODE>(__builtin_fmaf(z.x, z.x, z.y * z.y), a);
    }
  }
  float *const rk = wr;                 // row l + 32 s
  float *const rm = tile + L.rm();        // row (32 - l) + 32 (31 - s) = rm base + 32 (15 - s)
  // slot s: E = Z[k] + conj Z[M-k], D = Z[k] - conj Z[M-k], T = -i w D, X[k] = E + T, X[M-k] = conj(E - T); the generated
  // blocks return them as planes (re_k, re_(M-k)), (im_k, im_(M-k)) or as (|X_k|^2, |X_(M-k)|^2) = fma(re, re, im im) per half
  auto wtw = [&](int s) { return (s & 1) ? f2{tw[s >> 1].z, tw[s >> 1].w} : f2{tw[s >> 1].x, tw[s >> 1].y}; };
  auto put = [&](int s, f2 re, f2 im) {
    if constexpr (CPLX == 2) {
      mid.emit(s, re, im);
    } else if constexpr (CPLX == 1) {
      rk[kRowPitch32 * s] = re.x;
      rk[kTile32Floats + kRowPitch32 * s] = im.x;
      rm[kRowPitch32 * (15 - s)] = re.y;
      rm[kTile32Floats + kRowPitch32 * (15 - s)] = im.y;
    } else {
      rk[kRowPitch32 * s] = power_from_square<PMODE>(re.x, a);
      rm[kRowPitch32 * (15 - s)] = power_from_square<PMODE>(re.y, a);
    }
  };
#define SMX_PA(s) t[s], pp[s], wtw(s)
  static_assert(SMX_P32_STORE_AT == 1 && SMX_P32_LOAD_AT == 5, "the post-pass blocks end at slots 1, 5, 10, 15");
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
  SMX_FENCE(); mid.postpass_at(15); SMX_FENCE();   // (the complex kernel with carried lines requests its samples here: no registers before)
  mid.template stamp<7>();
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

// The same for a tile with frames that reach past either end of the signal (round 5: the border frames ride in the tile
// sequence instead of an epilogue of their own -- 30 us at the end of a 0.49 ms C2 launch on a quarter of the workgroups,
// profiles/r07/timeline_before.log).  Sample s of the clip at xc extended by the configuration's rule (stft.ml:300-338), for
// positions within ONE reflection of the signal (the launcher admits n >= fft_size only): reflect min(|s|, 2 (n - 1) - |s|),
// edge clamp(s), constant pad_value where s is outside; four-byte loads at clamped indices, so any tile may take this path.
// The values are those of fetch_padded: a frame has one value whichever path loads it.
__device__ __forceinline__ void load_frame32_padded(const FastArgs &a, const float *xc /* wave-uniform */, int s0, int l, float2 (&raw)[32]) {
  asm volatile("" : "+v"(s0));   // the index arithmetic below starts HERE (hoisted to the loop top it would hold registers through the frame)
  const int n = (int)a.n, top = 2 * (n - 1);
  const bool refl = a.pad == SMX_PAD_REFLECT;
  auto at = [&](int s) {   // (an unsigned index: the load takes the clip's scalar base and a 32-bit offset, no 64-bit address pair per sample)
    const int m = s < 0 ? -s : s;
    const int ir = m < top - m ? m : top - m;
    const int ie = s < 0 ? 0 : (s < n ? s : n - 1);
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(xc) + 4u * (unsigned)(refl ? ir : ie));
  };
  const int s00 = s0 + 2 * l;
#pragma unroll
  for (int j = 0; j < 32; ++j) raw[j] = make_float2(at(s00 + 64 * j), at(s00 + 64 * j + 1));
  if (a.pad != SMX_PAD_REFLECT && a.pad != SMX_PAD_EDGE) {   // constant padding (wave-uniform): the one mode that waits for the loads here
    const float pv = a.pad_value;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int s = s00 + 64 * j;
      if ((unsigned)s >= (unsigned)n) raw[j].x = pv;
      if ((unsigned)(s + 1) >= (unsigned)n) raw[j].y = pv;
    }
  }
}

// A wave's share of a finished tile: 8 parts of 16 rows (bins) x 4 frames per lane -> out[clip][bin][f0 + 4 g ..].
// Rows {0-3, 16-19} + 4 h per half-wave keep the LDS reads conflict free; a 4-lane group stores one 64-byte run.
// The tile is read into registers at one point of the frame and stored at a later one (see frame32_to_tile).
struct Flush32 {
  int src0;         // float offset in the tile of part 0: row0 * 17 + 4 g, row0 = 128 wave + (jj & 3) + 16 (jj >> 2) + 4 h
  unsigned goff0;   // byte offset of out[row0][4 g] from the tile's origin
  int g;
};
struct FlushRegs {
  float v[8][4];
  float nyq;        // bin 1024 (wave 0, lanes 0..15)
};
__device__ __forceinline__ void flush32_read(const float *tile, const Flush32 &fl, int wave, int lane, FlushRegs &r) {
  const float *src0 = tile + opaque32(fl.src0);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const float *src = src0 + (32 * (it >> 1) + 8 * (it & 1)) * kTileStride;
    r.v[it][0] = src[0]; r.v[it][1] = src[1]; r.v[it][2] = src[2]; r.v[it][3] = src[3];
  }
  r.nyq = tile[kM * kTileStride + (lane & 15)];   // row 1024 = Nyquist bin (every wave reads it, wave 0 stores it)
}
__device__ __forceinline__ void flush32_store(const FastArgs &a, const Flush32 &fl, float *obase, int frames_left, int wave,
                                              int lane, const FlushRegs &r) {
#ifdef SMX_DIAG
  if (a.abl_nostore == 1) {   // timing-only ablation: keep the LDS reads alive, drop the HBM stores
#pragma unroll
    for (int it = 0; it < 8; ++it) asm volatile("" ::"v"(r.v[it][0]), "v"(r.v[it][1]), "v"(r.v[it][2]), "v"(r.v[it][3]));
    asm volatile("" ::"v"(r.nyq));
    return;
  }
#endif
  const unsigned pitch = (unsigned)a.out_stride * 4u;
  const unsigned goff0 = opaque32(fl.goff0);
#ifdef SMX_DIAG
  if (a.abl_p32 & 6) {   // timing only (wrong places): runs on 64-byte boundaries / everything into the first tile's rows
    char *base = reinterpret_cast<char *>((a.abl_p32 & 4) ? a.out + a.out_offset : obase);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      uintptr_t ad = reinterpret_cast<uintptr_t>(base) + goff0 + (unsigned)(32 * (it >> 1) + 8 * (it & 1)) * pitch;
      if (a.abl_p32 & 2) ad = ((ad - 16u * fl.g) & ~(uintptr_t)63) + 16u * fl.g;
      *reinterpret_cast<float4 *>(ad) = make_float4(r.v[it][0], r.v[it][1], r.v[it][2], r.v[it][3]);
    }
    return;
  }
#endif
  if (frames_left >= kFT) {   // wave-uniform: a whole tile, no masks
#pragma unroll
    for (int it = 0; it < 8; ++it)
      store4_unaligned(obase, goff0 + (unsigned)(32 * (it >> 1) + 8 * (it & 1)) * pitch, r.v[it][0], r.v[it][1], r.v[it][2], r.v[it][3]);
  } else {
    const int fleft = frames_left - 4 * fl.g;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const unsigned goff = goff0 + (unsigned)(32 * (it >> 1) + 8 * (it & 1)) * pitch;
      if (fleft >= 4) {
        store4_unaligned(obase, goff, r.v[it][0], r.v[it][1], r.v[it][2], r.v[it][3]);
      } else {
        float *dst = reinterpret_cast<float *>(reinterpret_cast<char *>(obase) + goff);
        if (fleft > 0) dst[0] = r.v[it][0];
        if (fleft > 1) dst[1] = r.v[it][1];
        if (fleft > 2) dst[2] = r.v[it][2];
      }
    }
  }
  if (wave == 0 && lane < 16 && lane < frames_left) obase[(int64_t)kM * a.out_stride + lane] = r.nyq;
}

// ---- the flush in whole, aligned 64-byte blocks (round 4; `SKEW` kernels) --------------------------------------------
// Measured with nothing else running (tools/probes/store_shape_probe, profiles/r06/store_shape_probe.log): the 0.97 GB of a
// C2 spectrogram written as the tiles above write it -- a 64-byte run per row and tile, rows 3752 bytes apart, so that seven
// runs of eight straddle a 64-byte boundary -- take 0.42-0.46 ms: the memory side completes every partial 32-byte sector by
// a read-modify-write.  The same bytes in whole aligned 64-byte blocks take 0.27-0.29 ms (aligned 128-byte blocks 0.185).
// The kernel's 0.51 ms were that floor, not its arithmetic.  So a row's run of a tile is held back until it completes a
// block: row r's blocks begin e(r) frames into a tile (e even when the row pitch and the origin are: the launcher checks),
// the lanes of the first e frames of a row store the tile's values at once, the lanes of frames e..15 store what they
// carried from the previous tile 64 bytes further down -- one instruction, one whole block per row -- and carry theirs.
// A lane holds a pair of frames (8 lanes a row, 8 rows an instruction: rows {0,1,16,17} (+2 for the upper half-wave) keep
// the LDS reads conflict free); its 16 parts are rows row0 + 4 (p & 1) + 8 ((p >> 1) & 1) + 32 (p >> 2), and rows 8 apart
// share an alignment when the pitch is even, so a lane has two block offsets (p & 1), set once per clip.
// The workgroup must walk consecutive tiles of a clip (TileWalk's contiguous ranges); where a range or a clip begins there
// is nothing carried (those lanes store nothing), where it ends the carried pairs go out as the partial block they are.
struct Skew32 {
  unsigned goffc[2];           // part class c, from 64 bytes BEFORE a tile's origin (the offset register of a store is unsigned): byte offset of out[row0 + 4 c][2 g2] from a tile's origin, plus 64 for the lanes that store the current pair
  unsigned long long sel[2];   // part class c: the lanes whose pair of the current tile completes the block
};
struct SkewRegs {
  float2 cur[16];
  float nyq;
};
// (the lane's row and pair are re-derived from the lane index where they are used: see Lane32)
__device__ __forceinline__ void skew32_lane(int lane, int wave, int &row0, int &g2) {
  asm volatile("" : "+v"(lane));
  const int rsel = lane >> 3;
  g2 = lane & 7;
  row0 = 128 * wave + (rsel & 1) + 16 * ((rsel >> 1) & 1) + 2 * (rsel >> 2);
}
// the block offsets of the clip whose output begins at oclip (every tile origin of a clip is oclip + 16 t floats)
__device__ __forceinline__ void skew32_clip(const FastArgs &a, Skew32 &sk, const float *oclip, int lane, int wave) {
  int row0, g2;
  skew32_lane(lane, wave, row0, g2);
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const unsigned at = (unsigned)((reinterpret_cast<uintptr_t>(oclip) >> 2) + (uintptr_t)(row0 + 4 * c) * (uintptr_t)a.out_stride) & 15u;
    const unsigned e = at ? 16u - at : 16u;          // frames of a tile that end the row's open block
    const bool now = 2u * g2 < e;
    sk.sel[c] = __ballot(now);
    sk.goffc[c] = ((unsigned)(row0 + 4 * c) * (unsigned)a.out_stride + 2u * g2) * 4u + (now ? 64u : 0u);
  }
}
__device__ __forceinline__ constexpr int skew32_part_rows(int p) { return 8 * ((p >> 1) & 1) + 32 * (p >> 2); }
__device__ __forceinline__ void skew32_read(const float *tile, int lane, int wave, SkewRegs &r) {
  int row0, g2;
  skew32_lane(lane, wave, row0, g2);
  const float *src0 = tile + row0 * kTileStride + 2 * g2;
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const float *src = src0 + (4 * (p & 1) + skew32_part_rows(p)) * kTileStride;
    r.cur[p] = make_float2(src[0], src[1]);
  }
  r.nyq = tile[kM * kTileStride + (lane & 15)];   // row 1024 = Nyquist bin (every wave reads it, wave 0 stores it)
}
__device__ __forceinline__ void store2_at(float *base /* wave-uniform */, unsigned byte_off, float x, float y) {
  using f32x2 = __attribute__((ext_vector_type(2))) float;
  const f32x2 v = {x, y};
  asm volatile("global_store_dwordx2 %0, %1, %2" SMX_STORE_MOD : : "v"(byte_off), "v"(v), "s"(base) : "memory");   // (the wait state behind a store is for 12 bytes and more)
}
__device__ __forceinline__ float select_lanes(float other, float chosen, unsigned long long lanes) {   // lanes ? chosen : other
  float d;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(other), "v"(chosen), "s"(lanes));
  return d;
}
// obase: origin of the tile that `r` holds; fresh: nothing is carried into it; closing: nothing follows it in its clip (or
// in this workgroup's range); frames_left: its frames that exist (16 unless it is a clip's last tile)
__device__ __forceinline__ void skew32_store(const FastArgs &a, const Skew32 &sk, float *obase, int frames_left, bool fresh, bool closing,
                                             int wave, int lane, const SkewRegs &r, float2 (&carry)[16]) {
  const unsigned pitch = (unsigned)a.out_stride * 4u;
#ifdef SMX_DIAG
  if (a.abl_nostore == 1) {   // timing-only ablation: keep the LDS reads alive, drop the HBM stores
#pragma unroll
    for (int p = 0; p < 16; ++p) { asm volatile("" ::"v"(r.cur[p].x), "v"(r.cur[p].y)); carry[p] = r.cur[p]; }
    asm volatile("" ::"v"(r.nyq));
    return;
  }
#endif
  if (frames_left >= kFT && !fresh && !closing) {   // wave-uniform: one whole block per row and part
    const unsigned g0 = opaque32(sk.goffc[0]), g1 = opaque32(sk.goffc[1]);
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int c = p & 1;
      store2_at(obase - kFT, (c ? g1 : g0) + (unsigned)skew32_part_rows(p) * pitch, select_lanes(carry[p].x, r.cur[p].x, sk.sel[c]),
                select_lanes(carry[p].y, r.cur[p].y, sk.sel[c]));
      carry[p] = r.cur[p];
    }
  } else if ((frames_left & 1) == 0) {
    // a clip's (or a range's) first or last tile with whole pairs: lane-masked 8-byte stores, the current pair where it completes a
    // block or nothing follows, the carried pair 64 bytes down where one exists -- 16 to 32 instructions where the scalar path below
    // takes up to 64 (round 5: the first flush of a clip cost 1.8 us more than a steady one, the last 2.2)
    int row0, g2;
    skew32_lane(lane, wave, row0, g2);
    const unsigned gl = ((unsigned)row0 * (unsigned)a.out_stride + 2u * g2) * 4u;
    const int f = 2 * g2;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int c = p & 1;
      const bool now = (sk.sel[c] >> lane) & 1;
      const unsigned off = gl + (unsigned)(4 * c + skew32_part_rows(p)) * pitch;
      float *dst = reinterpret_cast<float *>(reinterpret_cast<char *>(obase) + off);
      if ((now || closing) && f < frames_left) *reinterpret_cast<float2 *>(dst) = r.cur[p];          // (8-byte aligned: the launcher admits even pitches and origins only)
      if (!now && !fresh) *reinterpret_cast<float2 *>(dst - 16) = carry[p];
      carry[p] = r.cur[p];
    }
  } else {
    int row0, g2;
    skew32_lane(lane, wave, row0, g2);
    const unsigned gl = ((unsigned)row0 * (unsigned)a.out_stride + 2u * g2) * 4u;
    const int f = 2 * g2;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int c = p & 1;
      const bool now = (sk.sel[c] >> lane) & 1;
      const unsigned off = gl + (unsigned)(4 * c + skew32_part_rows(p)) * pitch;
      float *dst = reinterpret_cast<float *>(reinterpret_cast<char *>(obase) + off);
      if (now || closing) {
        if (f < frames_left) dst[0] = r.cur[p].x;
        if (f + 1 < frames_left) dst[1] = r.cur[p].y;
      }
      if (!now && !fresh) { dst[-16] = carry[p].x; dst[-15] = carry[p].y; }
      carry[p] = r.cur[p];
    }
  }
  if (wave == 0 && lane < 16 && lane < frames_left) obase[(int64_t)kM * a.out_stride + lane] = r.nyq;
}

// ---- the same for ANY row pitch and origin (SKEW = 2): one frame per lane ------------------------------------------------
// With an odd row pitch (2813 frames: BASELINE C5's thirty-second clips) a row's open block ends an odd number of frames into
// a tile and the alignment differs from row to row with period 16, so a lane holds ONE frame of a row (16 lanes a row, 4 rows
// an instruction: rows {r, r + 16} per half-wave keep the LDS reads conflict free) and a wave takes the rows of two residues
// mod 16 -- rows 2 w + 16 k and 2 w + 1 + 16 k, k < 64 -- instead of 128 consecutive ones: all 32 parts of a lane (rows
// row_l + 32 p) then share one block offset, i.e. one lane mask and one offset adjustment per lane and clip.
struct SkewG32 {
  unsigned goff;             // from 64 bytes BEFORE a tile's origin: byte offset of out[row_l][col], plus 64 for the lanes that store the current frame
  unsigned long long sel;    // the lanes whose frame of the current tile completes the block
};
struct SkewGRegs {
  float cur[32];
  float nyq;
};
__device__ __forceinline__ void skewg32_lane(int lane, int wave, int &row_l, int &col) {
  asm volatile("" : "+v"(lane));
  const int rsel = lane >> 4;
  col = lane & 15;
  row_l = 2 * wave + (rsel >> 1) + 16 * (rsel & 1);
}
__device__ __forceinline__ void skewg32_clip(const FastArgs &a, SkewG32 &sk, const float *oclip, int lane, int wave) {
  int row_l, col;
  skewg32_lane(lane, wave, row_l, col);
  const unsigned at = (unsigned)((reinterpret_cast<uintptr_t>(oclip) >> 2) + (uintptr_t)row_l * (uintptr_t)a.out_stride) & 15u;
  const unsigned e = at ? 16u - at : 16u;          // frames of a tile that end the row's open block
  const bool now = (unsigned)col < e;
  sk.sel = __ballot(now);
  sk.goff = ((unsigned)row_l * (unsigned)a.out_stride + (unsigned)col) * 4u + (now ? 64u : 0u);
}
__device__ __forceinline__ void skewg32_read(const float *tile, int lane, int wave, SkewGRegs &r) {
  int row_l, col;
  skewg32_lane(lane, wave, row_l, col);
  const float *src0 = tile + row_l * kTileStride + col;
  const float *src1 = src0 + 16 * 32 * kTileStride;   // (ds offsets are 16 bits)
#pragma unroll
  for (int p = 0; p < 32; ++p) r.cur[p] = (p < 16 ? src0 : src1)[32 * kTileStride * (p & 15)];
  r.nyq = tile[kM * kTileStride + (lane & 15)];   // row 1024 = Nyquist bin (every wave reads it, wave 0 stores it)
}
__device__ __forceinline__ void store1_at(float *base /* wave-uniform */, unsigned byte_off, float x) {
  asm volatile("global_store_dword %0, %1, %2" SMX_STORE_MOD : : "v"(byte_off), "v"(x), "s"(base) : "memory");
}
__device__ __forceinline__ void skewg32_store(const FastArgs &a, const SkewG32 &sk, float *obase, int frames_left, bool fresh, bool closing,
                                              int wave, int lane, const SkewGRegs &r, float (&carry)[32]) {
  const unsigned pitch = (unsigned)a.out_stride * 4u;
#ifdef SMX_DIAG
  if (a.abl_nostore == 1) {   // timing-only ablation: keep the LDS reads alive, drop the HBM stores
#pragma unroll
    for (int p = 0; p < 32; ++p) { asm volatile("" ::"v"(r.cur[p])); carry[p] = r.cur[p]; }
    asm volatile("" ::"v"(r.nyq));
    return;
  }
#endif
  if (frames_left >= kFT && !fresh && !closing) {   // wave-uniform: one whole block per row and part
    const unsigned g = opaque32(sk.goff);
#pragma unroll
    for (int p = 0; p < 32; ++p) {
      store1_at(obase - kFT, g + 32u * (unsigned)p * pitch, select_lanes(carry[p], r.cur[p], sk.sel));
      carry[p] = r.cur[p];
    }
  } else {
    int row_l, col;
    skewg32_lane(lane, wave, row_l, col);
    const bool now = (sk.sel >> lane) & 1;
    const unsigned gl = ((unsigned)row_l * (unsigned)a.out_stride + (unsigned)col) * 4u;
#pragma unroll
    for (int p = 0; p < 32; ++p) {
      float *dst = reinterpret_cast<float *>(reinterpret_cast<char *>(obase) + gl + 32u * (unsigned)p * pitch);
      if ((now || closing) && col < frames_left) dst[0] = r.cur[p];
      if (!now && !fresh) dst[-16] = carry[p];
      carry[p] = r.cur[p];
    }
  }
  if (wave == 0 && lane < 16 && lane < frames_left) obase[(int64_t)kM * a.out_stride + lane] = r.nyq;
}

// what the power kernel does between the stages of a frame pair (see frame32_to_tile)
template <bool ALIGNED, int SKEW = 0>
struct PowerMid32 {
  const FastArgs &a;
  const Lds32 &lds;
  const Flush32 &fl;
  FlushRegs &fr;
  const Skew32 &sk;      // SKEW = 1: the flush in aligned blocks, a pair of frames per lane (even row pitch and origin)
  SkewRegs &sr;
  float2 (&carry)[16];
  const SkewG32 &skg;    // SKEW = 2: a frame per lane (any pitch and origin)
  SkewGRegs &srg;
  float (&carryg)[32];
  bool pend_fresh, pend_closing;
  float2 (&raw)[32];
  const float *src;      // the next frames' samples (per lane)
  const float *src_clip; // ... their clip (wave-uniform) and whether their tile holds a frame that reaches past the signal
  bool src_border;
  float *pend_out;       // output origin and frames of the previous tile
  int pend_left;
  int lane, wave, b, it;
  // A border tile's samples are requested at the END of the frame, by the padding rule, over an unconditional request behind
  // slot 5 that reads the clip's first frame for such a tile (in bounds, never used).  A request behind slot 5 that is skipped
  // for border tiles, or the padding rule there, made the loop spill (9 to 37 registers; the compiler's allocation, not a
  // count of live values): this form costs one register.  The two border tiles of a clip begin a memory round trip late.
  __device__ __forceinline__ void load_next() const {
    load_frame32<ALIGNED>(src_border ? src_clip : src, lane & 31, raw);
  }
  __device__ __forceinline__ void load_next_border() const {
    if (src_border) load_frame32_padded(a, src_clip, (int)(src - src_clip), lane & 31, raw);
  }
#ifdef SMX_STAMPS
  unsigned long long *stamp_sum, *stamp_prev_p;
  template <int I> __device__ __forceinline__ void stamp() const {
    unsigned long long &stamp_prev = *stamp_prev_p;
    SMX_STAMP(I);
  }
#else
  template <int I> __device__ __forceinline__ void stamp() const {}
#endif
  unsigned &pk_drained, &pk_filled;
  __device__ __forceinline__ void early() const {   // the counter is read a radix-32 pass before it is needed
    pk_drained = peek32(lds.drained + b * kTileStride);
  }
  __device__ __forceinline__ void before_cells() const {
    // buffer b last held tile it - 2, the (it >> 1)-th tile written there: every wave has read its share out
#ifdef SMX_STAMPS
    const unsigned long long w0 = __builtin_amdgcn_s_memtime();
#endif
    lds_wait32(lds.drained + b * kTileStride, 8u * ((unsigned)it >> 1), pk_drained);
#ifdef SMX_STAMPS
    stamp_sum[12] += __builtin_amdgcn_s_memtime() - w0;   // time in the wait itself (the stamps around it also hold sunk arithmetic)
#endif
  }
  __device__ __forceinline__ void after_transposition_issue() const {
    if (it > 0) pk_filled = peek32(lds.filled + (b ^ 1) * kTileStride);   // looked at after the second radix-32 pass
  }
  __device__ __forceinline__ void after_exchange_issue() const {
    if (it > 0) {
      // tile it - 1 (buffer b ^ 1, the ((it - 1) >> 1) + 1-th written there) is complete: every column is in
#ifdef SMX_STAMPS
      const unsigned long long w0 = __builtin_amdgcn_s_memtime();
#endif
      lds_wait32(lds.filled + (b ^ 1) * kTileStride, 8u * (((unsigned)(it - 1) >> 1) + 1), pk_filled);
#ifdef SMX_STAMPS
      stamp_sum[13] += __builtin_amdgcn_s_memtime() - w0;
#endif
      if constexpr (SKEW == 1) skew32_read(lds.tiles + (b ^ 1) * kTile32Floats, lane, wave, sr);
      else if constexpr (SKEW == 2) skewg32_read(lds.tiles + (b ^ 1) * kTile32Floats, lane, wave, srg);
      else flush32_read(lds.tiles + (b ^ 1) * kTile32Floats, fl, wave, lane, fr);
      // "read out" may be signalled as soon as the reads are ISSUED: the counter's add executes behind them in this wave's
      // LDS order, and nobody writes the buffer before seeing it.  Every frame fraction the signal comes earlier is slack
      // for the wave that waits for it: the two waves of a SIMD settle half a frame apart, and with the signal behind the
      // stores (0.8 of a frame, needed at 0.3 of the next) they spent 3500 cycles per tile polling (profiles/r05).
      lds_signal32(lds.drained + (b ^ 1) * kTileStride, lane);
    }
  }
  // slot s of the post-pass is done (SMX_P32_STORE_AT / SMX_P32_LOAD_AT: where the previous tile's stores and the next
  // frames' loads are issued; the same slot: SMX_P32_LOADS_FIRST says which goes first)
  __device__ __forceinline__ void postpass_at(int s) const {
    const bool same = SMX_P32_STORE_AT == SMX_P32_LOAD_AT;
    if (s == SMX_P32_LOAD_AT && same && SMX_P32_LOADS_FIRST) { load_next(); SMX_FENCE(); }
    if (s == SMX_P32_STORE_AT && it > 0) {
      if constexpr (SKEW == 1) skew32_store(a, sk, pend_out, pend_left, pend_fresh, pend_closing, wave, lane, sr, carry);
      else if constexpr (SKEW == 2) skewg32_store(a, skg, pend_out, pend_left, pend_fresh, pend_closing, wave, lane, srg, carryg);
      else flush32_store(a, fl, pend_out, pend_left, wave, lane, fr);
    }
    SMX_FENCE();
    if (s == SMX_P32_LOAD_AT && !(same && SMX_P32_LOADS_FIRST)) load_next();   // (half of the frame's registers are free by now)
    if (s == 15) load_next_border();
  }
};

template <bool ALIGNED, int PMODE, bool STRIP, int SKEW = 0>
__global__ void __launch_bounds__(512) stft2048_power32_kernel(FastArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SMX_STAMPS
  const unsigned long long tl_entry = __builtin_amdgcn_s_memrealtime();   // launch timeline (tools/launch_timeline.py): 100 MHz ticks, one clock for the chip
#endif
  const Lds32 lds = carve_lds32(smem);
  const Lane32 L = setup_lane32(lds, lane, wave);
  Tables32Regs tabs;
  tables32_request(a, tid, tabs);   // in flight under the tile walk's divisions
#ifdef SMX_STAMPS
  const unsigned long long tl_p1 = __builtin_amdgcn_s_memrealtime();
#endif
  TileWalk tw;
  tw.init(a, a.out + a.out_offset, kBins * a.out_stride);
  const int ntiles = tw.ntiles > 0 ? tw.ntiles : 0;
#ifdef SMX_STAMPS
  const unsigned long long tl_p2 = __builtin_amdgcn_s_memrealtime();
#endif

  // first sample of this lane's frame in tile t of the clip at xc (a lane-half without a frame re-reads the tile's
  // first frame and its results are never stored)
  auto frame_ptr = [&](const float *xc, int t) {
    const int64_t f0 = (int64_t)t * kFT;
    const int avail = (int)(a.count - f0 < kFT ? a.count - f0 : kFT) - 1;   // last frame of the tile that exists (wave-uniform)
    const int fi = 2 * wave + L.h;
    const int64_t p = a.p0 + f0 + (fi <= avail ? fi : 0);
    if (a.fold_frames == 1 && (p < a.border_i0 || p >= a.border_i1)) {
      const int64_t clip = (xc - a.x) / a.x_stride;
      return p < a.border_i0 ? a.strip_l + clip * a.strip_l_stride + (p - a.p0) * a.hop
                             : a.strip_r + clip * a.strip_r_stride + (p - a.border_i1) * a.hop;
    }
    return xc + (p * a.hop - a.left);   // (fold_frames == 2: possibly outside the clip -- then only its position is used)
  };
  // fold_frames == 2: one of THIS WAVE's two frames of tile t reaches past the signal: the wave takes load_frame32_padded for
  // the tile (wave-uniform; the other waves of such a tile keep the plain requests -- a C2 clip's first tile has its border
  // frames on wave 0, its last one on wave 4: with the whole tile on the padding rule the prologue's requests alone took
  // 4.7 us, profiles/r07/timeline_prologue_pieces.log)
  auto tile_border = [&](int t) {
    if (a.fold_frames != 2) return false;
    const int64_t f0 = (int64_t)t * kFT;
    const int avail = (int)(a.count - f0 < kFT ? a.count - f0 : kFT) - 1;
    bool any = false;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int fi = 2 * wave + hh;
      const int64_t p = a.p0 + f0 + (fi <= avail ? fi : 0);
      any = any || p < a.border_i0 || p >= a.border_i1;
    }
    return any;
  };

  float2 raw[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) raw[j] = make_float2(0.f, 0.f);
  // the first frames' samples are requested BEFORE the tables are filled: both round trips overlap (round 5: the prologue was
  // 5.8 us of a 0.49 ms C2 launch, profiles/r07/timeline_before.log)
  if (ntiles > 0) {
    const float *src0 = frame_ptr(tw.xclip, tw.ft);
    if (tile_border(tw.ft)) load_frame32_padded(a, tw.xclip, (int)(src0 - tw.xclip), L.l, raw);
    else load_frame32<ALIGNED>(src0, L.l, raw);
  }
#ifdef SMX_STAMPS
  const unsigned long long tl_p3 = __builtin_amdgcn_s_memrealtime();
#endif
  tables32_commit(lds, tid, tabs);
#ifdef SMX_STAMPS
  const unsigned long long tl_p4 = __builtin_amdgcn_s_memrealtime();
#endif
  __syncthreads();   // tables and zeroed counters visible: the only workgroup barrier of the main loop
  float *pend_out = nullptr;
  int pend_left = 0;
  Flush32 fl;
  {
    const int hsel = lane >> 5, jj = (lane & 31) >> 2;
    fl.g = lane & 3;
    const int row0 = 128 * wave + (jj & 3) + 16 * (jj >> 2) + 4 * hsel;
    fl.src0 = row0 * kTileStride + 4 * fl.g;
    fl.goff0 = ((unsigned)row0 * (unsigned)a.out_stride + 4u * fl.g) * 4u;
  }
  FlushRegs fr;
  Skew32 sk{};
  SkewRegs sr;
  float2 carry[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) carry[p] = make_float2(0.f, 0.f);
  SkewG32 skg{};
  SkewGRegs srg;
  float carryg[32];
#pragma unroll
  for (int p = 0; p < 32; ++p) carryg[p] = 0.f;
  bool pend_fresh = true, pend_closing = false;
  const float *pend_oclip = nullptr;
  unsigned pk_drained = 0, pk_filled = 0;
#ifdef SMX_STAMPS
  unsigned long long stamp_sum[kStampSlots] = {0}, stamp_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
  const unsigned long long clk_t0 = stamp_prev, clk_r0 = __builtin_amdgcn_s_memrealtime();
  stamp_sum[15] = tl_entry;
  stamp_sum[0] = tl_p1 - tl_entry;   // (slots 0-3 before the loop adds to them: the prologue's pieces, 100 MHz ticks; launch_timeline.py reads them from a launch of ZERO tiles ... or subtracts)
  stamp_sum[1] = tl_p2 - tl_p1;
  stamp_sum[2] = tl_p3 - tl_p2;
  stamp_sum[3] = tl_p4 - tl_p3;
  stamp_sum[4] = clk_r0 - tl_p4;
  stamp_sum[16] = clk_r0;          // tables in, first samples requested
  stamp_sum[22] = (unsigned long long)ntiles;
#endif
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
#ifdef SMX_DIAG
    if (a.abl_p32 & 1) src = frame_ptr(a.x, 1);   // timing only: every tile reads the same resident samples
#endif
    const bool have = (int64_t)tw.ft * kFT + 2 * wave < a.count;   // wave-uniform: at least the first half has a frame
    if constexpr (SKEW == 1) {
      if (it > 0 && pend_fresh) skew32_clip(a, sk, pend_oclip, lane, wave);   // (wave-uniform) the pending tile begins a clip or this workgroup's range
    } else if constexpr (SKEW == 2) {
      if (it > 0 && pend_fresh) skewg32_clip(a, skg, pend_oclip, lane, wave);
    }
#ifdef SMX_STAMPS
    const PowerMid32<ALIGNED, SKEW> mid{a, lds, fl, fr, sk, sr, carry, skg, srg, carryg, pend_fresh, pend_closing, raw, src, src_clip, src_border, pend_out, pend_left, lane, wave, b, it, stamp_sum, &stamp_prev, pk_drained, pk_filled};
#else
    const PowerMid32<ALIGNED, SKEW> mid{a, lds, fl, fr, sk, sr, carry, skg, srg, carryg, pend_fresh, pend_closing, raw, src, src_clip, src_border, pend_out, pend_left, lane, wave, b, it, pk_drained, pk_filled};
#endif
    mid.template stamp<0>();
#ifdef SMX_STAMPS
    if (it == 32) stamp_sum[14] = __builtin_amdgcn_s_memtime();   // when this wave starts its 33rd tile (wave offsets inside a workgroup)
    if (it == 1) stamp_sum[9] = __builtin_amdgcn_s_memrealtime();
    if (it == 2) stamp_sum[10] = __builtin_amdgcn_s_memrealtime();
    if (it == ntiles - 1) stamp_sum[11] = __builtin_amdgcn_s_memrealtime();
#endif
#if SMX_P32_HAVE_BRANCH
    if (have) {
      frame32_to_tile<PMODE>(a, L, raw, lds.tiles + b * kTile32Floats, mid);
    } else {
      mid.early();
      mid.before_cells();
      mid.after_transposition_issue();
      mid.after_exchange_issue();
      mid.postpass_at(SMX_P32_STORE_AT);
      if (SMX_P32_LOAD_AT != SMX_P32_STORE_AT) mid.postpass_at(SMX_P32_LOAD_AT);
    }
#else
    // (a wave without a frame in a clip's last tile runs the frame code all the same, on the tile's first frame; its columns
    // are never stored.  One path through the loop body: the branch cost ~130 register copies per tile at its join.)
    (void)have;
    frame32_to_tile<PMODE>(a, L, raw, lds.tiles + b * kTile32Floats, mid);
#endif
    lds_signal32(lds.filled + b * kTileStride, lane);
    mid.template stamp<8>();
    pend_out = tw.oclip + tw.ft * kFT;   // wave-uniform
    const int64_t left = a.count - (int64_t)tw.ft * kFT;
    pend_left = left < kFT ? (int)left : kFT;
    pend_oclip = tw.oclip;
    pend_fresh = it == 0 || tw.ft == 0;
    pend_closing = tw.ft == a.tiles_per_clip - 1;
    tw.xclip = xnext;
    tw.oclip = onext;
    tw.ft = ftnext;
  }
#ifdef SMX_STAMPS
  stamp_sum[17] = __builtin_amdgcn_s_memrealtime();   // tile loop left
#endif
  if (ntiles > 0) {   // the last tile of this workgroup
    const int b = (ntiles - 1) & 1;
    lds_wait(lds.filled + b * kTileStride, 8u * (((unsigned)(ntiles - 1) >> 1) + 1));
    if constexpr (SKEW == 1) {
      if (pend_fresh) skew32_clip(a, sk, pend_oclip, lane, wave);
      skew32_read(lds.tiles + b * kTile32Floats, lane, wave, sr);
      skew32_store(a, sk, pend_out, pend_left, pend_fresh, true, wave, lane, sr, carry);
    } else if constexpr (SKEW == 2) {
      if (pend_fresh) skewg32_clip(a, skg, pend_oclip, lane, wave);
      skewg32_read(lds.tiles + b * kTile32Floats, lane, wave, srg);
      skewg32_store(a, skg, pend_out, pend_left, pend_fresh, true, wave, lane, srg, carryg);
    } else {
      flush32_read(lds.tiles + b * kTile32Floats, fl, wave, lane, fr);
      flush32_store(a, fl, pend_out, pend_left, wave, lane, fr);
    }
  }
#ifdef SMX_STAMPS
  stamp_sum[20] = __builtin_amdgcn_s_memtime() - clk_t0;
  stamp_sum[21] = __builtin_amdgcn_s_memrealtime() - clk_r0;
  stamp_sum[18] = __builtin_amdgcn_s_memrealtime();   // last flush issued
  stamp_sum[23] = 0;
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
    for (int64_t bt = (int64_t)gridDim.x - 1 - tw.uid; bt * kFT < total; bt += gridDim.x) {   // from the last workgroup down: idle ones first
#ifdef SMX_STAMPS
      ++stamp_sum[23];
#endif
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
        frame32_to_tile<PMODE>(a, L, braw, bt_tile, NoMid32{});
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
#ifdef SMX_STAMPS
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the wave's stores are out
  stamp_sum[19] = __builtin_amdgcn_s_memrealtime();
  if (lane == 0 && blockIdx.x < 4096)
    for (int i = 0; i < kStampSlots; ++i) g_stamp_sums[(blockIdx.x * 16 + wave) * kStampSlots + i] = stamp_sum[i];
#endif
}

// ---- Stft.transform at fft 2048 on the same frame code (stft.ml:632-666): the spectrum's two planes fill both tile buffers, so
// the tile is single-buffered -- a wave reads the previous tile out in the middle of its next frame's first radix-32 (its
// stores then run under the rest of the frame) and nobody writes a cell before every wave has done so.
struct CplxFlush32 {
  int src0;          // float offset of (row0, frame 2 g) in a plane
  unsigned goff0;    // byte offset of out[row0][2 g] from the tile's origin (complex64: 8 bytes)
  int g;             // frames 2 g, 2 g + 1
};
__device__ __forceinline__ CplxFlush32 setup_cplx_flush32(const FastArgs &a, int lane, int wave) {
  // one store = 8 rows x 128 bytes; rows {0, 1, 16, 17} (+ 2 for the upper half-wave) keep a lane's LDS reads conflict free
  CplxFlush32 fl;
  const int half = lane >> 5, ridx = (lane & 31) >> 3;
  fl.g = lane & 7;
  const int row0 = 128 * wave + (ridx & 1) + 16 * (ridx >> 1) + 2 * half;
  fl.src0 = row0 * kTileStride + 2 * fl.g;
  fl.goff0 = ((unsigned)row0 * (unsigned)a.out_stride + 2u * fl.g) * 8u;
  return fl;
}
__device__ __forceinline__ void cplx_flush32(const FastArgs &a, const float *re, const CplxFlush32 &fl, float *obase, int frames_left,
                                             int wave, int lane) {
  const float *pr0 = re + opaque32(fl.src0);
  const int fleft = frames_left - 2 * fl.g;
  const unsigned pitch = (unsigned)a.out_stride * 8u, goff0 = opaque32(fl.goff0);
#pragma unroll
  for (int i = 0; i < 16; ++i) {   // rows row0 + 32 (i >> 2) + 4 (i & 3)
    const float *pr = pr0 + (32 * (i >> 2) + 4 * (i & 3)) * kTileStride, *pi = pr + kTile32Floats;
    const float r0 = pr[0], r1 = pr[1], i0 = pi[0], i1 = pi[1];
    const unsigned goff = goff0 + (unsigned)(32 * (i >> 2) + 4 * (i & 3)) * pitch;
    if (fleft >= 2) {
      store4_unaligned(obase, goff, r0, i0, r1, i1);
    } else if (fleft == 1) {
      float *dst = reinterpret_cast<float *>(reinterpret_cast<char *>(obase) + goff);
      dst[0] = r0;
      dst[1] = i0;
    }
  }
  if (wave == 0 && lane < 16 && lane < frames_left) {   // bin 1024: real
    float *dst = obase + ((int64_t)kM * a.out_stride + lane) * 2;
    dst[0] = re[kM * kTileStride + lane];
    dst[1] = 0.0f;
  }
}
// The flush in whole aligned 128-byte lines (SKEW, as skewg32_* of the power kernel with a complex value per lane: 16 lanes a
// row = 16 frames x 8 bytes, rows by residue mod 16 so that a lane has one block offset).  The tile's 128-byte runs as they
// stand -- rows 7504 bytes apart, so that 15 runs of 16 straddle a line -- take 0.4-0.56 ms per GB with nothing else running
// (profiles/r06/store_shape_probe.log), whole lines 0.185: the complex spectrogram of C2 is 1.97 GB.
struct CplxSkew32 {
  unsigned goff;             // from 128 bytes BEFORE a tile's origin: byte offset of out[row_l][col], plus 128 for the lanes that store the current frame
  unsigned long long sel;    // the lanes whose frame of the current tile completes the line
};
__device__ __forceinline__ void cplx_skew32_clip(const FastArgs &a, CplxSkew32 &sk, const float *oclip, int lane, int wave) {
  int row_l, col;
  skewg32_lane(lane, wave, row_l, col);
  const unsigned at = (unsigned)((reinterpret_cast<uintptr_t>(oclip) >> 3) + (uintptr_t)row_l * (uintptr_t)a.out_stride) & 15u;
  const unsigned e = at ? 16u - at : 16u;          // frames of a tile that end the row's open line
  const bool now = (unsigned)col < e;
  sk.sel = __ballot(now);
  sk.goff = ((unsigned)row_l * (unsigned)a.out_stride + (unsigned)col) * 8u + (now ? 128u : 0u);
}
// reads the tile (re, im planes) and stores it; carry: the frames of the previous tile that opened a line
__device__ __forceinline__ void cplx_skew32_flush(const FastArgs &a, const float *re, const CplxSkew32 &sk, float *obase, int frames_left,
                                                  bool fresh, bool closing, int wave, int lane_, float2 (&carry)[32]) {
  // (the lane index made opaque: everything derived from it below is loop-invariant, and hoisted out of the tile loop a dozen such
  // values did not fit beside the 64 carried registers -- they were spilled, and every scratch reload in the loop waits for ALL of
  // the wave's outstanding loads and stores (s_waitcnt vmcnt(0)); re-derived per tile they cost a handful of instructions)
  int lane = lane_;
  asm volatile("" : "+v"(lane));
  const unsigned pitch = (unsigned)a.out_stride * 8u;
  int row_l, col;
  skewg32_lane(lane, wave, row_l, col);
  const float *src0 = re + row_l * kTileStride + col;
  const float *src1 = src0 + 16 * 32 * kTileStride;   // (ds offsets are 16 bits)
  if (frames_left >= kFT && !fresh && !closing) {   // wave-uniform: one whole line per row and part
    const unsigned g = sk.goff;
#pragma unroll
    for (int p = 0; p < 32; ++p) {
      const float *pr = (p < 16 ? src0 : src1) + 32 * kTileStride * (p & 15);
      const float2 cur = make_float2(pr[0], pr[kTile32Floats]);
      store2_at(obase - 2 * kFT, g + 32u * (unsigned)p * pitch, select_lanes(carry[p].x, cur.x, sk.sel), select_lanes(carry[p].y, cur.y, sk.sel));
      carry[p] = cur;
    }
  } else {
    const bool now = (sk.sel >> lane) & 1;
    const unsigned gl = ((unsigned)row_l * (unsigned)a.out_stride + (unsigned)col) * 8u;
#pragma unroll
    for (int p = 0; p < 32; ++p) {
      const float *pr = (p < 16 ? src0 : src1) + 32 * kTileStride * (p & 15);
      const float2 cur = make_float2(pr[0], pr[kTile32Floats]);
      float *dst = reinterpret_cast<float *>(reinterpret_cast<char *>(obase) + gl + 32u * (unsigned)p * pitch);
      if ((now || closing) && col < frames_left) { dst[0] = cur.x; dst[1] = cur.y; }
      if (!now && !fresh) { dst[-2 * kFT] = carry[p].x; dst[-2 * kFT + 1] = carry[p].y; }
      carry[p] = cur;
    }
  }
  if (wave == 0 && lane < 16 && lane < frames_left) {   // bin 1024: real
    float *dst = obase + ((int64_t)kM * a.out_stride + lane) * 2;
    dst[0] = re[kM * kTileStride + lane];
    dst[1] = 0.0f;
  }
}
template <bool ALIGNED, bool SKEW = false>
struct CplxMid32 {
  const FastArgs &a;
  const Lds32 &lds;
  const CplxFlush32 &fl;
  const CplxSkew32 &sk;
  float2 (&carry)[32];
  bool pend_fresh, pend_closing;
  float2 (&raw)[32];
  const float *src;
  const float *src_clip;   // the next frames' clip and whether their tile holds a frame that reaches past the signal (see PowerMid32)
  bool src_border;
  float *pend_out;
  int pend_left;
  int lane, wave, it;
  template <int I> __device__ __forceinline__ void stamp() const {}
  __device__ __forceinline__ void early() const {   // between the two 16-point transforms of the first radix-32
    if (it > 0) {
      lds_wait(lds.filled, 8u * (unsigned)it);       // every wave's columns of the previous tile are in
      if constexpr (SKEW) cplx_skew32_flush(a, lds.tiles, sk, pend_out, pend_left, pend_fresh, pend_closing, wave, lane, carry);
      else cplx_flush32(a, lds.tiles, fl, pend_out, pend_left, wave, lane);
      lds_signal32(lds.drained, lane);               // behind this wave's reads in LDS order
    }
  }
  __device__ __forceinline__ void before_cells() const {
    if (it > 0) lds_wait(lds.drained, 8u * (unsigned)it);   // every wave has read the previous tile out
  }
  __device__ __forceinline__ void after_transposition_issue() const {}
  __device__ __forceinline__ void after_exchange_issue() const {}
  __device__ __forceinline__ void postpass_at(int s) const {
    if (s == (SKEW ? 15 : SMX_P32_LOAD_AT)) load_frame32<ALIGNED>(src_border ? src_clip : src, lane & 31, raw);
    if (s == 15 && src_border) load_frame32_padded(a, src_clip, (int)(src - src_clip), lane & 31, raw);   // (wave-uniform; see PowerMid32::load_next)
  }
};

template <bool ALIGNED, bool SKEW = false>
__global__ void __launch_bounds__(512) stft2048_complex32_kernel(FastArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Lds32 lds = carve_lds32(smem);
  const Lane32 L = setup_lane32(lds, lane, wave);
  fill_tables32(a, lds, tid, 512);
  TileWalk tw;
  tw.init(a, a.out + 2 * a.out_offset, 2 * kBins * a.out_stride);
  const int ntiles = tw.ntiles > 0 ? tw.ntiles : 0;
  auto frame_ptr = [&](const float *xc, int t) {   // as stft2048_power32_kernel
    const int64_t f0 = (int64_t)t * kFT;
    const int avail = (int)(a.count - f0 < kFT ? a.count - f0 : kFT) - 1;
    const int fi = 2 * wave + L.h;
    const int64_t p = a.p0 + f0 + (fi <= avail ? fi : 0);
    if (a.fold_frames == 1 && (p < a.border_i0 || p >= a.border_i1)) {
      const int64_t clip = (xc - a.x) / a.x_stride;
      return p < a.border_i0 ? a.strip_l + clip * a.strip_l_stride + (p - a.p0) * a.hop
                             : a.strip_r + clip * a.strip_r_stride + (p - a.border_i1) * a.hop;
    }
    return xc + (p * a.hop - a.left);
  };
  auto tile_border = [&](int t) {   // fold_frames == 2 (as stft2048_power32_kernel): one of this wave's two frames of tile t reaches past the signal
    if (a.fold_frames != 2) return false;
    const int64_t f0 = (int64_t)t * kFT;
    const int avail = (int)(a.count - f0 < kFT ? a.count - f0 : kFT) - 1;
    bool any = false;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int fi = 2 * wave + hh;
      const int64_t p = a.p0 + f0 + (fi <= avail ? fi : 0);
      any = any || p < a.border_i0 || p >= a.border_i1;
    }
    return any;
  };
  float2 raw[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) raw[j] = make_float2(0.f, 0.f);
  if (ntiles > 0) {
    const float *src0 = frame_ptr(tw.xclip, tw.ft);
    if (tile_border(tw.ft)) load_frame32_padded(a, tw.xclip, (int)(src0 - tw.xclip), L.l, raw);
    else load_frame32<ALIGNED>(src0, L.l, raw);
  }
  __syncthreads();
  const CplxFlush32 fl = setup_cplx_flush32(a, lane, wave);
  CplxSkew32 sk{};
  float2 carry[32];
#pragma unroll
  for (int p = 0; p < 32; ++p) carry[p] = make_float2(0.f, 0.f);
  bool pend_fresh = true, pend_closing = false;
  const float *pend_oclip = nullptr;
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
    if constexpr (SKEW) {
      if (it > 0 && pend_fresh) cplx_skew32_clip(a, sk, pend_oclip, lane, wave);   // (wave-uniform) the pending tile begins a clip or this workgroup's range
    }
    const CplxMid32<ALIGNED, SKEW> mid{a, lds, fl, sk, carry, pend_fresh, pend_closing, raw, src, src_clip, src_border, pend_out, pend_left, lane, wave, it};
    // (SKEW: the twiddle rows are requested behind the flush -- their 62 registers beside the 64 carried ones do not fit)
    frame32_to_tile<2, CplxMid32<ALIGNED, SKEW>, true, SKEW ? 0 : 15>(a, L, raw, lds.tiles, mid);
    lds_signal32(lds.filled, lane);
    pend_out = tw.oclip + 2 * tw.ft * kFT;
    const int64_t left = a.count - (int64_t)tw.ft * kFT;
    pend_left = left < kFT ? (int)left : kFT;
    pend_oclip = tw.oclip;
    pend_fresh = it == 0 || tw.ft == 0;
    pend_closing = tw.ft == a.tiles_per_clip - 1;
    tw.xclip = xnext;
    tw.oclip = onext;
    tw.ft = ftnext;
  }
  if (ntiles > 0) {
    lds_wait(lds.filled, 8u * (unsigned)ntiles);
    if constexpr (SKEW) {
      if (pend_fresh) cplx_skew32_clip(a, sk, pend_oclip, lane, wave);
      cplx_skew32_flush(a, lds.tiles, sk, pend_out, pend_left, pend_fresh, true, wave, lane, carry);
    } else {
      cplx_flush32(a, lds.tiles, fl, pend_out, pend_left, wave, lane);
    }
  }
}

// ---- the spectrum FRAME-MAJOR: out[clip][frame][k], k < 1025, rows of `a.out_stride` floats (Griffin-Lim's rebuilt spectra, which
// only this library's own synthesis reads: capi.cpp).  A lane holds bins l + 32 s and 1024 - l - 32 s of its frame, so 32 lanes
// store 256 contiguous bytes of the frame's row per slot straight from their registers: no tile, no flush, no counters -- the
// frame's column of LDS carries its transposition and exchange only, and the waves of a workgroup never wait for each other.
// Every clip owns tiles_per_clip x 16 rows: the frames a clip's last tile does not have land in rows nobody reads.
template <bool ALIGNED>
struct FmMid32 {
  const FastArgs &a;
  float2 (&raw)[32];
  const float *src;
  const float *src_clip;
  bool src_border;
  int lane;
  char *pk;   // the frame's row, bin l                        (slot s: + 256 s)
  char *pm;   // the frame's row, bin 1024 - l - 32 x 15       (slot s: + 256 (15 - s))
  template <int I> __device__ __forceinline__ void stamp() const {}
  __device__ __forceinline__ void early() const {}
  __device__ __forceinline__ void before_cells() const {}
  __device__ __forceinline__ void after_transposition_issue() const {}
  __device__ __forceinline__ void after_exchange_issue() const {}
  __device__ __forceinline__ void emit(int s, f2 re, f2 im) const {
    *reinterpret_cast<float2 *>(pk + 256 * s) = make_float2(re.x, im.x);
    *reinterpret_cast<float2 *>(pm + 256 * (15 - s)) = make_float2(re.y, im.y);
  }
  __device__ __forceinline__ void emit_self(float x, float y) const {   // bin 512: lane 0's
    if ((lane & 31) == 0) *reinterpret_cast<float2 *>(pk + 8 * 512) = make_float2(x, y);
  }
  __device__ __forceinline__ void postpass_at(int s) const {
    if (s == SMX_P32_LOAD_AT) load_frame32<ALIGNED>(src_border ? src_clip : src, lane & 31, raw);
    if (s == 15 && src_border) load_frame32_padded(a, src_clip, (int)(src - src_clip), lane & 31, raw);
  }
};

template <bool ALIGNED>
__global__ void __launch_bounds__(512) stft2048_complex_fm_kernel(FastArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Lds32 lds = carve_lds32(smem);
  const Lane32 L = setup_lane32(lds, lane, wave);
  fill_tables32(a, lds, tid, 512);
  TileWalk tw;
  tw.init(a, a.out, (int64_t)a.tiles_per_clip * kFT * a.out_stride);
  const int ntiles = tw.ntiles > 0 ? tw.ntiles : 0;
  auto frame_ptr = [&](const float *xc, int t) {   // as stft2048_power32_kernel (the border frames ride in the tile sequence or there are none)
    const int64_t f0 = (int64_t)t * kFT;
    const int avail = (int)(a.count - f0 < kFT ? a.count - f0 : kFT) - 1;
    const int fi = 2 * wave + L.h;
    const int64_t p = a.p0 + f0 + (fi <= avail ? fi : 0);
    return xc + (p * a.hop - a.left);
  };
  auto tile_border = [&](int t) {
    if (a.fold_frames != 2) return false;
    const int64_t f0 = (int64_t)t * kFT;
    const int avail = (int)(a.count - f0 < kFT ? a.count - f0 : kFT) - 1;
    bool any = false;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int fi = 2 * wave + hh;
      const int64_t p = a.p0 + f0 + (fi <= avail ? fi : 0);
      any = any || p < a.border_i0 || p >= a.border_i1;
    }
    return any;
  };
  float2 raw[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) raw[j] = make_float2(0.f, 0.f);
  if (ntiles > 0) {
    const float *src0 = frame_ptr(tw.xclip, tw.ft);
    if (tile_border(tw.ft)) load_frame32_padded(a, tw.xclip, (int)(src0 - tw.xclip), L.l, raw);
    else load_frame32<ALIGNED>(src0, L.l, raw);
  }
  __syncthreads();   // the tables
  for (int it = 0; it < ntiles; ++it) {
    int ftnext;
    const float *xnext;
    float *onext;
    tw.peek(a, ftnext, xnext, onext);
    const bool more = it + 1 < ntiles;
    const float *src_clip = more ? xnext : tw.xclip;
    const float *src = frame_ptr(src_clip, more ? ftnext : tw.ft);
    const bool src_border = tile_border(more ? ftnext : tw.ft);
    char *row = reinterpret_cast<char *>(tw.oclip + ((int64_t)tw.ft * kFT + L.col) * a.out_stride);   // this lane's frame
    const FmMid32<ALIGNED> mid{a, raw, src, src_clip, src_border, lane, row + 8 * L.li(), row + 8 * (1024 - 32 * 15) - 8 * L.li()};
    frame32_to_tile<2, FmMid32<ALIGNED>, 2>(a, L, raw, lds.tiles, mid);
    tw.xclip = xnext;
    tw.oclip = onext;
    tw.ft = ftnext;
  }
}
