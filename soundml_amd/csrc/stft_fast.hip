// Fused framing + window + real FFT(2048) + output stage for gfx950 (MI355X): the three fft-2048 kernels
//   stft2048_power32_kernel   |X|^p               Stft.power_spectrum          stft.ml:670-691   (stft_fast_p32.hpp, 32-lane frame pipeline)
//   stft2048_complex32_kernel X                   Stft.transform / _range      stft.ml:632-666   (the same)
//   stft2048_mel_kernel       W |X|^p (MFMA)      Soundml.mel_spectrogram      soundml.ml:12-24
// They replace, for float32 audio, the reference's hot call
//   Nx.stft cdtype ~window:fft ~step:hop ~win (to_double samples)   stft.ml:356-364
// Each audio sample is read from HBM once (hop-strided overlapping frames are re-read through L1 / L2) and
// the [bins; frames] result is written once, in 64-byte (power) or 128-byte (complex) runs along the frame axis.
//
// Frame pipeline (M = N/2 = 1024 complex points, z[n] = x[2n] + i x[2n+1]); one 64-lane wavefront owns one
// frame, lane l holds 16 complex points:
//   A. n = l + 64 j      : radix-16 over j in registers        -> y_l[k1],  twiddle W_M^(l k1)
//   X. in-wave 16x16 transpose (permlane32/16_swap + DPP row ops, no LDS):
//      lane l' = 4 k1 + a receives y_(4i+a)[k1], i = 0..15
//   B. radix-16 over i in registers, twiddle W_64^(a q)
//   C. radix-4 over a across the 4 lanes of a quad (v_fmac_f32_dpp)
//      -> lane (k1, rr), register q holds Z[k1 + 16 q + 256 r], r = bitrev2(rr)
//   P. real-FFT post-pass: partner Z[M-k] fetched with ds_bpermute (lane 67-l', register 15-q; lanes 0..3 are
//      the k1 = 0 column and pair inside themselves), X[k] = E - i w_k D with the 1/2 folded into the window.
//   T. the result goes into a workgroup tile [1024 bins (+ Nyquist in the pad column)][16 frames] in LDS with a
//      bank-conflict-free row permutation.
// A workgroup is 16 waves = 16 consecutive frames of one clip, one persistent workgroup per CU (LDS is full:
// two 69.6 KB tiles + 24.5 KB of tables incl. the window).  There is no workgroup barrier in the loops: waves
// synchronise through monotonic LDS counters and wait only for what they consume (see the kernels).
//
// Algorithmic HBM bytes per frame at hop 512: power 2048 + 4100 = 6148 B, complex 2048 + 8200 = 10248 B,
// mel 2048 + 4 n_mels.
#include <cstdlib>
#include <type_traits>

#include "fft_device.hpp"
#include "smx_internal.hpp"

namespace smx {
namespace {

using namespace fftdev;

template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_mov(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                               __builtin_bit_cast(int, src), CTRL, 0xF,
                                                               BANK_MASK, false));
}
// x[i] += k * x[i] of the quad partner (quad_perm QP), 8 complex values per statement:
// v_fmac_f32 with a DPP source does the cross-lane read and the butterfly in one instruction.
// hipcc does not form it from the builtins, and its hazard recogniser does not look inside asm, so
// the statement starts with the 2 wait states a DPP read needs after a VALU write.
#define SMX_FD(n, QP) "v_fmac_f32_dpp %" #n ", %" #n ", %16 quad_perm:" QP " row_mask:0xf bank_mask:0xf\n\t"
#define SMX_FMAC_DPP8(QP, V, K)                                                                              \
  asm("s_nop 1\n\t" SMX_FD(0, QP) SMX_FD(1, QP) SMX_FD(2, QP) SMX_FD(3, QP) SMX_FD(4, QP) SMX_FD(5, QP)     \
      SMX_FD(6, QP) SMX_FD(7, QP) SMX_FD(8, QP) SMX_FD(9, QP) SMX_FD(10, QP) SMX_FD(11, QP) SMX_FD(12, QP)   \
      SMX_FD(13, QP) SMX_FD(14, QP) SMX_FD(15, QP)                                                           \
      : "+v"((V)[0].x), "+v"((V)[0].y), "+v"((V)[1].x), "+v"((V)[1].y), "+v"((V)[2].x), "+v"((V)[2].y),       \
        "+v"((V)[3].x), "+v"((V)[3].y), "+v"((V)[4].x), "+v"((V)[4].y), "+v"((V)[5].x), "+v"((V)[5].y),       \
        "+v"((V)[6].x), "+v"((V)[6].y), "+v"((V)[7].x), "+v"((V)[7].y)                                        \
      : "v"(K))

// lanes 32-63 of a <-> lanes 0-31 of b / odd rows of a <-> even rows of b.  Inline asm: this
// hipcc drops the second result of __builtin_amdgcn_permlane{32,16}_swap.  "s_nop 1" covers the
// VALU-write -> v_permlane-read hazard (2 wait states) inside the statement.
__device__ __forceinline__ void swap32(float &a, float &b) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap16(float &a, float &b) {
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// In-wave 16x16 transpose between (lane >> 2) and the register index, for each lane & 3:
// lane (i, a) register k  ->  lane (k, a) register i.  Four butterfly exchanges at lane
// distances 32, 16 (permlane swaps) and 8, 4 (DPP row rotate / shift with bank masks);
// no LDS.  Verified on hardware by tools/probes/transpose_probe.hip.
// eight swaps in ONE statement: the 2 wait states a v_permlane* read needs after a VALU write are paid once per
// batch (an s_nop costs a wave an issue slot like a vector instruction does); inside the batch no swap reads a
// register another one has just written
#define SMX_SWAP8(OP, V, A0, B0, A1, B1, A2, B2, A3, B3, A4, B4, A5, B5, A6, B6, A7, B7)                              \
  asm("s_nop 1\n\t" OP " %0, %1\n\t" OP " %2, %3\n\t" OP " %4, %5\n\t" OP " %6, %7\n\t" OP " %8, %9\n\t" OP            \
      " %10, %11\n\t" OP " %12, %13\n\t" OP " %14, %15"                                                             \
      : "+v"((V)[A0]), "+v"((V)[B0]), "+v"((V)[A1]), "+v"((V)[B1]), "+v"((V)[A2]), "+v"((V)[B2]), "+v"((V)[A3]),     \
        "+v"((V)[B3]), "+v"((V)[A4]), "+v"((V)[B4]), "+v"((V)[A5]), "+v"((V)[B5]), "+v"((V)[A6]), "+v"((V)[B6]),     \
        "+v"((V)[A7]), "+v"((V)[B7]))
__device__ __forceinline__ void transpose16(float (&v)[16]) {
  SMX_SWAP8("v_permlane32_swap_b32", v, 0, 8, 1, 9, 2, 10, 3, 11, 4, 12, 5, 13, 6, 14, 7, 15);
  SMX_SWAP8("v_permlane16_swap_b32", v, 0, 4, 1, 5, 2, 6, 3, 7, 8, 12, 9, 13, 10, 14, 11, 15);
  // (the two DPP steps as one asm block per eight pairs -- no per-pair s_nop from hipcc's hazard recogniser, 34 -> 15 per
  // frame -- measured the same: 0.5705 vs 0.5701 / 0.5732 ms, profiles/r04/ab_dpp_batch.log)
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (!(k & 2)) {
      const float A = v[k], B = v[k | 2];
      v[k | 2] = dpp_mov<0x128, 0x3>(B, A);   // row_ror:8 into lanes 0-7 of each row
      v[k] = dpp_mov<0x128, 0xC>(A, B);       // row_ror:8 into lanes 8-15
    }
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (!(k & 1)) {
      const float A = v[k], B = v[k | 1];
      v[k | 1] = dpp_mov<0x104, 0x5>(B, A);   // row_shl:4 into banks 0, 2
      v[k] = dpp_mov<0x114, 0xA>(A, B);       // row_shr:4 into banks 1, 3
    }
}

// one 16-byte store to a 4-byte-aligned address (rows of [bins; frames] start anywhere).
// hipcc splits an under-aligned 16-byte store into dwordx3 + dword; the hardware takes
// dwordx4 at dword alignment, so emit it directly.  The trailing s_nop keeps the data
// registers intact until the store has read them (asm stores are invisible to hipcc's
// hazard and waitcnt bookkeeping; an uncounted younger store only makes its waits stricter).
using f32x4 = __attribute__((ext_vector_type(4))) float;
#ifndef SMX_STORE_MOD
#define SMX_STORE_MOD ""   // cache-policy bits of the tile stores (" nt", " sc1", ...): A/B builds
#endif
__device__ __forceinline__ void store4_unaligned(float *base /* wave-uniform */, unsigned byte_off, float a,
                                                 float b, float c, float d) {
  const f32x4 v = {a, b, c, d};
  asm volatile("global_store_dwordx4 %0, %1, %2" SMX_STORE_MOD "\n\ts_nop 1" : : "v"(byte_off), "v"(v), "s"(base) : "memory");
}

constexpr int kN = 2048, kM = 1024, kBins = 1025;

struct FastArgs {
  const float *x;
  int64_t n, x_stride;
  int64_t hop, left;
  int pad;
  float pad_value;
  int64_t p0, count;
  float *out;
  int64_t out_stride, out_offset;
  const float *hwin;    // 0.5 * window, 2048
  const float2 *w_m;    // exp(-2 pi i j / 1024)
  const float2 *w_n;    // exp(-2 pi i k / 2048), k <= 1024
  int tiles_per_clip;
  int64_t total_tiles;     // lead * tiles_per_clip: a flat (clip, tile) sequence
  int64_t blocks;          // persistent workgroups; each owns a contiguous range of the sequence
  // power kernel: the frames that touch a border of the signal (reflect / edge / constant padding) are
  // computed by the same launch, after the interior tiles (0 = none: they come as gathered strips)
  int border_left, border_right;       // border frames per clip before / after the interior range
  int64_t border_p0, border_i1;        // first frame of the request, first frame after the interior range
  int64_t border_out_offset;           // output frame offset of frame border_p0
  // complex / mel kernels: the launch covers ALL frames of the request; a frame that touches a border of the signal reads its
  // samples from a gathered, already padded strip (left: frames [p0, border_i0), right: frames [border_i1, ..)) -- only the
  // frame's base pointer differs (a scalar select), so the border frames cost no launches and no registers
  int fold_frames;
  int64_t border_i0;                   // first interior frame (border_i1: first frame after the interior range)
  const float *strip_l, *strip_r;
  int64_t strip_l_stride, strip_r_stride;
  int interleave;       // power kernel: workgroups of an XCD share a chunk of the sequence tile by tile
  int pmode;            // 2: power 2, 1: power 1, 0: general
  int abl_nostore;      // diagnostic builds only
  int abl_noskew;       // diagnostic builds only (ring kernel: rows not moved to their 64-byte boundaries)
  int abl_p32;          // diagnostic builds only (32-lane kernel, timing: 1 loads from one resident tile, 2 stores moved to 64-byte boundaries, 4 stores into one resident tile)
  float half_power;
};

// |X|^p from |X|^2 = pw.  PMODE 2: the square itself.  1: one v_sqrt_f32 (1 ulp; the IEEE expansion of sqrtf costs three
// selects on vcc at 19 cycles each, tools/probes/issue_probe.hip).  0: the general power pw^(p/2) = 2^(h e) 2^(h log2 m),
// pw = m 2^e with m in [0.5, 1): h e is split into its rounded value and the exact residual (one fma), the integer part goes
// to v_ldexp and only a fraction of a few units reaches v_exp_f32, so the result is good to 2-3 ulp over the whole range
// (the library powf: 190 instructions and 13 selects per value, 2.6x the kernel's time).  pw = 0, inf and p = 0 come out as
// powf gives them (0 or inf by the sign of p; 1): the logarithm is clamped to +-FLT_MAX, so that 0 x it is 0, not NaN; a NaN stays one.
// -1 (the 64-lane power kernels kept for A/B timing): the run-time choice of rounds 1-2, both forms evaluated and selected.
template <int PMODE>
__device__ __forceinline__ float power_from_square(float pw, const FastArgs &a) {
#pragma clang fp contract(off)
  if constexpr (PMODE == 2) return pw;
  else if constexpr (PMODE == 1) return __builtin_amdgcn_sqrtf(pw);
  else if constexpr (PMODE == 0) {
    const float h = a.half_power;
    const float ef = (float)__builtin_amdgcn_frexp_expf(pw);
    const float lm = __builtin_amdgcn_fmed3f(__builtin_amdgcn_logf(__builtin_amdgcn_frexp_mantf(pw)), -3.402823466e38f, 3.402823466e38f);
    const float thi = h * ef;
    const float tlo = __builtin_fmaf(h, ef, -thi);
    const float n = __builtin_rintf(thi);
    const float fr = (thi - n) + __builtin_fmaf(h, lm, tlo);
    const float r = __builtin_ldexpf(__builtin_amdgcn_exp2f(fr), (int)n);
    // v_med3_f32 returns min3 when an operand is NaN and min3 drops the NaN: the clamp above turns log(NaN) into -FLT_MAX.
    // A NaN power stays a NaN (as powf and the reference give it): an unordered compare, one select on an SGPR-pair mask.
    return pw != pw ? pw : r;
  } else return a.pmode == 1 ? sqrtf(pw) : __powf(pw, a.half_power);
}

#ifdef SMX_NOFENCE
#define SMX_FENCE() do { } while (0)
#else
#define SMX_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef SMX_PRE
#define SMX_PRE 1
#endif
constexpr bool kPre = SMX_PRE != 0;   // power kernel: twiddle tables read one stage ahead
#ifndef SMX_EPILOGUE_PRE
#define SMX_EPILOGUE_PRE kPre
#endif
constexpr int kFT = 16;                         // frames per tile: one per wave, 16 waves per workgroup
constexpr int kTileStride = kFT + 1;            // floats per tile row (pad column 16)
// A tile holds bins 0..1023 as rows; bin 1024 (Nyquist) of frame f lives in the otherwise
// unused pad slot of row f.  This, and dropping the trivial k1 = 0 twiddle row, is what makes
// two tiles + all tables + the window fit the 160 KB of LDS exactly.
constexpr size_t kTileBytes = (size_t)kM * kTileStride * sizeof(float);                  // 69,632
constexpr size_t kTabABytes = 15 * 64 * sizeof(float2);   // W_M^(l k1), k1 = 1..15     [k1-1][lane]
constexpr size_t kTabPBytes = 16 * 64 * sizeof(float2);   // post-pass twiddles         [q][lane]
constexpr size_t kTabBBytes = 16 * 4 * sizeof(float2);    // W_64^(a q)                 [q][a]
constexpr size_t kWinBytes = (size_t)kM * sizeof(float2); // 0.5 * window as (even, odd) pairs
constexpr size_t kFastLds = 2 * kTileBytes + kTabABytes + kTabPBytes + kTabBBytes + kWinBytes;
static_assert(kFastLds <= 160 * 1024, "LDS budget");

// 16 coalesced 8-byte loads of one 8 KB window: lane l takes elements l + 64 j.
// `base` is wave-uniform, so the loads use two SGPR bases (base, base + 4 KB) plus
// ONE shared 32-bit lane offset with 12-bit immediates.  The +4 KB step is made
// opaque to the optimiser: otherwise hipcc materialises a separate 64-bit VGPR
// address per unrolled load beyond the immediate range (and spills them).
// (__builtin_amdgcn_raw_buffer_load_b64/_b128 are not usable on this toolchain:
// ROCm 7.2 hipcc emits a single-dword load for them.)
__device__ __forceinline__ void load16_f2(const float2 *base, int lane, float2 (&dst)[16]) {
  long hi_off = 512;
  asm volatile("" : "+s"(hi_off));
  const float2 *hi = base + hi_off;
#pragma unroll
  for (int j = 0; j < 16; ++j)
    dst[j] = j < 8 ? base[(unsigned)lane + 64u * j] : hi[(unsigned)lane + 64u * (j - 8)];
}

// raw (unwindowed) samples of frame p: lane l takes z[n] = (x[2n], x[2n+1]), n = l + 64 j.
// Every frame given to this kernel lies inside the signal (border frames arrive
// through gathered, already padded strips -- see launch_stft_fast).  p is wave-uniform.
template <bool ALIGNED>
__device__ __forceinline__ void load_frame(const float *src /* first sample of the frame */, int lane,
                                           float2 (&raw)[16]) {
  if constexpr (ALIGNED) {
    load16_f2(reinterpret_cast<const float2 *>(src), lane, raw);
  } else {
    long hi_off = 1024;
    asm volatile("" : "+s"(hi_off));
    const float *hi = src + hi_off;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float *b = j < 8 ? src : hi;
      const unsigned e = 2u * lane + 128u * (j & 7);
      raw[j] = make_float2(b[e], b[e + 1u]);
    }
  }
}

// One quarter of a wave's share of a finished tile: 16 tile rows (bins) x 4 frames per lane ->
// out[clip][bin][f0 + 4g .. +3].  LDS reads are bank-conflict free (row set {0-3,16-19}+4h per
// half-wave); each 4-lane group stores one 64-byte run.  Called between the FFT stages of the
// NEXT frame so the store traffic is spread over the arithmetic instead of bursting.
// Addresses: `obase` (clip / tile origin) is wave-uniform and stays in SGPRs; lanes carry one
// 32-bit byte offset (goff0) and one LDS offset (row0), the four parts differ by constants.
// Pad-column row that holds the Nyquist bin of frame f: the otherwise unused pad slot of row f.
__device__ __forceinline__ constexpr int nyquist_row(int f) { return f; }

struct FlushLane {
  int row0;          // tile row of part 0: 32 * wave + rloc
  unsigned goff0;    // byte offset of out[bin0][4 g] from the tile origin
  int g;
};
__device__ __forceinline__ void flush_part(const FastArgs &a, const float *tile, int it, const FlushLane &fl,
                                           float *obase, int frames_left, int wave, int lane, int ft = 0) {
  const int row = fl.row0 + 512 * (it >> 1) + 8 * (it & 1);
  const float *src = tile + row * kTileStride + 4 * fl.g;
  const float v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
  // bin(row + 8) = bin + 2, bin(row + 512) = bin + 128
  const unsigned goff = fl.goff0 + (unsigned)(2 * (it & 1) + 128 * (it >> 1)) * (unsigned)a.out_stride * 4u;
  const int fleft = frames_left - 4 * fl.g;    // frames remaining from this column group
#ifdef SMX_DIAG
  if (a.abl_nostore == 1) {   // timing-only ablation: keep the LDS reads alive, drop the HBM stores
    asm volatile("" ::"v"(v0), "v"(v1), "v"(v2), "v"(v3));
    return;
  }
  if (a.abl_nostore == 5) {   // timing-only: every 64-byte run moved down to a 64-byte boundary (no partial sectors)
    const uintptr_t addr = (reinterpret_cast<uintptr_t>(obase) + goff) & ~uintptr_t(63);
    *reinterpret_cast<f32x4 *>(addr + 16 * fl.g) = f32x4{v0, v1, v2, v3};
    return;
  }
  if (a.abl_nostore >= 2) {   // timing-only: the same bytes as runs of 64 << k bytes (k = abl_nostore - 1)
    const int k = a.abl_nostore - 1, lr = 4 << k, rpi = 64 / lr, msk = (1 << k) - 1;
    const int r = (4 * wave + it) * rpi + lane / lr;
    const int bin = (r << k) + (ft & msk);
    const unsigned off = ((unsigned)bin * (unsigned)a.out_stride + 4u * (lane % lr)) * 4u;
    store4_unaligned(obase - 16 * (ft & msk), off, v0, v1, v2, v3);
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
  if (it == 3 && wave == 0 && lane < 16) {   // bin 1024 (row 1024): 16 frames by 16 lanes
    if (lane < frames_left) obase[(int64_t)kM * a.out_stride + lane] = tile[nyquist_row(lane) * kTileStride + kFT];
  }
}

// SQUARE = power 2; otherwise |X|^p through pmode / half_power (power_from_square<-1>)
#if defined(SMX_STAMPS) && !defined(SMX_DIAG)
#define SMX_DIAG 1
#endif
#ifdef SMX_DIAG
#define SMX_ABL_PARAM , int ABL
#define SMX_ABL_ARG , ABL
#define SMX_ABL_ZERO , 0
#define SMX_ABL(n) (ABL == (n))
#else
#define SMX_ABL_PARAM
#define SMX_ABL_ARG
#define SMX_ABL_ZERO
#define SMX_ABL(n) false
#endif

// Where one frame's results go in the LDS tile(s), per lane.  Register q of a lane belongs to tile row
// tile_row0 + 64 q, so its cell is own + kCellStep q.  Before the results are written, the same cells carry the
// post-pass exchange: a lane parks Z[k] there and reads Z[M - k] out of its partner lane's cells (same wave, and
// the LDS executes one wave's operations in order) -- see step P of frame_to_tile.
constexpr int kCellStep = 64 * (kFT + 1);   // floats between the cells of registers q and q + 1
struct Cells {
  float *own;         // cell of register 0
  const float *pg;    // partner lane's cell of ITS register 0 (one register further for the k1 = 0 lanes, whose
                      // register q pairs with register 16 - q): register q >= 1 reads pg + kCellStep (15 - q)
  const float *p0;    // the cell register 0 pairs with (k1 = 0 lanes: register 0 of lane bitrev2((4 - r) & 3))
  float *nyq;         // lane 0: where the frame's Nyquist bin goes
};

// Per-lane constants of the frame pipeline (see the header comment for the digit layout).
struct LaneConst {
  int k1, qa, r;
  float s12, kap1, kap2;    // quad radix-4: sign folded into the data, butterfly multipliers
  bool rot, low4;
  int prow_g, prow_0;       // tile rows of the post-pass partners: of register q >= 1 (prow_g + 64 (15 - q)) and of register 0
  int tile_row0;            // tile row of register q is tile_row0 + 64 q
  const float2 *tabA_l, *winL_l, *tabP_l, *tabB_l;
  const float2 *tabP_g;     // the same twiddles in global memory: exp(-2 pi i k / N), k = k1 + 256 r + 16 q
};

struct Lds {
  float *tiles;
  float2 *tabA, *tabP, *tabB, *winL;
};

__device__ __forceinline__ Lds carve_lds(unsigned char *smem) {
  Lds l;
  l.tiles = reinterpret_cast<float *>(smem);   // two [1024][17] tiles, used alternately
  l.tabA = reinterpret_cast<float2 *>(smem + 2 * kTileBytes);
  l.tabP = reinterpret_cast<float2 *>(smem + 2 * kTileBytes + kTabABytes);
  l.tabB = reinterpret_cast<float2 *>(smem + 2 * kTileBytes + kTabABytes + kTabPBytes);
  l.winL = reinterpret_cast<float2 *>(smem + 2 * kTileBytes + kTabABytes + kTabPBytes + kTabBBytes);
  return l;
}

// The cells of the frame that owns column `wave` of `tile` (the unskewed kernels: complex, mel, border frames).
__device__ __forceinline__ Cells cells_of_column(const LaneConst &L, float *tile, int wave);

// Fills the workgroup-shared LDS tables (wave w writes row w of each) and returns this lane's constants.
// The caller must __syncthreads() before the tables are read.
template <bool FILL_TABP = true>
__device__ __forceinline__ LaneConst setup_lane(const FastArgs &a, const Lds &lds, int tid, int lane, int wave) {
  LaneConst L;
  L.k1 = lane >> 2;
  L.qa = lane & 3;
  L.r = ((L.qa & 1) << 1) | (L.qa >> 1);
  if (wave > 0) lds.tabA[(wave - 1) * 64 + lane] = a.w_m[lane * wave];          // W_M^(l k1), k1 = wave
  lds.winL[tid] = reinterpret_cast<const float2 *>(a.hwin)[tid];                // the whole window, once per workgroup
  if constexpr (FILL_TABP) lds.tabP[wave * 64 + lane] = a.w_n[L.k1 + 256 * L.r + 16 * wave];   // exp(-2 pi i k / N), k = k1 + 16 q + 256 r
  L.tabP_g = a.w_n + L.k1 + 256 * L.r;
  if (lane < 4) {
    // row q = 0 (all ones) is never read: it holds the power kernel's synchronisation counters
    if (wave > 0) {                                                             // s1 s2 W_64^(a q), a = lane
      const float sg = ((lane < 2) != ((lane & 1) == 0)) ? -1.0f : 1.0f;
      const float2 w = a.w_m[16 * lane * wave];
      lds.tabB[wave * 4 + lane] = make_float2(sg * w.x, sg * w.y);
    }
    else reinterpret_cast<unsigned *>(lds.tabB)[lane] = 0u;
  }
  L.tabA_l = lds.tabA + lane - 64;   // row k1 - 1
  L.winL_l = lds.winL + lane;
  L.tabP_l = lds.tabP + lane;
  L.tabB_l = lds.tabB + L.qa;
  const float s1 = L.qa < 2 ? 1.0f : -1.0f, s2 = (L.qa & 1) ? -1.0f : 1.0f;
  L.s12 = s1 * s2;
  L.kap1 = -s1;
  L.kap2 = -s2;
  L.rot = L.qa == 3;
  L.low4 = lane < 4;
  L.tile_row0 = 4 * L.k1 + L.r;                              // row' = 4 (k1 + 16 q) + r
  // partner of bin k = k1 + 16 q + 256 r is M - k: lane 67 - l, register 15 - q  (k1' = 16 - k1, r' = 3 - r);
  // in the k1 = 0 column: register 16 - q of the lane with r' = 3 - r, and for q = 0 register 0 of r' = (4 - r) & 3
  if (lane >= 4) {
    L.prow_g = 4 * (16 - L.k1) + (3 - L.r);
    L.prow_0 = L.prow_g + 64 * 15;
  } else {
    L.prow_g = (3 - L.r) + 64;
    L.prow_0 = (4 - L.r) & 3;
  }
  return L;
}

__device__ __forceinline__ Cells cells_of_column(const LaneConst &L, float *tile, int wave) {
  Cells c;
  float *col = tile + wave;
  c.own = col + L.tile_row0 * kTileStride;
  c.pg = col + L.prow_g * kTileStride;
  c.p0 = col + L.prow_0 * kTileStride;
  c.nyq = tile + nyquist_row(wave) * kTileStride + kFT;
  return c;
}

// Persistent workgroups: block b owns a contiguous range of the flat (clip, tile) sequence.
// XCD-aware: blocks that share an XCD (b % 8) get neighbouring ranges, i.e. whole runs of clips,
// so halo re-reads and the partial output lines of neighbouring tiles meet in one L2.
__device__ __forceinline__ void block_to_range(const FastArgs &a, int64_t &tau_begin, int64_t &tau_end) {
  int64_t vb = blockIdx.x;
  const int64_t nb = a.blocks, q = nb / 8, r = nb % 8, xcd = vb % 8, idx = vb / 8;
  vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int64_t base = a.total_tiles / nb, extra = a.total_tiles % nb;
  tau_begin = vb * base + (vb < extra ? vb : extra);
  tau_end = tau_begin + base + (vb < extra ? 1 : 0);
}

// sample s of a signal of n samples extended by the configuration's padding (stft.ml:300-338); 32-bit positions
__device__ __forceinline__ float fetch_padded(const float *x, int n, int s, int pad, float pad_value) {
  if ((unsigned)s < (unsigned)n) return x[s];
  if (pad == SMX_PAD_REFLECT) {
    if (n == 1) return x[0];
    const int period = 2 * (n - 1);
    int m = s % period;
    if (m < 0) m += period;
    return x[m < n ? m : period - m];
  }
  if (pad == SMX_PAD_EDGE) return x[s < 0 ? 0 : n - 1];
  return pad_value;
}

// The tiles of one workgroup: tau0, tau0 + step, ... (ntiles of them) of the flat (clip, tile)
// sequence, walked without divisions (a scalar 64-bit division costs a few hundred dependent SALU
// instructions; the only ones are in init()).
struct TileWalk {
  int ntiles, ft, step_clips, step_tiles;   // ft: tile index inside the clip
  int uid;                                  // this workgroup's rank among the launch's workgroups in tile order (its first tile when it has one)
  const float *xclip;                       // first sample of the current clip
  float *oclip;                             // output origin of the current clip
  int64_t x_step, o_step;                   // per-clip strides of input and output

  __device__ __forceinline__ void init(const FastArgs &a, float *out, int64_t out_clip_floats) {
    int64_t tau0;
    int step;
    uid = (int)blockIdx.x;
    if (a.interleave) {
      // The workgroups of one XCD walk a contiguous chunk of the sequence side by side: at any time the
      // XCD is writing ~32 neighbouring tiles of the same clip, i.e. for every bin one contiguous run of
      // ~2 KB, which its L2 can assemble into whole lines before they go to HBM.
      const int64_t nb = a.blocks, xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
      if (a.interleave == 2) {   // all workgroups side by side; an XCD holds 32 neighbouring tiles of every 256
        const int64_t q = nb / 8, r = nb % 8;
        tau0 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        uid = (int)tau0;
        step = (int)nb;
        ntiles = tau0 < a.total_tiles ? (int)((a.total_tiles - tau0 + nb - 1) / nb) : 0;
      } else {
        const int64_t nx = (nb - xcd + 7) / 8;                            // workgroups on this XCD
        const int64_t nxcd = nb < 8 ? nb : 8;                             // XCDs that received a workgroup
        const int64_t x0 = a.total_tiles * xcd / nxcd, x1 = a.total_tiles * (xcd + 1) / nxcd;
        tau0 = x0 + idx;
        step = (int)nx;
        ntiles = tau0 < x1 ? (int)((x1 - tau0 + nx - 1) / nx) : 0;
      }
    } else {
      int64_t tau_end;
      block_to_range(a, tau0, tau_end);
      step = 1;
      ntiles = (int)(tau_end - tau0);
    }
    if (ntiles <= 0) return;
    ft = (int)(tau0 % a.tiles_per_clip);
    x_step = a.x_stride;
    o_step = out_clip_floats;
    xclip = a.x + (tau0 / a.tiles_per_clip) * x_step;
    oclip = out + (tau0 / a.tiles_per_clip) * o_step;
    step_clips = step / a.tiles_per_clip;
    step_tiles = step % a.tiles_per_clip;
  }
  // the tile after the current one
  __device__ __forceinline__ void peek(const FastArgs &a, int &ftn, const float *&xn, float *&on) const {
    ftn = ft + step_tiles;
    int dclip = step_clips;
    if (ftn >= a.tiles_per_clip) {
      ftn -= a.tiles_per_clip;
      ++dclip;
    }
    xn = xclip + dclip * x_step;
    on = oclip + dclip * o_step;
  }
};

// One frame: raw samples (registers) -> window -> FFT(1024 complex) -> real post-pass -> |X|^p
// written to the frame's cells in the LDS tile (`cells`; for the unskewed kernels column `wave` of a tile:
// cells_of_column).  `hook.at<P>()` is called at 16 points between the stages;
// the power kernel uses them to trickle out the previous tile's stores.
// PRE: each twiddle table is read from LDS one stage before it is used (30 more live registers),
// so its latency -- long when 16 waves queue on the LDS pipe -- hides behind the stage in between.
// CPLX: the spectrum itself goes to the tile (real parts in `tile`, imaginary parts in the plane after it).
// TABPG: the post-pass twiddles come from global memory (a.w_n, L2 resident), requested before stage C,
//        for the kernel that uses the LDS space of that table for something else (mel).
template <int PMODE, bool PRE, bool CPLX, bool TABPG SMX_ABL_PARAM, class Hook>
__device__ __forceinline__ void frame_to_tile(const FastArgs &a, const LaneConst &L, float2 (&raw)[16],
                                              Cells cells, int lane, const Hook &hook) {
  c32 v[16];
  float2 win[16], tw[16];
  constexpr bool kNoLds = SMX_ABL(12) || SMX_ABL(13) || SMX_ABL(14);   // timing-only: the arithmetic and its table reads alone (no exchange, no tile, no memory, no waits)
#pragma unroll
  for (int j = 0; j < 16; ++j) win[j] = L.winL_l[64 * j];
  if constexpr (PRE) {
#pragma unroll
    for (int k = 1; k < 16; ++k) tw[k] = L.tabA_l[64 * k];
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = {raw[j].x * win[j].x, raw[j].y * win[j].y};   // hipcc fuses these into the first butterflies (a w_a +- c w_c: one product, two fmas)
  if constexpr (SMX_ABL(6) || SMX_ABL(9)) {   // timing-only: memory traffic and synchronisation without the FFT (9: no loads either)
    hook.template at<0>(); hook.template at<2>(); hook.template at<5>(); hook.template at<8>(); hook.template at<11>();
    hook.ready(cells);
#pragma unroll
    for (int q = 0; q < 16; ++q) cells.own[kCellStep * q] = v[q].x + v[q].y;
    return;
  }
  hook.template at<0>();
  SMX_FENCE();
  // A: radix-16 over j, twiddle W_M^(l k1)
  fft16_pass1(v);
  hook.template at<1>();
  fft16_pass2(v);
  hook.template at<2>();
  SMX_FENCE();
#pragma unroll
  for (int k = 1; k < 16; ++k) {
    const float2 w = PRE ? tw[k] : L.tabA_l[64 * k];
    v[k] = cmul(v[k], c32{w.x, w.y});
  }
  if constexpr (PRE) {
#pragma unroll
    for (int q = 1; q < 16; ++q) tw[q] = L.tabB_l[4 * q];
  }
  hook.template at<3>();
  SMX_FENCE();
  // X: transpose lane (i, a) register k1 -> lane (k1, a) register i, in registers (permlane swaps + DPP); a
  // variant through the wave's own tile column in LDS removed ~500 VALU cycles per frame but measured slower
  // (it ties the transpose to the tile buffer's availability)
  {
    float re[16], im[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { re[k] = v[k].x; im[k] = v[k].y; }
    if constexpr (!SMX_ABL(5) && !SMX_ABL(11) && !SMX_ABL(13)) transpose16(re);
    hook.template at<4>();
    if constexpr (!SMX_ABL(5) && !SMX_ABL(11) && !SMX_ABL(13)) transpose16(im);
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = {re[k], im[k]};
  }
  hook.template at<5>();
  SMX_FENCE();
  // B: radix-16 over i, twiddle W_64^(a q)
  fft16_pass1(v);
  hook.template at<6>();
  fft16_pass2(v);
  hook.template at<7>();
  SMX_FENCE();
#pragma unroll
  for (int q = 1; q < 16; ++q) {
    const float2 wb = PRE ? tw[q] : L.tabB_l[4 * q];
    v[q] = cmul(v[q], c32{wb.x, wb.y});
  }
  if constexpr (TABPG) {
#pragma unroll
    for (int q = 0; q < 16; ++q) tw[q] = L.tabP_g[16 * q];
  } else if constexpr (PRE) {
#pragma unroll
    for (int q = 0; q < 16; ++q) tw[q] = L.tabP_l[64 * q];
  }
  hook.template at<8>();
  SMX_FENCE();
  // C: radix-4 across the quad.  lane a ends with r = bitrev2(a).
  // Butterfly 1 pairs lane a with a ^ 2 (u = +-v + partner), then lane 3 multiplies by -i, butterfly 2
  // pairs a with a ^ 1.  The lane signs s1 (a < 2 ? + : -) and s2 (a even ? + : -) are folded into the
  // W_64 table (and into v[0]), which turns each butterfly into x += kappa * partner(x) with
  // kappa1 = -s1, kappa2 = -s2: one v_fmac_f32_dpp per component.  Sign flips are exact, so the values
  // are those of the plain formulation bit for bit.
  if constexpr (!SMX_ABL(14)) {
  v[0].x *= L.s12;
  v[0].y *= L.s12;
  SMX_FMAC_DPP8("[2,3,0,1]", v, L.kap1);
  SMX_FMAC_DPP8("[2,3,0,1]", v + 8, L.kap1);
  hook.template at<9>();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const c32 u = v[q];
    v[q] = L.rot ? c32{u.y, -u.x} : u;
  }
  SMX_FMAC_DPP8("[1,0,3,2]", v, L.kap2);
  SMX_FMAC_DPP8("[1,0,3,2]", v + 8, L.kap2);
  }
  hook.template at<10>();
  SMX_FENCE();
  // P: real-FFT post-pass X[k] = E - i w_k D, E = Z[k] + conj Z[M-k], D = Z[k] - conj Z[M-k] (the 1/2 is in the
  // window).  Z[M-k] sits in another lane (67 - l, register 15 - q).  The exchange goes through the frame's own
  // cells of the tile, which are free from here on: every lane parks its 16 real parts, reads its partners',
  // then the same with the imaginary parts -- 64 plain LDS accesses (2-4 cycles of the LDS pipe each) where
  // 32 ds_bpermute_b32 (~24 cycles each) were the single largest load on that pipe.  One wave's LDS operations
  // execute in order, so no wait separates the rounds.
  hook.ready(cells);   // waits until the frame's cells are free; the ring kernel derives them here
  const float nyq = 2.0f * (v[0].x - v[0].y);   // X[M] = Re Z0 - Im Z0 (true scale), lane 0
  float px[16], py[16];
  constexpr int kPlane = kTileBytes / sizeof(float);   // CPLX: the imaginary plane
  if constexpr (SMX_ABL(4) || kNoLds) {   // timing-only: no exchange
#pragma unroll
    for (int q = 0; q < 16; ++q) { px[q] = v[15 - q].x; py[q] = v[15 - q].y; }
  } else if constexpr (CPLX) {   // both planes are this tile's: one round
#pragma unroll
    for (int q = 0; q < 16; ++q) { cells.own[kCellStep * q] = v[q].x; cells.own[kCellStep * q + kPlane] = v[q].y; }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float *src = q == 0 ? cells.p0 : cells.pg + kCellStep * (15 - q);
      px[q] = src[0];
      py[q] = src[kPlane];
    }
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) cells.own[kCellStep * q] = v[q].x;
#pragma unroll
    for (int q = 0; q < 16; ++q) px[q] = q == 0 ? cells.p0[0] : cells.pg[kCellStep * (15 - q)];
#pragma unroll
    for (int q = 0; q < 16; ++q) cells.own[kCellStep * q] = v[q].y;
#pragma unroll
    for (int q = 0; q < 16; ++q) py[q] = q == 0 ? cells.p0[0] : cells.pg[kCellStep * (15 - q)];
  }
  SMX_FENCE();
  auto finish_bin = [&](int q, float2 w) {
    const c32 e = {v[q].x + px[q], v[q].y - py[q]};
    const c32 d = {v[q].x - px[q], v[q].y + py[q]};
    const float tr = e.x + w.x * d.y + w.y * d.x;
    const float ti = e.y - w.x * d.x + w.y * d.y;
    if constexpr (CPLX) {
      cells.own[kCellStep * q] = tr;
      cells.own[kCellStep * q + kPlane] = ti;
    } else {
      const float pw = power_from_square<PMODE>(tr * tr + ti * ti, a);
      if constexpr (kNoLds) asm volatile("" ::"v"(pw));
      else cells.own[kCellStep * q] = pw;
    }
  };
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    finish_bin(q, (PRE || TABPG) ? tw[q] : L.tabP_l[64 * q]);
    if (q == 3) hook.template at<11>();
    if (q == 7) hook.template at<12>();
    if (q == 11) hook.template at<13>();
    if (q == 15) { hook.template at<14>(); hook.template at<15>(); }
  }
  if (lane == 0 && !kNoLds) {
    float pw = nyq;                       // CPLX: X[M] is real
    if constexpr (!CPLX) {
      pw = PMODE == 1 || (PMODE == -1 && a.pmode == 1) ? fabsf(nyq) : power_from_square<PMODE>(nyq * nyq, a);
    }
    *cells.nyq = pw;   // Nyquist bin of this frame: a pad slot
  }
}

template <bool ALIGNED SMX_ABL_PARAM>
__device__ __forceinline__ void prefetch_frame(const FastArgs &a, const float *src, int lane, float2 (&raw)[16]) {
  if constexpr (SMX_ABL(2) || SMX_ABL(3) || SMX_ABL(9) || SMX_ABL(12) || SMX_ABL(13) || SMX_ABL(14)) {
#pragma unroll
    for (int j = 0; j < 16; ++j) raw[j] = make_float2((float)(lane + j) + raw[j].x * 0.0f, (float)(lane - j));
  } else if constexpr (SMX_ABL(10) || SMX_ABL(11)) {   // timing-only: 4 of the 16 loads (what re-using the 75 % overlap of consecutive frames would leave)
    const float2 *b = reinterpret_cast<const float2 *>(src);
#pragma unroll
    for (int j = 0; j < 4; ++j) raw[j] = b[(unsigned)lane + 64u * j];
#pragma unroll
    for (int j = 4; j < 16; ++j) raw[j] = make_float2(raw[j & 3].x + (float)j, raw[j & 3].y - (float)j);
  } else {
    load_frame<ALIGNED>(src, lane, raw);
  }
}

// ---- power spectrogram kernel -------------------------------------------------------------------
// Wave-level synchronisation through two pairs of monotonic LDS counters instead of a workgroup
// barrier per tile.  All 16 waves of a barrier-synchronised workgroup sit in the same phase of the
// frame at the same time (all reading twiddles, all in the butterflies, ...), so the LDS pipe and the
// VALUs are busy alternately, never together -- measured: tile time = VALU time + LDS time.  With
// counters a wave only waits for what it really depends on, one frame of slack each way:
//   filled[b]  += 1 by every wave once its column of the tile in buffer b is written (or skipped);
//                 a wave stores its share of that tile only when filled[b] reaches 16 per tile;
//   drained[b] += 1 by every wave once it has read its share of the tile out of buffer b;
//                 a wave writes into buffer b again only when drained[b] reaches 16 per tile.
// The waves drift apart by up to a frame and overlap each other's LDS and VALU phases.
struct Counters {
  unsigned *filled, *drained;   // [2] each, in the unused q = 0 row of the W_64 table
};
#ifndef SMX_WAIT_SLEEP
#define SMX_WAIT_SLEEP 2
#endif
#ifndef SMX_RING_PRIO
#define SMX_RING_PRIO 0
#endif
__device__ __forceinline__ void lds_signal(unsigned *c, int lane) {
  if (lane == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_wait(unsigned *c, unsigned target) {
  while ((unsigned)__builtin_amdgcn_readfirstlane(
             (int)__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
    __builtin_amdgcn_s_sleep(SMX_WAIT_SLEEP);
}

struct NoHook {
  template <int P>
  __device__ __forceinline__ void at() const {}
  __device__ __forceinline__ void ready(Cells &) const {}
};

struct SyncHook {
  unsigned *drained;
  unsigned target;
#ifdef SMX_STAMPS
  unsigned long long *stamp_sum, *stamp_prev_p;
#endif
  // the tile buffer about to be written has been read out by every wave
  __device__ __forceinline__ void ready(Cells &) const { lds_wait(drained, target); }
  template <int P>
  __device__ __forceinline__ void at() const {
#ifdef SMX_STAMPS
    unsigned long long &stamp_prev = *stamp_prev_p;
    SMX_STAMP(1 + P);
#endif
  }
};

#include "stft_fast_p32.hpp"   // the 32-lane frame pipeline: stft2048_power32_kernel
#include "stft_fast_mel32.hpp" // the fused audio -> mel kernel on the 32-lane pipeline: stft2048_mel32_kernel
#include "stft_fast_p16.hpp"   // the same pipeline with a frame in 16 / 8 lanes: stft_power_lanes_kernel (power spectrogram at fft 1024 / 512)


// ---- fused audio -> mel kernel -------------------------------------------------------------------
// Soundml.mel_spectrogram (soundml.ml:12-24) = Mel.apply (Stft.power_spectrum x) with the power
// tile kept in LDS: the finished tile [1024 bins][16 frames] is the B operand of
// v_mfma_f32_16x16x4_f32 (k = bins, n = frames), the banded filterbank the A operand, and only
// [n_mels][16] floats per tile reach HBM (8 KB instead of 65.6 KB).  The filters are banded, so
// each 16-mel block walks only the union of its rows' supports (274 MFMAs per tile for 128 mels
// at 2048 / 48 kHz instead of 2056).  A host-built plan gives every wave one item
// (mel block, K range): heavy blocks are split in K between an owner wave and up to three helper
// waves whose partial 16x16 accumulators travel through the spare pad column of the tile buffer
// and are added by the owner in a fixed order (deterministic, no atomics).
// Pipeline per tile t (one barrier): owner finishes tile t-2 | every wave runs its MFMA item on
// tile t-1 | every wave computes its frame of tile t.
using f32x4v = __attribute__((ext_vector_type(4))) float;

// Partial 16x16 sums travel from helper to owner through LDS: slots 0-2 in the pad column of the tile's
// own buffer, slots 3-6 in the space of the post-pass twiddle table (which this kernel reads from global
// memory instead), one set per buffer parity.
constexpr int kMelHelpers = 11, kMelPadSlots = 3, kMelMaxSteps = 24;   // 11 helper pieces: 20 .. 128 mels all get a plan (7 covered 128 only)
struct MelItem {          // one per wave; wave-uniform, read through scalar loads
  int block;              // 16-mel block index
  int k4_begin, k4_count; // MFMA steps: bins [4 k4_begin, 4 (k4_begin + k4_count))
  int a_offset;           // offset (in 64-float rows) of this item's A operands in w_mfma
  int slot;               // helper: partial slot 0..7; owner / idle: -1
  int owner;              // 1: this wave stores the block's result
  int nslots;             // owner: number of helper partials to add
  int slots[kMelHelpers];
};

struct MelFusedArgs {
  const MelItem *items;   // [16]
  const float *w_mfma;    // [rows][64]: A operand of MFMA step i of an item, in lane order
  float *scratch;         // per workgroup: partial-sum records of helper slots >= kMelPadSlots
  float *out;             // [lead; n_mels; out_stride]
  int64_t out_stride, out_offset;
  int n_mels;
};


struct ReadyHook {   // frame_to_tile calls ready() just before the powers overwrite the tile buffer
  unsigned *c;
  unsigned target;
  template <int P>
  __device__ __forceinline__ void at() const {}
  __device__ __forceinline__ void ready(Cells &) const { lds_wait(c, target); }
};

template <bool ALIGNED, int PMODE, bool STRIP>
__global__ void __launch_bounds__(1024) stft2048_mel_kernel(FastArgs a, MelFusedArgs m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Lds lds = carve_lds(smem);
  const LaneConst L = setup_lane(a, lds, tid, lane, wave);
  TileWalk tw;
  tw.init(a, m.out + m.out_offset, (int64_t)m.n_mels * m.out_stride);
  if (tw.ntiles <= 0) return;                  // uniform for the workgroup
  const int ntiles = tw.ntiles;
  // This wave's item, every field in an SGPR.  (Only static indices below: a dynamically indexed copy of the
  // struct would live in scratch memory and come back as vector values.)
  struct {
    int block, k4_begin, k4_count, a_offset, slot, owner, nslots, slots[kMelHelpers];
  } item;
  {
    const MelItem *src = m.items + wave;
    item.block = __builtin_amdgcn_readfirstlane(src->block);
    item.k4_begin = __builtin_amdgcn_readfirstlane(src->k4_begin);
    item.k4_count = __builtin_amdgcn_readfirstlane(src->k4_count);
    item.a_offset = __builtin_amdgcn_readfirstlane(src->a_offset);
    item.slot = __builtin_amdgcn_readfirstlane(src->slot);
    item.owner = __builtin_amdgcn_readfirstlane(src->owner);
    item.nslots = __builtin_amdgcn_readfirstlane(src->nslots);
#pragma unroll
    for (int i = 0; i < kMelHelpers; ++i) item.slots[i] = __builtin_amdgcn_readfirstlane(src->slots[i]);
  }
  // This wave's A operands (filterbank weights of its item in MFMA lane order) stay in registers for the
  // whole kernel: they are the same for every tile.  Rows past the item belong to the next items or to
  // the table's zero padding and are never multiplied.
  float aw[kMelMaxSteps];
  {
    const float *abase = m.w_mfma + (int64_t)item.a_offset * 64;    // wave-uniform
#pragma unroll
    for (int j = 0; j < kMelMaxSteps; ++j) {
      const float w = abase[(unsigned)lane + 64u * (unsigned)j];
      aw[j] = j < item.k4_count ? w : 0.0f;   // steps past the item multiply by zero
    }
  }

  auto frame_ptr = [&](const float *xc, int t, bool &hv) {
    const int64_t f0 = (int64_t)t * kFT;
    hv = f0 + wave < a.count;
    const int64_t p = a.p0 + f0 + (hv ? wave : 0);
    if (a.fold_frames && (p < a.border_i0 || p >= a.border_i1)) {   // wave-uniform; four frames per clip at C3
      const int64_t clip = (xc - a.x) / a.x_stride;
      return p < a.border_i0 ? a.strip_l + clip * a.strip_l_stride + (p - a.p0) * a.hop
                             : a.strip_r + clip * a.strip_r_stride + (p - a.border_i1) * a.hop;
    }
    return xc + (p * a.hop - a.left);
  };
  float2 raw[16];
  bool have;
  load_frame<ALIGNED>(frame_ptr(tw.xclip, tw.ft, have), lane, raw);
  __syncthreads();   // tables and zeroed counters visible: the only workgroup barrier of the kernel
  const int pad_lane = (16 + lane) * kTileStride + kFT;     // pad-column slot of this lane (rows >= 16)
  constexpr int kTileFloats = kTileBytes / sizeof(float);
  // output origin (clip, first frame) and frames left in the clip, of tile t-1 / t-2
  float *out_cur = nullptr, *out_m1 = nullptr, *out_m2 = nullptr;
  int64_t left_cur = 0, left_m1 = 0, left_m2 = 0;
  f32x4v acc_prev = {0.f, 0.f, 0.f, 0.f};
  // Wave-level synchronisation through monotonic LDS counters (see the power kernel), per buffer:
  //   filled: the wave's column of the tile is written;  mdone: its MFMA item over the tile (and its
  //   partial sums, if it is a helper) is done;  fin: the owner has read the partials of the tile.
  unsigned *const c_filled = reinterpret_cast<unsigned *>(lds.tabB);
  unsigned *const c_mdone = c_filled + 2, *const c_fin = c_filled + 4;
  auto nth = [](int tile) { return 16u * (((unsigned)tile >> 1) + 1u); };   // counter value once tile `tile` is through

  // Partial slot sl of the tile in buffer `buf` (parity `par`).  Slots 0-2 are in LDS (pad column of the
  // tile's buffer); further slots are 1 KB records of this workgroup in global memory (L2): one 16-byte
  // store per helper lane, read back by the owner (same CU, same L1: workgroup scope) an iteration later.
  float *const gslots = m.scratch + (int64_t)blockIdx.x * (2 * (kMelHelpers - kMelPadSlots) * 256);
  auto put_partial = [&](float *buf, int par, int sl, f32x4v acc) {
    if (sl < kMelPadSlots) {
      float *pp = buf + pad_lane + (256 * sl) * kTileStride;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) pp[(64 * reg) * kTileStride] = acc[reg];
    } else {
      float *g = gslots + (((kMelHelpers - kMelPadSlots) * par + (sl - kMelPadSlots)) * 64 + lane) * 4;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        __hip_atomic_store(g + reg, acc[reg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // complete before this wave signals mdone
    }
  };
  auto get_partial = [&](const float *buf, int par, int sl) -> f32x4v {
    f32x4v v;
    if (sl < kMelPadSlots) {
      const float *pp = buf + pad_lane + (256 * sl) * kTileStride;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) v[reg] = pp[(64 * reg) * kTileStride];
    } else {
      const float *g = gslots + (((kMelHelpers - kMelPadSlots) * par + (sl - kMelPadSlots)) * 64 + lane) * 4;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) v[reg] = __hip_atomic_load(g + reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return v;
  };
  // (A) owner: finish the tile whose MFMA partials were produced one iteration ago
  auto finish = [&](float *buf, int par, float *obase, int64_t frames_left) {
    if (!item.owner) return;
    f32x4v total = acc_prev;
#pragma unroll
    for (int s = 0; s < kMelHelpers; ++s) {   // fixed order: the sum does not depend on timing
      if (s < item.nslots) {
        const f32x4v part = get_partial(buf, par, item.slots[s]);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) total[reg] += part[reg];
      }
    }
    const int f = lane & 15;
    if (f < frames_left) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int mel = 16 * item.block + 4 * (lane >> 4) + reg;
        if (mel < m.n_mels) obase[(int64_t)mel * m.out_stride + f] = total[reg];
      }
    }
  };
  // (B) every wave: its MFMA item over the finished tile in `buf`: A from registers, B from the tile.
  // Tile rows advance by 16 per step (4 bins); an item spans at most 128 bins, so it crosses a 256-bin
  // block boundary (where the row index wraps) at most once: two lane bases, chosen per step by a scalar.
  const int kk = lane >> 4, fcol = lane & 15;
  const int kabs0 = 4 * item.k4_begin + kk;
  const int cross = (256 - ((4 * item.k4_begin) & 255)) >> 2;                 // first step of the next block (scalar)
  const int boff0 = (4 * (kabs0 & 255) + (kabs0 >> 8)) * kTileStride + fcol;
  constexpr int kWrap = (4 * 256 - 1) * kTileStride;                          // after the wrap: row - 1023
  const bool touches_end = 4 * (item.k4_begin + item.k4_count) > kM;          // the item of bins >= 1024 (scalar)
  auto mfma_item = [&](float *buf, int par) {
    if (item.k4_count <= 0) return;
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    // Steps in groups of 4; inside the last group, steps past the item have zero weights and re-read the
    // item's last rows.  One lane base that changes with the buffer (so the per-step addresses are formed
    // here, one add each, instead of living in registers across the FFT) plus a scalar offset per step.
    // Two accumulators halve the dependent MFMA chain; they are added in a fixed order.
    const float *bp = buf + boff0;
    const float nyq = buf[fcol * kTileStride + kFT];
    f32x4v acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < kMelMaxSteps; g += 4) {   // groups of 4 steps; whole groups past the item are skipped (scalar branch)
      if (g < item.k4_count) {
        float bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int j = g + i;
          const int jj = j < item.k4_count ? j : item.k4_count - 1;   // scalar
          const int sofs = __builtin_amdgcn_readfirstlane(jj < cross ? 16 * kTileStride * jj : 16 * kTileStride * jj - kWrap);
          bv[i] = bp[sofs];
          // bins 1024..1027: the Nyquist bin of frame f sits in the pad slot of row f, the rest do not exist
          if (touches_end && jj == item.k4_count - 1) bv[i] = kk == 0 ? nyq : 0.0f;
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[g], bv[0], acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[g + 1], bv[1], acc1, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[g + 2], bv[2], acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[g + 3], bv[3], acc1, 0, 0, 0);
      }
    }
    acc += acc1;
    if (item.owner) {
      acc_prev = acc;
    } else {
      put_partial(buf, par, item.slot, acc);
    }
  };

  // Iteration t: frame of tile t -> buffer t & 1 | owner finishes tile t-2 | MFMA item over tile t-1.
  // Each step waits only for what it consumes, a whole frame behind the producers, so the waves
  // drift apart and overlap each other's LDS, MFMA and VALU phases; two extra iterations drain.
  for (int t = 0; t < ntiles + 2; ++t) {
    const int b = t & 1;
    float *tcur = lds.tiles + b * kTileFloats;
    float *tprev = lds.tiles + (b ^ 1) * kTileFloats;
    if (t < ntiles) {
      // every MFMA read of tile t-2 (same buffer) must be over before the powers of tile t land
      if (have) {
        frame_to_tile<PMODE, false, false, false SMX_ABL_ZERO>(a, L, raw, cells_of_column(L, tcur, wave), lane,
                                                                        ReadyHook{c_mdone + b, t >= 2 ? nth(t - 2) : 0u});
      }
      lds_signal(c_filled + b, lane);
      int ftnext;
      const float *xnext;
      float *onext;
      tw.peek(a, ftnext, xnext, onext);
      bool have_next;
      const float *xsrc = t + 1 < ntiles ? xnext : tw.xclip;
      const float *src = frame_ptr(xsrc, t + 1 < ntiles ? ftnext : tw.ft, have_next);
      have_next = have_next && t + 1 < ntiles;
      load_frame<ALIGNED>(src, lane, raw);
      out_cur = tw.oclip + tw.ft * kFT;   // wave-uniform
      left_cur = a.count - (int64_t)tw.ft * kFT;
      have = have_next;
      tw.xclip = xnext;
      tw.oclip = onext;
      tw.ft = ftnext;
    }
    if (t >= 2) {   // tile t-2: its partials sit in this buffer's pad column once every helper is through
      lds_wait(c_mdone + b, nth(t - 2));
      finish(tcur, b, out_m2, left_m2);
      lds_signal(c_fin + b, lane);
    }
    if (t >= 1 && t - 1 < ntiles) {   // tile t-1
      lds_wait(c_filled + (b ^ 1), nth(t - 1));
      if (t >= 3) lds_wait(c_fin + (b ^ 1), nth(t - 3));   // the partial slots of tile t-3 have been read
      mfma_item(tprev, b ^ 1);
      lds_signal(c_mdone + (b ^ 1), lane);
    }
    out_m2 = out_m1;
    left_m2 = left_m1;
    out_m1 = out_cur;
    left_m1 = left_cur;
  }
}

}  // namespace

namespace {

// Border strips: out[clip][j] = padded sample at signal position pos0 + j
// (reflect / edge / constant extension of stft.ml:300-338), so that border
// frames run through the SAME kernel and arithmetic as interior frames.
struct GatherSpan {   // one strip: len samples from signal position pos0 on, rows of out_stride floats
  int64_t pos0, len, out_stride;
  float *out;
};
// blockIdx.z picks the strip (left border / right border of the clips): both in one launch
__global__ void __launch_bounds__(256) gather_padded_kernel(const float *x, int64_t n, int64_t x_stride, GatherSpan s0, GatherSpan s1,
                                                            int pad, float pad_value) {
  const GatherSpan sp = blockIdx.z == 0 ? s0 : s1;
  const int64_t clip = blockIdx.y, pos0 = sp.pos0, len = sp.len;
  const float *src = x + clip * x_stride;
  float *dst = sp.out + clip * sp.out_stride;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < len; j += (int64_t)gridDim.x * 256) {
    int64_t s = pos0 + j;
    float v;
    if (s >= 0 && s < n) {
      v = src[s];
    } else if (pad == SMX_PAD_REFLECT) {
      if (n == 1) {
        s = 0;
      } else {
        const int64_t period = 2 * (n - 1);
        int64_t m = s % period;
        if (m < 0) m += period;
        s = m < n ? m : period - m;
      }
      v = src[s];
    } else if (pad == SMX_PAD_EDGE) {
      v = src[s < 0 ? 0 : n - 1];
    } else {
      v = pad_value;
    }
    dst[j] = v;
  }
}

// What a launch produces: the power spectrogram (mel == nullptr) or the fused mel spectrogram.
struct FastTarget {
  void *out = nullptr;
  int64_t out_stride = 0;       // frames dimension of the output
  int64_t out_offset = 0;       // frame offset of this job's first frame
  const MelFusedArgs *mel = nullptr;
  const Mel32Args *mel32 = nullptr;   // with mel: the 32-lane kernel's plan (nullptr: the 64-lane kernel)
  bool complex_out = false;     // Stft.transform: interleaved (re, im)
  // power kernel only: border frames folded into the interior launch (see stft2048_power_kernel's epilogue)
  int border_left = 0, border_right = 0;
  int64_t border_p0 = 0, border_i1 = 0;
  bool fold_frames = false;     // complex / mel kernels: one launch over every frame of the request (FastArgs::fold_frames)
  const float *strip_l = nullptr, *strip_r = nullptr;
  int64_t strip_l_stride = 0, strip_r_stride = 0;
};

// one launch of the fused kernel over frames that all lie inside [0, n)
void launch_interior(const StftJob &job, const FastTarget &tg, const float *x, int64_t n, int64_t x_stride,
                     int64_t left, int64_t p0, int64_t count, int64_t out_offset, bool strip = false) {
  if (count <= 0) return;
  const smx_stft_config &c = *job.cfg;
  const StftTables &t = c.tables();
  FastArgs a{};
  a.x = x;
  a.n = n;
  a.x_stride = x_stride;
  a.hop = c.hop;
  a.left = left;
  a.pad = job.pad;
  a.pad_value = (float)job.pad_value;
  a.p0 = p0;
  a.count = count;
  a.out = reinterpret_cast<float *>(tg.out);
  a.out_stride = tg.out_stride;
  a.out_offset = out_offset;
  a.hwin = t.fast_window;
  a.w_m = t.fast_w_m;
  a.w_n = t.fast_w_n;
  const int lanes = c.fft_size == kN16 ? 16 : c.fft_size == kN8 ? 8 : 0;   // the power spectrogram at fft 1024 / 512 (launch_stft_fast admits nothing else of these sizes)
  const int64_t ft = lanes == 16 ? PL<16>::FT : lanes == 8 ? PL<8>::FT : kFT, bins = c.fft_size / 2 + 1;
  const int64_t tiles = (count + ft - 1) / ft;
  if (tiles > 0x7fffffff) throw Failure("stft: too many frame tiles for one launch");
  a.tiles_per_clip = (int)tiles;
  // persistent workgroups: one per CU (160 KB of LDS each), every one walks a contiguous range of the
  // flat (clip, tile) sequence, so the table fill / first-load latency / final flush are paid once
  a.total_tiles = job.lead * tiles;
  static int cu_count = 0;
  if (cu_count == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    SMX_HIP_CHECK(hipGetDevice(&dev));
    SMX_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  a.blocks = a.total_tiles < cu_count ? a.total_tiles : cu_count;
  if (!strip && !tg.mel && !tg.complex_out && tg.border_left + tg.border_right > 0 && a.blocks < cu_count) {
    // a small launch: the border tiles of the 32- / 16- / 8-lane kernels' epilogue go to workgroups of their own (they are
    // handed out from the last workgroup down), so that a short clip does not pay two tile latencies in a row
    const int64_t border_tiles = (job.lead * (tg.border_left + tg.border_right) + ft - 1) / ft;
    a.blocks = a.blocks + border_tiles < cu_count ? a.blocks + border_tiles : cu_count;
  }
  {
    // Tile order (TileWalk::init).  Measured on whole batches: while input + output stay within ~2 GB the
    // chip-wide order (2) is 0-3 % ahead; beyond that the per-XCD chunks (1) win by 3-33 % (C5, 71 GB:
    // 409 vs 308 Mframes/s) -- every XCD then stays inside one clip's pages for several tiles.
    const double footprint = (double)job.lead * ((double)n + (double)bins * (double)count) * 4.0;
    a.interleave = (int)diag_int("SMX_INTERLEAVE", footprint <= 2.0e9 ? 2 : 1);
  }
  a.pmode = job.power == 2.0 ? 2 : (job.power == 1.0 ? 1 : 0);
  a.half_power = (float)(0.5 * job.power);
  a.border_left = strip ? 0 : tg.border_left;
  a.border_right = strip ? 0 : tg.border_right;
  a.border_p0 = tg.border_p0;
  a.border_i1 = tg.border_i1;
  a.border_out_offset = tg.out_offset;
  a.fold_frames = (!strip && tg.fold_frames) ? 1 : 0;
  a.border_i0 = tg.border_p0;   // folded complex / mel launches: border_p0 carries the first interior frame
  a.strip_l = tg.strip_l;
  a.strip_r = tg.strip_r;
  a.strip_l_stride = tg.strip_l_stride;
  a.strip_r_stride = tg.strip_r_stride;
  const bool aligned = (c.hop % 2 == 0) && (left % 2 == 0) && (x_stride % 2 == 0) &&
                       (reinterpret_cast<uintptr_t>(x) % 8 == 0);
  const bool square = a.pmode == 2;
  if (lanes && tg.complex_out) {   // Stft.transform at fft 1024 / 512
    auto launch_cplx_lanes = [&](auto ll) {
      constexpr int LL = decltype(ll)::value;
      auto kl = aligned ? stft_complex_lanes_kernel<LL, true> : stft_complex_lanes_kernel<LL, false>;
      SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kl), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PL<LL>::Lds));
      SMX_LAUNCH(kl, dim3((unsigned)a.blocks), dim3(512), PL<LL>::Lds, job.stream, a);
    };
    if (lanes == 16) launch_cplx_lanes(std::integral_constant<int, 16>{});
    else launch_cplx_lanes(std::integral_constant<int, 8>{});
    SMX_HIP_CHECK(hipGetLastError());
    return;
  }
  if (lanes && tg.mel32) {   // fused mel at fft 1024 / 512
    Mel32Args m = *tg.mel32;
    m.out_offset = out_offset;
    auto launch_mel_lanes = [&](auto ll) {
      constexpr int LL = decltype(ll)::value;
      auto by_power = [&](auto al) {
        constexpr bool A = decltype(al)::value;
        return a.pmode == 2 ? stft_mel_lanes_kernel<LL, A, 2> : a.pmode == 1 ? stft_mel_lanes_kernel<LL, A, 1> : stft_mel_lanes_kernel<LL, A, 0>;
      };
      auto kl = aligned ? by_power(std::true_type{}) : by_power(std::false_type{});
      SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kl), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PL<LL>::Lds));
      SMX_LAUNCH(kl, dim3((unsigned)a.blocks), dim3(512), PL<LL>::Lds, job.stream, a, m);
    };
    if (lanes == 16) launch_mel_lanes(std::integral_constant<int, 16>{});
    else launch_mel_lanes(std::integral_constant<int, 8>{});
    SMX_HIP_CHECK(hipGetLastError());
    return;
  }
  if (lanes) {
    auto launch_lanes = [&](auto ll) {
      constexpr int LL = decltype(ll)::value;
      auto pick = [&](auto strip_tag) {
        constexpr bool S = decltype(strip_tag)::value;
        auto by_power = [&](auto al) {
          constexpr bool A = decltype(al)::value;
          return a.pmode == 2 ? stft_power_lanes_kernel<LL, A, 2, S> : a.pmode == 1 ? stft_power_lanes_kernel<LL, A, 1, S> : stft_power_lanes_kernel<LL, A, 0, S>;
        };
        return aligned ? by_power(std::true_type{}) : by_power(std::false_type{});
      };
      auto kl = strip ? pick(std::true_type{}) : pick(std::false_type{});
      SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kl), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PL<LL>::Lds));
      SMX_LAUNCH(kl, dim3((unsigned)a.blocks), dim3(512), PL<LL>::Lds, job.stream, a);
    };
    if (lanes == 16) launch_lanes(std::integral_constant<int, 16>{});
    else launch_lanes(std::integral_constant<int, 8>{});
    SMX_HIP_CHECK(hipGetLastError());
    return;
  }
  if (tg.mel && tg.mel32) {
    Mel32Args m = *tg.mel32;
    m.out_offset = out_offset;
    auto by_power = [&](auto al) {
      constexpr bool A = decltype(al)::value;
      return a.pmode == 2 ? stft2048_mel32_kernel<A, 2> : a.pmode == 1 ? stft2048_mel32_kernel<A, 1> : stft2048_mel32_kernel<A, 0>;
    };
    auto k32 = aligned ? by_power(std::true_type{}) : by_power(std::false_type{});
    SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFast32Lds));
    SMX_LAUNCH(k32, dim3((unsigned)a.blocks), dim3(512), kFast32Lds, job.stream, a, m);
    SMX_HIP_CHECK(hipGetLastError());
    return;
  }
  if (tg.mel) {
    MelFusedArgs m = *tg.mel;
    m.out_offset = out_offset;
    auto pick = [&](auto strip_tag) {
      constexpr bool S = decltype(strip_tag)::value;
      auto by_power = [&](auto al) {
        constexpr bool A = decltype(al)::value;
        return a.pmode == 2 ? stft2048_mel_kernel<A, 2, S> : a.pmode == 1 ? stft2048_mel_kernel<A, 1, S> : stft2048_mel_kernel<A, 0, S>;
      };
      return aligned ? by_power(std::true_type{}) : by_power(std::false_type{});
    };
    auto kernel = strip ? pick(std::true_type{}) : pick(std::false_type{});
    SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFastLds));
    SMX_LAUNCH(kernel, dim3((unsigned)a.blocks), dim3(1024), kFastLds, job.stream, a, m);
    SMX_HIP_CHECK(hipGetLastError());
    return;
  }
  if (tg.complex_out) {   // Stft.transform on the 32-lane frame pipeline
    // the flush in whole aligned 128-byte lines (stft_fast_p32.hpp, cplx_skew32_*): consecutive tiles of a clip on one workgroup
    const bool cskew = reinterpret_cast<uintptr_t>(a.out) % 8 == 0 && env_flag("SMX_COMPLEX_SKEW") != 0;
    if (cskew) a.interleave = 0;
    auto k32 = cskew ? (aligned ? stft2048_complex32_kernel<true, true> : stft2048_complex32_kernel<false, true>)
                     : (aligned ? stft2048_complex32_kernel<true, false> : stft2048_complex32_kernel<false, false>);
    SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFast32Lds));
    SMX_LAUNCH(k32, dim3((unsigned)a.blocks), dim3(512), kFast32Lds, job.stream, a);
    SMX_HIP_CHECK(hipGetLastError());
    return;
  }
#ifdef SMX_DIAG
  if (diag_flag("SMX_NOSTORE") == 1) a.abl_nostore = 1;   // timing-only ablations of the 32-lane kernel (results wrong by construction)
  a.abl_p32 = (int)diag_int("SMX_P32_ABL", 0);
#endif
  (void)square;
  // The power spectrogram at fft 2048: the 32-lane frame pipeline (stft_fast_p32.hpp).
  {
    auto pick32 = [&](auto strip_tag) {
      constexpr bool S = decltype(strip_tag)::value;
      auto by_power = [&](auto al) {
        constexpr bool A = decltype(al)::value;
        return a.pmode == 2 ? stft2048_power32_kernel<A, 2, S> : a.pmode == 1 ? stft2048_power32_kernel<A, 1, S> : stft2048_power32_kernel<A, 0, S>;
      };
      return aligned ? by_power(std::true_type{}) : by_power(std::false_type{});
    };
    // The flush in whole aligned 64-byte blocks (stft_fast_p32.hpp, SKEW): needs even block offsets in every row -- an even row
    // pitch and an origin on an 8-byte boundary -- and consecutive tiles of a clip on one workgroup (contiguous ranges).
    const bool even = a.out_stride % 2 == 0 && ((reinterpret_cast<uintptr_t>(a.out) >> 2) + (uintptr_t)a.out_offset) % 2 == 0;
    const int skew_env = env_flag("SMX_POWER_SKEW");   // unset: a pair of frames per lane where the geometry is even, else a frame per lane; "0": the unskewed flush
    const int skew_form = (int)diag_int("SMX_POWER_SKEW_FORM", 0);   // diagnostic builds: 1 / 2 force a form
    const int skew = (strip || reinterpret_cast<uintptr_t>(a.out) % 4 != 0 || skew_env == 0) ? 0 : (skew_form == 2 || !even) ? 2 : 1;
    if (skew) {
      a.interleave = 0;
      auto by_power = [&](auto al, auto sk) {
        constexpr bool A = decltype(al)::value;
        constexpr int K = decltype(sk)::value;
        return a.pmode == 2 ? stft2048_power32_kernel<A, 2, false, K> : a.pmode == 1 ? stft2048_power32_kernel<A, 1, false, K> : stft2048_power32_kernel<A, 0, false, K>;
      };
      auto by_al = [&](auto sk) { return aligned ? by_power(std::true_type{}, sk) : by_power(std::false_type{}, sk); };
      auto k32s = skew == 1 ? by_al(std::integral_constant<int, 1>{}) : by_al(std::integral_constant<int, 2>{});
      SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k32s), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFast32Lds));
      SMX_LAUNCH(k32s, dim3((unsigned)a.blocks), dim3(512), kFast32Lds, job.stream, a);
      SMX_HIP_CHECK(hipGetLastError());
      return;
    }
    auto k32 = strip ? pick32(std::true_type{}) : pick32(std::false_type{});
    SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFast32Lds));
    SMX_LAUNCH(k32, dim3((unsigned)a.blocks), dim3(512), kFast32Lds, job.stream, a);
    SMX_HIP_CHECK(hipGetLastError());
    return;
  }
}

// frames [pa, pb) that touch a border: gather their padded span, then run the fused kernel on it
void launch_border(const StftJob &job, const FastTarget &tg, int64_t pa, int64_t pb) {
  if (pb <= pa) return;
  const smx_stft_config &c = *job.cfg;
  const int64_t pos0 = pa * c.hop - job.left;               // signal position of the strip's first sample
  const int64_t len = (pb - pa - 1) * c.hop + c.fft_size;
  const int64_t stride = (len + 1) & ~int64_t(1);            // even: keeps 8-byte aligned rows
  float *strip = nullptr;
  SMX_HIP_CHECK(smx::pool_malloc_async((void **)&strip, (size_t)job.lead * (size_t)stride * sizeof(float), job.stream));
  dim3 grid((unsigned)((len + 255) / 256 < 64 ? (len + 255) / 256 : 64), (unsigned)job.lead);
  const GatherSpan span{pos0, len, stride, strip};
  SMX_LAUNCH(gather_padded_kernel, grid, dim3(256), 0, job.stream, reinterpret_cast<const float *>(job.x), job.n, job.x_stride, span, span,
             job.pad, (float)job.pad_value);
  SMX_HIP_CHECK(hipGetLastError());
  launch_interior(job, tg, strip, len, stride, 0, 0, pb - pa, tg.out_offset + (pa - job.p0), /*strip=*/true);
  SMX_HIP_CHECK(hipFreeAsync(strip, job.stream));
}

bool fast_eligible(const StftJob &job, bool power_face = false) {
  const smx_stft_config &c = *job.cfg;
  if (fast_path_disabled()) return false;
  const bool size_ok = c.fft_size == kN || (power_face && (c.fft_size == kN16 || c.fft_size == kN8) && diag_flag("SMX_POWER16_OFF") != 1 &&
                                              true);
  if (!size_ok || job.in_bytes != 4 || job.interior != SMX_INTERIOR_F32) return false;
  if (diag_flag("SMX_GENERIC_2048") == 1) return false;   // diagnostic: time the stage-free generic kernels at fft 2048
  if (job.lead > 65535) return false;
  return true;
}

// splits frames [p0, p0 + count) into left border / interior / right border launches
void launch_ranges(const StftJob &job, const FastTarget &tg) {
  const smx_stft_config &c = *job.cfg;
  // frame p lies inside the signal iff 0 <= p*hop - left and p*hop - left + N <= n
  const int64_t p0 = job.p0, p1 = job.p0 + job.count;
  int64_t i0 = job.left > 0 ? (job.left + c.hop - 1) / c.hop : 0;
  int64_t i1 = job.n + job.left - c.fft_size >= 0 ? (job.n + job.left - c.fft_size) / c.hop + 1 : 0;
  if (i0 < p0) i0 = p0;
  if (i1 > p1) i1 = p1;
  if (i1 <= i0) {          // no interior frame in range: one strip for everything
    launch_border(job, tg, p0, p1);
    return;
  }
  // power spectrogram: the border frames ride in the interior launch (no gathers, no extra launches) whenever
  // 32-bit sample positions suffice for the padding rule
  static const bool fold_off = diag_flag("SMX_NO_BORDER_FOLD") == 1;
  // ... while the border frames of the whole batch are few (C2: 1024 of 240 128).  Batches of many short clips (16 384 one-second
  // clips: 65 536 border frames of 524 288) take the gathered strips below instead: the epilogue's barrier-separated tiles and
  // element-wise stores cost them more than the whole interior (2.76 ms against 1.1).
  const int64_t border_total = job.lead * ((i0 - p0) + (p1 - i1));
  static const int64_t epilogue_max = (int64_t)diag_int("SMX_BORDER_EPILOGUE_MAX", 20000);
  if (!tg.mel && !tg.complex_out && !fold_off && job.n < (int64_t(1) << 30) && (i0 - p0) + (p1 - i1) > 0 &&
      (i0 - p0) + (p1 - i1) < 4096 && border_total <= epilogue_max) {
    FastTarget folded = tg;
    folded.border_left = (int)(i0 - p0);
    folded.border_right = (int)(p1 - i1);
    folded.border_p0 = p0;
    folded.border_i1 = i1;
    launch_interior(job, folded, reinterpret_cast<const float *>(job.x), job.n, job.x_stride, job.left, i0, i1 - i0,
                    tg.out_offset + (i0 - p0));
    return;
  }
  // complex spectrogram / fused mel: ONE launch of the fused kernel over the whole request; the border frames' padded
  // spans are gathered first (two small launches) and the kernel takes those frames from the strips (C3: 0.66 -> 0.60 ms)
  if (!fold_off && (i0 - p0) + (p1 - i1) > 0) {   // (the power kernel too when the epilogue above was not taken)
    FastTarget folded = tg;
    folded.fold_frames = true;
    folded.border_p0 = i0;
    folded.border_i1 = i1;
    float *strips[2] = {nullptr, nullptr};
    GatherSpan spans[2] = {};
    auto plan = [&](int which, int64_t pa, int64_t pb, const float *&dst, int64_t &dst_stride) {
      if (pb <= pa) return;
      const int64_t pos0 = pa * c.hop - job.left, len = (pb - pa - 1) * c.hop + c.fft_size;
      const int64_t stride = (len + 1) & ~int64_t(1);            // even: keeps 8-byte aligned rows
      SMX_HIP_CHECK(smx::pool_malloc_async((void **)&strips[which], (size_t)job.lead * (size_t)stride * sizeof(float), job.stream));
      spans[which] = GatherSpan{pos0, len, stride, strips[which]};
      dst = strips[which];
      dst_stride = stride;
    };
    plan(0, p0, i0, folded.strip_l, folded.strip_l_stride);
    plan(1, i1, p1, folded.strip_r, folded.strip_r_stride);
    {   // both strips in ONE launch (a span of length 0 has no work)
      const int64_t longest = spans[0].len > spans[1].len ? spans[0].len : spans[1].len;
      dim3 grid((unsigned)((longest + 255) / 256 < 64 ? (longest + 255) / 256 : 64), (unsigned)job.lead, 2);
      SMX_LAUNCH(gather_padded_kernel, grid, dim3(256), 0, job.stream, reinterpret_cast<const float *>(job.x), job.n, job.x_stride,
                 spans[0], spans[1], job.pad, (float)job.pad_value);
      SMX_HIP_CHECK(hipGetLastError());
    }
    launch_interior(job, folded, reinterpret_cast<const float *>(job.x), job.n, job.x_stride, job.left, p0, p1 - p0, tg.out_offset);
    for (float *sp : strips)
      if (sp) SMX_HIP_CHECK(hipFreeAsync(sp, job.stream));
    return;
  }
  launch_border(job, tg, p0, i0);
  launch_interior(job, tg, reinterpret_cast<const float *>(job.x), job.n, job.x_stride, job.left, i0, i1 - i0,
                  tg.out_offset + (i0 - p0));
  launch_border(job, tg, i1, p1);
}

}  // namespace

#ifdef SMX_STAMPS
extern "C" int smx_debug_read_stamps(unsigned long long *out, int count) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp_sums), sizeof(unsigned long long) * (size_t)count);
}
#endif

bool launch_stft_fast(const StftJob &job) {
  if (!fast_eligible(job, /*power_face=*/true)) return false;
  if (job.count <= 0 || job.lead <= 0) return true;
  const int64_t elem = job.mode == OUT_COMPLEX ? 8 : 4;
  if ((job.cfg->fft_size / 2 + 1) * job.out_stride * elem >= (int64_t(1) << 32)) return false;   // 32-bit row offsets
  FastTarget tg;
  tg.complex_out = job.mode == OUT_COMPLEX;
  tg.out = job.out;
  tg.out_stride = job.out_stride;
  tg.out_offset = job.out_offset;
  launch_ranges(job, tg);
  return true;
}

bool launch_mel_spectrogram_fused(const MelSpecJob &job) {
  {   // fft 1024 / 512: the 16- / 8-lane frame pipeline with the filterbank product over its tiles (stft_mel_lanes_kernel)
    const StftJob &sj = job.stft;
    if ((sj.cfg->fft_size == kN16 || sj.cfg->fft_size == kN8) && sj.in_bytes == 4 && sj.interior == SMX_INTERIOR_F32 && sj.mode != OUT_COMPLEX && !fast_path_disabled() &&
        sj.lead <= 65535 && env_flag("SMX_MEL_V1") != 1 && (int64_t)job.mel->n_mels * sj.count * 4 < (int64_t(1) << 32)) {
      const MelFusedPlan &pl = job.mel->fused32_plan();
      if (pl.state == 1) {
        if (sj.count <= 0 || sj.lead <= 0) return true;
        MelFusedArgs m{};
        m.out = reinterpret_cast<float *>(job.out);
        m.out_stride = sj.count;
        m.n_mels = (int)job.mel->n_mels;
        Mel32Args m32{};
        m32.items = reinterpret_cast<const Mel32Item *>(pl.items);
        m32.w = pl.w_mfma;
        m32.out = m.out;
        m32.out_stride = m.out_stride;
        m32.n_mels = m.n_mels;
        FastTarget tg;
        tg.out = job.out;
        tg.out_stride = sj.count;
        tg.out_offset = 0;
        tg.mel = &m;
        tg.mel32 = &m32;
        launch_ranges(sj, tg);
        return true;
      }
    }
  }
  if (launch_mel_spectrogram_16(job)) return true;
  if (!fast_eligible(job.stft)) return false;
  if (job.stft.count <= 0 || job.stft.lead <= 0) return true;
  const MelFusedPlan &plan = job.mel->fused_plan();
  if (plan.state != 1) return false;
  MelFusedArgs m{};
  m.items = reinterpret_cast<const MelItem *>(plan.items);
  m.w_mfma = plan.w_mfma;
  // partial-sum records of the global helper slots: one set per persistent workgroup (at most one per CU),
  // stream-ordered so that concurrent calls on other streams never share them
  {
    static const int cus = [] {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
      return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }();
    SMX_HIP_CHECK(smx::pool_malloc_async((void **)&m.scratch, (size_t)cus * 2 * (kMelHelpers - kMelPadSlots) * 256 * sizeof(float),
                                 job.stft.stream));
  }
  m.out = reinterpret_cast<float *>(job.out);
  m.out_stride = job.stft.count;
  m.out_offset = 0;
  m.n_mels = (int)job.mel->n_mels;
  FastTarget tg;
  tg.out = job.out;
  tg.out_stride = job.stft.count;
  tg.out_offset = 0;
  tg.mel = &m;
  Mel32Args m32{};
  if (env_flag("SMX_MEL_V1") != 1) {   // SMX_MEL_V1=1: the 64-lane kernel (A/B timing, tests: two implementations of one contract)
    const MelFusedPlan &p32 = job.mel->fused32_plan();
    if (p32.state == 1) {
      m32.items = reinterpret_cast<const Mel32Item *>(p32.items);
      m32.w = p32.w_mfma;
      m32.out = m.out;
      m32.out_stride = m.out_stride;
      m32.out_offset = 0;
      m32.n_mels = m.n_mels;
      tg.mel32 = &m32;
    }
  }
  launch_ranges(job.stft, tg);
  SMX_HIP_CHECK(hipFreeAsync(m.scratch, job.stft.stream));
  return true;
}

}  // namespace smx

// ---- MFMA work plan of the fused mel kernel (per mel configuration and device) ---------------------
const smx::MelFusedPlan &smx_mel_config::fused_plan() const {
  using namespace smx;
  int device = 0;
  SMX_HIP_CHECK(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(mutex_);
  MelFusedPlan &plan = fused_[device];
  if (plan.state != 0) return plan;
  plan.state = -1;
  const int64_t nb = bins();
  if (fft_size != kN || n_mels < 1 || n_mels > 256) return plan;
  const int blocks = (int)((n_mels + 15) / 16);
  if (blocks > 16) return plan;
  std::vector<int> k4lo((size_t)blocks, 0), k4n((size_t)blocks, 0), pieces((size_t)blocks, 1);
  for (int b = 0; b < blocks; ++b) {
    int lo = (int)nb, hi = 0;
    for (int64_t mm = 16 * b; mm < 16 * (b + 1) && mm < n_mels; ++mm)
      for (int64_t k = 0; k < nb; ++k)
        if (weights[(size_t)(mm * nb + k)] != 0.0) {
          if (k < lo) lo = (int)k;
          if (k + 1 > hi) hi = (int)k + 1;
        }
    if (hi > lo) {
      k4lo[(size_t)b] = lo / 4;
      k4n[(size_t)b] = (hi + 3) / 4 - lo / 4;
    }
  }
  // split the heaviest blocks in K among the waves that own no block: at most kMelHelpers helper pieces
  int total_pieces = blocks, helpers = 0;
  while (helpers < kMelHelpers && total_pieces < 16) {
    int best = -1;
    double load = 8.0;    // not worth splitting below ~8 MFMAs per piece
    for (int b = 0; b < blocks; ++b) {
      const double l = (double)k4n[(size_t)b] / pieces[(size_t)b];
      if (l > load) { load = l; best = b; }
    }
    if (best < 0) break;
    ++pieces[(size_t)best];
    ++helpers;
    ++total_pieces;
  }
  std::vector<MelItem> items(16);
  for (auto &it : items) { it = MelItem{}; it.slot = -1; }
  std::vector<float> wm;
  int next_wave = blocks, next_slot = 0;
  auto emit_operands = [&](MelItem &it) {
    it.a_offset = (int)(wm.size() / 64);
    for (int i = 0; i < it.k4_count; ++i)
      for (int lane = 0; lane < 64; ++lane) {
        const int64_t mm = 16 * (int64_t)it.block + (lane & 15);
        const int64_t k = 4 * (int64_t)(it.k4_begin + i) + (lane >> 4);
        wm.push_back((mm < n_mels && k < nb) ? (float)weights[(size_t)(mm * nb + k)] : 0.0f);
      }
  };
  for (int b = 0; b < blocks; ++b) {
    const int np = pieces[(size_t)b], n = k4n[(size_t)b];
    int begin = k4lo[(size_t)b];
    for (int pc = 0; pc < np; ++pc) {
      const int cnt = n / np + (pc < n % np ? 1 : 0);
      MelItem &it = pc == 0 ? items[(size_t)b] : items[(size_t)next_wave];
      it.block = b;
      it.k4_begin = begin;
      it.k4_count = cnt;
      if (pc == 0) {
        it.owner = 1;
      } else {
        it.slot = next_slot;
        MelItem &own = items[(size_t)b];
        own.slots[own.nslots++] = next_slot;
        ++next_slot;
        ++next_wave;
      }
      emit_operands(it);
      begin += cnt;
    }
  }
  for (const auto &it : items)
    if (it.k4_count > kMelMaxSteps) return plan;   // the A operands of an item must fit its wave's registers
#ifdef SMX_DIAG   // result-altering timing switches exist in diagnostic builds only (make DIAG=1)
  if (diag_flag("SMX_MEL_ONESTEP") == 1)  // one MFMA step per item (wrong results): the cost of the protocol alone
    for (auto &it : items) it.k4_count = it.k4_count > 0 ? 1 : 0;
  if (diag_flag("SMX_MEL_NOMFMA") == 1)   // plan without MFMA work (results are zeros)
    for (auto &it : items) it.k4_count = 0;
#endif
  wm.resize(wm.size() + 64 * (size_t)kMelMaxSteps, 0.0f);   // every item can be read kMelMaxSteps rows deep
  SMX_HIP_CHECK(hipMalloc(&plan.items, items.size() * sizeof(MelItem)));
  SMX_HIP_CHECK(hipMemcpy(plan.items, items.data(), items.size() * sizeof(MelItem), hipMemcpyHostToDevice));
  SMX_HIP_CHECK(hipMalloc((void **)&plan.w_mfma, wm.size() * sizeof(float)));
  SMX_HIP_CHECK(hipMemcpy(plan.w_mfma, wm.data(), wm.size() * sizeof(float), hipMemcpyHostToDevice));
  plan.state = 1;
  return plan;
}

// ---- work plan of the 32-lane fused mel kernel (stft_fast_mel32.hpp): items of up to 16 mel rows, each summed over its own
// band by one wave; the longest items are split by rows until no wave holds much more than an eighth of the MFMA steps
const smx::MelFusedPlan &smx_mel_config::fused32_plan() const {
  using namespace smx;
  int device = 0;
  SMX_HIP_CHECK(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(mutex_);
  MelFusedPlan &plan = fused32_[device];
  if (plan.state != 0) return plan;
  plan.state = -1;
  const int64_t nb = bins();
  if ((fft_size != kN && fft_size != kN16 && fft_size != kN8) || n_mels < 1 || n_mels > 256) return plan;
  std::vector<int> lo((size_t)n_mels, (int)nb), hi((size_t)n_mels, 0);
  for (int64_t mm = 0; mm < n_mels; ++mm)
    for (int64_t k = 0; k < nb; ++k)
      if ((float)weights[(size_t)(mm * nb + k)] != 0.0f) {
        if (k < lo[(size_t)mm]) lo[(size_t)mm] = (int)k;
        hi[(size_t)mm] = (int)k + 1;
      }
  struct Piece { int row0, nrows, k4b, k4n; };   // k4n: steps, a multiple of 4
  auto make = [&](int row0, int nrows) {
    int l = (int)nb, h = 0;
    for (int r = row0; r < row0 + nrows; ++r)
      if (hi[(size_t)r] > lo[(size_t)r]) {
        l = std::min(l, lo[(size_t)r]);
        h = std::max(h, hi[(size_t)r]);
      }
    Piece p{row0, nrows, 0, 4};                    // rows without a weight: four steps over zeros write their zeros
    if (h > l) {
      p.k4b = l / 4;
      p.k4n = (((h + 3) / 4 - l / 4) + 3) / 4 * 4;
    }
    return p;
  };
  std::vector<Piece> pieces;
  for (int row0 = 0; row0 < n_mels; row0 += 16) pieces.push_back(make(row0, (int)std::min<int64_t>(16, n_mels - row0)));
  auto total = [&] { int t = 0; for (const auto &p : pieces) t += p.k4n; return t; };
  for (;;) {   // split the longest piece by rows while that shortens the longest wave
    size_t big = 0;
    for (size_t i = 1; i < pieces.size(); ++i)
      if (pieces[i].k4n > pieces[big].k4n) big = i;
    const int target = (total() + 7) / 8;
    if (pieces[big].k4n <= target + target / 4 || pieces[big].nrows < 2 || pieces.size() >= 8 * kMel32MaxItems) break;
    const Piece p = pieces[big];
    const Piece a1 = make(p.row0, p.nrows / 2), a2 = make(p.row0 + p.nrows / 2, p.nrows - p.nrows / 2);
    if (std::max(a1.k4n, a2.k4n) >= p.k4n) break;   // rows with one common band: nothing to gain
    pieces[big] = a1;
    pieces.push_back(a2);
  }
  if (total() > 1024) return plan;   // a dense filterbank: the 64-lane kernel's K-split plan (or the composition) serves it
  std::sort(pieces.begin(), pieces.end(), [](const Piece &x, const Piece &y) { return x.k4n > y.k4n; });   // longest first onto the least loaded wave
  std::vector<Mel32Item> items(8 * kMel32MaxItems, Mel32Item{});
  int load[8] = {0}, count[8] = {0};
  std::vector<float> wm;
  for (const auto &p : pieces) {
    int w = -1;
    for (int i = 0; i < 8; ++i)
      if (count[i] < kMel32MaxItems && (w < 0 || load[i] < load[w])) w = i;
    if (w < 0) return plan;
    Mel32Item &it = items[(size_t)(w * kMel32MaxItems + count[w]++)];
    it.row0 = p.row0;
    it.nrows = p.nrows;
    it.k4_begin = p.k4b;
    it.k4_count = p.k4n;
    it.a_offset = (int)(wm.size() / 64);
    it.last_bin = (int)nb - 1;
    load[w] += p.k4n;
    for (int i = 0; i < p.k4n; ++i)
      for (int lane = 0; lane < 64; ++lane) {
        const int64_t mm = p.row0 + (lane & 15), k = 4 * (int64_t)(p.k4b + i) + (lane >> 4);
        wm.push_back(((lane & 15) < p.nrows && k < nb) ? (float)weights[(size_t)(mm * nb + k)] : 0.0f);
      }
  }
  // Measured against the 64-lane kernel (profiles/r05/ab_mel32.log): ahead by 6 % at 40 steps on the longest wave (128 mels at
  // 48 kHz), by 4.5 % at 48 (80 mels at 16 kHz), behind by 5 % at 72 (40 mels at 22.05 kHz: a mel's own band cannot be split by rows)
  const int max_load = (int)diag_int("SMX_MEL32_MAXLOAD", 52);
  for (int i = 0; i < 8; ++i)
    if (load[i] > max_load) return plan;
  SMX_HIP_CHECK(hipMalloc(&plan.items, items.size() * sizeof(Mel32Item)));
  SMX_HIP_CHECK(hipMemcpy(plan.items, items.data(), items.size() * sizeof(Mel32Item), hipMemcpyHostToDevice));
  SMX_HIP_CHECK(hipMalloc((void **)&plan.w_mfma, wm.size() * sizeof(float)));
  SMX_HIP_CHECK(hipMemcpy(plan.w_mfma, wm.data(), wm.size() * sizeof(float), hipMemcpyHostToDevice));
  plan.state = 1;
  return plan;
}
