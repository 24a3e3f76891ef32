// Fused framing + window + real FFT + output stage for gfx950 (MI355X): the launchers, the tile walk and the border handling of
//   stft2048_power32_kernel    |X|^p               Stft.power_spectrum          stft.ml:670-691   (stft_fast_p32.hpp: the 32-lane frame pipeline)
//   stft2048_complex32_kernel  X                   Stft.transform / _range      stft.ml:632-666   (the same)
//   stft2048_mel32_kernel      W |X|^p (MFMA)      Soundml.mel_spectrogram      soundml.ml:12-24  (stft_fast_mel32.hpp: the filterbank product over the pipeline's tiles)
//   stft_power_lanes_kernel / stft_complex_lanes_kernel / stft_mel_lanes_kernel<16 | 8>    the same at fft 1024 / 512 (stft_fast_p16.hpp)
//   stft2048_complex_fm_kernel                                                             the complex spectrum frame-major, bins stored straight from the registers (Griffin-Lim's own layout)
//   stft_power_lanes_kernel / stft_complex_lanes_kernel<4>                                  ... and at fft 256 (a frame in 4 lanes, 128-frame tiles)
//   stft4096_power64_kernel    |X|^p at fft 4096                                            a frame in a WHOLE wave, 8-frame single-buffered tiles (stft_fast_p64.hpp; round 6)
// They replace, for float32 audio, the reference's hot call
//   Nx.stft cdtype ~window:fft ~step:hop ~win (to_double samples)   stft.ml:356-364
// Each audio sample is read from HBM once (hop-strided overlapping frames are re-read through L1 / L2) and the [bins; frames]
// result is written once, in whole aligned 64-byte blocks (power) or 128-byte lines (complex) along the frame axis.
//
// The frame pipeline itself (a frame in 32 / 16 / 8 lanes with 32 points per lane, radix-32 stages in registers, one transposition
// through the frame's own column of the output tile in LDS, the real-FFT post-pass on pairs of bins) is described at the head of
// stft_fast_p32.hpp; a workgroup is 8 waves = one tile of 16 (32, 64) consecutive frames of one clip, two tile buffers, one
// persistent workgroup per CU walking a contiguous range of the flat (clip, tile) sequence (TileWalk below), waves meeting at
// monotonic LDS counters instead of barriers.  (The 64-lane kernels of rounds 1-2 -- in-wave transposes by permlane / DPP -- are gone.)
//
// Frames that reach past either end of the signal: their tile takes its samples through the padding rule (FastArgs::fold_frames
// == 2, load_frame32_padded / load_frameL_padded: round 5); for clips shorter than a frame, from gathered, already padded strips
// (fold_frames == 1) or the power kernels' border epilogue.
//
// Algorithmic HBM bytes per frame at fft 2048 / hop 512: power 2048 + 4100 = 6148 B, complex 2048 + 8200 = 10248 B,
// mel 2048 + 4 n_mels.
#include <cstdlib>
#include <type_traits>

#include "fft_device.hpp"
#include "smx_internal.hpp"

namespace smx {
namespace {

using namespace fftdev;

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
  int64_t range_base, range_extra;   // total_tiles / blocks and the remainder (block_to_range divides nothing)
  // power kernel: the frames that touch a border of the signal (reflect / edge / constant padding) are
  // computed by the same launch, after the interior tiles (0 = none: they come as gathered strips)
  int border_left, border_right;       // border frames per clip before / after the interior range
  int64_t border_p0, border_i1;        // first frame of the request, first frame after the interior range
  int64_t border_out_offset;           // output frame offset of frame border_p0
  // complex / mel kernels: the launch covers ALL frames of the request; a frame that touches a border of the signal reads its
  // samples from a gathered, already padded strip (left: frames [p0, border_i0), right: frames [border_i1, ..)) -- only the
  // frame's base pointer differs (a scalar select), so the border frames cost no launches and no registers
  int fold_frames;                     // 1: strips; 2 (stft2048_power32_kernel): no strips, a tile with such a frame loads through load_frame32_padded
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
constexpr int kFT = 16;                         // frames per tile: one per wave, 16 waves per workgroup
constexpr int kTileStride = kFT + 1;            // floats per tile row (pad column 16)
constexpr size_t kWinBytes = (size_t)kM * sizeof(float2); // 0.5 * window as (even, odd) pairs

// Persistent workgroups: block b owns a contiguous range of the flat (clip, tile) sequence.
// XCD-aware: blocks that share an XCD (b % 8) get neighbouring ranges, i.e. whole runs of clips,
// so halo re-reads and the partial output lines of neighbouring tiles meet in one L2.
// (No 64-bit division here or in TileWalk::init when the launch has fewer than 2^31 tiles: as scalar code one costs a few
// hundred dependent instructions, and the ten of them were 5 us of every workgroup's start -- a C2 launch is 59 tiles of 7 us:
// profiles/r07/timeline_inline_border.log.  The launcher supplies total_tiles / blocks and its remainder.)
__device__ __forceinline__ void block_to_range(const FastArgs &a, int64_t &tau_begin, int64_t &tau_end) {
  const unsigned nb = (unsigned)a.blocks, q = nb / 8u, r = nb % 8u, xcd = blockIdx.x % 8u, idx = blockIdx.x / 8u;
  const unsigned vb = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;
  const int64_t base = a.range_base, extra = a.range_extra;
  tau_begin = (int64_t)vb * base + ((int64_t)vb < extra ? (int64_t)vb : extra);
  tau_end = tau_begin + base + ((int64_t)vb < extra ? 1 : 0);
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

  // I: the integer type of the tile arithmetic -- unsigned when the launch has fewer than 2^31 tiles, else int64_t
  template <class I>
  __device__ __forceinline__ void init_as(const FastArgs &a, float *out, int64_t out_clip_floats) {
    I tau0;
    int step;
    uid = (int)blockIdx.x;
    const I total = (I)a.total_tiles, tpc = (I)a.tiles_per_clip;
    if (a.interleave) {
      // The workgroups of one XCD walk a contiguous chunk of the sequence side by side: at any time the
      // XCD is writing ~32 neighbouring tiles of the same clip, i.e. for every bin one contiguous run of
      // ~2 KB, which its L2 can assemble into whole lines before they go to HBM.
      const I nb = (I)a.blocks, xcd = blockIdx.x % 8u, idx = blockIdx.x / 8u;
      if (a.interleave == 2) {   // all workgroups side by side; an XCD holds 32 neighbouring tiles of every 256
        const I q = nb / 8, r = nb % 8;
        tau0 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        uid = (int)tau0;
        step = (int)nb;
        ntiles = tau0 < total ? (int)((total - tau0 + nb - 1) / nb) : 0;
      } else {
        const I nx = (nb - xcd + 7) / 8;                            // workgroups on this XCD
        const I nxcd = nb < 8 ? nb : 8;                             // XCDs that received a workgroup
        // total * xcd / nxcd without the wide product: floor((Q n + R) x / n) = Q x + floor(R x / n)
        const I tq = total / nxcd, tr = total % nxcd;
        const I x0 = tq * xcd + tr * xcd / nxcd, x1 = tq * (xcd + 1) + tr * (xcd + 1) / nxcd;
        tau0 = x0 + idx;
        step = (int)nx;
        ntiles = tau0 < x1 ? (int)((x1 - tau0 + nx - 1) / nx) : 0;
      }
    } else {
      int64_t tb, te;
      block_to_range(a, tb, te);
      tau0 = (I)tb;
      step = 1;
      ntiles = (int)(te - tb);
    }
    if (ntiles <= 0) return;
    const I clip0 = tau0 / tpc;
    ft = (int)(tau0 - clip0 * tpc);
    x_step = a.x_stride;
    o_step = out_clip_floats;
    xclip = a.x + (int64_t)clip0 * x_step;
    oclip = out + (int64_t)clip0 * o_step;
    step_clips = step == 1 ? 0 : (int)((unsigned)step / (unsigned)a.tiles_per_clip);
    step_tiles = step == 1 ? 1 : (int)((unsigned)step % (unsigned)a.tiles_per_clip);
    if (step == 1 && a.tiles_per_clip == 1) { step_clips = 1; step_tiles = 0; }
  }
  __device__ __forceinline__ void init(const FastArgs &a, float *out, int64_t out_clip_floats) {
    if (a.total_tiles < (int64_t(1) << 31)) init_as<unsigned>(a, out, out_clip_floats);   // (wave-uniform)
    else init_as<int64_t>(a, out, out_clip_floats);
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

#ifndef SMX_WAIT_SLEEP
#define SMX_WAIT_SLEEP 2
#endif
__device__ __forceinline__ void lds_wait(unsigned *c, unsigned target) {
  while ((unsigned)__builtin_amdgcn_readfirstlane(
             (int)__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
    __builtin_amdgcn_s_sleep(SMX_WAIT_SLEEP);
}

#include "stft_fast_p32.hpp"   // the 32-lane frame pipeline: stft2048_power32_kernel
#include "stft_fast_mel32.hpp" // the fused audio -> mel kernel on the 32-lane pipeline: stft2048_mel32_kernel
#include "stft_fast_p16.hpp"   // the same pipeline with a frame in 16 / 8 / 4 lanes: stft_power_lanes_kernel (power spectrogram at fft 1024 / 512 / 256)
#include "stft_fast_p64.hpp"   // ... and with a frame in a whole wave: stft4096_power64_kernel (power spectrogram at fft 4096; round 6)
#ifdef SMX_ISA_ONE   // tools/isa_one.py: ONE instantiation of a register-pipeline kernel (registers / scratch / instruction mix in seconds, no GPU)
#ifndef SMX_ISA_KERNEL
#define SMX_ISA_KERNEL 0
#endif
#if SMX_ISA_KERNEL == 1
template __global__ void stft2048_complex32_kernel<true, (SMX_ISA_ONE != 0)>(FastArgs);
#elif SMX_ISA_KERNEL == 2
template __global__ void stft2048_mel32_kernel<true, 2, SMX_ISA_ONE>(FastArgs, Mel32Args);
#elif SMX_ISA_KERNEL == 3
template __global__ void stft_power_lanes_kernel<SMX_ISA_ONE, true, 2, false>(FastArgs);
#elif SMX_ISA_KERNEL == 7
template __global__ void stft_power_lanes_kernel<SMX_ISA_ONE, true, 2, false, true>(FastArgs);
#elif SMX_ISA_KERNEL == 4
template __global__ void stft_mel_lanes_kernel<SMX_ISA_ONE, true, 2, true>(FastArgs, Mel32Args);
#elif SMX_ISA_KERNEL == 6
template __global__ void stft2048_complex_fm_kernel<(SMX_ISA_ONE != 0)>(FastArgs);
#elif SMX_ISA_KERNEL == 5
template __global__ void stft_complex_lanes_kernel<SMX_ISA_ONE, true>(FastArgs);
#elif SMX_ISA_KERNEL == 8
template __global__ void stft4096_power64_kernel<true, 2, (SMX_ISA_ONE != 0)>(FastArgs);
#else
template __global__ void stft2048_power32_kernel<true, 2, false, SMX_ISA_ONE>(FastArgs);
#endif
}  // namespace
}  // namespace smx
#else
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
  bool mel = false;              // a fused mel launch (the plan and the output: mel32)
  const Mel32Args *mel32 = nullptr;   // with mel: the 32-lane kernel's plan (nullptr: the 64-lane kernel)
  bool complex_out = false;     // Stft.transform: interleaved (re, im)
  bool frame_major = false;     // ... as out[clip][frame][bin] in rows of out_stride floats, tiles_per_clip x 16 rows a clip (Griffin-Lim's own spectra)
  // power kernel only: border frames folded into the interior launch (see stft2048_power_kernel's epilogue)
  int border_left = 0, border_right = 0;
  int64_t border_p0 = 0, border_i1 = 0;
  bool fold_frames = false;     // complex / mel kernels: one launch over every frame of the request (FastArgs::fold_frames)
  bool inline_border = false;   // power kernel at fft 2048 (round 5): one launch over every frame, the border tiles load through the padding rule (FastArgs::fold_frames == 2)
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
  const int lanes = c.fft_size == kN16 ? 16 : c.fft_size == kN8 ? 8 : c.fft_size == kN4 ? 4 : 0;   // fft 1024 / 512 / 256 (launch_stft_fast admits nothing else of these sizes)
  const bool p64 = c.fft_size == k64N;   // fft 4096: a frame in a whole wave, 8-frame tiles (stft_fast_p64.hpp; power output only)
  const int64_t ft = p64 ? k64FT : lanes == 16 ? PL<16>::FT : lanes == 8 ? PL<8>::FT : lanes == 4 ? PL<4>::FT : kFT, bins = c.fft_size / 2 + 1;
  const int64_t tiles = (count + ft - 1) / ft;
  if (tiles > 0x7fffffff) throw Failure("stft: too many frame tiles for one launch");
  a.tiles_per_clip = (int)tiles;
  // persistent workgroups: one per CU (160 KB of LDS each), every one walks a contiguous range of the
  // flat (clip, tile) sequence, so the table fill / first-load latency / final flush are paid once
  a.total_tiles = job.lead * tiles;
  const int cu_count = device_cu_count();   // (per device, thread-safe: tables.cpp)
  a.blocks = a.total_tiles < cu_count ? a.total_tiles : cu_count;
  auto set_ranges = [&] { a.range_base = a.total_tiles / a.blocks; a.range_extra = a.total_tiles % a.blocks; };
  set_ranges();
  if (!strip && !tg.mel && !tg.complex_out && tg.border_left + tg.border_right > 0 && a.blocks < cu_count) {
    // a small launch: the border tiles of the 32- / 16- / 8-lane kernels' epilogue go to workgroups of their own (they are
    // handed out from the last workgroup down), so that a short clip does not pay two tile latencies in a row
    const int64_t border_tiles = (job.lead * (tg.border_left + tg.border_right) + ft - 1) / ft;
    a.blocks = a.blocks + border_tiles < cu_count ? a.blocks + border_tiles : cu_count;
    set_ranges();
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
  a.fold_frames = strip ? 0 : tg.inline_border ? 2 : tg.fold_frames ? 1 : 0;
  a.border_i0 = tg.border_p0;   // folded complex / mel launches: border_p0 carries the first interior frame
  a.strip_l = tg.strip_l;
  a.strip_r = tg.strip_r;
  a.strip_l_stride = tg.strip_l_stride;
  a.strip_r_stride = tg.strip_r_stride;
  const bool aligned = (c.hop % 2 == 0) && (left % 2 == 0) && (x_stride % 2 == 0) &&
                       (reinterpret_cast<uintptr_t>(x) % 8 == 0);
  const bool square = a.pmode == 2;
  if (p64) {
    if (tg.complex_out || tg.mel) throw Failure("stft: the fft-4096 pipeline has the power face only");
    // a.interleave as chosen above (2 chip-wide / 1 per XCD): this kernel's 32-byte row runs need the neighbouring tiles written at the
    // same time (stft_fast_p64.hpp); SMX_P64_CONTIGUOUS=1 in diagnostic builds: contiguous ranges (A/B timing: 0.74 against 0.54 ms)
    if (diag_flag("SMX_P64_CONTIGUOUS") == 1) a.interleave = 0;
    a.pmode = job.power == 2.0 ? 2 : (job.power == 1.0 ? 1 : 0);
    const bool even = a.out_stride % 2 == 0 && ((reinterpret_cast<uintptr_t>(a.out) >> 2) + (uintptr_t)a.out_offset) % 2 == 0 &&
                      reinterpret_cast<uintptr_t>(a.out) % 8 == 0;
    auto by_power = [&](auto al, auto ev) {
      constexpr bool A = decltype(al)::value, E = decltype(ev)::value;
      return a.pmode == 2 ? stft4096_power64_kernel<A, 2, E> : a.pmode == 1 ? stft4096_power64_kernel<A, 1, E> : stft4096_power64_kernel<A, 0, E>;
    };
    auto by_al = [&](auto ev) { return aligned ? by_power(std::true_type{}, ev) : by_power(std::false_type{}, ev); };
    auto k64 = even ? by_al(std::true_type{}) : by_al(std::false_type{});
    SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)k64Lds));
    SMX_LAUNCH(k64, dim3((unsigned)a.blocks), dim3(512), k64Lds, job.stream, a);
    SMX_HIP_CHECK(hipGetLastError());
    return;
  }
  if (lanes && tg.complex_out) {   // Stft.transform at fft 1024 / 512 / 256
    auto launch_cplx_lanes = [&](auto ll) {
      constexpr int LL = decltype(ll)::value;
      auto kl = aligned ? stft_complex_lanes_kernel<LL, true> : stft_complex_lanes_kernel<LL, false>;
      SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kl), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PL<LL>::Lds));
      SMX_LAUNCH(kl, dim3((unsigned)a.blocks), dim3(512), PL<LL>::Lds, job.stream, a);
    };
    if (lanes == 16) launch_cplx_lanes(std::integral_constant<int, 16>{});
    else if (lanes == 8) launch_cplx_lanes(std::integral_constant<int, 8>{});
    else launch_cplx_lanes(std::integral_constant<int, 4>{});
    SMX_HIP_CHECK(hipGetLastError());
    return;
  }
  if (lanes >= 8 && tg.mel32) {   // fused mel at fft 1024 / 512
    Mel32Args m = *tg.mel32;
    m.out_offset = out_offset;
    auto launch_mel_lanes = [&](auto ll) {
      constexpr int LL = decltype(ll)::value;
      auto by_power = [&](auto al, auto fr) {
        constexpr bool A = decltype(al)::value, F = decltype(fr)::value;
        return a.pmode == 2 ? stft_mel_lanes_kernel<LL, A, 2, F> : a.pmode == 1 ? stft_mel_lanes_kernel<LL, A, 1, F> : stft_mel_lanes_kernel<LL, A, 0, F>;
      };
      auto by_al = [&](auto fr) { return aligned ? by_power(std::true_type{}, fr) : by_power(std::false_type{}, fr); };
      auto kl = m.four == 2 ? by_al(std::true_type{}) : by_al(std::false_type{});
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
      // The flush in whole aligned 128-byte lines (stft_fast_p16.hpp, SkL: a frame per lane, the row's trailing frames carried in
      // registers): fft 1024, consecutive tiles of a clip on one workgroup.  SMX_POWER_SKEW=0: the plain per-tile flush.
      // (Sustained, interleaved, 256 clips of C1's length: fft 1024 0.435 -> 0.400 ms, 0.458 -> 0.404 at 1729 frames a clip; fft 512,
      // whose plain runs are 256 bytes, 0.368 -> 0.379: it keeps the plain flush.  profiles/r07/ab_lanes_skew.log)
      if constexpr (LL == 16) {
        // (a launch whose workgroups hold a single tile each -- C1 itself: one clip, 54 tiles -- has nothing to carry: the plain flush, 18.4 against 19.4 us a call)
        if (!strip && env_flag("SMX_POWER_SKEW") != 0 && reinterpret_cast<uintptr_t>(a.out) % 4 == 0 && a.total_tiles >= 2 * (int64_t)a.blocks) {
          a.interleave = 0;
          auto by_power = [&](auto al) {
            constexpr bool A = decltype(al)::value;
            return a.pmode == 2 ? stft_power_lanes_kernel<LL, A, 2, false, true> : a.pmode == 1 ? stft_power_lanes_kernel<LL, A, 1, false, true> : stft_power_lanes_kernel<LL, A, 0, false, true>;
          };
          kl = aligned ? by_power(std::true_type{}) : by_power(std::false_type{});
        }
      }
      SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kl), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PL<LL>::Lds));
      SMX_LAUNCH(kl, dim3((unsigned)a.blocks), dim3(512), PL<LL>::Lds, job.stream, a);
    };
    if (lanes == 16) launch_lanes(std::integral_constant<int, 16>{});
    else if (lanes == 8) launch_lanes(std::integral_constant<int, 8>{});
    else launch_lanes(std::integral_constant<int, 4>{});
    SMX_HIP_CHECK(hipGetLastError());
    return;
  }
  if (tg.mel && tg.mel32) {
    Mel32Args m = *tg.mel32;
    m.out_offset = out_offset;
    auto by_power = [&](auto al, auto fr) {
      constexpr bool A = decltype(al)::value;
      constexpr int F = decltype(fr)::value;
      return a.pmode == 2 ? stft2048_mel32_kernel<A, 2, F> : a.pmode == 1 ? stft2048_mel32_kernel<A, 1, F> : stft2048_mel32_kernel<A, 0, F>;
    };
    auto by_al = [&](auto fr) { return aligned ? by_power(std::true_type{}, fr) : by_power(std::false_type{}, fr); };
    auto k32 = m.four == 2 ? by_al(std::integral_constant<int, 2>{}) : m.four == 1 ? by_al(std::integral_constant<int, 1>{}) : by_al(std::integral_constant<int, 0>{});
    SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFast32Lds));
    SMX_LAUNCH(k32, dim3((unsigned)a.blocks), dim3(512), kFast32Lds, job.stream, a, m);
    SMX_HIP_CHECK(hipGetLastError());
    return;
  }
  if (tg.complex_out && tg.frame_major) {   // the spectrum frame-major, straight from the registers (stft_fast_p32.hpp: stft2048_complex_fm_kernel)
    auto kf = aligned ? stft2048_complex_fm_kernel<true> : stft2048_complex_fm_kernel<false>;
    SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFast32Lds));
    SMX_LAUNCH(kf, dim3((unsigned)a.blocks), dim3(512), kFast32Lds, job.stream, a);
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
  const bool size_ok = c.fft_size == kN || (power_face && (c.fft_size == kN16 || c.fft_size == kN8 || (c.fft_size == kN4 && diag_flag("SMX_FFT256_OFF") != 1)) &&
                                              diag_flag("SMX_POWER16_OFF") != 1) ||
                       (power_face && c.fft_size == k64N && job.mode != OUT_COMPLEX && diag_flag("SMX_P64_OFF") != 1);   // fft 4096: the power face (round 6)
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
  // Round 5, fft 2048: the border frames ride in the tile sequence itself -- ONE launch over every frame of the request, a tile
  // that holds such a frame takes its samples through the padding rule (load_frame32_padded: one reflection, hence n >= fft).
  // The epilogue below ran 4 tile times deep on a quarter of the workgroups at C2: 30 of the launch's 495 us
  // (profiles/r07/timeline_before.log).  SMX_BORDER_INLINE=0: the epilogue / strips as before (same values: tested).
  // (the complex spectrogram and the fused mel kernel at fft 2048 the same way: no gather launches before them)
  // (and the fft 1024 / 512 / 256 kernels of stft_fast_p16.hpp: load_frameL_padded)
  if ((c.fft_size == kN || c.fft_size == kN16 || c.fft_size == kN8 || c.fft_size == kN4) && (i0 - p0) + (p1 - i1) > 0 && job.n >= c.fft_size &&
      job.n < (int64_t(1) << 30) && env_flag("SMX_BORDER_INLINE") != 0) {
    FastTarget folded = tg;
    folded.inline_border = true;
    folded.border_p0 = i0;
    folded.border_i1 = i1;
    launch_interior(job, folded, reinterpret_cast<const float *>(job.x), job.n, job.x_stride, job.left, p0, p1 - p0, tg.out_offset);
    return;
  }
  if (!tg.mel && !tg.complex_out && !fold_off && job.n < (int64_t(1) << 30) && (i0 - p0) + (p1 - i1) > 0 &&
      (i0 - p0) + (p1 - i1) < 4096 && border_total <= epilogue_max && c.fft_size != k64N) {   // (fft 4096: gathered strips, below)
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

// Griffin-Lim's analysis step (capi.cpp): the complex spectrum of every frame of the request frame-major -- out[clip][frame][bin] in
// rows of pitch_floats floats, rows_per_clip (= a multiple of 16, at least the frames) rows a clip.  false = not taken (the caller
// keeps the reference layout): fft 2048 / float32 interior only, and a request whose border frames can ride in the tile sequence.
bool launch_stft_complex_fm(const StftJob &job, void *out, int64_t pitch_floats, int64_t rows_per_clip, bool only_ask) {
  const smx_stft_config &c = *job.cfg;
  if (!fast_eligible(job, /*power_face=*/false) || c.fft_size != kN || job.mode != OUT_COMPLEX) return false;
  if (job.count <= 0 || job.lead <= 0) return true;
  if (rows_per_clip != (job.count + kFT - 1) / kFT * kFT || pitch_floats < 2 * (kN / 2 + 1) || pitch_floats % 2 != 0) return false;
  const int64_t p0 = job.p0, p1 = job.p0 + job.count;
  int64_t i0 = job.left > 0 ? (job.left + c.hop - 1) / c.hop : 0;
  int64_t i1 = job.n + job.left - c.fft_size >= 0 ? (job.n + job.left - c.fft_size) / c.hop + 1 : 0;
  if (i0 < p0) i0 = p0;
  if (i1 > p1) i1 = p1;
  if (i1 <= i0) return false;
  const bool border = (i0 - p0) + (p1 - i1) > 0;
  if (border && !(job.n >= c.fft_size && job.n < (int64_t(1) << 30) && env_flag("SMX_BORDER_INLINE") != 0)) return false;
  if (only_ask) return true;
  FastTarget tg;
  tg.complex_out = true;
  tg.frame_major = true;
  tg.out = out;
  tg.out_stride = pitch_floats;
  tg.out_offset = 0;
  if (border) {
    tg.inline_border = true;
    tg.border_p0 = i0;
    tg.border_i1 = i1;
  }
  launch_interior(job, tg, reinterpret_cast<const float *>(job.x), job.n, job.x_stride, job.left, p0, p1 - p0, 0);
  return true;
}

// The fused audio -> mel spectrogram: the register frame pipelines with the filterbank product over their tiles (fft 2048:
// stft2048_mel32_kernel, fft 1024 / 512: stft_mel_lanes_kernel), one kernel family for every plan the planner accepts
// (smx_mel_config::fused32_plan); false = not taken (the caller composes power spectrogram + Mel.apply on the device).
// Batches beyond 65 535 clips, or whose output passes 2^32 bytes, go through the same kernel in chunks of clips: a clip's values
// never depend on the size of its batch (mel_props.ml:136-155).
bool launch_mel_spectrogram_fused(const MelSpecJob &job) {
  const StftJob &sj0 = job.stft;
  const int64_t fft = sj0.cfg->fft_size;
  const bool lanes = fft == kN16 || fft == kN8;
  if (!lanes && fft != kN) return launch_mel_spectrogram_16(job);
  if (sj0.in_bytes != 4 || sj0.interior != SMX_INTERIOR_F32 || sj0.mode == OUT_COMPLEX || fast_path_disabled())
    return launch_mel_spectrogram_16(job);
  if (!lanes && !fast_eligible(sj0)) return launch_mel_spectrogram_16(job);
  // fft 2048: the banded 4 x 4 x 1 product where its plan exists (SMX_MEL_DENSE=1: the dense 16 x 16 x 4 one, A/B and tests)
  const MelFusedPlan *p4 = env_flag("SMX_MEL_DENSE") != 1 ? &job.mel->fused4_plan() : nullptr;
  const bool four = p4 && p4->state == 1 && (!lanes || p4->resident);   // (the lanes kernels have the resident form only)
  const MelFusedPlan &pl = four ? *p4 : job.mel->fused32_plan();
  if (pl.state != 1) return launch_mel_spectrogram_16(job);
  if (sj0.count <= 0 || sj0.lead <= 0) return true;
  const int64_t n_mels = job.mel->n_mels;
  int64_t chunk = 65535;
  while (chunk > 1 && n_mels * sj0.count * 4 * chunk >= (int64_t(1) << 32)) chunk /= 2;
  if (n_mels * sj0.count * 4 >= (int64_t(1) << 32)) return launch_mel_spectrogram_16(job);   // one clip alone passes the 32-bit offsets
  for (int64_t c0 = 0; c0 < sj0.lead; c0 += chunk) {
    StftJob sj = sj0;
    sj.lead = std::min<int64_t>(chunk, sj0.lead - c0);
    sj.x = reinterpret_cast<const unsigned char *>(sj0.x) + (size_t)c0 * (size_t)sj0.x_stride * 4u;
    Mel32Args m32{};
    m32.items = reinterpret_cast<const Mel32Item *>(pl.items);
    m32.w = pl.w_mfma;
    m32.out = reinterpret_cast<float *>(job.out) + (size_t)c0 * (size_t)n_mels * (size_t)sj0.count;
    m32.out_stride = sj0.count;
    m32.out_offset = 0;
    m32.n_mels = (int)n_mels;
    m32.last_bin = (int)(fft / 2);
    m32.four = four ? (pl.resident ? 2 : 1) : 0;
    FastTarget tg;
    tg.out = m32.out;
    tg.out_stride = sj0.count;
    tg.out_offset = 0;
    tg.mel = true;
    tg.mel32 = &m32;
    launch_ranges(sj, tg);
  }
  return true;
}

}  // namespace smx

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
  if (total() > 4096) return plan;   // a dense filterbank of many rows: the composition (power spectrogram + Mel.apply) serves it
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
  // (No bound on a wave's steps any more: round 3 sent plans with more than 52 steps on one wave -- few wide mels, 40 at
  // 22.05 kHz -- to the 64-lane kernel, which was 5 % ahead there; that kernel family is gone.)
  SMX_HIP_CHECK(hipMalloc(&plan.items, items.size() * sizeof(Mel32Item)));
  SMX_HIP_CHECK(hipMemcpy(plan.items, items.data(), items.size() * sizeof(Mel32Item), hipMemcpyHostToDevice));
  SMX_HIP_CHECK(hipMalloc((void **)&plan.w_mfma, wm.size() * sizeof(float)));
  SMX_HIP_CHECK(hipMemcpy(plan.w_mfma, wm.data(), wm.size() * sizeof(float), hipMemcpyHostToDevice));
  plan.state = 1;
  return plan;
}

// ---- work plan of the banded product (stft_fast_mel32.hpp, Mel4Item): mel groups of 4 rows, each over its own band; a band much
// longer than a wave's share is cut in 2 or 4 K-parts inside one item; items dealt to the 8 waves longest first
const smx::MelFusedPlan &smx_mel_config::fused4_plan() const {
  using namespace smx;
  int device = 0;
  SMX_HIP_CHECK(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(mutex_);
  MelFusedPlan &plan = fused4_[device];
  if (plan.state != 0) return plan;
  plan.state = -1;
  const int64_t nb = bins();
  if ((fft_size != kN && fft_size != kN16 && fft_size != kN8) || n_mels < 1 || n_mels > 252) return plan;
  const int max_chunks = fft_size == kN ? kMel4rChunks : kMel4rChunksL;   // operand registers / 8 of the kernel that takes the plan
  struct Group { int row0, nr, lo, len; };
  std::vector<Group> groups;
  int64_t total = 0;
  for (int r0 = 0; r0 < n_mels; r0 += 4) {
    Group g{r0, (int)std::min<int64_t>(4, n_mels - r0), 0, 0};
    int lo = (int)nb, hi = 0;
    for (int r = r0; r < r0 + g.nr; ++r)
      for (int64_t k = 0; k < nb; ++k)
        if ((float)weights[(size_t)(r * nb + k)] != 0.0f) {
          lo = std::min(lo, (int)k);
          hi = std::max(hi, (int)k + 1);
        }
    if (hi > lo) {
      g.lo = lo;
      g.len = hi - lo;
    } else {
      g.len = 1;   // rows without a weight: a step over zeros writes their zeros
    }
    total += g.len;
    groups.push_back(g);
  }
  const int target = (int)std::max<int64_t>(8, (total + 31) / 32);   // steps of a wave when every lane group of every item is busy
  struct Piece { int row0, nr, kb, kn; };
  struct Item { Piece p[4]; int mode, steps; };
  std::vector<Item> items;
  std::vector<const Group *> ones, twos;
  auto part = [](const Group &g, int parts, int q) {
    const int per = (g.len + parts - 1) / parts, kb = g.lo + q * per;
    const int kn = std::max(0, std::min(per, g.lo + g.len - kb));
    return Piece{g.row0, g.nr, kn > 0 ? kb : g.lo, kn};
  };
  auto finish = [&](Item it) {
    int longest = 1;
    for (const Piece &q : it.p) longest = std::max(longest, q.kn);
    it.steps = (longest + 3) / 4 * 4;
    items.push_back(it);
  };
  for (const Group &g : groups) {
    if (g.len > target * 5 / 2) finish(Item{{part(g, 4, 0), part(g, 4, 1), part(g, 4, 2), part(g, 4, 3)}, 4, 0});
    else if (g.len > target * 5 / 4) twos.push_back(&g);
    else ones.push_back(&g);
  }
  const Piece idle{0, 0, 0, 0};
  for (size_t i = 0; i < twos.size(); i += 2) {
    Item it{{part(*twos[i], 2, 0), part(*twos[i], 2, 1), idle, idle}, 2, 0};
    if (i + 1 < twos.size()) {
      it.p[2] = part(*twos[i + 1], 2, 0);
      it.p[3] = part(*twos[i + 1], 2, 1);
    }
    finish(it);
  }
  for (size_t i = 0; i < ones.size(); i += 4) {
    Item it{{idle, idle, idle, idle}, 1, 0};
    for (size_t q = 0; q < 4 && i + q < ones.size(); ++q) it.p[q] = part(*ones[i + q], 1, 0);
    finish(it);
  }
  if (items.size() > 8 * (size_t)kMel32MaxItems) return plan;
  auto weight_row = [&](const Item &it, int st, std::vector<float> &dst) {   // the 64 A operands of step st
    for (int lane = 0; lane < 64; ++lane) {
      const Piece &q = it.p[lane >> 4];
      const int r = lane & 3;
      const int64_t k = (int64_t)q.kb + st;
      dst.push_back((r < q.nr && st < q.kn && k < nb) ? (float)weights[(size_t)((q.row0 + r) * nb + k)] : 0.0f);
    }
  };
  auto pack = [](const Item &it, int steps, int a_offset) {
    Mel4Item d{};
    unsigned rows = 0, nrows = 0;
    for (int q = 0; q < 4; ++q) {
      rows |= (unsigned)it.p[q].row0 << (8 * q);
      nrows |= (unsigned)it.p[q].nr << (8 * q);
      d.kb[q] = it.p[q].kb;
    }
    d.rows = (int)rows;
    d.nrows = (int)nrows;
    d.steps_mode = steps | (it.mode << 16);
    d.a_offset = a_offset;
    return d;
  };
  std::sort(items.begin(), items.end(), [](const Item &x, const Item &y) { return x.steps > y.steps; });
  {   // resident form: items in whole chunks of 8 steps, at most 8 chunks (64 operand registers) and 8 items per wave
    std::vector<std::vector<const Item *>> per_wave(8);
    int chunks[8] = {0};
    bool fits = true;
    for (const Item &it : items) {
      const int need = (it.steps + 7) / 8;
      int w = -1;
      for (int i = 0; i < 8; ++i)
        if ((int)per_wave[(size_t)i].size() < kMel32MaxItems && chunks[i] + need <= max_chunks && (w < 0 || chunks[i] < chunks[w])) w = i;
      if (w < 0) { fits = false; break; }
      per_wave[(size_t)w].push_back(&it);
      chunks[w] += need;
    }
    if (fits) {
      std::vector<Mel4Item> table(8 * kMel32MaxItems, Mel4Item{});
      std::vector<float> wm;
      wm.reserve((size_t)8 * 8 * max_chunks * 64);
      for (int w = 0; w < 8; ++w) {
        int n = 0, used = 0;
        for (const Item *it : per_wave[(size_t)w]) {
          const int steps = (it->steps + 7) / 8 * 8;
          table[(size_t)(w * kMel32MaxItems + n++)] = pack(*it, steps, 0);
          for (int st = 0; st < steps; ++st) weight_row(*it, st, wm);
          used += steps;
        }
        wm.resize(wm.size() + (size_t)(8 * max_chunks - used) * 64, 0.0f);
      }
      SMX_HIP_CHECK(hipMalloc(&plan.items, table.size() * sizeof(Mel4Item)));
      SMX_HIP_CHECK(hipMemcpy(plan.items, table.data(), table.size() * sizeof(Mel4Item), hipMemcpyHostToDevice));
      SMX_HIP_CHECK(hipMalloc((void **)&plan.w_mfma, wm.size() * sizeof(float)));
      SMX_HIP_CHECK(hipMemcpy(plan.w_mfma, wm.data(), wm.size() * sizeof(float), hipMemcpyHostToDevice));
      plan.resident = 1;
      plan.state = 1;
      return plan;
    }
  }
  if (fft_size != kN) return plan;   // the streamed form exists for the fft-2048 kernel only
  std::vector<Mel4Item> table(8 * kMel32MaxItems, Mel4Item{});
  int load[8] = {0}, count[8] = {0};
  std::vector<float> wm;
  for (const Item &it : items) {
    int w = -1;
    for (int i = 0; i < 8; ++i)
      if (count[i] < kMel32MaxItems && (w < 0 || load[i] < load[w])) w = i;
    if (w < 0) return plan;
    table[(size_t)(w * kMel32MaxItems + count[w]++)] = pack(it, it.steps, (int)(wm.size() / 64));
    load[w] += it.steps;
    for (int st = 0; st < it.steps; ++st) weight_row(it, st, wm);
  }
  SMX_HIP_CHECK(hipMalloc(&plan.items, table.size() * sizeof(Mel4Item)));
  SMX_HIP_CHECK(hipMemcpy(plan.items, table.data(), table.size() * sizeof(Mel4Item), hipMemcpyHostToDevice));
  SMX_HIP_CHECK(hipMalloc((void **)&plan.w_mfma, wm.size() * sizeof(float)));
  SMX_HIP_CHECK(hipMemcpy(plan.w_mfma, wm.data(), wm.size() * sizeof(float), hipMemcpyHostToDevice));
  plan.state = 1;
  return plan;
}
#endif   // SMX_ISA_ONE
