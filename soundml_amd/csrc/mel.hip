// Mel filterbank product, the one dense contraction of the path:
//   Mel.apply c s = cast dtype (matmul W_f64 (cast f64 s))        mel.ml:202-231
// out[l][m][t] = sum_b W[m][b] * S[l][b][t], W = [n_mels; bins] triangular filters.
//
// float32: v_mfma_f32_32x32x2_f32 (exact f32 FMA chain, 64 FLOP/clk/SIMD).  A
// wave owns a 32-mel x 64-frame output block (two 32x32 accumulators sharing
// the A operand).  The filters are banded (each mel row touches a contiguous
// bin range), so every 32-mel block only walks the union of its rows' supports:
// for 128 mels over 1025 bins that is ~1/3.8 of the dense K loop.
// float64 (and float32 audio with the float64 interior): VALU dot products over
// the band, float64 accumulation, one rounding -- the reference's arithmetic.
#include "smx_internal.hpp"

namespace smx {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
#ifndef SMX_MEL_DEPTH
#define SMX_MEL_DEPTH 2   // k-steps whose operands are fetched together (C3: 1: 0.34 ms, 2: 0.29, 4: 0.30; next trip prefetched: 0.35)
#endif

struct MelArgs {
  const void *s;
  void *out;
  const float *w_block;  // [n_mels_pad / 32][k_pad / 2][64]: weights in MFMA A-operand order
  const int *block_lo, *block_hi;
  const double *w64;     // [n_mels][bins]
  const int *band_lo, *band_hi;
  int64_t lead, frames;
  int n_mels, n_mels_pad, bins, k_pad;
  int mel_blocks, frame_tiles;
};

// BY_BLOCK = false: grid = lead * frame_tiles workgroups of 4 waves; wave w walks mel blocks w, w+4, ...
// BY_BLOCK = true: a workgroup's 4 waves take 4 neighbouring frame tiles of ONE mel block (equal work per wave), and
// the blocks are dealt longest band first (the highest mels), so the launch ends on the short ones.
template <bool BY_BLOCK>
__global__ void __launch_bounds__(256) mel_apply_mfma_kernel(MelArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t tile = blockIdx.x;
  int mb_first = wave, mb_step = 4, mb_end = a.mel_blocks;
  if (BY_BLOCK) {
    const int64_t groups = (a.lead * a.frame_tiles + 3) / 4;
    mb_first = a.mel_blocks - 1 - (int)(blockIdx.x / groups);
    mb_end = mb_first + 1;
    mb_step = 1;
    tile = (blockIdx.x % groups) * 4 + wave;
    if (tile >= a.lead * a.frame_tiles) return;
  }
  const int64_t clip = tile / a.frame_tiles;
  const int64_t t0 = (tile % a.frame_tiles) * 64;
  const float *S = reinterpret_cast<const float *>(a.s) + clip * (int64_t)a.bins * a.frames;
  float *O = reinterpret_cast<float *>(a.out) + clip * (int64_t)a.n_mels * a.frames;
  const int col = lane & 31, half = lane >> 5;
  const int64_t ta = t0 + col, tb = t0 + 32 + col;
  const int64_t ta_c = ta < a.frames ? ta : a.frames - 1;
  const int64_t tb_c = tb < a.frames ? tb : a.frames - 1;
  for (int mb = mb_first; mb < mb_end; mb += mb_step) {
    // the block's band (host-built, wave-uniform) and its weights in operand order: one 256-byte run per k-step
    const int klo = __builtin_amdgcn_readfirstlane(a.block_lo[mb]) & ~7;
    const int khi = __builtin_amdgcn_readfirstlane(a.block_hi[mb]);
    f32x16 acc0 = {0}, acc1 = {0};
    const float *wb = a.w_block + (int64_t)mb * (a.k_pad / 2) * 64 + lane;
    // four k-steps (8 bins) per trip, every operand fetched before the first multiply-add; trips may run past the
    // band (W is zero there; k_pad is a multiple of 32) but never past the spectrogram's last bin: whole trips walk a
    // uniform row pointer with no guards and no 64-bit multiplies, the one trip that can cross the last bin is guarded
    const int whole_end = khi < (a.bins & ~7) ? khi : (a.bins & ~7);
    const int64_t lane_a = (int64_t)half * a.frames + ta_c, lane_b = (int64_t)half * a.frames + tb_c;
    const int64_t two_rows = 2 * a.frames;
    const float *rows = S + (int64_t)klo * a.frames;     // wave-uniform
    int k = klo;
    for (; k + 8 <= whole_end; k += 8, rows += 4 * two_rows) {
#pragma unroll
      for (int h = 0; h < 4; h += SMX_MEL_DEPTH) {
        float av[SMX_MEL_DEPTH], b0[SMX_MEL_DEPTH], b1[SMX_MEL_DEPTH];
#pragma unroll
        for (int j = 0; j < SMX_MEL_DEPTH; ++j) {
          av[j] = wb[(int64_t)(k / 2 + h + j) * 64];
          b0[j] = rows[(h + j) * two_rows + lane_a];
          b1[j] = rows[(h + j) * two_rows + lane_b];
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the loads ahead of the first multiply-add
#pragma unroll
        for (int j = 0; j < SMX_MEL_DEPTH; ++j) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b0[j], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b1[j], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    for (; k < khi; k += 8) {
      float av[4], b0[4], b1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = k + 2 * j + half;
        av[j] = wb[(int64_t)(k / 2 + j) * 64];                     // W[m0 + col][k + 2 j + half]
        const bool ok = kk < a.bins;
        const int64_t row = (int64_t)(ok ? kk : 0) * a.frames;
        b0[j] = ok ? S[row + ta_c] : 0.0f;                         // S[kk][t0 + col]
        b1[j] = ok ? S[row + tb_c] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b0[j], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b1[j], acc1, 0, 0, 0);
      }
    }
    // C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = mb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
      if (m < a.n_mels) {
        if (ta < a.frames) O[(int64_t)m * a.frames + ta] = acc0[reg];
        if (tb < a.frames) O[(int64_t)m * a.frames + tb] = acc1[reg];
      }
    }
  }
}

// float64 accumulation over the band; Tio = float (f64 interior on f32 data) or double
template <typename Tio>
__global__ void __launch_bounds__(256) mel_apply_f64_kernel(MelArgs a) {
  const int64_t clip = blockIdx.y;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= a.frames) return;
  const Tio *S = reinterpret_cast<const Tio *>(a.s) + clip * (int64_t)a.bins * a.frames;
  Tio *O = reinterpret_cast<Tio *>(a.out) + clip * (int64_t)a.n_mels * a.frames;
  for (int m = 0; m < a.n_mels; ++m) {
    const double *w = a.w64 + (int64_t)m * a.bins;
    double acc = 0.0;
    for (int b = a.band_lo[m]; b < a.band_hi[m]; ++b) acc += w[b] * (double)S[(int64_t)b * a.frames + t];
    O[(int64_t)m * a.frames + t] = (Tio)acc;
  }
}

}  // namespace

void launch_mel_apply(const MelJob &job) {
  if (job.lead <= 0 || job.frames <= 0) return;
  const smx_mel_config &c = *job.cfg;
  const smx_mel_config::Tables &t = c.tables();
  MelArgs a{};
  a.s = job.s;
  a.out = job.out;
  a.w_block = t.w_block;
  a.block_lo = t.block_lo;
  a.block_hi = t.block_hi;
  a.w64 = t.w_f64;
  a.band_lo = t.band_lo;
  a.band_hi = t.band_hi;
  a.lead = job.lead;
  a.frames = job.frames;
  a.n_mels = (int)c.n_mels;
  a.n_mels_pad = (int)t.n_mels_pad;
  a.bins = (int)c.bins();
  a.k_pad = (int)t.k_pad;
  a.mel_blocks = (int)(t.n_mels_pad / 32);
  a.frame_tiles = (int)((job.frames + 63) / 64);
  const bool f64_interior = job.elem_bytes == 8 || smx_get_interior() == SMX_INTERIOR_F64;
  if (!f64_interior) {
    static const bool by_tile = diag_flag("SMX_MEL_APPLY_BY_TILE") == 1;   // diagnostic: all blocks in one workgroup
    if (by_tile) {
      const int64_t blocks = job.lead * a.frame_tiles;
      if (blocks > 0x7fffffff) throw Failure("apply: too many frame tiles for one launch");
      SMX_LAUNCH(mel_apply_mfma_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, job.stream, a);
    } else {
      const int64_t blocks = (job.lead * a.frame_tiles + 3) / 4 * a.mel_blocks;
      if (blocks > 0x7fffffff) throw Failure("apply: too many frame tiles for one launch");
      SMX_LAUNCH(mel_apply_mfma_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, job.stream, a);
    }
  } else {
    if (job.lead > 65535) throw Failure("apply: too many leading slices for one launch");
    dim3 grid((unsigned)((job.frames + 255) / 256), (unsigned)job.lead);
    if (job.elem_bytes == 8)
      SMX_LAUNCH(mel_apply_f64_kernel<double>, grid, dim3(256), 0, job.stream, a);
    else
      SMX_LAUNCH(mel_apply_f64_kernel<float>, grid, dim3(256), 0, job.stream, a);
  }
  SMX_HIP_CHECK(hipGetLastError());
}

}  // namespace smx
