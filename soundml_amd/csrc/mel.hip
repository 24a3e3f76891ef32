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

struct MelArgs {
  const void *s;
  void *out;
  const float *w32;      // [n_mels_pad][k_pad]
  const double *w64;     // [n_mels][bins]
  const int *band_lo, *band_hi;
  int64_t lead, frames;
  int n_mels, n_mels_pad, bins, k_pad;
  int mel_blocks, frame_tiles;
};

// grid: lead * frame_tiles workgroups of 4 waves; wave w walks mel blocks w, w+4, ...
__global__ void __launch_bounds__(256) mel_apply_mfma_kernel(MelArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t clip = blockIdx.x / a.frame_tiles;
  const int64_t t0 = (int64_t)(blockIdx.x % a.frame_tiles) * 64;
  const float *S = reinterpret_cast<const float *>(a.s) + clip * (int64_t)a.bins * a.frames;
  float *O = reinterpret_cast<float *>(a.out) + clip * (int64_t)a.n_mels * a.frames;
  const int col = lane & 31, half = lane >> 5;
  const int64_t ta = t0 + col, tb = t0 + 32 + col;
  const int64_t ta_c = ta < a.frames ? ta : a.frames - 1;
  const int64_t tb_c = tb < a.frames ? tb : a.frames - 1;
  for (int mb = wave; mb < a.mel_blocks; mb += 4) {
    // union of the block's band supports (wave-uniform)
    int klo = a.bins, khi = 0;
    for (int m = mb * 32; m < mb * 32 + 32 && m < a.n_mels; ++m) {
      const int lo = a.band_lo[m], hi = a.band_hi[m];
      klo = lo < klo ? lo : klo;
      khi = hi > khi ? hi : khi;
    }
    klo &= ~1;
    f32x16 acc0 = {0}, acc1 = {0};
    const float *wrow = a.w32 + (int64_t)(mb * 32 + col) * a.k_pad + half;
    for (int k = klo; k < khi; k += 2) {
      const int kk = k + half;
      const float av = wrow[k];                                   // W[m0 + col][k + half] (zero padded)
      const bool ok = kk < a.bins;
      const int64_t row = (int64_t)(ok ? kk : 0) * a.frames;
      const float b0 = ok ? S[row + ta_c] : 0.0f;                 // S[k + half][t0 + col]
      const float b1 = ok ? S[row + tb_c] : 0.0f;
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc1, 0, 0, 0);
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
  a.w32 = t.w_f32;
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
    const int64_t blocks = job.lead * a.frame_tiles;
    if (blocks > 0x7fffffff) throw Failure("apply: too many frame tiles for one launch");
    hipLaunchKernelGGL(mel_apply_mfma_kernel, dim3((unsigned)blocks), dim3(256), 0, job.stream, a);
  } else {
    if (job.lead > 65535) throw Failure("apply: too many leading slices for one launch");
    dim3 grid((unsigned)((job.frames + 255) / 256), (unsigned)job.lead);
    if (job.elem_bytes == 8)
      hipLaunchKernelGGL(mel_apply_f64_kernel<double>, grid, dim3(256), 0, job.stream, a);
    else
      hipLaunchKernelGGL(mel_apply_f64_kernel<float>, grid, dim3(256), 0, job.stream, a);
  }
  SMX_HIP_CHECK(hipGetLastError());
}

}  // namespace smx
