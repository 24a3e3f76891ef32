// Fused framing + window + real FFT(2048) + |.|^p for gfx950 (MI355X).
//
// Replaces, for float32 audio and fft_size = 2048, the reference's hot call
//   Nx.stft cdtype ~window:fft ~step:hop ~win (to_double samples)   stft.ml:356-364
//   |> Nx.magnitude |> Nx.square                                     stft.ml:670-674
// with one kernel that reads each audio sample from HBM once (hop-strided
// overlapping frames are re-read through L2) and writes the [bins; frames]
// power spectrogram once, in 64-byte runs along the frame axis.
//
// Decomposition (M = N/2 = 1024 complex points, z[n] = x[2n] + i x[2n+1]):
//   one 64-lane wavefront owns one frame; lane l holds 16 complex points.
//   A. n = l + 64 j      : radix-16 over j in registers        -> y_l[k1]
//      twiddle W_M^(l k1)
//   X. one LDS transpose (8.5 KB per wave, padded rows, conflict-free b64)
//      lane l' = 4 k1 + a receives y_(4i+a)[k1], i = 0..15
//   B. radix-16 over i in registers, twiddle W_64^(a q)
//   C. radix-4 over a across the 4 lanes of a quad with DPP quad_perm
//      -> lane (k1, rr), register q holds Z[k1 + 16 q + 256 r], r = bitrev2(rr)
//   P. real-FFT post-pass: partner Z[M-k] fetched with ds_bpermute
//      (lane 67-l', register 15-q; lanes 0..3 are the k1 = 0 column and pair
//      inside themselves), X[k] = E - i w_k D with the 1/2 folded into the window.
//   T. |X|^2 is written into a workgroup tile [1025 bins][16 frames] in LDS
//      (bank-conflict-free row permutation), and the 8 waves flush the tile to
//      HBM frames-fastest.
// A workgroup is 8 waves = 16 frames (2 per wave); LDS = 69.6 KB exchange +
// 69.7 KB tile; 1 workgroup per CU, 2 waves per SIMD, <= 256 VGPRs.
//
// Algorithmic HBM bytes per frame: hop*4 read + 1025*4 written = 6148 B at
// hop 512.  Flops per frame ~= 50k VALU lane-ops.
#include "smx_internal.hpp"

namespace smx {
namespace {

struct c32 {
  float x, y;
};
__device__ __forceinline__ c32 operator+(c32 a, c32 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c32 operator-(c32 a, c32 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c32 cmul(c32 a, c32 w) {
  return {a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x};
}
__device__ __forceinline__ c32 mul_neg_i(c32 a) { return {a.y, -a.x}; }

// 4-point forward DFT in place: (a,b,c,d) -> (X0,X1,X2,X3)
__device__ __forceinline__ void fft4(c32 &a, c32 &b, c32 &c, c32 &d) {
  const c32 t0 = a + c, t1 = a - c, t2 = b + d, t3 = mul_neg_i(b - d);
  a = t0 + t2;
  b = t1 + t3;
  c = t0 - t2;
  d = t1 - t3;
}

// 16-point forward DFT, natural order in and out, fully in registers (4 x 4).
__device__ __forceinline__ void fft16(c32 (&v)[16]) {
  constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f;
  constexpr float h = 0.70710678118654752f;
#pragma unroll
  for (int n0 = 0; n0 < 4; ++n0) fft4(v[n0], v[4 + n0], v[8 + n0], v[12 + n0]);
  // u[n0][k0] sits at v[4 k0 + n0]; multiply by W16^(n0 k0)
  v[4 * 1 + 1] = cmul(v[4 * 1 + 1], c32{c1, -s1});   // W^1
  v[4 * 1 + 2] = cmul(v[4 * 1 + 2], c32{h, -h});     // W^2
  v[4 * 1 + 3] = cmul(v[4 * 1 + 3], c32{s1, -c1});   // W^3
  v[4 * 2 + 1] = cmul(v[4 * 2 + 1], c32{h, -h});     // W^2
  v[4 * 2 + 2] = mul_neg_i(v[4 * 2 + 2]);            // W^4
  v[4 * 2 + 3] = cmul(v[4 * 2 + 3], c32{-h, -h});    // W^6
  v[4 * 3 + 1] = cmul(v[4 * 3 + 1], c32{s1, -c1});   // W^3
  v[4 * 3 + 2] = cmul(v[4 * 3 + 2], c32{-h, -h});    // W^6
  v[4 * 3 + 3] = cmul(v[4 * 3 + 3], c32{-c1, s1});   // W^9
#pragma unroll
  for (int k0 = 0; k0 < 4; ++k0) fft4(v[4 * k0], v[4 * k0 + 1], v[4 * k0 + 2], v[4 * k0 + 3]);
  // X[k0 + 4 k1] sits at v[4 k0 + k1]: transpose the 4x4 index
  c32 t[16];
#pragma unroll
  for (int k0 = 0; k0 < 4; ++k0)
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) t[k0 + 4 * k1] = v[4 * k0 + k1];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = t[i];
}

template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

__device__ __forceinline__ float bperm(int byte_addr, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}

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
  int tiles_per_clip, groups_per_clip, tiles_per_group;
  int64_t blocks;
  int pmode;            // 2: power 2, 1: power 1, 0: general
  float half_power;
};

constexpr int kN = 2048, kM = 1024, kBins = 1025;
constexpr int kWaves = 8, kFT = 16;
constexpr int kXRow = 68;                       // float2 per exchange row (64 + 4 pad)
constexpr int kXWave = 16 * kXRow;              // float2 per wave
constexpr int kTileStride = kFT + 1;            // floats per tile row
constexpr size_t kExchBytes = (size_t)kWaves * kXWave * sizeof(float2);
constexpr size_t kTileBytes = ((size_t)kBins * kTileStride * sizeof(float) + 15) / 16 * 16;
constexpr size_t kTabPBytes = 16 * 64 * sizeof(float2);   // post-pass twiddles [q][lane]
constexpr size_t kTabBBytes = 16 * 4 * sizeof(float2);    // W_64^(a q)       [q][a]
constexpr size_t kFastLds = kExchBytes + kTileBytes + kTabPBytes + kTabBBytes;

__device__ __forceinline__ float fetch_padded(const float *x, int64_t n, int64_t s, int pad,
                                              float pad_value) {
  if (s >= 0 && s < n) return x[s];
  if (pad == SMX_PAD_REFLECT) {  // stft.ml:300-305
    if (n == 1) return x[0];
    const int64_t period = 2 * (n - 1);
    int64_t m = s % period;
    if (m < 0) m += period;
    return x[m < n ? m : period - m];
  }
  if (pad == SMX_PAD_EDGE) return x[s < 0 ? 0 : n - 1];
  return pad_value;
}

template <bool ALIGNED>
__global__ void __launch_bounds__(512, 2) stft2048_power_kernel(FastArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  float2 *exch = reinterpret_cast<float2 *>(smem) + wave * kXWave;
  float *tile = reinterpret_cast<float *>(smem + kExchBytes);
  float2 *tabP = reinterpret_cast<float2 *>(smem + kExchBytes + kTileBytes);
  float2 *tabB = reinterpret_cast<float2 *>(smem + kExchBytes + kTileBytes + kTabPBytes);

  // XCD-aware block -> (clip, tile group): blocks that share an XCD (b % 8) get a
  // contiguous range of virtual ids, i.e. whole clips, so halo re-reads and the
  // partial output lines of neighbouring tiles meet in one L2.
  int64_t vb = blockIdx.x;
  {
    const int64_t nb = a.blocks, q = nb / 8, r = nb % 8, xcd = vb % 8, idx = vb / 8;
    vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int64_t clip = vb / a.groups_per_clip;
  const int group = (int)(vb % a.groups_per_clip);
  const float *x = a.x + clip * a.x_stride;

  // ---- per-lane constants ----------------------------------------------------
  const int k1 = lane >> 2, qa = lane & 3;
  const int r = ((qa & 1) << 1) | (qa >> 1);
  float2 win[16];
  c32 twA[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    win[j] = *reinterpret_cast<const float2 *>(a.hwin + 2 * lane + 128 * j);
    const float2 wa = a.w_m[lane * j];            // W_M^(l k1), k1 = j
    twA[j] = {wa.x, wa.y};
  }
  // workgroup-shared twiddle tables in LDS (each wave fills 2 of the 16 rows)
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int j = 2 * wave + jj;
    tabP[j * 64 + lane] = a.w_n[k1 + 256 * r + 16 * j];   // exp(-2 pi i k / N), k = k1 + 16 j + 256 r
    if (lane < 4) tabB[j * 4 + lane] = a.w_m[16 * lane * j]; // W_64^(a j)
  }
  __syncthreads();
  const float2 *tabP_l = tabP + lane;
  const float2 *tabB_l = tabB + qa;
  const float s1 = qa < 2 ? 1.0f : -1.0f;
  const float s2 = (qa & 1) ? -1.0f : 1.0f;
  const bool rot = qa == 3;
  // partner lanes for the real-FFT post-pass (byte addresses for ds_bpermute)
  int addr_g, addr_0;
  if (lane >= 4) {
    addr_g = addr_0 = (67 - lane) * 4;
  } else {
    addr_g = (3 - lane) * 4;
    const int r0 = (4 - r) & 3;                          // partner r for q = 0
    addr_0 = ((((r0 & 1) << 1) | (r0 >> 1))) * 4;        // its lane = bitrev2(r0)
  }
  const bool low4 = lane < 4;
  const int exch_wr = lane;                               // + k1 * kXRow
  const int exch_rd = k1 * kXRow + qa;                    // + 4 i
  const int tile_row0 = 4 * k1 + r;                       // + 64 q   (row' = 4 (k1 + 16 q) + r)

  const int t_begin = group * a.tiles_per_group;
  int t_end = t_begin + a.tiles_per_group;
  if (t_end > a.tiles_per_clip) t_end = a.tiles_per_clip;

  for (int t = t_begin; t < t_end; ++t) {
    const int64_t f0 = (int64_t)t * kFT;
#pragma unroll 1
    for (int ff = 0; ff < 2; ++ff) {
      const int f = 2 * wave + ff;
      if (f0 + f >= a.count) break;  // wave-uniform
      const int64_t p = a.p0 + f0 + f;
      const int64_t s0 = p * a.hop - a.left;
      c32 v[16];
      if (s0 >= 0 && s0 + kN <= a.n) {
        if constexpr (ALIGNED) {
          const float2 *src = reinterpret_cast<const float2 *>(x + s0) + lane;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const float2 d = src[64 * j];
            v[j] = {d.x * win[j].x, d.y * win[j].y};
          }
        } else {
          const float *src = x + s0 + 2 * lane;
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = {src[128 * j] * win[j].x, src[128 * j + 1] * win[j].y};
        }
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int64_t s = s0 + 2 * lane + 128 * j;
          v[j] = {fetch_padded(x, a.n, s, a.pad, a.pad_value) * win[j].x,
                  fetch_padded(x, a.n, s + 1, a.pad, a.pad_value) * win[j].y};
        }
      }
      // A: radix-16 over j, twiddle
      fft16(v);
#pragma unroll
      for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], twA[k]);
      // X: transpose through LDS (wave-private region; LDS ops of one wave are in order)
#pragma unroll
      for (int k = 0; k < 16; ++k) exch[k * kXRow + exch_wr] = make_float2(v[k].x, v[k].y);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float2 d = exch[exch_rd + 4 * i];
        v[i] = {d.x, d.y};
      }
      __builtin_amdgcn_wave_barrier();
      // B: radix-16 over i, twiddle W_64^(a q)
      fft16(v);
#pragma unroll
      for (int q = 1; q < 16; ++q) {
        const float2 wb = tabB_l[4 * q];
        v[q] = cmul(v[q], c32{wb.x, wb.y});
      }
      // C: radix-4 across the quad (DPP).  lane a ends with r = bitrev2(a).
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        c32 u;
        u.x = fmaf(v[q].x, s1, dpp_quad<0x4E>(v[q].x));
        u.y = fmaf(v[q].y, s1, dpp_quad<0x4E>(v[q].y));
        const c32 w = rot ? c32{u.y, -u.x} : u;
        v[q].x = fmaf(w.x, s2, dpp_quad<0xB1>(w.x));
        v[q].y = fmaf(w.y, s2, dpp_quad<0xB1>(w.y));
      }
      // P: real-FFT post-pass.  provider rotation for the k1 = 0 lanes.
      c32 prov[16];
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        prov[m].x = low4 ? v[(m + 1) & 15].x : v[m].x;
        prov[m].y = low4 ? v[(m + 1) & 15].y : v[m].y;
      }
      float *col = tile + f;
      const float nyq = 2.0f * (v[0].x - v[0].y);   // X[M] = Re Z0 - Im Z0 (true scale), lane 0
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int addr = q == 0 ? addr_0 : addr_g;
        const float px = bperm(addr, prov[15 - q].x);
        const float py = bperm(addr, prov[15 - q].y);
        const c32 e = {v[q].x + px, v[q].y - py};
        const c32 d = {v[q].x - px, v[q].y + py};
        const float2 w = tabP_l[64 * q];
        const float tr = e.x + w.x * d.y + w.y * d.x;
        const float ti = e.y - w.x * d.x + w.y * d.y;
        float pw = tr * tr + ti * ti;
        if (a.pmode == 1) pw = sqrtf(pw);
        else if (a.pmode == 0) pw = __powf(pw, a.half_power);
        col[(tile_row0 + 64 * q) * kTileStride] = pw;
      }
      if (lane == 0) {
        float pw = nyq * nyq;
        if (a.pmode == 1) pw = fabsf(nyq);
        else if (a.pmode == 0) pw = __powf(pw, a.half_power);
        col[kM * kTileStride] = pw;
      }
    }
    __syncthreads();
    // ---- flush tile -> out[clip][bin][frame], frames fastest -------------------
    {
      const int hsel = lane >> 5, jj = (lane & 31) >> 2, g = lane & 3;
      const int rloc = (jj & 3) + 16 * (jj >> 2) + 4 * hsel;
      float *obase = a.out + (clip * kBins) * a.out_stride + a.out_offset + f0 + 4 * g;
      const int64_t fleft = a.count - f0 - 4 * g;    // frames remaining from this column group
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = 32 * (wave + 8 * (it >> 1)) + 8 * (it & 1) + rloc;
        const int bin = (row & 3) * 256 + (row >> 2);
        const float *src = tile + row * kTileStride + 4 * g;
        const float v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
        float *dst = obase + (int64_t)bin * a.out_stride;
        if (fleft >= 4) {
          struct __attribute__((packed, aligned(4))) f4 { float a, b, c, d; };
          *reinterpret_cast<f4 *>(dst) = f4{v0, v1, v2, v3};
        } else {
          if (fleft > 0) dst[0] = v0;
          if (fleft > 1) dst[1] = v1;
          if (fleft > 2) dst[2] = v2;
        }
      }
      if (wave == 0 && lane < 16) {   // bin 1024 (row 1024): 16 frames by 16 lanes
        const int64_t fl = a.count - f0;
        if (lane < fl)
          a.out[(clip * kBins + kM) * a.out_stride + a.out_offset + f0 + lane] =
              tile[kM * kTileStride + lane];
      }
    }
    __syncthreads();
  }
}

}  // namespace

bool launch_stft_fast(const StftJob &job) {
  const smx_stft_config &c = *job.cfg;
  if (fast_path_disabled()) return false;
  if (c.fft_size != kN || job.in_bytes != 4 || job.interior != SMX_INTERIOR_F32) return false;
  if (job.mode != OUT_POWER) return false;
  if (job.count <= 0 || job.lead <= 0) return true;
  const StftTables &t = c.tables();
  FastArgs a{};
  a.x = reinterpret_cast<const float *>(job.x);
  a.n = job.n;
  a.x_stride = job.x_stride;
  a.hop = c.hop;
  a.left = job.left;
  a.pad = job.pad;
  a.pad_value = (float)job.pad_value;
  a.p0 = job.p0;
  a.count = job.count;
  a.out = reinterpret_cast<float *>(job.out);
  a.out_stride = job.out_stride;
  a.out_offset = job.out_offset;
  a.hwin = t.fast_window;
  a.w_m = t.fast_w_m;
  a.w_n = t.fast_w_n;
  const int64_t tiles = (job.count + kFT - 1) / kFT;
  if (tiles > 0x7fffffff) return false;
  a.tiles_per_clip = (int)tiles;
  // enough workgroups to fill 256 CUs several times over, while amortising the
  // per-lane twiddle loads over a few tiles
  int tpg = 8;
  while (tpg > 1 && job.lead * ((tiles + tpg - 1) / tpg) < 2048) tpg >>= 1;
  a.tiles_per_group = tpg;
  a.groups_per_clip = (int)((tiles + tpg - 1) / tpg);
  a.blocks = job.lead * a.groups_per_clip;
  if (a.blocks > 0x7fffffff) return false;
  a.pmode = job.power == 2.0 ? 2 : (job.power == 1.0 ? 1 : 0);
  a.half_power = (float)(0.5 * job.power);
  const bool aligned = (c.hop % 2 == 0) && (job.left % 2 == 0) && (job.x_stride % 2 == 0) &&
                       (reinterpret_cast<uintptr_t>(job.x) % 8 == 0);
  auto kernel = aligned ? stft2048_power_kernel<true> : stft2048_power_kernel<false>;
  SMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFastLds));
  hipLaunchKernelGGL(kernel, dim3((unsigned)a.blocks), dim3(512), kFastLds, job.stream, a);
  SMX_HIP_CHECK(hipGetLastError());
  return true;
}

}  // namespace smx
