// stft2048_mel32_kernel -- the fused audio -> mel spectrogram at fft 2048 on the 32-lane frame pipeline of stft_fast_p32.hpp
// (included by stft_fast.hip after it, inside its anonymous namespace).  Replaces Soundml.mel_spectrogram =
// Mel.apply (Stft.power_spectrum ...), soundml.ml:12-24 + mel.ml:202-231, for banded filterbanks.
//
// The frame code is frame32_to_tile unchanged: every tile of 16 frames x 1025 powers lands in LDS exactly as the power
// kernel leaves it.  Where that kernel reads its share of the previous tile out to HBM, this one multiplies the tile by the
// filterbank: v_mfma_f32_16x16x4_f32 with A = W[16 mels][4 bins] (global memory, pre-arranged in lane order, L2 resident),
// B = P[4 bins][16 frames] (one LDS read per lane: tile rows ARE bins in this pipeline), K running over the union band of
// the item's mels only.  The work is cut into ITEMS of up to 16 mel rows, each summed over its whole band by ONE wave in
// ascending bin order -- no partial sums to combine, one value per (mel, frame) whatever the batch -- and the items are
// dealt to the 8 waves by length (the planner splits the rows of the longest items until no wave holds much more than an
// eighth of the steps: mel_config::fused32_plan).  The operands of kMel32Chunk steps (8: measured best of 4 / 8 / 12 / 16 / 24 / 32) are requested together and multiplied
// together: an LDS round trip takes ~2000 cycles under this kernel's load, a dependent read -> multiply loop pays it per step.
// The spectrogram never reaches HBM: 2048 + 4 n_mels bytes per frame.
constexpr int kMel32MaxItems = 8;      // per wave
#ifndef SMX_MEL32_CHUNK
#define SMX_MEL32_CHUNK 8
#endif
constexpr int kMel32Chunk = SMX_MEL32_CHUNK;
#ifndef SMX_MEL32_PIPE
#define SMX_MEL32_PIPE 1   // two chunks of operands in flight (0.566 -> 0.551 ms at C3)
#endif
struct Mel32Item {
  int row0, nrows;        // mel rows [row0, row0 + nrows), nrows <= 16 (nrows = 0: no item)
  int k4_begin, k4_count; // bins 4 k4_begin .. 4 (k4_begin + k4_count); k4_count is padded to a multiple of 4 with zero weights
  int a_offset;           // offset (in 64-float rows) of the item's A operands in w
  int last_bin;           // the last spectrum row this item may read (rows beyond hold transposition cells)
  int pad[2];
};
struct Mel32Args {
  const Mel32Item *items; // [8 waves][kMel32MaxItems]
  const float *w;         // [steps][64]: A operand of each MFMA step in lane order
  float *out;             // [lead; n_mels; out_stride]
  int64_t out_stride, out_offset;
  int n_mels;
};

// the wave's items over the finished tile `tile` (rows = bins, 17 floats apart): out[mel][f0 + f] for its mels.
// `iv`: the wave's eight items, one int per lane (item i's field q in lane 8 i + q), read once per kernel: a field is one
// v_readlane away, where a load from the plan in global memory would put its latency into every tile.
template <int CH, int TS, class Acc>
__device__ __forceinline__ void mel32_chunk(const float *ap, const float *tile, int k4, int kk, int f, int last, Acc &acc0, Acc &acc1) {
  float av[CH], bv[CH];
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    av[u] = ap[64 * u];
    const int row = 4 * (k4 + u) + kk;
    bv[u] = tile[(row < last ? row : last) * TS + f];   // rows past the band multiply zero weights, but must be spectrum
  }
#pragma unroll
  for (int u = 0; u < CH; u += 2) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u + 1], bv[u + 1], acc1, 0, 0, 0);
  }
}
// TS: floats per tile row; `tile` points at the first of the 16 frame columns multiplied, `obase` at their outputs
template <int TS = kTileStride>
__device__ __forceinline__ void mel32_items(const Mel32Args &m, int iv, const float *tile, float *obase, int frames_left, int lane) {
  using f32x4m = __attribute__((ext_vector_type(4))) float;
  const int kk = lane >> 4, f = lane & 15;
#pragma unroll 1
  for (int i = 0; i < kMel32MaxItems; ++i) {
    const int nrows = __builtin_amdgcn_readlane(iv, 8 * i + 1);
    if (nrows == 0) break;
    const int row0 = __builtin_amdgcn_readlane(iv, 8 * i);
    const int k4b = __builtin_amdgcn_readlane(iv, 8 * i + 2);
    const int k4n = __builtin_amdgcn_readlane(iv, 8 * i + 3);
    const int last = __builtin_amdgcn_readlane(iv, 8 * i + 5);
    const float *ap = m.w + (int64_t)__builtin_amdgcn_readlane(iv, 8 * i + 4) * 64 + lane;
    f32x4m acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int j = 0;
#if SMX_MEL32_PIPE
    {   // two chunks in flight: the operands of chunk c + 1 are requested before chunk c is multiplied
      constexpr int CH = kMel32Chunk;
      float av[2][CH], bv[2][CH];
      auto request = [&](int slot, int jj) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          av[slot][u] = ap[64 * (jj + u)];
          const int row = 4 * (k4b + jj + u) + kk;
          bv[slot][u] = tile[(row < last ? row : last) * TS + f];
        }
      };
      auto multiply = [&](int slot) {
#pragma unroll
        for (int u = 0; u < CH; u += 2) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[slot][u], bv[slot][u], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[slot][u + 1], bv[slot][u + 1], acc1, 0, 0, 0);
        }
      };
      const int nch = k4n / CH;   // whole chunks
      if (nch > 0) {
        request(0, 0);
#pragma unroll 1
        for (int c = 0; c + 2 <= nch; c += 2) {   // slots alternate with static indices
          request(1, CH * (c + 1));
          multiply(0);
          if (c + 2 < nch) request(0, CH * (c + 2));
          multiply(1);
        }
        if (nch & 1) multiply(0);
        j = nch * CH;
      }
    }
#else
#pragma unroll 1
    for (; j + kMel32Chunk <= k4n; j += kMel32Chunk) mel32_chunk<kMel32Chunk, TS>(ap + 64 * j, tile, k4b + j, kk, f, last, acc0, acc1);
#endif
#pragma unroll 1
    for (; j < k4n; j += 4) mel32_chunk<4, TS>(ap + 64 * j, tile, k4b + j, kk, f, last, acc0, acc1);
    const f32x4m acc = acc0 + acc1;
    if (f < frames_left) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * kk + r;
        if (row < nrows) obase[(int64_t)(row0 + row) * m.out_stride + f] = acc[r];
      }
    }
  }
}

// the same over NCG groups of 16 frame columns at once (the tiles of the 16- / 8-lane kernels hold 32 / 64 frames): an item's
// A operands are requested once per chunk and multiplied into every group; two chunks of CH steps in flight
template <int TS, int NCG, int CH>
__device__ __forceinline__ void mel32_items_multi(const Mel32Args &m, int iv, const float *tile, float *obase, int frames_left, int lane) {
  using f32x4m = __attribute__((ext_vector_type(4))) float;
  const int kk = lane >> 4, f = lane & 15;
#pragma unroll 1
  for (int i = 0; i < kMel32MaxItems; ++i) {
    const int nrows = __builtin_amdgcn_readlane(iv, 8 * i + 1);
    if (nrows == 0) break;
    const int row0 = __builtin_amdgcn_readlane(iv, 8 * i);
    const int k4b = __builtin_amdgcn_readlane(iv, 8 * i + 2);
    const int k4n = __builtin_amdgcn_readlane(iv, 8 * i + 3);
    const int last = __builtin_amdgcn_readlane(iv, 8 * i + 5);
    const float *ap = m.w + (int64_t)__builtin_amdgcn_readlane(iv, 8 * i + 4) * 64 + lane;
    f32x4m acc[NCG];
#pragma unroll
    for (int c = 0; c < NCG; ++c) acc[c] = f32x4m{0.f, 0.f, 0.f, 0.f};
    float av[2][CH], bv[2][CH][NCG];
    auto request = [&](int slot, int jj) {
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        av[slot][u] = ap[64 * (jj + u)];
        const int row = 4 * (k4b + jj + u) + kk;
        const float *bp = tile + (row < last ? row : last) * TS + f;
#pragma unroll
        for (int c = 0; c < NCG; ++c) bv[slot][u][c] = bp[16 * c];
      }
    };
    auto multiply = [&](int slot) {
#pragma unroll
      for (int u = 0; u < CH; ++u)
#pragma unroll
        for (int c = 0; c < NCG; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[slot][u], bv[slot][u][c], acc[c], 0, 0, 0);
    };
    const int nch = k4n / CH;   // k4n is a multiple of 4 and CH divides 4
    request(0, 0);
#pragma unroll 1
    for (int c = 0; c + 2 <= nch; c += 2) {
      request(1, CH * (c + 1));
      multiply(0);
      if (c + 2 < nch) request(0, CH * (c + 2));
      multiply(1);
    }
    if (nch & 1) multiply(0);
#pragma unroll
    for (int c = 0; c < NCG; ++c)
      if (16 * c + f < frames_left) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * kk + r;
          if (row < nrows) obase[(int64_t)(row0 + row) * m.out_stride + 16 * c + f] = acc[c][r];
        }
      }
  }
}

// what the mel kernel does between the stages of a frame pair (see frame32_to_tile / PowerMid32)
template <bool ALIGNED>
struct MelMid32 {
  const FastArgs &a;
  const Mel32Args &m;
  int iv;
  const Lds32 &lds;
  float2 (&raw)[32];
  const float *src;      // the next frames' samples (per lane)
  float *pend_out;       // output origin and frames of the previous tile
  int pend_left;
  int lane, wave, b, it;
  unsigned &pk_drained, &pk_filled;
  template <int I> __device__ __forceinline__ void stamp() const {}
  __device__ __forceinline__ void early() const { pk_drained = peek32(lds.drained + b * kTileStride); }
  __device__ __forceinline__ void before_cells() const {
    lds_wait32(lds.drained + b * kTileStride, 8u * ((unsigned)it >> 1), pk_drained);
  }
  __device__ __forceinline__ void after_transposition_issue() const {
    if (it > 0) pk_filled = peek32(lds.filled + (b ^ 1) * kTileStride);
  }
  __device__ __forceinline__ void after_exchange_issue() const {
    if (it > 0) {
      lds_wait32(lds.filled + (b ^ 1) * kTileStride, 8u * (((unsigned)(it - 1) >> 1) + 1), pk_filled);
      mel32_items(m, iv, lds.tiles + (b ^ 1) * kTile32Floats, pend_out, pend_left, lane);
      lds_signal32(lds.drained + (b ^ 1) * kTileStride, lane);   // behind the item's last LDS read in this wave's order
    }
  }
  __device__ __forceinline__ void postpass_at(int s) const {
    if (s == SMX_P32_LOAD_AT) load_frame32<ALIGNED>(src, lane & 31, raw);
  }
};

template <bool ALIGNED, int PMODE>
__global__ void __launch_bounds__(512) stft2048_mel32_kernel(FastArgs a, Mel32Args m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Lds32 lds = carve_lds32(smem);
  const Lane32 L = setup_lane32(lds, lane, wave);
  fill_tables32(a, lds, tid, 512);
  TileWalk tw;
  tw.init(a, m.out + m.out_offset, (int64_t)m.n_mels * m.out_stride);
  const int ntiles = tw.ntiles > 0 ? tw.ntiles : 0;
  auto frame_ptr = [&](const float *xc, int t) {   // as stft2048_power32_kernel
    const int64_t f0 = (int64_t)t * kFT;
    const int avail = (int)(a.count - f0 < kFT ? a.count - f0 : kFT) - 1;
    const int fi = 2 * wave + L.h;
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
  if (ntiles > 0) load_frame32<ALIGNED>(frame_ptr(tw.xclip, tw.ft), L.l, raw);
  __syncthreads();   // tables and zeroed counters visible: the only workgroup barrier
  float *pend_out = nullptr;
  int pend_left = 0;
  unsigned pk_drained = 0, pk_filled = 0;
  const int iv = reinterpret_cast<const int *>(m.items + wave * kMel32MaxItems)[lane];   // this wave's items (8 x 8 ints)
  for (int it = 0; it < ntiles; ++it) {   // tile `it` of this workgroup lives in buffer it & 1
    const int b = it & 1;
    int ftnext;
    const float *xnext;
    float *onext;
    tw.peek(a, ftnext, xnext, onext);
    const bool more = it + 1 < ntiles;
    const float *src = frame_ptr(more ? xnext : tw.xclip, more ? ftnext : tw.ft);
    const MelMid32<ALIGNED> mid{a, m, iv, lds, raw, src, pend_out, pend_left, lane, wave, b, it, pk_drained, pk_filled};
    frame32_to_tile<PMODE>(a, L, raw, lds.tiles + b * kTile32Floats, mid);
    lds_signal32(lds.filled + b * kTileStride, lane);
    pend_out = tw.oclip + tw.ft * kFT;   // wave-uniform
    const int64_t left = a.count - (int64_t)tw.ft * kFT;
    pend_left = left < kFT ? (int)left : kFT;
    tw.xclip = xnext;
    tw.oclip = onext;
    tw.ft = ftnext;
  }
  if (ntiles > 0) {   // the last tile of this workgroup
    const int b = (ntiles - 1) & 1;
    lds_wait(lds.filled + b * kTileStride, 8u * (((unsigned)(ntiles - 1) >> 1) + 1));
    mel32_items(m, iv, lds.tiles + b * kTile32Floats, pend_out, pend_left, lane);
  }
}
