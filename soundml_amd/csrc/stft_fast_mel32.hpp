// stft2048_mel32_kernel -- the fused audio -> mel spectrogram at fft 2048 on the 32-lane frame pipeline of stft_fast_p32.hpp
// (included by stft_fast.hip after it, inside its anonymous namespace).  Replaces Soundml.mel_spectrogram =
// Mel.apply (Stft.power_spectrum ...), soundml.ml:12-24 + mel.ml:202-231, for banded filterbanks.
//
// The frame code is frame32_to_tile unchanged: every tile of 16 frames x 1025 powers lands in LDS exactly as the power
// kernel leaves it.  Where that kernel reads its share of the previous tile out to HBM, this one multiplies the tile by the
// filterbank; the spectrogram never reaches HBM: 2048 + 4 n_mels bytes per frame.  Two forms of the product (chosen by the
// host: Mel32Args::four):
//  * DENSE (Mel32Item; the form of rounds 1-3, SMX_MEL_DENSE=1, and of plans the banded form does not take):
//    v_mfma_f32_16x16x4_f32 with A = W[16 mels][4 bins] (global memory, pre-arranged in lane order, L2 resident), B = P[4 bins]
//    [16 frames] (one LDS read per lane: tile rows ARE bins in this pipeline), K running over the union band of the item's mels.
//    The work is cut into ITEMS of up to 16 mel rows, each summed over its whole band by ONE wave in ascending bin order, dealt to
//    the 8 waves by length (mel_config::fused32_plan); the operands of two chunks of 8 steps are in flight at a time.
//  * BANDED (Mel4Item, further down; the default): v_mfma_f32_4x4x1_16B_f32, every 4-mel group over its own band, the A
//    operands resident in registers (mel4r_items) or streamed (mel4_items).
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
  int last_bin;           // Mel4Item plans: the last spectrum row (rows beyond hold transposition cells)
  int four;               // host side: 1 items are Mel4Item (the 4 x 4 x 1 product), 2 and the A operands are laid out per wave for residence
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

// ---- the product on v_mfma_f32_4x4x1_16B_f32 (fft 2048) ------------------------------------------------------------------
// A mel filterbank is banded: of the 16 mels of a dense 16 x 16 x 4 step two have a weight at any bin, the other 14 multiply
// zeros (C3: 284 steps of 32 cycles per tile).  The 4 x 4 x 1 form is 16 independent blocks D_b[4][4] += A_b[4] B_b[4]: block
// b = 4 mg + fg is mel GROUP mg (4 consecutive mels) over frames 4 fg .. 4 fg + 3, and every mel group walks its OWN band --
// lane (mg, fg, j) reads B = P[kb_mg + step][4 fg + j] from the tile (a per-lane row base, the same step offsets) and
// A = W[row_mg + (lane & 3)][kb_mg + step] from the plan's table in lane order.  A step is 8 cycles instead of 32 and a
// group's band is 5 half-widths instead of the 17 of a 16-mel union: C3 takes 341 steps of 8 cycles.  A long band is cut in
// 2 or 4 K-parts that sit in the lane groups of ONE item (mode 2: groups a a b b, mode 4: a a a a) and are added at the end
// in a fixed order, (p0 + p1) + (p2 + p3), across the lane groups: one wave still owns a (mel, frame) value, the plan depends
// on the configuration alone, so a clip's values do not depend on its batch (mel_props.ml:136-155).
struct Mel4Item {
  int rows;        // first mel row of lane group mg in byte mg
  int nrows;       // rows of lane group mg in byte mg (0: the group is idle)
  int kb[4];       // first bin of lane group mg
  int steps_mode;  // steps (a multiple of 4; 0: no item) | mode << 16 (1: four groups, 2: two groups x 2 K-parts, 4: one group x 4)
  int a_offset;    // offset (in 64-float rows) of the item's A operands in w
};
static_assert(sizeof(Mel4Item) == sizeof(Mel32Item), "both plans are read as 8 ints per item");

template <int TS = kTileStride>
__device__ __forceinline__ void mel4_items(const Mel32Args &m, int iv, const float *tile, float *obase, int frames_left, int lane) {
  using f32x4m = __attribute__((ext_vector_type(4))) float;
  const int mg = lane >> 4, f = lane & 15;
#pragma unroll 1
  for (int i = 0; i < kMel32MaxItems; ++i) {
    const int sm = __builtin_amdgcn_readlane(iv, 8 * i + 6);
    const int nsteps = sm & 0xffff, mode = sm >> 16;
    if (nsteps == 0) break;
    const int rows = __builtin_amdgcn_readlane(iv, 8 * i), nrw = __builtin_amdgcn_readlane(iv, 8 * i + 1);
    const int k0 = __builtin_amdgcn_readlane(iv, 8 * i + 2), k1 = __builtin_amdgcn_readlane(iv, 8 * i + 3);
    const int k2 = __builtin_amdgcn_readlane(iv, 8 * i + 4), k3 = __builtin_amdgcn_readlane(iv, 8 * i + 5);
    const int kbl = mg == 0 ? k0 : mg == 1 ? k1 : mg == 2 ? k2 : k3;
    const float *ap = m.w + (int64_t)__builtin_amdgcn_readlane(iv, 8 * i + 7) * 64 + lane;
    const float *bp = tile + kbl * TS + f;
    const int room = m.last_bin - kbl;   // steps past it would leave the spectrum: they multiply zero weights, but must read spectrum
    f32x4m acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    constexpr int CH = kMel32Chunk;
    float av[2][CH], bv[2][CH];
    auto request = [&](int slot, int jj, auto n) {
      constexpr int N = decltype(n)::value;
#pragma unroll
      for (int u = 0; u < N; ++u) {
        av[slot][u] = ap[64 * (jj + u)];
        const int st = jj + u;
        bv[slot][u] = bp[(st < room ? st : room) * TS];
      }
    };
    auto multiply = [&](int slot, auto n) {
      constexpr int N = decltype(n)::value;
#pragma unroll
      for (int u = 0; u < N; u += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[slot][u], bv[slot][u], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[slot][u + 1], bv[slot][u + 1], acc1, 0, 0, 0);
      }
    };
    const std::integral_constant<int, CH> whole{};
    const std::integral_constant<int, 4> tail{};
    const int nch = nsteps / CH;
    int j = 0;
    if (nch > 0) {   // two chunks in flight, as mel32_items
      request(0, 0, whole);
#pragma unroll 1
      for (int c = 0; c + 2 <= nch; c += 2) {
        request(1, CH * (c + 1), whole);
        multiply(0, whole);
        if (c + 2 < nch) request(0, CH * (c + 2), whole);
        multiply(1, whole);
      }
      if (nch & 1) multiply(0, whole);
      j = nch * CH;
    }
#pragma unroll 1
    for (; j < nsteps; j += 4) {
      request(0, j, tail);
      multiply(0, tail);
    }
    f32x4m acc = acc0 + acc1;
    if (mode >= 2) {   // wave-uniform: add the K-parts across the lane groups, (p0 + p1) + (p2 + p3)
      const int from16 = ((lane + 16) & 63) << 2, from32 = ((lane + 32) & 63) << 2;
      // (element by element through scalars: __builtin_bit_cast of acc[r] itself made hipcc permute element 0 four times)
      auto from = [&](int addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); };
      const float a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
      float s0 = a0 + from(from16, a0), s1 = a1 + from(from16, a1), s2 = a2 + from(from16, a2), s3 = a3 + from(from16, a3);
      if (mode == 4) {
        s0 = s0 + from(from32, s0);
        s1 = s1 + from(from32, s1);
        s2 = s2 + from(from32, s2);
        s3 = s3 + from(from32, s3);
      }
      acc = f32x4m{s0, s1, s2, s3};
    }
    const bool owner = mode == 1 || (mode == 2 ? (mg & 1) == 0 : mg == 0);
    const int row0 = (int)(((unsigned)rows >> (8 * mg)) & 255u), nr = (int)(((unsigned)nrw >> (8 * mg)) & 255u);
    if (owner && f < frames_left) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r < nr) obase[(int64_t)(row0 + r) * m.out_stride + f] = acc[r];
    }
  }
}

// The same product with the A operands RESIDENT: a wave's items take at most 8 chunks of 8 steps (C3: 7), i.e. 64 values per
// lane, and the kernel has that many registers to spare -- they are loaded once per kernel and the tile loop reads nothing
// but the tile.  (With the operands streamed from L2, two chunks in flight, a wave's ~46 steps took ~4 000 cycles per tile, all
// of it L2 latency; the MFMA work itself is 46 x 8 cycles.)  Register indices must be static: the chunk loop is written out,
// an item is a whole number of chunks (zero weights pad it), and the item that a chunk belongs to is scalar state.
constexpr int kMel4rChunks = 8;
// TS: floats per tile row; NCG: groups of 16 frame columns multiplied at once (the tiles of the 16- / 8-lane kernels hold 32 / 64
// frames: an operand register serves every group); CHUNKS: operand registers / 8; SUB: steps per request (B values in flight:
// 2 x SUB x NCG registers)
// Where a wave's requests begin, worked out ONCE per kernel (the plan is the same for every tile): per lane the tile offset (in floats)
// of the first B row of each of the 8 sub-chunks, two to a register, and a scalar mask of the sub-chunks whose rows pass the last
// spectrum row for some lane group (those take the clamped path).  With it a request is one address register and 4 ds_read2 with
// immediate offsets; without it (the lanes kernels) the band of the lane's group is re-derived from the plan at every item.
struct Mel4rSlots {
  unsigned boff[4];
  unsigned slow_mask;
};
template <int TS = kTileStride, int CHUNKS = kMel4rChunks, int SUB = 8>
__device__ __forceinline__ Mel4rSlots mel4r_slots(const Mel32Args &m, int iv, int lane) {
  constexpr int SLOTS = 8 * CHUNKS / SUB;
  static_assert(SLOTS <= 8, "two 16-bit offsets to each of four registers");
  Mel4rSlots p{{0u, 0u, 0u, 0u}, 0u};
  const int mg = lane >> 4, f = lane & 15;
  int item = 0, st = 0, steps = 0, kbl = 0, room_min = 0;
  auto open = [&]() {
    steps = __builtin_amdgcn_readlane(iv, 8 * item + 6) & 0xffff;
    const int k0 = __builtin_amdgcn_readlane(iv, 8 * item + 2), k1 = __builtin_amdgcn_readlane(iv, 8 * item + 3);
    const int k2 = __builtin_amdgcn_readlane(iv, 8 * item + 4), k3 = __builtin_amdgcn_readlane(iv, 8 * item + 5);
    kbl = k0;
    kbl = mg == 1 ? k1 : kbl;
    kbl = mg == 2 ? k2 : kbl;
    kbl = mg == 3 ? k3 : kbl;
    const int k01 = k0 > k1 ? k0 : k1, k23 = k2 > k3 ? k2 : k3;
    room_min = m.last_bin - (k01 > k23 ? k01 : k23);
    st = 0;
  };
  open();
#pragma unroll
  for (int c = 0; c < SLOTS; ++c) {
    if (steps != 0) {
      const unsigned off = (unsigned)((kbl + st) * TS + f);
      p.boff[c >> 1] |= (c & 1) ? off << 16 : off;
      if (st + SUB - 1 > room_min) p.slow_mask |= 1u << c;
      if (st + SUB >= steps) {
        ++item;
        if (item < kMel32MaxItems) open();
        else steps = 0;
      } else {
        st += SUB;
      }
    }
  }
  return p;
}
template <int TS = kTileStride, int NCG = 1, int CHUNKS = kMel4rChunks, int SUB = 8, bool PRE = false>
__device__ __forceinline__ void mel4r_items(const Mel32Args &m, int iv, const float (&areg)[8 * CHUNKS], const float *tile, float *obase,
                                            int frames_left, int lane, const Mel4rSlots &pre = Mel4rSlots{}) {
  using f32x4m = __attribute__((ext_vector_type(4))) float;
  constexpr int SLOTS = 8 * CHUNKS / SUB;
  constexpr int NA = NCG == 1 ? 2 : NCG;   // accumulators: two alternating ones for a single column group, one per group otherwise
  const int mg = lane >> 4, f = lane & 15;
  // scalar state of the sub-chunk being REQUESTED (one ahead of the one being multiplied)
  int item = 0, st = 0, steps = 0, mode = 0, rows = 0, nrw = 0;
  // (the tile through an LDS-address-space pointer: 32-bit address arithmetic and the rows of a request as IMMEDIATE offsets of one
  // base register; through the generic pointer every read cost a v_min and a 64-bit v_mad beside the ds_read itself)
  using lds_f = const __attribute__((address_space(3))) float;
  lds_f *const tile3 = (lds_f *)tile;
  // (the row pitch of the result as a scalar of its own: read out of the argument block at every use, the compiler reloaded the
  // block's eight spilled registers around each of an item's four stores)
  int64_t out_pitch = m.out_stride;
  asm volatile("" : "+s"(out_pitch));
  lds_f *bp = tile3;
  int room = 0, room_min = 0;
  auto open_item = [&]() {
    const int sm = __builtin_amdgcn_readlane(iv, 8 * item + 6);
    steps = sm & 0xffff;
    mode = sm >> 16;
    rows = __builtin_amdgcn_readlane(iv, 8 * item);
    nrw = __builtin_amdgcn_readlane(iv, 8 * item + 1);
    st = 0;
    if constexpr (PRE) return;   // (the band of the lane's group is in pre.boff; the clamped path derives it itself)
    const int k0 = __builtin_amdgcn_readlane(iv, 8 * item + 2), k1 = __builtin_amdgcn_readlane(iv, 8 * item + 3);
    const int k2 = __builtin_amdgcn_readlane(iv, 8 * item + 4), k3 = __builtin_amdgcn_readlane(iv, 8 * item + 5);
    int kbl = k0;   // (three selects: the nested form compiled to branches on the execution mask)
    kbl = mg == 1 ? k1 : kbl;
    kbl = mg == 2 ? k2 : kbl;
    kbl = mg == 3 ? k3 : kbl;
    bp = tile3 + kbl * TS + f;
    room = m.last_bin - kbl;
    const int k01 = k0 > k1 ? k0 : k1, k23 = k2 > k3 ? k2 : k3;
    room_min = m.last_bin - (k01 > k23 ? k01 : k23);   // (scalar: the group that comes closest to the last spectrum row)
    st = 0;
  };
  open_item();
  if (steps == 0) return;
  float bv[2][SUB][NCG];
  auto band_of_item = [&]() {   // (PRE, clamped path only)
    const int k0 = __builtin_amdgcn_readlane(iv, 8 * item + 2), k1 = __builtin_amdgcn_readlane(iv, 8 * item + 3);
    const int k2 = __builtin_amdgcn_readlane(iv, 8 * item + 4), k3 = __builtin_amdgcn_readlane(iv, 8 * item + 5);
    int kbl = k0;
    kbl = mg == 1 ? k1 : kbl;
    kbl = mg == 2 ? k2 : kbl;
    kbl = mg == 3 ? k3 : kbl;
    bp = tile3 + kbl * TS + f;
    room = m.last_bin - kbl;
  };
  auto request = [&](int slot, int c) {   // c: the sub-chunk requested (a constant once the loop is unrolled)
    if constexpr (PRE) {
      if (((pre.slow_mask >> c) & 1u) == 0u) {
        lds_f *row0 = tile3 + ((c & 1) ? pre.boff[c >> 1] >> 16 : pre.boff[c >> 1] & 0xffffu);
#pragma unroll
        for (int u = 0; u < SUB; ++u)
#pragma unroll
          for (int g = 0; g < NCG; ++g) bv[slot][u][g] = row0[u * TS + 16 * g];
        return;
      }
      band_of_item();
    }
    if (!PRE && st + SUB - 1 <= room_min) {   // wave-uniform: no group's rows of this request pass the last spectrum row (all but a filterbank's top items)
      lds_f *row0 = bp + st * TS;
#pragma unroll
      for (int u = 0; u < SUB; ++u)
#pragma unroll
        for (int g = 0; g < NCG; ++g) bv[slot][u][g] = row0[u * TS + 16 * g];
      return;
    }
#pragma unroll
    for (int u = 0; u < SUB; ++u) {
      const int q = st + u;
      lds_f *row = bp + (q < room ? q : room) * TS;
#pragma unroll
      for (int g = 0; g < NCG; ++g) bv[slot][u][g] = row[16 * g];
    }
  };
  f32x4m acc[NA];
#pragma unroll
  for (int g = 0; g < NA; ++g) acc[g] = f32x4m{0.f, 0.f, 0.f, 0.f};
  request(0, 0);
#pragma unroll
  for (int c = 0; c < SLOTS; ++c) {
    // the sub-chunk being multiplied: c.  Its item's output description, before the state moves on
    const int c_mode = mode, c_rows = rows, c_nrw = nrw;
    const bool c_last = st + SUB >= steps;
    bool more = false;
    if (c + 1 < SLOTS) {   // advance the request state to sub-chunk c + 1 and request it
      if (c_last) {
        ++item;
        if (item < kMel32MaxItems) {
          open_item();
          more = steps != 0;
        }
      } else {
        st += SUB;
        more = true;
      }
      if (more) request((c + 1) & 1, c + 1);
    }
#pragma unroll
    for (int u = 0; u < SUB; ++u)
#pragma unroll
      for (int g = 0; g < NCG; ++g) {
        const int t = NCG == 1 ? (u & 1) : g;
        acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(areg[SUB * c + u], bv[c & 1][u][g], acc[t], 0, 0, 0);
      }
    if (c_last) {
      const bool owner = (mg & (c_mode - 1)) == 0;   // mode 1: every lane group its own rows; 2: groups 0 and 2; 4: group 0
      const int row0 = (int)(((unsigned)c_rows >> (8 * mg)) & 255u), nr = (int)(((unsigned)c_nrw >> (8 * mg)) & 255u);
#pragma unroll
      for (int g = 0; g < NCG; ++g) {
        f32x4m sum = NCG == 1 ? acc[0] + acc[1] : acc[g];
        if (c_mode >= 2) {   // wave-uniform: add the K-parts across the lane groups, (p0 + p1) + (p2 + p3)
          const int from16 = ((lane + 16) & 63) << 2, from32 = ((lane + 32) & 63) << 2;
          auto from = [&](int addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); };
          const float a0 = sum[0], a1 = sum[1], a2 = sum[2], a3 = sum[3];
          // (the four permutes of a level issued together, then the four additions: written value by value the compiler waited for
          // every permute before it issued the next one -- eight LDS round trips in a row at an item's end instead of two)
          const float t0 = from(from16, a0), t1 = from(from16, a1), t2 = from(from16, a2), t3 = from(from16, a3);
          SMX_FENCE();
          float s0 = a0 + t0, s1 = a1 + t1, s2 = a2 + t2, s3 = a3 + t3;
          if (c_mode == 4) {
            const float u0 = from(from32, s0), u1 = from(from32, s1), u2 = from(from32, s2), u3 = from(from32, s3);
            SMX_FENCE();
            s0 = s0 + u0;
            s1 = s1 + u1;
            s2 = s2 + u2;
            s3 = s3 + u3;
          }
          sum = f32x4m{s0, s1, s2, s3};
        }
        if (owner && 16 * g + f < frames_left) {   // (one 64-bit row address, the other rows a pitch further each)
          float *po = obase + (int64_t)row0 * out_pitch + (16 * g + f);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (r < nr) po[0] = sum[r];   // (an idle lane group of an item has no rows)
            po += out_pitch;
          }
        }
      }
#pragma unroll
      for (int g = 0; g < NA; ++g) acc[g] = f32x4m{0.f, 0.f, 0.f, 0.f};
    }
    if (!more) break;
  }
}

// what the mel kernel does between the stages of a frame pair (see frame32_to_tile / PowerMid32)
template <bool ALIGNED, int FOUR>
struct MelMid32 {
  const FastArgs &a;
  const Mel32Args &m;
  const float (&areg)[8 * kMel4rChunks];
  const Mel4rSlots &slots;
  int iv;
  const Lds32 &lds;
  float2 (&raw)[32];
  const float *src;      // the next frames' samples (per lane)
  const float *src_clip; // ... their clip and whether their tile holds a frame that reaches past the signal (see PowerMid32)
  bool src_border;
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
      if constexpr (FOUR == 2) mel4r_items<kTileStride, 1, kMel4rChunks, 8, true>(m, iv, areg, lds.tiles + (b ^ 1) * kTile32Floats, pend_out, pend_left, lane, slots);
      else if constexpr (FOUR == 1) mel4_items(m, iv, lds.tiles + (b ^ 1) * kTile32Floats, pend_out, pend_left, lane);
      else mel32_items(m, iv, lds.tiles + (b ^ 1) * kTile32Floats, pend_out, pend_left, lane);
      lds_signal32(lds.drained + (b ^ 1) * kTileStride, lane);   // behind the item's last LDS read in this wave's order
    }
  }
  __device__ __forceinline__ void postpass_at(int s) const {
    if (s == SMX_P32_LOAD_AT) load_frame32<ALIGNED>(src_border ? src_clip : src, lane & 31, raw);
    if (s == 15 && src_border) load_frame32_padded(a, src_clip, (int)(src - src_clip), lane & 31, raw);   // (wave-uniform; see PowerMid32::load_next)
  }
};

// FOUR: 0 the dense 16 x 16 x 4 product (Mel32Item plan), 1 the banded 4 x 4 x 1 one with streamed operands, 2 with resident ones
template <bool ALIGNED, int PMODE, int FOUR = 0>
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
  __syncthreads();   // tables and zeroed counters visible: the only workgroup barrier
  float *pend_out = nullptr;
  int pend_left = 0;
  unsigned pk_drained = 0, pk_filled = 0;
  const int iv = reinterpret_cast<const int *>(m.items + wave * kMel32MaxItems)[lane];   // this wave's items (8 x 8 ints)
  float areg[8 * kMel4rChunks];
  if constexpr (FOUR == 2) {   // the wave's A operands: [wave][step][lane], loaded once
#pragma unroll
    for (int q = 0; q < 8 * kMel4rChunks; ++q) areg[q] = m.w[(wave * 8 * kMel4rChunks + q) * 64 + lane];
  }
  Mel4rSlots slots{};
  if constexpr (FOUR == 2) slots = mel4r_slots(m, iv, lane);
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
    const MelMid32<ALIGNED, FOUR> mid{a, m, areg, slots, iv, lds, raw, src, src_clip, src_border, pend_out, pend_left, lane, wave, b, it, pk_drained, pk_filled};
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
    if constexpr (FOUR == 2) mel4r_items<kTileStride, 1, kMel4rChunks, 8, true>(m, iv, areg, lds.tiles + b * kTile32Floats, pend_out, pend_left, lane, slots);
    else if constexpr (FOUR == 1) mel4_items(m, iv, lds.tiles + b * kTile32Floats, pend_out, pend_left, lane);
    else mel32_items(m, iv, lds.tiles + b * kTile32Floats, pend_out, pend_left, lane);
  }
}
