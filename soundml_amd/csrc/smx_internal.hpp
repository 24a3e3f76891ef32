// Internal declarations shared by the host layer, the HIP kernels and the C ABI.
// Nothing here is part of the public boundary (include/soundml_amd.h).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/soundml_amd.h"

namespace smx {

// ---- errors -------------------------------------------------------------
// Invalid_argument of the reference (user-facing precondition, message verbatim)
struct InvalidArgument : std::runtime_error {
  using std::runtime_error::runtime_error;
};
// Failure of the reference (bookkeeping / runtime error)
struct Failure : std::runtime_error {
  using std::runtime_error::runtime_error;
};

std::string format(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
// OCaml's "%g" differs from C's only in corner cases we never print; keep C's.

void set_last_error(const std::string &message);

#define SMX_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t err__ = (expr);                                                           \
    if (err__ != hipSuccess)                                                             \
      throw ::smx::Failure(::smx::format("%s: HIP error %d (%s) at %s:%d", #expr,        \
                                         (int)err__, hipGetErrorString(err__), __FILE__, \
                                         __LINE__));                                     \
  } while (0)

// ---- environment switches ------------------------------------------------------------------------------------------
// The shipped library reads thirteen variables, each once per use and parsed ONE way (env_flag: unset -> -1, "0" / "" /
// "false" / "off" -> 0, anything else -> 1); none alters results except by selecting another kernel of the same contract:
//   SMX_DISABLE_FAST   the generic kernels instead of the hand-laid ones (tests: two implementations of one contract)
//   SMX_POWER_SKEW=0   fft-2048 / fft-1024 power spectrogram: the plain per-tile flush instead of whole aligned 64-byte blocks / 128-byte lines (tests, A/B timing)
//   SMX_BORDER_INLINE=0 fft 2048 / 1024 / 512 (power / complex spectrogram, fused mel): the border frames in an epilogue / gathered strips instead of the tile sequence (tests: same values)
//   SMX_COMPLEX_SKEW=0 fft-2048 Stft.transform: the plain per-tile flush instead of whole aligned 128-byte lines (tests, A/B timing)
//   SMX_INVERT_PIPELINE=0 Stft.invert at fft 2048 / hop 512: the one-tile-per-workgroup kernel of rounds 1-4 instead of the persistent pipeline (tests, A/B timing)
//   SMX_GL_FRAME_MAJOR=0 Griffin-Lim at fft 2048 / hop 512: its rebuilt spectra in the reference layout [bin][frame] instead of frame-major (tests: bit-identical; A/B timing)
//   SMX_WIDE_PIPELINE=0 float64 interior at fft 2048: the one-tile-per-workgroup kernel instead of the persistent one (tests: bit-identical)
//   SMX_MEL_DENSE=1    fused mel at fft 2048: the dense 16 x 16 x 4 product instead of the banded 4 x 4 x 1 one (tests, A/B timing)
//   SMX_MIXED_OFF      chirp-z instead of the mixed-radix kernels (tests: the two agree)
//   SMX_HOST_TRACE     print where a host-pointer call's time goes
//   SMX_HOST_PIPELINE=0 host-pointer STFT calls: upload, kernels, download one after the other instead of overlapped clip units
//   SMX_COPY_THREADS / SMX_COPY_PLAIN   host <-> device staging of the host-pointer entry points
// Every other switch (kernel selection for timing, ablations) exists in diagnostic builds only (make DIAG=1): diag_flag /
// diag_int answer "unset" in the shipped build, so what a launcher picks there is a function of the call alone.
int env_flag(const char *name);
long env_int(const char *name, long fallback);
#ifdef SMX_DIAG
inline int diag_flag(const char *name) { return env_flag(name); }
inline long diag_int(const char *name, long fallback) { return env_int(name, fallback); }
#else
inline int diag_flag(const char *) { return -1; }
inline long diag_int(const char *, long fallback) { return fallback; }
#endif

// Every kernel launch of the library goes through SMX_LAUNCH, which counts it: smx_debug_kernel_launches() lets a
// caller assert how many launches one entry point costs (bench.py: one step of the hot path = ONE launch, so the
// HIP events around a step are that kernel's duration).
extern std::atomic<unsigned long long> g_kernel_launches;
#define SMX_LAUNCH(...)                                                       \
  do {                                                                        \
    ::smx::g_kernel_launches.fetch_add(1ull, std::memory_order_relaxed);      \
    hipLaunchKernelGGL(__VA_ARGS__);                                          \
  } while (0)

// ---- device-resident tables owned by a config, one set per HIP device ------
struct DeviceBuffer {
  void *ptr = nullptr;
  size_t bytes = 0;
};

struct StftTables {
  // analysis window, fft_size entries
  double *window_f64 = nullptr;
  float *window_f32 = nullptr;
  // generic pow2 kernel: exp(-2 pi i j / N), j < N/2 ; direct DFT kernel: j < N
  double2 *twiddle_f64 = nullptr;
  float2 *twiddle_f32 = nullptr;
  int64_t twiddle_len = 0;
  // fast kernels (N = 2048 family): half-scaled window + split twiddle tables
  float *fast_window = nullptr;    // 0.5 * window, f32
  float2 *fast_w_m = nullptr;      // exp(-2 pi i j / M), j < M   (M = N/2)
  double2 *fast_w_m_f64 = nullptr; // the same in float64 (float64-interior Stockham form, fft 512 .. 4096)
  float2 *fast_w_n = nullptr;      // exp(-2 pi i k / N), k <= M
  // chirp-z (Bluestein) path of the generic float32 kernels for sizes that are not powers of two: an N-point DFT
  // as one circular convolution of length blu_m = 2^blu_log2m >= 2 N - 1
  float2 *blu_chirp = nullptr;     // exp(-i pi n^2 / N) * window[n], n < N   (window folded in)
  float2 *blu_post = nullptr;      // exp(-i pi k^2 / N), k < N
  float2 *blu_filter = nullptr;    // FFT_M of exp(+i pi m^2 / N) (wrapped), times 1/M, natural order
  float2 *blu_tw = nullptr;        // exp(-2 pi i j / M), j < M/2
  int blu_log2m = 0;
  // even sizes: chirp-z of length L = N/2 over (even, odd) sample pairs + the real-input post-pass
  float2 *blu2_chirp = nullptr;    // exp(-i pi n^2 / L), n < L (input chirp and post factor)
  float2 *blu2_filter = nullptr;   // FFT_M2 of exp(+i pi m^2 / L) (wrapped), times 1/M2
  float2 *blu2_tw = nullptr;       // exp(-2 pi i j / M2), j < M2/2
  float *blu2_window = nullptr;    // 0.5 * window, N entries
  int blu2_log2m = 0;
  // even sizes with N / 2 = 2^a 3^b 5^c <= 1024, not a power of two: the mixed-radix kernel's plan (stft_mixed_power16_kernel)
  float2 *mixed_tw = nullptr;      // exp(-2 pi i j / L), j < L = N / 2
  double2 *mixed_tw_f64 = nullptr; // the same in float64 (the float64 interior)
  int mixed_npass = 0;
  int mixed_full = 0;              // odd N: the plan is for a complex transform of N points (no half-size trick)
  int mixed_radix[10] = {};
  float2 *fast_synth_window = nullptr;   // (w[2j], -w[2j+1]) / (2M): synthesis window of the fast inverse kernel
  double2 *fast_synth_window_f64 = nullptr;   // the same in float64 (fft 512 .. 4096)
};

}  // namespace smx

namespace smx {
// least-squares synthesis envelope of `count` frames (stft.ml:836-889) on the device: head | period | tail | 1.0
struct EnvelopeTable {
  double *dev = nullptr;
  // keeps the device copy alive: the cache and every caller that took a descriptor share it, so an eviction by another
  // thread cannot free a table a launch is about to read; the last holder's hipFree waits for the device, i.e. for every
  // launch enqueued while a holder existed
  std::shared_ptr<void> owner;
  size_t head = 0, period = 0, tail = 0;   // doubles in each piece
  int64_t head_n = 0, stop = 0;
  uint64_t serial = 0;   // insertion order in the owning config's cache
};
}  // namespace smx

// ---- opaque handle bodies (C ABI names) ----------------------------------------
struct smx_stft_config {
  int64_t fft_size = 0, win_length = 0, hop = 0;
  int alignment = SMX_ALIGN_CENTERED, pad = SMX_PAD_REFLECT, scale = SMX_SCALE_NONE;
  double pad_value = 0.0;
  int window_kind = SMX_WINDOW_HANN;
  std::vector<double> analysis_window;  // fft_size doubles (stft.ml:57-59)

  int64_t bins() const { return fft_size / 2 + 1; }
  int64_t left_width() const;   // stft.ml:132-140
  int64_t right_width() const;  // stft.ml:141-142
  int64_t frames(int64_t n) const;

  // lazily built device tables, keyed by HIP device ordinal; internally locked
  // (SURVEY 8b "Threading": a plan cache must be per-handle or locked).
  const smx::StftTables &tables() const;
  // the envelope of a `count`-frame synthesis, built once per (device, count) on the host in float64 (the
  // reference's summation order) and kept: repeated inversions of one geometry (Griffin-Lim) upload nothing
  smx::EnvelopeTable envelope(int64_t count) const;   // tables.cpp; by value: the cache may evict the entry later
  ~smx_stft_config();

 private:
  mutable std::mutex mutex_;
  mutable std::map<int, smx::StftTables> tables_;
  mutable std::map<std::pair<int, int64_t>, smx::EnvelopeTable> envelopes_;
  mutable uint64_t envelope_serial_ = 0;
};

namespace smx {
struct MelFusedPlan {      // per device; built lazily by stft_fast.hip
  void *items = nullptr;   // Mel32Item[8][8]
  float *w_mfma = nullptr; // MFMA A operands in lane order
  int state = 0;           // 0 not built, 1 usable, -1 this configuration is not eligible
  int resident = 0;        // fused4 plans: w_mfma is [wave][64 steps][64 lanes], a wave's items in order (the kernel keeps them in registers)
};
}  // namespace smx

struct smx_mel_config {
  double f_min = 0, f_max = 0;
  int scale = SMX_MEL_SLANEY, norm = SMX_NORM_SLANEY;
  int64_t n_mels = 0, sample_rate = 0, fft_size = 0;
  std::vector<double> weights;  // [n_mels; bins] float64 (mel.ml:31-33)
  int64_t bins() const { return fft_size / 2 + 1; }

  struct Tables {
    double *w_f64 = nullptr;   // [n_mels; bins]
    float *w_f32 = nullptr;    // [n_mels_pad; k_pad] zero padded for the MFMA kernel
    int64_t n_mels_pad = 0, k_pad = 0;
    // banded form: first/last non-zero bin per mel row
    int *band_lo = nullptr, *band_hi = nullptr;
    // the same per tile of 16 rows (rows without weights do not count; an empty tile is lo = hi = 0)
    int *tile_lo = nullptr, *tile_hi = nullptr;
    float *w_tile = nullptr;   // [n_mels_pad / 16][k_pad / 4][64]: w_f32 in MFMA A-operand order (16x16x4)
    float *w_block = nullptr;  // [n_mels_pad / 32][k_pad / 2][64]: the same for 32x32x2 (Mel.apply)
    int *block_lo = nullptr, *block_hi = nullptr;   // band per 32-row block
  };
  const Tables &tables() const;
  const smx::MelFusedPlan &fused32_plan() const; // stft_fast.hip: items of the 32-lane mel kernel (items = Mel32Item[8][8])
  const smx::MelFusedPlan &fused4_plan() const;  // stft_fast.hip: the same product on 4 x 4 x 1 blocks (items = Mel4Item[8][8]; fft 2048)
  ~smx_mel_config();

 private:
  mutable std::mutex mutex_;
  mutable std::map<int, Tables> tables_;
  mutable std::map<int, smx::MelFusedPlan> fused32_, fused4_;
};

// Chroma.Config.t (chroma.ml:95-107): the [n_chroma; bins] projection matrix, float64, built once on the host
struct smx_chroma_config {
  int64_t n_chroma = 12, sample_rate = 0, fft_size = 0;
  double tuning = 0.0, ctroct = 5.0, octwidth = 2.0;
  bool has_octwidth = true, base_c = true;
  std::vector<double> weights;   // [n_chroma; bins]
  int64_t bins() const { return fft_size / 2 + 1; }
  const double *device_weights() const;   // tables.cpp: transposed [bins; n_chroma] on the current device
  ~smx_chroma_config();

 private:
  mutable std::mutex mutex_;
  mutable std::map<int, double *> tables_;
};

namespace smx {

// ---- host <-> device transfers of the host-pointer entry points (transfer.cpp) ----
// Blocking; large copies are staged through pinned buffers with the DMA and a few host threads overlapped.
void copy_to_device(void *d_dst, const void *src, size_t bytes);
void copy_to_host(void *dst, const void *d_src, size_t bytes);
// page-locked result arrays (smx_host_alloc / smx_host_free): the DMA engine reads / writes them directly, no staging
void *host_alloc(size_t bytes);
void host_free(void *p);
bool host_is_pinned(const void *p, size_t bytes);   // [p, p + bytes) lies inside a live block of host_alloc
bool staging_available();                            // a pipelined call can stage both directions on the current device
void set_transfer_share(int parts);                  // this thread's transfers take 1 / parts of the copying threads (shards of a device-list call)
void staging_peak(int *up, int *down, bool reset);   // most staged transfers ever in flight at once, per direction (tests)
int device_cu_count();                               // compute units of the CURRENT device (cached per device, thread-safe; tables.cpp)
// transfer.cpp: a host call cut into units of clips whose upload, kernels and download overlap (three host threads, HIP events)
void pipelined_host_call(const void *src, size_t in_clip_bytes, void *dst, size_t out_clip_bytes, int64_t clips, int64_t unit,
                         void *d_in, void *d_out, const std::function<void(int64_t, int64_t, hipStream_t)> &launch);

// ---- host logic (host_config.cpp) ----------------------------------------------
void window_make(int kind, bool periodic, int64_t n, double *out);             // window.ml:374-405
void window_make_param(int kind, double param, bool periodic, int64_t n, double *out);   // Kaiser / Gaussian / Tukey too
bool window_cola(int kind, double param, int64_t length, int64_t hop);          // window.ml:407-434
smx_stft_config *stft_config_create(int64_t fft_size, int64_t win_length, int64_t hop,
                                    int alignment, int pad, double pad_value, int scale,
                                    int window_kind, const double *custom_window);
int64_t stft_first_complete(const smx_stft_config &c);                           // stft.ml:225-227
int64_t stft_last_complete(const smx_stft_config &c, int64_t n);                 // stft.ml:229-235
// source sample for signal-relative position q (may be <0 or >=n); -1 = constant pad
int64_t source_index(const smx_stft_config &c, int64_t n, int64_t q);            // stft.ml:300-338
smx_mel_config *mel_config_create(int64_t n_mels, int64_t sample_rate, int64_t fft_size,
                                  double f_min, bool has_f_max, double f_max, int scale, int norm);
smx_chroma_config *chroma_config_create(int64_t n_chroma, double tuning, double ctroct, bool has_octwidth,
                                        double octwidth, bool base_c, int64_t sample_rate, int64_t fft_size);
bool stft_nola(const smx_stft_config &c);                                        // stft.ml:731-743
int64_t stft_output_length(const smx_stft_config &c, int64_t frames);            // stft.ml:792-796
// envelope over the span of `frames` frames as three pieces: positions [0, head) and [stop, span) summed tap by
// tap, one period of the folded squared window for [head, stop) (stft.ml:836-889); all guarded against 0
void stft_envelope(const smx_stft_config &c, int64_t frames, std::vector<double> &head, std::vector<double> &period,
                   std::vector<double> &tail, int64_t &head_n, int64_t &stop);
double hz_to_mel(double f, int scale);                                           // convert.ml:80-90
double mel_to_hz(double m, int scale);                                           // convert.ml:92-102
double kaiser_beta(double att);                                                  // resample.ml:105-109
double bessel_i0(double x);                                                      // resample.ml:128-139
void design_lowpass(int64_t taps, double fc, double beta, double *h);

// ---- kernel launchers ----------------------------------------------------------
enum OutMode { OUT_COMPLEX = 0, OUT_POWER = 1 };

struct StftJob {
  const smx_stft_config *cfg = nullptr;
  const void *x = nullptr;       // device, [lead; n] with x_stride
  int in_bytes = 4;              // 4 = float32 audio, 8 = float64
  int interior = SMX_INTERIOR_F32;
  int64_t lead = 0, n = 0, x_stride = 0;
  // frame p covers padded positions [p*hop, p*hop + fft), padded q <-> source q - left
  int64_t left = 0;
  int pad = SMX_PAD_REFLECT;     // applied to positions outside [0, n)
  double pad_value = 0.0;
  int64_t p0 = 0, count = 0;     // frames [p0, p0 + count)
  OutMode mode = OUT_POWER;
  double power = 2.0;
  void *out = nullptr;           // device, [lead; bins; out_stride] (+ out_offset frames)
  int64_t out_stride = 0, out_offset = 0;
  hipStream_t stream = nullptr;
};

void launch_stft(const StftJob &job);             // dispatch: fast path or generic
void launch_stft_generic(const StftJob &job);     // stft_generic.hip
bool launch_stft_fast(const StftJob &job);        // stft_fast.hip; false = not eligible
// stft_fast.hip: the complex spectrum frame-major (Griffin-Lim's own layout: out[clip][frame][bin], rows of pitch_floats floats,
// rows_per_clip = the frames rounded up to 16); false = not eligible
bool launch_stft_complex_fm(const StftJob &job, void *out, int64_t pitch_floats, int64_t rows_per_clip, bool only_ask = false);
bool fast_path_disabled();                        // env SMX_DISABLE_FAST=1 (tests)
void init_device_pool();                          // tables.cpp: the library's own stream-ordered pool of the current device
hipError_t pool_malloc_async(void **ptr, size_t bytes, hipStream_t stream);   // tables.cpp: scratch from that pool
void set_scratch_retention(int64_t bytes);        // tables.cpp

// Stft.invert (stft.ml:902-939) on device-resident data
struct IstftJob {
  const smx_stft_config *cfg = nullptr;
  const void *z = nullptr;       // device [lead; bins; frames] complex (interleaved), frames fastest
  int z_bytes = 8;               // 8 = complex64, 16 = complex128
  int interior = SMX_INTERIOR_F32;
  int64_t lead = 0, frames = 0;
  int64_t count = 0;             // frames that reach the output
  int64_t out_len = 0;
  void *out = nullptr;           // device [lead; out_len], float32 for complex64 input, float64 for complex128
  // optional real factors [lead; bins; frames] (element type of z's components) multiplied into z as it is read:
  // Griffin-Lim's S * angles without materialising the product.  Only where istft_takes_factors(job) says so.
  const void *mag = nullptr;
  // with `unit` (same condition): z is Griffin-Lim's rebuilt spectrum c_k and the kernel inverts
  // mag * unit(c_k - beta * prev) -- the phase update of stft.ml:1003-1012 folded into the staging; prev may be null
  bool unit = false;
  const void *prev = nullptr;
  double beta = 0.0;
  hipStream_t stream = nullptr;
  // Streaming synthesis (Stft.Synthesis, stft.ml:1181-1269) runs the same kernels on the LAST frames of a longer synthesis:
  // the array's padded position 0 is position env_q0 of that synthesis (a multiple of the hop, negative while the stream is
  // younger than a frame), out[0] is the array's padded position `left` (default: the configuration's left width), the
  // envelope is that of an env_count-frame synthesis (default: count), and with env_open the synthesis goes on after these
  // frames (no tail region: every position read is settled, stft.ml:1137-1142)
  int64_t left = -1, env_q0 = 0, env_count = 0;
  bool env_open = false;
  // Griffin-Lim's own spectra (capi.cpp): z, prev and mag FRAME-MAJOR, [lead; fm_rows; fm_pitch] elements (fm_pitch > 0; fft 2048 /
  // hop 512 / complex64 / float32 interior on the persistent pipeline only: istft_frame_major_ok(job))
  int64_t fm_pitch = 0, fm_rows = 0;
};
void launch_istft(const IstftJob &job);           // istft.hip
bool istft_frame_major_ok(const IstftJob &job);   // the frame-major form of the job (fm_pitch / fm_rows set) has a kernel
// Stft.Synthesis' release (stft.ml:1172-1179, 1229-1241): stream = carry ++ quot per channel; out gets stream[drop, release),
// carry_out gets stream[release, carry_len + nq).  Rows: carry / carry_out `hold` apart, quot `nq` apart, out `out_stride` apart.
void launch_synthesis_release(const void *carry, int64_t carry_len, const void *quot, int64_t nq, int64_t channels, int64_t drop,
                              int64_t release, int64_t hold, void *out, int64_t out_stride, void *carry_out, int elem_bytes,
                              hipStream_t stream);   // istft.hip
bool istft_takes_factors(const IstftJob &job);    // istft.hip: the fused fft-2048 / hop-512 kernel does

// elementwise steps of Stft.griffin_lim (griffinlim.hip); elem_bytes 4 = float32 / complex64, 8 = float64 / complex128
void launch_gl_widen(const float *src, double *dst, int64_t total, hipStream_t stream);    // float32 -> float64
void launch_gl_narrow(const double *src, float *dst, int64_t total, hipStream_t stream);   // float64 -> float32
void launch_gl_init(const void *phase, void *angles, int64_t total, int elem_bytes, hipStream_t stream);
// [lead][bins][frames] -> [lead][rows][pitch] of elem_bytes-wide elements (4: magnitudes, 8: complex64 angles)
void launch_gl_to_frame_major(const void *src, void *dst, int64_t lead, int64_t bins, int64_t frames, int64_t rows, int64_t pitch, int elem_bytes,
                              hipStream_t stream);
void launch_gl_apply(const void *mag, const void *angles, void *z, int64_t total, int elem_bytes, hipStream_t stream);
void launch_gl_update(const void *rebuilt, const void *previous, double beta, void *angles, int64_t total,
                      int elem_bytes, hipStream_t stream);

struct MelJob {
  const smx_mel_config *cfg = nullptr;
  const void *s = nullptr;       // device [lead; bins; frames]
  int elem_bytes = 4;
  int64_t lead = 0, frames = 0;
  void *out = nullptr;           // device [lead; n_mels; frames]
  hipStream_t stream = nullptr;
};
void launch_mel_apply(const MelJob &job);         // mel.hip

// log-mel + DCT tail of Soundml.mfcc (soundml.ml:50-95) on a device-resident mel spectrogram
struct MfccJob {
  const void *mel = nullptr;     // device [lead; n_mels; frames]
  int elem_bytes = 4;
  int64_t lead = 0, frames = 0;
  int n_mels = 0, n_mfcc = 0;
  double lifter = 0.0;           // 0: none
  void *out = nullptr;           // device [lead; n_mfcc; frames]
  hipStream_t stream = nullptr;
};
void launch_mfcc(const MfccJob &job);             // mfcc.hip

// Convert.power_to_db / amplitude_to_db (convert.ml:30-62) on device-resident data of either dtype
struct ToDbJob {
  const void *s = nullptr;
  void *out = nullptr;           // same shape and dtype; may alias s
  int elem_bytes = 4;
  int64_t total = 0;
  double gain = 10.0;            // 10 powers, 20 amplitudes
  bool magnitude = false;        // |s| first (amplitudes)
  double reference = 1.0, amin = 1e-10, top_db = 0.0;
  bool has_top_db = false;
  hipStream_t stream = nullptr;
};
void launch_to_db(const ToDbJob &job);            // mfcc.hip

// spectral-shape features over a device-resident spectrogram [lead; bins; frames] (spectral.ml:171-255)
enum SpectralFeature { SPECTRAL_CENTROID = 0, SPECTRAL_BANDWIDTH = 1, SPECTRAL_ROLLOFF = 2, SPECTRAL_FLATNESS = 3 };
struct SpectralJob {
  int feature = SPECTRAL_CENTROID;
  const void *s = nullptr;       // device [lead; bins; frames]
  int elem_bytes = 4;
  int64_t lead = 0, bins = 0, frames = 0;
  const double *freqs = nullptr; // HOST [bins] custom grid, or null: bin k at k * step
  double step = 0.0;
  double p = 2.0;                // bandwidth exponent | roll_percent | flatness amin
  double power = 2.0;            // flatness power
  const void *centroid = nullptr;   // device [lead; 1; frames] (bandwidth), element type of s, or null
  void *out = nullptr;           // device [lead; 1; frames]
  hipStream_t stream = nullptr;
};
// returns false when the spectrogram holds a negative or NaN entry (nothing useful is written then);
// synchronises the stream to read that verdict
bool launch_spectral(const SpectralJob &job);   // spectral.hip

// Chroma.apply (chroma.ml:285-317): float64 projection + per-frame normalisation, one rounding
struct ChromaJob {
  const smx_chroma_config *config = nullptr;
  const void *s = nullptr;       // device [lead; bins; frames]
  int elem_bytes = 4;
  int64_t lead = 0, frames = 0;
  int norm = SMX_CHROMA_NORM_INF;
  double norm_p = 0.0;           // exponent for SMX_CHROMA_NORM_P
  void *out = nullptr;           // device [lead; n_chroma; frames]
  hipStream_t stream = nullptr;
};
void launch_chroma(const ChromaJob &job);       // spectral.hip

struct MelSpecJob {
  StftJob stft;                  // out/out_stride unused; mode/power used
  const smx_mel_config *mel = nullptr;
  void *out = nullptr;           // device [lead; n_mels; count]
};
bool launch_mel_spectrogram_fused(const MelSpecJob &job);   // stft_fast.hip; false = not eligible
bool launch_mel_spectrogram_16(const MelSpecJob &job);      // stft_generic.hip: fft 512 / 1024

}  // namespace smx
