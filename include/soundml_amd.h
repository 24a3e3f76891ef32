/*
 * soundml_amd.h -- C ABI of the MI355X-native spectral path for SoundML.
 *
 * This is the drop-in boundary: plain C, raw pointers + int64 extents, an int
 * status and a thread-local error string.  Each entry point names the reference
 * interface it replaces (paths relative to /root/reference/soundml/lib).  The
 * OCaml side binds these with `external` + CAMLprim stubs exactly like the
 * reference binds resample_stubs.c (resample.ml:1162-1196); see INTEGRATION.md
 * and ocaml/soundml_amd_stubs.c.
 *
 * Conventions (reference: soundml.mli:3-24, stft.mli:211-250)
 *   - time axis last; audio is [lead; n], spectra are [lead; bins; frames]
 *     (frames fastest); `lead` is the product of all leading axes;
 *   - inputs are borrowed for the call, outputs are caller-allocated and fully
 *     overwritten, nothing is retained (stft.mli:454-458);
 *   - complex outputs are interleaved (re, im) pairs of the component type;
 *   - user-facing precondition failures return SMX_INVALID_ARGUMENT with the
 *     reference's own message (OCaml: raise Invalid_argument); geometry /
 *     runtime (HIP) failures return SMX_FAILURE (OCaml: Failure);
 *   - `*_dev` entry points take DEVICE pointers on the current HIP device and a
 *     hipStream_t (as void*); they enqueue work and return without
 *     synchronising.  The host-pointer entry points upload, run and download
 *     (this is what an nx-tensor caller uses).
 *   - every extent is validated before any pointer is formed or any device
 *     work is enqueued (discipline of resample_stubs.c:228-276).
 *   - the library never falls back to a CPU implementation: without a HIP
 *     device every compute entry point returns SMX_FAILURE.
 */
#ifndef SOUNDML_AMD_H
#define SOUNDML_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMX_OK 0
#define SMX_INVALID_ARGUMENT 1
#define SMX_FAILURE 2

/* "argument not given" for optional integer arguments (OCaml ?win_length / ?hop) */
#define SMX_DEFAULT INT64_MIN

/* Stft.Config.alignment (stft.ml:53) */
#define SMX_ALIGN_CENTERED 0
#define SMX_ALIGN_LEFT 1
#define SMX_ALIGN_RIGHT 2
/* Stft.Config.pad (stft.ml:54) */
#define SMX_PAD_REFLECT 0
#define SMX_PAD_CONSTANT 1
#define SMX_PAD_EDGE 2
/* Stft.Config.scale (stft.ml:55) */
#define SMX_SCALE_NONE 0
#define SMX_SCALE_MAGNITUDE 1
#define SMX_SCALE_PSD 2
/* Window.t families built by the library (window.ml:362-375); any other family
 * is built by the host in float64 and passed as a table (SMX_WINDOW_CUSTOM). */
#define SMX_WINDOW_HANN 0
#define SMX_WINDOW_RECTANGULAR 1
#define SMX_WINDOW_HAMMING 2
#define SMX_WINDOW_BLACKMAN 3
#define SMX_WINDOW_BLACKMAN_HARRIS 4
#define SMX_WINDOW_NUTTALL 5
#define SMX_WINDOW_FLAT_TOP 6
#define SMX_WINDOW_BARTLETT 7
#define SMX_WINDOW_KAISER 8      /* shape parameter: beta, finite and non-negative      */
#define SMX_WINDOW_GAUSSIAN 9    /* shape parameter: standard deviation in samples, > 0 */
#define SMX_WINDOW_TUKEY 10      /* shape parameter: taper fraction in [0, 1]           */
#define SMX_WINDOW_CUSTOM 100
/* Mel.Config.scale / norm (mel.ml:26-27) */
#define SMX_MEL_SLANEY 0
#define SMX_MEL_HTK 1
#define SMX_NORM_SLANEY 0
#define SMX_NORM_NONE 1
/* Interior arithmetic for float32 audio.  The reference computes in float64
 * whatever the I/O dtype (stft.ml:28-35); SMX_INTERIOR_F32 is the documented
 * fast deviation (north_star: 1e-5 relative), SMX_INTERIOR_F64 reproduces the
 * reference's interior (widen, f64 window multiply, f64 FFT, one rounding).
 * float64 audio always uses the float64 interior. */
#define SMX_INTERIOR_F32 0
#define SMX_INTERIOR_F64 1

typedef struct smx_stft_config smx_stft_config;
typedef struct smx_mel_config smx_mel_config;
typedef struct smx_chroma_config smx_chroma_config;
typedef struct smx_stft_kernel smx_stft_kernel;
typedef struct smx_fir_plan smx_fir_plan;
typedef struct smx_resample_stage smx_resample_stage;
typedef struct smx_resample_kernel smx_resample_kernel;

/* ---- library ------------------------------------------------------------ */
const char *smx_last_error(void);      /* message of the last failing call on this thread */
int smx_version(void);
/* diagnostics: kernel launches issued by this process through the library so far (bench.py asserts that one step
 * of the hot path is one launch, so that HIP events around a step time that kernel) */
unsigned long long smx_debug_kernel_launches(void);
/* diagnostics (tests of Griffin-Lim's internal layout): Stft.transform of device-resident float32 audio at fft 2048 with the
 * spectrum FRAME-MAJOR -- out[clip][frame][bin] complex64 in rows of pitch_floats floats, rows_per_clip (the frames rounded up to
 * a multiple of 16) rows per clip; fails where the frame-major kernel does not apply */
int smx_debug_stft_transform_frame_major_f32_dev(const smx_stft_config *c, const float *d_x, int64_t lead, int64_t n, float *d_out,
                                                 int64_t pitch_floats, int64_t rows_per_clip, void *stream);
int smx_device_count(int *count);
int smx_set_device(int device);        /* device used by this thread's subsequent calls */
/* Clip sharding for the HOST-POINTER batch calls (round 6).  The reference's caller is one process that hands over host
 * tensors, "a batch of clips is one call" (stft.mli:211-250, soundml.mli:3-24), and its leading axes are independent by contract
 * (stft.mli:214-218; per-slice law tested in stft_grid.ml:180-205, mel_props.ml:136-155).  With a device list set, every
 * host-pointer entry point of Stft.transform / transform_range / power_spectrum / power_range / invert and
 * Soundml.mel_spectrogram splits `lead` into contiguous clip ranges -- shard s of S owns clips [s*q + min(s, r), ...) with
 * q = lead / S, r = lead % S, the first r shards one clip more (soundml_amd/shard.py clip_range) -- and runs each range on its
 * device from a host thread of its own, with that device's own staging rings and PCIe link; every shard writes its slice of the
 * caller's `out`, so the result is the single-device call's bit for bit and no collective is involved.  A device may be listed
 * more than once (virtual shards).  n = 0 clears the list: calls run on the calling thread's current device (smx_set_device).
 * Process-wide; the `*_dev` entry points and the streaming states are not affected (they live on one device by construction).
 * smx_get_devices writes min(*n, capacity) ordinals. */
int smx_set_devices(const int *devices, int n);
int smx_get_devices(int *devices, int capacity, int *n);
/* the rule itself (no device needed): clips [*lo, *hi) of shard `shard` of `shards` -- what a caller that gathers per-device results
 * itself, or a test, checks the split against (soundml_amd/shard.py clip_range is the same function) */
int smx_shard_clip_range(int64_t total_clips, int64_t shards, int64_t shard, int64_t *lo, int64_t *hi);
/* diagnostics (tests of the per-device staging): the most staged uploads / downloads that were ever in flight at once */
int smx_debug_staging_peak(int *uploads, int *downloads, int reset);
int smx_set_interior(int interior);    /* SMX_INTERIOR_*, process-wide default for f32 audio */
/* Scratch arrays (Griffin-Lim's spectra, scratch spectrograms, small tables) come from a stream-ordered memory pool
 * that the library creates for itself on each device (the process-wide default pool is never touched); it keeps up to
 * `bytes` of freed scratch per device for reuse instead of returning it to the driver at every synchronisation
 * (default: 1/8 of the device's memory, at most 16 GiB; 0 = keep nothing; -1 = the default again).  Memory held this
 * way is not visible to another allocator in the process (e.g. torch's): lower it when embedding. */
int smx_set_scratch_retention(int64_t bytes);
int smx_get_interior(void);
int smx_synchronize(void *stream);
/* Result arrays for the host-pointer entry points.  The reference's faces return a fresh host tensor per call (`analyse`
 * allocates its result inside Nx.stft: stft.ml:356-364; Stft.power_spectrum: stft.ml:670-691).  Fresh pageable memory is the slow
 * half of such a call -- every page of a 0.98 GB spectrogram faults as the copying threads reach it, and the bytes cross host
 * memory twice (DMA into a staging ring, then memcpy).  A block from smx_host_alloc is page-locked: when the `out` (or `x`)
 * pointer of a host-pointer entry point lies inside one, the DMA engine writes (reads) it directly.  smx_host_free returns the
 * block to the library, which keeps up to 3 GiB of released blocks for the next result of about the same size.  A binding
 * allocates its result tensors here (the OCaml stub wraps the block in a Bigarray whose finaliser calls smx_host_free:
 * INTEGRATION.md section 2); a pointer from any other allocator takes the staged path, as before.  Thread-safe. */
int smx_host_alloc(size_t bytes, void **ptr);
int smx_host_free(void *ptr);

/* ---- Window.make (window.ml:374-405), float64, Hann & friends ------------ */
int smx_window_make(int kind, int periodic, int64_t n, double *out);
/* the parametric families too (window.ml:77-97 validate their shape parameter; ignored by the others) */
int smx_window_make_param(int kind, double param, int periodic, int64_t n, double *out);
/* Window.cola (window.ml:407-434): 1 iff the `length`-point periodic window overlap-adds to a constant at `hop` */
int smx_window_cola(int kind, double param, int64_t length, int64_t hop, int *cola);

/* ---- Stft.Config (stft.ml:48-129) --------------------------------------- */
int smx_stft_config_create(int64_t fft_size, int64_t win_length /* SMX_DEFAULT = fft_size */,
                           int64_t hop /* SMX_DEFAULT = max 1 (fft_size/4) */, int alignment,
                           int pad, double pad_value, int scale, int window_kind,
                           const double *custom_window /* win_length doubles iff SMX_WINDOW_CUSTOM */,
                           smx_stft_config **out);
void smx_stft_config_destroy(smx_stft_config *c);
int64_t smx_stft_config_fft_size(const smx_stft_config *c);
int64_t smx_stft_config_hop(const smx_stft_config *c);
int64_t smx_stft_config_win_length(const smx_stft_config *c);
int64_t smx_stft_config_bins(const smx_stft_config *c);          /* stft.ml:127 */
int64_t smx_stft_config_left_width(const smx_stft_config *c);    /* stft.ml:132-140 */
int64_t smx_stft_config_right_width(const smx_stft_config *c);   /* stft.ml:141-142 */
int64_t smx_stft_config_latency(const smx_stft_config *c);       /* stft.ml:144-145 */
int smx_stft_config_analysis_window(const smx_stft_config *c, double *out /* fft_size */);

/* ---- frame grid (stft.ml:217-261) ---------------------------------------- */
int smx_stft_frames(const smx_stft_config *c, int64_t n, int64_t *out);
int smx_stft_first_complete(const smx_stft_config *c, int64_t *out);
int smx_stft_last_complete(const smx_stft_config *c, int64_t n, int64_t *out);
int smx_stft_times(const smx_stft_config *c, int64_t sample_rate, int64_t n, double *out /* frames */);
int smx_stft_frequencies(const smx_stft_config *c, int64_t sample_rate, double *out /* bins */);

/* ---- Stft.transform / transform_range / power_spectrum (stft.ml:632-691) ---
 * x: [lead; n] contiguous (x_stride = elements between consecutive signals, >= n).
 * out: transform -> complex [lead; bins; frames]; range -> [lead; bins; p1-p0];
 *      power_spectrum -> real [lead; bins; frames].                           */
int smx_stft_transform_f32(const smx_stft_config *c, const float *x, int64_t lead, int64_t n,
                           float *out_c64);
int smx_stft_transform_f64(const smx_stft_config *c, const double *x, int64_t lead, int64_t n,
                           double *out_c128);
int smx_stft_transform_range_f32(const smx_stft_config *c, const float *x, int64_t lead, int64_t n,
                                 int64_t p0, int64_t p1, float *out_c64);
int smx_stft_transform_range_f64(const smx_stft_config *c, const double *x, int64_t lead, int64_t n,
                                 int64_t p0, int64_t p1, double *out_c128);
int smx_stft_power_spectrum_f32(const smx_stft_config *c, const float *x, int64_t lead, int64_t n,
                                double power, float *out);
int smx_stft_power_spectrum_f64(const smx_stft_config *c, const double *x, int64_t lead, int64_t n,
                                double power, double *out);
/* |frames [p0, p1)|^power: transform_range (stft.ml:652-666) followed by magnitude_pow (stft.ml:670-674) */
int smx_stft_power_range_f32(const smx_stft_config *c, const float *x, int64_t lead, int64_t n, int64_t p0,
                             int64_t p1, double power, float *out);
int smx_stft_power_range_f64(const smx_stft_config *c, const double *x, int64_t lead, int64_t n, int64_t p0,
                             int64_t p1, double power, double *out);
/* device-resident forms (the batch API the benchmark and the sharded driver use) */
int smx_stft_transform_range_f32_dev(const smx_stft_config *c, const float *d_x, int64_t lead,
                                     int64_t n, int64_t x_stride, int64_t p0, int64_t p1,
                                     float *d_out_c64, void *stream);
int smx_stft_transform_range_f64_dev(const smx_stft_config *c, const double *d_x, int64_t lead,
                                     int64_t n, int64_t x_stride, int64_t p0, int64_t p1,
                                     double *d_out_c128, void *stream);
int smx_stft_power_range_f32_dev(const smx_stft_config *c, const float *d_x, int64_t lead,
                                 int64_t n, int64_t x_stride, int64_t p0, int64_t p1, double power,
                                 float *d_out, void *stream);
int smx_stft_power_range_f64_dev(const smx_stft_config *c, const double *d_x, int64_t lead,
                                 int64_t n, int64_t x_stride, int64_t p0, int64_t p1, double power,
                                 double *d_out, void *stream);

/* ---- Stft.Kernel (stft.ml:597-622): streaming analysis, device-resident carry
 * dtype_bytes: 4 (float32 audio, complex64 spectra) or 8.
 * step/flush write at most `capacity` frames into out [channels; bins; capacity]
 * laid out with `capacity` as the frame stride, and report how many they emitted
 * (0 = the reference's None).  frame_bound = stft.ml:1316-1317.               */
int smx_stft_kernel_prepare(const smx_stft_config *c, int dtype_bytes, int64_t channels,
                            int64_t max_block, smx_stft_kernel **out);
/* the same state machine emitting |spectrum|^power in the chunk's dtype ([channels; bins; capacity] real): the body
 * of Stft.power_stage (stft.ml:1364-1409); stage_latency / frame_bound / stage_rate of stft.ml:1307-1348 are
 * smx_stft_stage_latency, smx_stft_frame_bound and 1 / hop.                                                   */
int smx_stft_kernel_prepare_power(const smx_stft_config *c, int dtype_bytes, int64_t channels, int64_t max_block,
                                  double power, smx_stft_kernel **out);
int64_t smx_stft_stage_latency(const smx_stft_config *c);                 /* stft.ml:1307-1308 */
int64_t smx_stft_frame_bound(const smx_stft_config *c, int64_t max_items); /* stft.ml:1316-1317 */
void smx_stft_kernel_destroy(smx_stft_kernel *k);
int smx_stft_kernel_frame_bound(const smx_stft_kernel *k, int64_t *out);
int smx_stft_kernel_step(smx_stft_kernel *k, const void *chunk /* host [channels; m] */, int64_t m,
                         void *out_complex, int64_t capacity, int64_t *emitted);
int smx_stft_kernel_flush(smx_stft_kernel *k, void *out_complex, int64_t capacity, int64_t *emitted);
int smx_stft_kernel_reset(smx_stft_kernel *k);
/* the same state machine fed from and emitting into DEVICE memory (chunk rows x_stride samples apart; the window
 * [channels; bins; capacity] as above): a push moves nothing over the host link.  The frame count comes back on the host
 * (it is bookkeeping of lengths, known before any kernel runs); the frames are ready in stream order.                */
int smx_stft_kernel_step_dev(smx_stft_kernel *k, const void *d_chunk, int64_t m, int64_t x_stride, void *d_out,
                             int64_t capacity, int64_t *emitted, void *stream);
int smx_stft_kernel_flush_dev(smx_stft_kernel *k, void *d_out, int64_t capacity, int64_t *emitted, void *stream);
/* The reference's state takes its leading shape from the chunks it is fed (stft.ml:521-559: only `channels >= 1` is checked
 * at prepare, stft.ml:603-617), so a binding must be able to follow the first chunk: channels() reports the count the
 * kernel was prepared for (step / flush read and write exactly that many rows: a caller passes buffers of that extent),
 * set_channels() re-shapes a kernel that has not received a sample since prepare / reset and refuses one that has
 * (SMX_INVALID_ARGUMENT: the stream's chunks disagree in their leading shape).                                      */
int smx_stft_kernel_channels(const smx_stft_kernel *k, int64_t *out);
int smx_stft_kernel_set_channels(smx_stft_kernel *k, int64_t channels);
const smx_stft_config *smx_stft_kernel_config(const smx_stft_kernel *k);   /* the configuration it was prepared with (borrowed) */

/* ---- Stft.Synthesis (stft.ml:1271-1298; stft.mli:519-591) and the body of synthesis_stage (stft.ml:1417-1442):
 * incremental least-squares synthesis with the state in device memory.  dtype_bytes 4: complex64 frames in, float32
 * samples out; 8: complex128 / float64.  step takes k frames [channels; bins; k] (frames fastest) and releases every sample
 * they settle into out [channels; capacity] (row stride = capacity), reporting how many per channel (0 = the reference's
 * None); flush drains the trimmed tail.  Any chunking of a stream totals smx_stft_invert of the whole stream, bit for
 * bit.  sample_bound = the most samples one step of max_block frames or the flush can release (synthesis_stage's
 * max_items, stft.ml:1421-1427); smx_stft_synthesis_latency = Config.synthesis_latency (stft.ml:152-153).            */
typedef struct smx_stft_synthesis smx_stft_synthesis;
int smx_stft_synthesis_prepare(const smx_stft_config *c, int dtype_bytes, int64_t channels, int64_t max_block,
                               smx_stft_synthesis **out);
void smx_stft_synthesis_destroy(smx_stft_synthesis *s);
int64_t smx_stft_synthesis_latency(const smx_stft_config *c);
int smx_stft_synthesis_sample_bound(const smx_stft_synthesis *s, int64_t *out);
int smx_stft_synthesis_step(smx_stft_synthesis *s, const void *z, int64_t bins, int64_t k, void *out, int64_t capacity,
                            int64_t *emitted);
int smx_stft_synthesis_flush(smx_stft_synthesis *s, void *out, int64_t capacity, int64_t *emitted);
int smx_stft_synthesis_reset(smx_stft_synthesis *s);
/* the same on device-resident chunks and outputs (no host copy per push) */
int smx_stft_synthesis_step_dev(smx_stft_synthesis *s, const void *d_z, int64_t bins, int64_t k, void *d_out,
                                int64_t capacity, int64_t *emitted, void *stream);
int smx_stft_synthesis_flush_dev(smx_stft_synthesis *s, void *d_out, int64_t capacity, int64_t *emitted, void *stream);

/* ---- Mel.Config / Mel.apply (mel.ml:22-233) ------------------------------- */
/* Convert.hz_to_mel / mel_to_hz (convert.ml:70-102): the scalar maps behind the filterbank's breakpoints; host float64 */
int smx_hz_to_mel(int scale /* SMX_MEL_* */, const double *f, int64_t n, double *out);
int smx_mel_to_hz(int scale, const double *m, int64_t n, double *out);
int smx_mel_config_create(int64_t n_mels, int64_t sample_rate, int64_t fft_size, double f_min,
                          int has_f_max, double f_max, int scale, int norm, smx_mel_config **out);
/* A filterbank handle over caller-supplied float64 weights [rows; fft_size/2 + 1] (copied): every projection of
 * the reference that is `matmul W (power spectrum)` with its own W -- Chroma.apply's chroma filters
 * (chroma.ml:307), a mel bank built elsewhere -- runs through smx_mel_apply_* / smx_mel_spectrogram_*.       */
int smx_mel_config_from_weights(int64_t rows, int64_t fft_size, const double *weights, smx_mel_config **out);
void smx_mel_config_destroy(smx_mel_config *c);
int64_t smx_mel_config_n_mels(const smx_mel_config *c);
int64_t smx_mel_config_bins(const smx_mel_config *c);
int64_t smx_mel_config_fft_size(const smx_mel_config *c);
double smx_mel_config_f_max(const smx_mel_config *c);
int smx_mel_filterbank(const smx_mel_config *c, double *out /* [n_mels; bins] */); /* mel.ml:200 */
/* s: [lead; bins; frames] -> out [lead; n_mels; frames]; `bins` is the caller's
 * declared axis size and is checked against the config (mel.ml:210-218).     */
int smx_mel_apply_f32(const smx_mel_config *c, const float *s, int64_t lead, int64_t bins,
                      int64_t frames, float *out);
int smx_mel_apply_f64(const smx_mel_config *c, const double *s, int64_t lead, int64_t bins,
                      int64_t frames, double *out);
int smx_mel_apply_f32_dev(const smx_mel_config *c, const float *d_s, int64_t lead, int64_t bins,
                          int64_t frames, float *d_out, void *stream);
int smx_mel_apply_f64_dev(const smx_mel_config *c, const double *d_s, int64_t lead, int64_t bins,
                          int64_t frames, double *d_out, void *stream);

/* ---- Soundml.mel_spectrogram (soundml.ml:12-24): fused audio -> mel -------- */
int smx_mel_spectrogram_f32(const smx_stft_config *sc, const smx_mel_config *mc, const float *x,
                            int64_t lead, int64_t n, double power, float *out);
int smx_mel_spectrogram_f64(const smx_stft_config *sc, const smx_mel_config *mc, const double *x,
                            int64_t lead, int64_t n, double power, double *out);
int smx_mel_spectrogram_f32_dev(const smx_stft_config *sc, const smx_mel_config *mc,
                                const float *d_x, int64_t lead, int64_t n, int64_t x_stride,
                                double power, float *d_out, void *stream);

/* ---- Stft.griffin_lim (stft.ml:941-1017): phase reconstruction from magnitudes s [lead; bins; frames].
 * c_k = analyse (synthesise (s * angles_k)),  angles_{k+1} = unit (c_k - a c_{k-1}),  a = momentum / (1 + momentum);
 * the loop runs at the natural length, `length` applies to the final synthesis only.  init_phase (radians,
 * same shape) or NULL for the all-ones phase.  out is [lead; out_len] as for smx_stft_invert_*.
 * Invalid_argument: the synthesis checks, n_iter < 1, momentum < 0 (messages of stft.ml:963-976).           */
int smx_stft_griffin_lim_f32(const smx_stft_config *c, const float *s, int64_t lead, int64_t bins, int64_t frames,
                             int64_t n_iter, double momentum, const float *init_phase, int has_length,
                             int64_t length, float *out);
int smx_stft_griffin_lim_f64(const smx_stft_config *c, const double *s, int64_t lead, int64_t bins, int64_t frames,
                             int64_t n_iter, double momentum, const double *init_phase, int has_length,
                             int64_t length, double *out);
int smx_stft_griffin_lim_f32_dev(const smx_stft_config *c, const float *d_s, int64_t lead, int64_t bins,
                                 int64_t frames, int64_t n_iter, double momentum, const float *d_init_phase,
                                 int has_length, int64_t length, float *d_out, void *stream);

/* ---- Convert.power_to_db / amplitude_to_db (convert.ml:30-62): decibels in the data's own dtype -------
 * db = scale * ln(max(v, amin)) - scale * ln(max(amin, reference)), scale = gain / ln 10 (gain 10 for powers, 20 for
 * amplitudes, which take |s| first: a negative power sits at the floor); with has_top_db the result is clamped at
 * (the maximum over the WHOLE tensor) - top_db.  `total` elements of any shape; out may alias s on the device.
 * Invalid_argument (convert.ml:3-16): reference / amin not finite and positive, top_db negative or not finite. */
int smx_power_to_db_f32(const float *s, int64_t total, double reference, double amin, int has_top_db, double top_db,
                        float *out);
int smx_power_to_db_f64(const double *s, int64_t total, double reference, double amin, int has_top_db, double top_db,
                        double *out);
int smx_power_to_db_f32_dev(const float *d_s, int64_t total, double reference, double amin, int has_top_db,
                            double top_db, float *d_out, void *stream);
int smx_amplitude_to_db_f32(const float *s, int64_t total, double reference, double amin, int has_top_db,
                            double top_db, float *out);
int smx_amplitude_to_db_f64(const double *s, int64_t total, double reference, double amin, int has_top_db,
                            double top_db, double *out);
int smx_amplitude_to_db_f32_dev(const float *d_s, int64_t total, double reference, double amin, int has_top_db,
                                double top_db, float *d_out, void *stream);

/* ---- Soundml.mfcc (soundml.ml:50-95): mel_spectrogram (power 2) -> power_to_db with the 80 dB clamp under
 * the maximum of the WHOLE tensor (convert.ml:30-50) -> orthonormal DCT-II along the mel axis, first n_mfcc
 * rows -> optional sinusoidal lifter.  float64 interior after the mel spectrogram, one rounding to the audio
 * dtype.  out is [lead; n_mfcc; frames].  Invalid_argument: fft sizes differ, n_mfcc outside [1, n_mels],
 * lifter negative or not finite (messages of soundml.ml:52-70).                                          */
int smx_mfcc_f32(const smx_stft_config *sc, const smx_mel_config *mc, const float *x, int64_t lead, int64_t n,
                 int64_t n_mfcc, int has_lifter, double lifter, float *out);
int smx_mfcc_f64(const smx_stft_config *sc, const smx_mel_config *mc, const double *x, int64_t lead, int64_t n,
                 int64_t n_mfcc, int has_lifter, double lifter, double *out);
int smx_mfcc_f32_dev(const smx_stft_config *sc, const smx_mel_config *mc, const float *d_x, int64_t lead,
                     int64_t n, int64_t x_stride, int64_t n_mfcc, int has_lifter, double lifter, float *d_out,
                     void *stream);

/* ---- Spectral-shape features (spectral.ml:171-255; Soundml.spectral_* re-exports, soundml.ml:119-133) ------
 * One reduction along the bin axis of a magnitude spectrogram s [lead; bins; frames] (frames fastest) into
 * [lead; 1; frames]; float64 interior in the reference's operation order, one rounding to the dtype of s.
 *   centroid   sum_k f_k * (s_k / max(sum s, guarded))            (frames whose sum is below the smallest
 *                                                                   normal double divide by 1)
 *   bandwidth  (sum_k (s_k / sum s) * |centroid - f_k|^p)^(1/p)    centroid: the caller's [lead; 1; frames]
 *                                                                   (c_rows / c_frames are its last two
 *                                                                   extents, checked) or NULL = computed
 *   rolloff    min { f_k : cumsum_k >= roll_percent * cumsum_last }
 *   flatness   exp(mean log m) / mean m,  m = max(s^power, amin)
 * freqs: HOST pointer to `n_freqs` bin frequencies (float64), or NULL = the FFT grid of the 2 (bins - 1)
 * point transform, bin k at k * (1 / (fft_size * (1 / sample_rate))) (spectral.ml:105-136).
 * Invalid_argument (messages of spectral.ml:27-98): sample_rate < 1, n_freqs != bins, bins < 2 without
 * freqs, p / roll_percent / amin / power out of range, a centroid of the wrong shape, and a spectrogram
 * holding a negative or NaN entry.  That last check reads the data, so the _dev entry points synchronise
 * the stream before returning.  Empty extents write nothing (the result is all zero by contract).         */
int smx_spectral_centroid_f32(const float *s, int64_t lead, int64_t bins, int64_t frames, const double *freqs,
                              int64_t n_freqs, int64_t sample_rate, float *out);
int smx_spectral_centroid_f64(const double *s, int64_t lead, int64_t bins, int64_t frames, const double *freqs,
                              int64_t n_freqs, int64_t sample_rate, double *out);
int smx_spectral_centroid_f32_dev(const float *d_s, int64_t lead, int64_t bins, int64_t frames,
                                  const double *freqs, int64_t n_freqs, int64_t sample_rate, float *d_out,
                                  void *stream);
int smx_spectral_bandwidth_f32(const float *s, int64_t lead, int64_t bins, int64_t frames, double p,
                               const double *freqs, int64_t n_freqs, const float *centroid, int64_t c_rows,
                               int64_t c_frames, int64_t sample_rate, float *out);
int smx_spectral_bandwidth_f64(const double *s, int64_t lead, int64_t bins, int64_t frames, double p,
                               const double *freqs, int64_t n_freqs, const double *centroid, int64_t c_rows,
                               int64_t c_frames, int64_t sample_rate, double *out);
int smx_spectral_bandwidth_f32_dev(const float *d_s, int64_t lead, int64_t bins, int64_t frames, double p,
                                   const double *freqs, int64_t n_freqs, const float *d_centroid, int64_t c_rows,
                                   int64_t c_frames, int64_t sample_rate, float *d_out, void *stream);
int smx_spectral_rolloff_f32(const float *s, int64_t lead, int64_t bins, int64_t frames, double roll_percent,
                             const double *freqs, int64_t n_freqs, int64_t sample_rate, float *out);
int smx_spectral_rolloff_f64(const double *s, int64_t lead, int64_t bins, int64_t frames, double roll_percent,
                             const double *freqs, int64_t n_freqs, int64_t sample_rate, double *out);
int smx_spectral_rolloff_f32_dev(const float *d_s, int64_t lead, int64_t bins, int64_t frames,
                                 double roll_percent, const double *freqs, int64_t n_freqs, int64_t sample_rate,
                                 float *d_out, void *stream);
int smx_spectral_flatness_f32(const float *s, int64_t lead, int64_t bins, int64_t frames, double amin,
                              double power, float *out);
int smx_spectral_flatness_f64(const double *s, int64_t lead, int64_t bins, int64_t frames, double amin,
                              double power, double *out);
int smx_spectral_flatness_f32_dev(const float *d_s, int64_t lead, int64_t bins, int64_t frames, double amin,
                                  double power, float *d_out, void *stream);

/* ---- Chroma over a linear-frequency spectrum (chroma.ml:95-317; Soundml.chroma_stft, soundml.ml:97-107) --
 * Config: the float64 [n_chroma; bins] projection of chroma.ml:109-175 (Gaussian bumps in the wrapped
 * chroma distance, unit euclidean columns, optional octave envelope, rows rolled so row 0 is C).
 * apply: out = cast(normalise(W x cast_f64(s))) with the per-frame norm of chroma.ml:58-88: lengths below
 * the smallest normal of the dtype of s divide by 1.  s is [lead; bins; frames], out [lead; n_chroma; frames].
 * chroma_stft = apply . Stft.power_spectrum ~power (fft sizes must agree).                                 */
#define SMX_CHROMA_NORM_NONE 0
#define SMX_CHROMA_NORM_INF 1
#define SMX_CHROMA_NORM_P 2      /* norm_p: finite, positive */
int smx_chroma_config_create(int64_t n_chroma, double tuning, double ctroct, int has_octwidth, double octwidth,
                             int base_c, int64_t sample_rate, int64_t fft_size, smx_chroma_config **out);
void smx_chroma_config_destroy(smx_chroma_config *c);
int64_t smx_chroma_config_n_chroma(const smx_chroma_config *c);
int64_t smx_chroma_config_bins(const smx_chroma_config *c);
int64_t smx_chroma_config_fft_size(const smx_chroma_config *c);
int smx_chroma_filterbank(const smx_chroma_config *c, double *out /* [n_chroma; bins] */); /* chroma.ml:259 */
int smx_chroma_apply_f32(const smx_chroma_config *c, const float *s, int64_t lead, int64_t bins, int64_t frames,
                         int norm, double norm_p, float *out);
int smx_chroma_apply_f64(const smx_chroma_config *c, const double *s, int64_t lead, int64_t bins, int64_t frames,
                         int norm, double norm_p, double *out);
int smx_chroma_apply_f32_dev(const smx_chroma_config *c, const float *d_s, int64_t lead, int64_t bins,
                             int64_t frames, int norm, double norm_p, float *d_out, void *stream);
int smx_chroma_stft_f32(const smx_stft_config *sc, const smx_chroma_config *cc, const float *x, int64_t lead,
                        int64_t n, double power, int norm, double norm_p, float *out);
int smx_chroma_stft_f64(const smx_stft_config *sc, const smx_chroma_config *cc, const double *x, int64_t lead,
                        int64_t n, double power, int norm, double norm_p, double *out);
int smx_chroma_stft_f32_dev(const smx_stft_config *sc, const smx_chroma_config *cc, const float *d_x,
                            int64_t lead, int64_t n, int64_t x_stride, double power, int norm, double norm_p,
                            float *d_out, void *stream);

/* ---- Least-squares synthesis: Stft.invert (stft.ml:902-939, stft.mli "invert") --------------------
 * x[m] = (sum_p w[m - p hop] irfft(Z[:, p])[m - p hop]) / (sum_p w^2[m - p hop]) in padded coordinates,
 * boundary extension trimmed, cut or zero-extended to `length`.  z is [lead; bins; frames] complex
 * (interleaved re, im; frames fastest), out is [lead; out_len] with out_len = length when has_length,
 * else smx_stft_output_length(frames).  The imaginary parts of the DC and Nyquist bins are ignored.
 * Invalid_argument (messages of stft.ml:745-786): wrong bin count, negative length, a window / hop pair
 * whose overlap-added squared window does not stay above 1e-10 of its maximum (smx_stft_nola).        */
int smx_stft_nola(const smx_stft_config *c, int *invertible);                         /* stft.ml:731-743 */
int smx_stft_output_length(const smx_stft_config *c, int64_t frames, int64_t *length); /* stft.ml:792-796 */
int smx_stft_invert_f32(const smx_stft_config *c, const float *z_c64, int64_t lead, int64_t bins,
                        int64_t frames, int has_length, int64_t length, float *out);
int smx_stft_invert_f64(const smx_stft_config *c, const double *z_c128, int64_t lead, int64_t bins,
                        int64_t frames, int has_length, int64_t length, double *out);
int smx_stft_invert_f32_dev(const smx_stft_config *c, const float *d_z_c64, int64_t lead, int64_t bins,
                            int64_t frames, int has_length, int64_t length, float *d_out, void *stream);
int smx_stft_invert_f64_dev(const smx_stft_config *c, const double *d_z_c128, int64_t lead, int64_t bins,
                            int64_t frames, int has_length, int64_t length, double *d_out, void *stream);

/* ---- FIR block convolution (BASELINE config 4; model resample.ml:383-415) --
 * y[c][i] = sum_k h[k] x[c][i-k], zeros before the stream start, i in [0, n).
 * Overlap-save on the library's own FFT core, spectrum of h precomputed.     */
int smx_fir_kaiser_beta(double attenuation_db, double *out);            /* resample.ml:105-109 */
int smx_fir_design_lowpass(int64_t taps, double cutoff, double beta, double *h); /* :145-163 */
int smx_fir_plan_create(const double *h, int64_t taps, smx_fir_plan **out);
void smx_fir_plan_destroy(smx_fir_plan *p);
int64_t smx_fir_plan_block(const smx_fir_plan *p);                      /* FFT length N */
int smx_fir_apply_f32(const smx_fir_plan *p, const float *x, int64_t channels, int64_t n, float *y);
int smx_fir_apply_f32_dev(const smx_fir_plan *p, const float *d_x, int64_t channels, int64_t n,
                          int64_t x_stride, float *d_y, int64_t y_stride, void *stream);


/* ---- Resample: the overlap-save executor's pieces ------------------------------------------------
 * The reference's planner, polyphase bank and cascade logic (resample.ml, 2100 lines) stay in OCaml; what moves to the
 * device is the block convolution of a stage and the block identity its own C stub computes.
 *
 * smx_resample_ols_geom    resample.ml:279-300 `ols_block_n` / `ols_geom`: (N, B, delta) of a xL or /M stage of group
 *                          delay K consuming `rate` Hz; *eligible = 0 past the 130 ms emission ceiling.
 * smx_resample_prototype   resample.ml:145-163 `design_prototype`: the 2 K L + 1 tap Kaiser-sinc, float64, sum = L.
 * smx_resample_shape_c128  replaces `soundml_resample_shape` (resample_stubs.c:329-422; bound in resample.ml:1186-1196
 *                          as `resample_shape_c`): interleaved complex128 half spectra x [lines; N/2+1] and plan
 *                          spectrum h -> y [lines; W/2+1], W = N sl | N / sm | N.  xL: periodic extension times h[k];
 *                          /M: product on the half grid, alias fold in ascending order.  Same operations in the same
 *                          order in float64 (no fused multiply-add): bit for bit the stub's result.  Geometry errors
 *                          are Failure with the stub's message ("soundml_resample_shape: invalid geometry").
 * smx_resample_stage_*     one stage y[i] = sum_t proto[t] xu[i M + K L - t], xu = x zero-stuffed by L, ceil(n L / M)
 *                          outputs (what `ols_run` + the virtual-silence drain emit, resample.ml:1456-1599,1745-1755),
 *                          float32 interior.  A pure xL or /M stage (the overlap-save eligible ones, resample.ml:279-300)
 *                          runs in its polyphase form block by block in the frequency domain -- L filters of the input at
 *                          the input rate / the sum of M filters of the input's phases at the output rate: the arithmetic
 *                          of the reference's spectral shortcut (resample_stubs.c:329-372) regrouped into power-of-two
 *                          transforms at the low rate; any other L / M as one block convolution at the interpolated
 *                          rate on the FIR kernel.  2 K L + 1 <= 16384 taps.
 * smx_resample_kernel_*    `Resample.Kernel.{prepare,step,flush,reset}` (resample.mli:270-319, executor `ols_run`
 *                          resample.ml:1456-1599) of ONE pure xL or /M stage: all channels in one state, the unconsumed
 *                          input (the block carry) on the device; a step emits the block pairs its samples complete
 *                          (burst emission, possibly nothing), flush the virtual-silence tail, and every partition of a
 *                          signal totals smx_resample_stage_apply bit for bit.  Errors as the reference's: a chunk
 *                          longer than max_block, a step after flush (reset first), channels or max_block < 1 are
 *                          SMX_INVALID_ARGUMENT.  The stage must outlive the kernel.  A second flush emits nothing.     */
int smx_resample_ols_geom(int64_t rate, int64_t l, int64_t m, int64_t k, int64_t *n, int64_t *b, int64_t *delta,
                          int *eligible);
int smx_resample_prototype(int64_t l, int64_t k, double fc, double beta, double *h /* 2 K L + 1 */);
int smx_resample_shape_c128(const double *x, const double *h, double *y, int64_t lines, int64_t n, int64_t sl, int64_t sm);
int smx_resample_shape_c128_dev(const double *d_x, const double *d_h, double *d_y, int64_t lines, int64_t n, int64_t sl,
                                int64_t sm, void *stream);
int smx_resample_stage_create(const double *proto /* 2 K L + 1 */, int64_t l, int64_t m, int64_t k, smx_resample_stage **out);
void smx_resample_stage_destroy(smx_resample_stage *s);
int64_t smx_resample_stage_out_length(const smx_resample_stage *s, int64_t n);       /* ceil(n L / M) */
int smx_resample_stage_apply_f32(const smx_resample_stage *s, const float *x, int64_t channels, int64_t n, float *y);
int smx_resample_stage_apply_f32_dev(const smx_resample_stage *s, const float *d_x, int64_t channels, int64_t n,
                                     int64_t x_stride, float *d_y, int64_t y_stride, void *stream);
int smx_resample_kernel_prepare(const smx_resample_stage *s, int64_t channels, int64_t max_block, smx_resample_kernel **out);
void smx_resample_kernel_destroy(smx_resample_kernel *k);
int smx_resample_kernel_reset(smx_resample_kernel *k);
int64_t smx_resample_kernel_out_bound(const smx_resample_kernel *k, int64_t n);   /* most samples per channel a step of n can emit */
int64_t smx_resample_kernel_pending(const smx_resample_kernel *k);                /* samples per channel the next flush emits */
/* x [channels; n] (row stride x_stride) -> y [channels; *n_out] (row stride y_stride >= out_bound(n) / pending) */
int smx_resample_kernel_step_f32(smx_resample_kernel *k, const float *x, int64_t n, int64_t x_stride, float *y, int64_t y_stride,
                                 int64_t *n_out);
int smx_resample_kernel_flush_f32(smx_resample_kernel *k, float *y, int64_t y_stride, int64_t *n_out);
int smx_resample_kernel_step_f32_dev(smx_resample_kernel *k, const float *d_x, int64_t n, int64_t x_stride, float *d_y,
                                     int64_t y_stride, int64_t *n_out, void *stream);
int smx_resample_kernel_flush_f32_dev(smx_resample_kernel *k, float *d_y, int64_t y_stride, int64_t *n_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SOUNDML_AMD_H */
