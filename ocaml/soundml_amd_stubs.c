/* soundml_amd_stubs.c -- CAMLprim layer binding libsoundml_amd.so (include/soundml_amd.h)
 * into SoundML, written to the reference's own stub conventions
 * (soundml/lib/resample_stubs.c:228-297): arguments are Bigarray.Array1 views of
 * contiguous nx storage plus OCaml ints; every extent is validated against the
 * arrays' real dimensions before a pointer is formed; failures before the runtime
 * lock is released raise (Invalid_argument for user-facing preconditions, Failure
 * for bookkeeping); the lock is released around the device work and nothing touches
 * the OCaml heap while it is released.
 *
 * NOT compiled in this repository (no OCaml toolchain in the build image); it is the
 * binding a SoundML maintainer adds next to resample_stubs.c -- see INTEGRATION.md.
 */
#define CAML_NAME_SPACE
#include <caml/alloc.h>
#include <caml/bigarray.h>
#include <caml/custom.h>
#include <caml/fail.h>
#include <caml/memory.h>
#include <caml/mlvalues.h>
#include <caml/threads.h>

#include <stdint.h>
#include <string.h>

#include "soundml_amd.h"

/* ---- handles: custom blocks whose finaliser destroys the C object --------------------- */
#define Stft_val(v) (*((smx_stft_config **)Data_custom_val(v)))
#define Mel_val(v) (*((smx_mel_config **)Data_custom_val(v)))

static void stft_finalize(value v) { smx_stft_config_destroy(Stft_val(v)); }
static void mel_finalize(value v) { smx_mel_config_destroy(Mel_val(v)); }
static struct custom_operations stft_ops = {"soundml.amd.stft_config", stft_finalize, custom_compare_default,
                                            custom_hash_default, custom_serialize_default,
                                            custom_deserialize_default, custom_compare_ext_default,
                                            custom_fixed_length_default};
static struct custom_operations mel_ops = {"soundml.amd.mel_config", mel_finalize, custom_compare_default,
                                           custom_hash_default, custom_serialize_default,
                                           custom_deserialize_default, custom_compare_ext_default,
                                           custom_fixed_length_default};

/* status -> OCaml exception, with the library's (= the reference's) message */
static void smx_raise(int status) {
  if (status == SMX_OK) return;
  if (status == SMX_INVALID_ARGUMENT) caml_invalid_argument(smx_last_error());
  caml_failwith(smx_last_error());
}

static int64_t ba_dim(value v) { return (int64_t)Caml_ba_array_val(v)->dim[0]; }
static int ba_kind(value v) { return Caml_ba_array_val(v)->flags & CAML_BA_KIND_MASK; }

/* Result tensors in page-locked memory (include/soundml_amd.h, smx_host_alloc), OPT-IN (Stft_amd.set_pinned_results true).
 * The block is owned by a custom value that (a) reports its out-of-heap bytes to the GC (caml_alloc_custom_mem: a loop producing
 * 1 GB results then triggers major collections at the rate of the page-locked memory it holds, not of the few words on the
 * heap) and (b) returns the block to the library's pool exactly once, from its finaliser or from an explicit release.  The
 * Bigarray over it is CAML_BA_EXTERNAL -- the runtime neither owns nor reference-counts the data -- so stft_amd.ml keeps the
 * owner reachable from a finaliser closure on THE Bigarray it hands to Nx (result_tensor); headers derived from that Bigarray
 * outside Nx (Array1.sub, reshape, genarray_of_array1) do not keep the owner alive: that is the documented condition of the
 * opt-in, and why the default is an ordinary Nx.empty. */
#define Block_val(v) (*((void **)Data_custom_val(v)))
static void block_finalize(value v) {
  void *p = Block_val(v);
  Block_val(v) = NULL;
  if (p) (void)smx_host_free(p);
}
static struct custom_operations block_ops = {"soundml.amd.host_block", block_finalize, custom_compare_default,
                                             custom_hash_default, custom_serialize_default,
                                             custom_deserialize_default, custom_compare_ext_default,
                                             custom_fixed_length_default};
/* bytes -> owner; Failure when the library has no page-locked memory to give (the caller falls back to Nx.empty) */
CAMLprim value soundml_amd_host_block(value v_bytes) {
  CAMLparam1(v_bytes);
  CAMLlocal1(v_block);
  const intnat bytes = Long_val(v_bytes);
  if (bytes < 0) caml_invalid_argument("soundml_amd_host_block: a non-negative size");
  void *block = NULL;
  smx_raise(smx_host_alloc((size_t)bytes, &block));
  v_block = caml_alloc_custom_mem(&block_ops, sizeof(void *), (mlsize_t)bytes);
  Block_val(v_block) = block;
  CAMLreturn(v_block);
}
/* a one-dimensional Bigarray of `n` elements of kind `kind` over the owner's block (the owner was sized for it) */
CAMLprim value soundml_amd_host_block_array(value v_block, value v_kind, value v_n) {
  CAMLparam3(v_block, v_kind, v_n);
  const int kind = Int_val(v_kind);
  const intnat n = Long_val(v_n);
  const size_t elem = kind == CAML_BA_FLOAT32 ? 4 : kind == CAML_BA_FLOAT64 ? 8 : kind == CAML_BA_COMPLEX32 ? 8 : kind == CAML_BA_COMPLEX64 ? 16 : 0;
  if (elem == 0 || n < 0 || !Block_val(v_block))
    caml_invalid_argument("soundml_amd_host_block_array: float32 / float64 / complex32 / complex64, a non-negative length and a live block");
  CAMLreturn(caml_ba_alloc_dims(kind | CAML_BA_C_LAYOUT | CAML_BA_EXTERNAL, 1, Block_val(v_block), n));
}
CAMLprim value soundml_amd_host_block_release(value v_block) {
  CAMLparam1(v_block);
  block_finalize(v_block);   /* (idempotent: the owner's own finaliser then finds nothing) */
  CAMLreturn(Val_unit);
}

/* smx_set_devices: the device list of the host-pointer batch calls (one process, several GPUs: clip ranges side by side, one
 * host thread + staging ring pair per device).  An int array of device ordinals; [||] restores the single-device behaviour. */
CAMLprim value soundml_amd_set_devices(value v_ids) {
  CAMLparam1(v_ids);
  const mlsize_t n = Wosize_val(v_ids);
  int ids[64];
  if (n > 64) caml_invalid_argument("soundml_amd_set_devices: at most 64 devices");
  for (mlsize_t i = 0; i < n; ++i) ids[i] = Int_val(Field(v_ids, i));
  smx_raise(smx_set_devices(ids, (int)n));
  CAMLreturn(Val_unit);
}
CAMLprim value soundml_amd_device_count(value v_unit) {
  CAMLparam1(v_unit);
  int n = 0;
  smx_raise(smx_device_count(&n));
  CAMLreturn(Val_int(n));
}

/* Stft.Config.create: window table (float64, win_length points) comes from Window.make on the
 * OCaml side, so every window family of the reference is supported unchanged. */
CAMLprim value soundml_amd_stft_config(value v_fft, value v_win_length, value v_hop, value v_alignment,
                                       value v_pad, value v_pad_value, value v_scale, value v_window) {
  CAMLparam5(v_fft, v_win_length, v_hop, v_alignment, v_pad);
  CAMLxparam3(v_pad_value, v_scale, v_window);
  CAMLlocal1(v_handle);
  const int64_t win_length = Long_val(v_win_length);
  if (ba_kind(v_window) != CAML_BA_FLOAT64 || ba_dim(v_window) < win_length)
    caml_failwith("soundml_amd: window table disagrees with win_length");
  smx_stft_config *c = NULL;
  smx_raise(smx_stft_config_create(Long_val(v_fft), win_length, Long_val(v_hop), Int_val(v_alignment),
                                   Int_val(v_pad), Double_val(v_pad_value), Int_val(v_scale),
                                   SMX_WINDOW_CUSTOM, (const double *)Caml_ba_data_val(v_window), &c));
  /* out-of-heap footprint of a handle: the float64 window and, once used, ~10 device tables (window, twiddles;
   * 100 - 300 KB for the usual sizes): tell the GC, so that dropped handles are finalised (device memory returned)
   * at a rate that matches what they hold */
  v_handle = caml_alloc_custom_mem(&stft_ops, sizeof(smx_stft_config *), (mlsize_t)(64 * 1024 + 40 * Long_val(v_fft)));
  Stft_val(v_handle) = c;
  CAMLreturn(v_handle);
}
CAMLprim value soundml_amd_stft_config_bc(value *argv, int argn) {
  (void)argn;
  return soundml_amd_stft_config(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5], argv[6], argv[7]);
}

/* analyse (stft.ml:356-364) + magnitude_pow (stft.ml:670-674):
 *   mode 0: complex spectrum of frames [p0, p1) -> out [lead; bins; p1-p0] (interleaved re, im)
 *   mode 1: |.|^power                            -> out [lead; bins; p1-p0]
 * x: [lead; n].  Kinds: float32 audio with complex64/float32 out, or float64 with complex128/float64. */
CAMLprim value soundml_amd_stft_range(value v_cfg, value v_x, value v_out, value v_lead, value v_n,
                                      value v_p0, value v_p1, value v_mode, value v_power) {
  CAMLparam5(v_cfg, v_x, v_out, v_lead, v_n);
  CAMLxparam4(v_p0, v_p1, v_mode, v_power);
  const smx_stft_config *c = Stft_val(v_cfg);
  const int64_t lead = Long_val(v_lead), n = Long_val(v_n), p0 = Long_val(v_p0), p1 = Long_val(v_p1);
  const int mode = Int_val(v_mode);
  const double power = Double_val(v_power);
  const int kind = ba_kind(v_x);
  if (kind != CAML_BA_FLOAT32 && kind != CAML_BA_FLOAT64) caml_failwith("soundml_amd: unsupported dtype");
  if (lead < 0 || n < 0 || p0 < 0 || p1 < p0) caml_failwith("soundml_amd: invalid geometry");
  const int64_t bins = smx_stft_config_bins(c);
  const int64_t out_elems = lead * bins * (p1 - p0);
  const int out_kind = ba_kind(v_out);
  const int want_out = mode == 0 ? (kind == CAML_BA_FLOAT32 ? CAML_BA_COMPLEX32 : CAML_BA_COMPLEX64) : kind;
  if (out_kind != want_out) caml_failwith("soundml_amd: output dtype disagrees with the input");
  if (ba_dim(v_x) < lead * n || ba_dim(v_out) < out_elems)
    caml_failwith("soundml_amd: buffer extents disagree with geometry");
  void *x = Caml_ba_data_val(v_x);
  void *out = Caml_ba_data_val(v_out);
  int status;
  caml_release_runtime_system();
  if (mode == 0)
    status = kind == CAML_BA_FLOAT32
                 ? smx_stft_transform_range_f32(c, (const float *)x, lead, n, p0, p1, (float *)out)
                 : smx_stft_transform_range_f64(c, (const double *)x, lead, n, p0, p1, (double *)out);
  else   /* the same frame range, |.|^power fused on the device: out holds exactly [lead; bins; p1 - p0] */
    status = kind == CAML_BA_FLOAT32
                 ? smx_stft_power_range_f32(c, (const float *)x, lead, n, p0, p1, power, (float *)out)
                 : smx_stft_power_range_f64(c, (const double *)x, lead, n, p0, p1, power, (double *)out);
  caml_acquire_runtime_system();
  smx_raise(status);
  CAMLreturn(Val_unit);
}
CAMLprim value soundml_amd_stft_range_bc(value *argv, int argn) {
  (void)argn;
  return soundml_amd_stft_range(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5], argv[6], argv[7], argv[8]);
}

/* Mel.Config: the float64 [n_mels; bins] weights are the config-owned matrix of mel.ml:31-33;
 * the library rebuilds them from the same scalars (bit-compatible construction, mel.ml:67-117). */
CAMLprim value soundml_amd_mel_config(value v_n_mels, value v_sample_rate, value v_fft, value v_f_min,
                                      value v_f_max, value v_scale, value v_norm) {
  CAMLparam5(v_n_mels, v_sample_rate, v_fft, v_f_min, v_f_max);
  CAMLxparam2(v_scale, v_norm);
  CAMLlocal1(v_handle);
  smx_mel_config *c = NULL;
  smx_raise(smx_mel_config_create(Long_val(v_n_mels), Long_val(v_sample_rate), Long_val(v_fft),
                                  Double_val(v_f_min), 1, Double_val(v_f_max), Int_val(v_scale),
                                  Int_val(v_norm), &c));
  v_handle = caml_alloc_custom_mem(&mel_ops, sizeof(smx_mel_config *),
                                   (mlsize_t)(20 * Long_val(v_n_mels) * (Long_val(v_fft) / 2 + 1)));   /* f64 + f32 weights */
  Mel_val(v_handle) = c;
  CAMLreturn(v_handle);
}
CAMLprim value soundml_amd_mel_config_bc(value *argv, int argn) {
  (void)argn;
  return soundml_amd_mel_config(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5], argv[6]);
}

/* Mel.apply (mel.ml:202-231): s [lead; bins; frames] -> out [lead; n_mels; frames] */
CAMLprim value soundml_amd_mel_apply(value v_cfg, value v_s, value v_out, value v_lead, value v_bins,
                                     value v_frames) {
  CAMLparam5(v_cfg, v_s, v_out, v_lead, v_bins);
  CAMLxparam1(v_frames);
  const smx_mel_config *c = Mel_val(v_cfg);
  const int64_t lead = Long_val(v_lead), bins = Long_val(v_bins), frames = Long_val(v_frames);
  const int kind = ba_kind(v_s);
  if ((kind != CAML_BA_FLOAT32 && kind != CAML_BA_FLOAT64) || ba_kind(v_out) != kind)
    caml_failwith("soundml_amd: unsupported or mixed dtypes");
  if (lead < 0 || bins < 0 || frames < 0 || ba_dim(v_s) < lead * bins * frames ||
      ba_dim(v_out) < lead * smx_mel_config_n_mels(c) * frames)
    caml_failwith("soundml_amd: buffer extents disagree with geometry");
  void *s = Caml_ba_data_val(v_s), *out = Caml_ba_data_val(v_out);
  int status;
  caml_release_runtime_system();
  status = kind == CAML_BA_FLOAT32 ? smx_mel_apply_f32(c, (const float *)s, lead, bins, frames, (float *)out)
                                   : smx_mel_apply_f64(c, (const double *)s, lead, bins, frames, (double *)out);
  caml_acquire_runtime_system();
  smx_raise(status);
  CAMLreturn(Val_unit);
}
CAMLprim value soundml_amd_mel_apply_bc(value *argv, int argn) {
  (void)argn;
  return soundml_amd_mel_apply(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5]);
}

/* Soundml.mel_spectrogram (soundml.ml:12-24), fused on the device */
CAMLprim value soundml_amd_mel_spectrogram(value v_stft, value v_mel, value v_x, value v_out, value v_lead,
                                           value v_n, value v_power) {
  CAMLparam5(v_stft, v_mel, v_x, v_out, v_lead);
  CAMLxparam2(v_n, v_power);
  const smx_stft_config *sc = Stft_val(v_stft);
  const smx_mel_config *mc = Mel_val(v_mel);
  const int64_t lead = Long_val(v_lead), n = Long_val(v_n);
  const double power = Double_val(v_power);
  const int kind = ba_kind(v_x);
  if ((kind != CAML_BA_FLOAT32 && kind != CAML_BA_FLOAT64) || ba_kind(v_out) != kind)
    caml_failwith("soundml_amd: unsupported or mixed dtypes");
  int64_t frames = 0;
  smx_raise(smx_stft_frames(sc, n, &frames));
  if (lead < 0 || ba_dim(v_x) < lead * n || ba_dim(v_out) < lead * smx_mel_config_n_mels(mc) * frames)
    caml_failwith("soundml_amd: buffer extents disagree with geometry");
  void *x = Caml_ba_data_val(v_x), *out = Caml_ba_data_val(v_out);
  int status;
  caml_release_runtime_system();
  status = kind == CAML_BA_FLOAT32
               ? smx_mel_spectrogram_f32(sc, mc, (const float *)x, lead, n, power, (float *)out)
               : smx_mel_spectrogram_f64(sc, mc, (const double *)x, lead, n, power, (double *)out);
  caml_acquire_runtime_system();
  smx_raise(status);
  CAMLreturn(Val_unit);
}
CAMLprim value soundml_amd_mel_spectrogram_bc(value *argv, int argn) {
  (void)argn;
  return soundml_amd_mel_spectrogram(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5], argv[6]);
}

/* Stft.invert (stft.ml:902-939): z is the complex spectrum [lead; bins; frames] as a flat Bigarray of
   complex32 / complex64 (interleaved components), out [lead; out_len]; length < 0 = not given */
CAMLprim value soundml_amd_stft_invert(value v_cfg, value v_z, value v_out, value v_lead, value v_bins,
                                       value v_frames, value v_length) {
  CAMLparam5(v_cfg, v_z, v_out, v_lead, v_bins);
  CAMLxparam2(v_frames, v_length);
  const smx_stft_config *c = Stft_val(v_cfg);
  const int64_t lead = Long_val(v_lead), bins = Long_val(v_bins), frames = Long_val(v_frames);
  const int64_t length = Long_val(v_length);
  const int has_length = length >= 0;
  const int zkind = ba_kind(v_z), okind = ba_kind(v_out);
  const int wide = zkind == CAML_BA_COMPLEX64;
  if ((zkind != CAML_BA_COMPLEX32 && zkind != CAML_BA_COMPLEX64) || okind != (wide ? CAML_BA_FLOAT64 : CAML_BA_FLOAT32))
    caml_failwith("soundml_amd: unsupported or mixed dtypes");
  int64_t out_len = length;
  if (!has_length) smx_raise(smx_stft_output_length(c, frames < 0 ? 0 : frames, &out_len));
  if (lead < 0 || bins < 0 || frames < 0 || ba_dim(v_z) < lead * bins * frames || ba_dim(v_out) < lead * out_len)
    caml_failwith("soundml_amd: buffer extents disagree with geometry");
  void *z = Caml_ba_data_val(v_z), *out = Caml_ba_data_val(v_out);
  int status;
  caml_release_runtime_system();
  status = wide ? smx_stft_invert_f64(c, (const double *)z, lead, bins, frames, has_length, length, (double *)out)
                : smx_stft_invert_f32(c, (const float *)z, lead, bins, frames, has_length, length, (float *)out);
  caml_acquire_runtime_system();
  smx_raise(status);
  CAMLreturn(Val_unit);
}
CAMLprim value soundml_amd_stft_invert_bc(value *argv, int argn) {
  (void)argn;
  return soundml_amd_stft_invert(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5], argv[6]);
}

/* Soundml.mfcc (soundml.ml:50-95): lifter < 0 = not given (the OCaml side validates the user's value first) */
CAMLprim value soundml_amd_mfcc(value v_stft, value v_mel, value v_x, value v_out, value v_lead, value v_n,
                                value v_n_mfcc, value v_lifter) {
  CAMLparam5(v_stft, v_mel, v_x, v_out, v_lead);
  CAMLxparam3(v_n, v_n_mfcc, v_lifter);
  const smx_stft_config *sc = Stft_val(v_stft);
  const smx_mel_config *mc = Mel_val(v_mel);
  const int64_t lead = Long_val(v_lead), n = Long_val(v_n), n_mfcc = Long_val(v_n_mfcc);
  const double lifter = Double_val(v_lifter);
  const int has_lifter = lifter >= 0.0;
  const int kind = ba_kind(v_x);
  if ((kind != CAML_BA_FLOAT32 && kind != CAML_BA_FLOAT64) || ba_kind(v_out) != kind)
    caml_failwith("soundml_amd: unsupported or mixed dtypes");
  int64_t frames = 0;
  smx_raise(smx_stft_frames(sc, n, &frames));
  if (lead < 0 || n_mfcc < 0 || ba_dim(v_x) < lead * n || ba_dim(v_out) < lead * n_mfcc * frames)
    caml_failwith("soundml_amd: buffer extents disagree with geometry");
  void *x = Caml_ba_data_val(v_x), *out = Caml_ba_data_val(v_out);
  int status;
  caml_release_runtime_system();
  status = kind == CAML_BA_FLOAT32
               ? smx_mfcc_f32(sc, mc, (const float *)x, lead, n, n_mfcc, has_lifter, lifter, (float *)out)
               : smx_mfcc_f64(sc, mc, (const double *)x, lead, n, n_mfcc, has_lifter, lifter, (double *)out);
  caml_acquire_runtime_system();
  smx_raise(status);
  CAMLreturn(Val_unit);
}
CAMLprim value soundml_amd_mfcc_bc(value *argv, int argn) {
  (void)argn;
  return soundml_amd_mfcc(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5], argv[6], argv[7]);
}

/* Spectral.{centroid,bandwidth,rolloff,flatness} (spectral.ml:171-255): one stub, the feature tagged.
 * v_feature: 0 centroid, 1 bandwidth, 2 roll-off, 3 flatness; v_a: p | roll_percent | amin; v_b: flatness power.
 * The OCaml side keeps check_rank / check_p / ... (same messages) and allocates [...; 1; frames]; freqs is the
 * float64 grid or an empty Bigarray (= the FFT grid); the reused centroid is passed as zero frames when absent. */
CAMLprim value soundml_amd_spectral(value v_feature, value v_s, value v_out, value v_lead, value v_bins,
                                    value v_frames, value v_a, value v_b, value v_freqs, value v_centroid,
                                    value v_sample_rate) {
  CAMLparam5(v_feature, v_s, v_out, v_lead, v_bins);
  CAMLxparam5(v_frames, v_a, v_b, v_freqs, v_centroid);
  CAMLxparam1(v_sample_rate);
  const int feature = Int_val(v_feature);
  const int64_t lead = Long_val(v_lead), bins = Long_val(v_bins), frames = Long_val(v_frames);
  const int64_t sample_rate = Long_val(v_sample_rate);
  const double a = Double_val(v_a), b = Double_val(v_b);
  const int kind = ba_kind(v_s);
  if ((kind != CAML_BA_FLOAT32 && kind != CAML_BA_FLOAT64) || ba_kind(v_out) != kind)
    caml_failwith("soundml_amd: unsupported or mixed dtypes");
  if (lead < 0 || bins < 0 || frames < 0 || ba_dim(v_s) < lead * bins * frames || ba_dim(v_out) < lead * frames)
    caml_failwith("soundml_amd: buffer extents disagree with geometry");
  const int64_t n_freqs = ba_dim(v_freqs);
  if (n_freqs > 0 && ba_kind(v_freqs) != CAML_BA_FLOAT64) caml_failwith("soundml_amd: freqs must be float64");
  const double *freqs = n_freqs > 0 ? (const double *)Caml_ba_data_val(v_freqs) : NULL;
  const int has_centroid = ba_dim(v_centroid) > 0;
  if (has_centroid && (ba_kind(v_centroid) != kind || ba_dim(v_centroid) < lead * frames))
    caml_failwith("soundml_amd: centroid buffer disagrees with the spectrogram");
  const void *cen = has_centroid ? Caml_ba_data_val(v_centroid) : NULL;
  void *s = Caml_ba_data_val(v_s), *out = Caml_ba_data_val(v_out);
  const int f32 = kind == CAML_BA_FLOAT32;
  int status = SMX_FAILURE;
  caml_release_runtime_system();
  switch (feature) {
    case 0:
      status = f32 ? smx_spectral_centroid_f32((const float *)s, lead, bins, frames, freqs, n_freqs, sample_rate, (float *)out)
                   : smx_spectral_centroid_f64((const double *)s, lead, bins, frames, freqs, n_freqs, sample_rate, (double *)out);
      break;
    case 1:
      status = f32 ? smx_spectral_bandwidth_f32((const float *)s, lead, bins, frames, a, freqs, n_freqs, (const float *)cen,
                                                1, frames, sample_rate, (float *)out)
                   : smx_spectral_bandwidth_f64((const double *)s, lead, bins, frames, a, freqs, n_freqs, (const double *)cen,
                                                1, frames, sample_rate, (double *)out);
      break;
    case 2:
      status = f32 ? smx_spectral_rolloff_f32((const float *)s, lead, bins, frames, a, freqs, n_freqs, sample_rate, (float *)out)
                   : smx_spectral_rolloff_f64((const double *)s, lead, bins, frames, a, freqs, n_freqs, sample_rate, (double *)out);
      break;
    case 3:
      status = f32 ? smx_spectral_flatness_f32((const float *)s, lead, bins, frames, a, b, (float *)out)
                   : smx_spectral_flatness_f64((const double *)s, lead, bins, frames, a, b, (double *)out);
      break;
    default: break;
  }
  caml_acquire_runtime_system();
  smx_raise(status);
  CAMLreturn(Val_unit);
}
CAMLprim value soundml_amd_spectral_bc(value *argv, int argn) {
  (void)argn;
  return soundml_amd_spectral(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5], argv[6], argv[7], argv[8], argv[9],
                              argv[10]);
}

/* Chroma.Config handle built from the config's own scalar fields (chroma.ml:177-222); the library rebuilds
 * the float64 weights with the reference's arithmetic (bit-identical on the same libm).  octwidth < 0 = None. */
#define Chroma_val(v) (*((smx_chroma_config **)Data_custom_val(v)))
static void chroma_finalize(value v) { smx_chroma_config_destroy(Chroma_val(v)); }
static struct custom_operations chroma_ops = {"soundml.amd.chroma_config", chroma_finalize, custom_compare_default,
                                              custom_hash_default, custom_serialize_default,
                                              custom_deserialize_default, custom_compare_ext_default,
                                              custom_fixed_length_default};
CAMLprim value soundml_amd_chroma_config(value v_n_chroma, value v_tuning, value v_ctroct, value v_octwidth,
                                         value v_base_c, value v_sample_rate, value v_fft) {
  CAMLparam5(v_n_chroma, v_tuning, v_ctroct, v_octwidth, v_base_c);
  CAMLxparam2(v_sample_rate, v_fft);
  CAMLlocal1(v_handle);
  const double octwidth = Double_val(v_octwidth);
  smx_chroma_config *c = NULL;
  smx_raise(smx_chroma_config_create(Long_val(v_n_chroma), Double_val(v_tuning), Double_val(v_ctroct), octwidth >= 0.0,
                                     octwidth, Bool_val(v_base_c), Long_val(v_sample_rate), Long_val(v_fft), &c));
  v_handle = caml_alloc_custom_mem(&chroma_ops, sizeof(smx_chroma_config *),
                                   (mlsize_t)(20 * Long_val(v_n_chroma) * (Long_val(v_fft) / 2 + 1)));
  Chroma_val(v_handle) = c;
  CAMLreturn(v_handle);
}
CAMLprim value soundml_amd_chroma_config_bc(value *argv, int argn) {
  (void)argn;
  return soundml_amd_chroma_config(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5], argv[6]);
}

/* Chroma.apply (chroma.ml:285-317): v_norm 0 none, 1 inf, 2 the exponent v_p */
CAMLprim value soundml_amd_chroma_apply(value v_cfg, value v_s, value v_out, value v_lead, value v_bins,
                                        value v_frames, value v_norm, value v_p) {
  CAMLparam5(v_cfg, v_s, v_out, v_lead, v_bins);
  CAMLxparam3(v_frames, v_norm, v_p);
  const smx_chroma_config *c = Chroma_val(v_cfg);
  const int64_t lead = Long_val(v_lead), bins = Long_val(v_bins), frames = Long_val(v_frames);
  const int64_t rows = smx_chroma_config_n_chroma(c);
  const int kind = ba_kind(v_s);
  if ((kind != CAML_BA_FLOAT32 && kind != CAML_BA_FLOAT64) || ba_kind(v_out) != kind)
    caml_failwith("soundml_amd: unsupported or mixed dtypes");
  if (lead < 0 || bins < 0 || frames < 0 || ba_dim(v_s) < lead * bins * frames || ba_dim(v_out) < lead * rows * frames)
    caml_failwith("soundml_amd: buffer extents disagree with geometry");
  void *s = Caml_ba_data_val(v_s), *out = Caml_ba_data_val(v_out);
  const int norm = Int_val(v_norm);
  const double p = Double_val(v_p);
  int status;
  caml_release_runtime_system();
  status = kind == CAML_BA_FLOAT32 ? smx_chroma_apply_f32(c, (const float *)s, lead, bins, frames, norm, p, (float *)out)
                                   : smx_chroma_apply_f64(c, (const double *)s, lead, bins, frames, norm, p, (double *)out);
  caml_acquire_runtime_system();
  smx_raise(status);
  CAMLreturn(Val_unit);
}
CAMLprim value soundml_amd_chroma_apply_bc(value *argv, int argn) {
  (void)argn;
  return soundml_amd_chroma_apply(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5], argv[6], argv[7]);
}

/* Convert.power_to_db / amplitude_to_db (convert.ml:52-62): v_amplitude selects the amplitude form; top_db < 0 = None
 * (the OCaml side validates the user's values first, with the reference's messages; the C side repeats the checks). */
CAMLprim value soundml_amd_to_db(value v_amplitude, value v_s, value v_out, value v_reference, value v_amin, value v_top_db) {
  CAMLparam5(v_amplitude, v_s, v_out, v_reference, v_amin);
  CAMLxparam1(v_top_db);
  const int kind = ba_kind(v_s);
  if ((kind != CAML_BA_FLOAT32 && kind != CAML_BA_FLOAT64) || ba_kind(v_out) != kind)
    caml_failwith("soundml_amd: unsupported or mixed dtypes");
  const int64_t total = ba_dim(v_s);
  if (ba_dim(v_out) < total) caml_failwith("soundml_amd: buffer extents disagree with geometry");
  const double reference = Double_val(v_reference), amin = Double_val(v_amin), top_db = Double_val(v_top_db);
  const int has_top = top_db >= 0.0, amplitude = Bool_val(v_amplitude);
  void *s = Caml_ba_data_val(v_s), *out = Caml_ba_data_val(v_out);
  int status;
  caml_release_runtime_system();
  if (kind == CAML_BA_FLOAT32)
    status = amplitude ? smx_amplitude_to_db_f32((const float *)s, total, reference, amin, has_top, top_db, (float *)out)
                       : smx_power_to_db_f32((const float *)s, total, reference, amin, has_top, top_db, (float *)out);
  else
    status = amplitude ? smx_amplitude_to_db_f64((const double *)s, total, reference, amin, has_top, top_db, (double *)out)
                       : smx_power_to_db_f64((const double *)s, total, reference, amin, has_top, top_db, (double *)out);
  caml_acquire_runtime_system();
  smx_raise(status);
  CAMLreturn(Val_unit);
}
CAMLprim value soundml_amd_to_db_bc(value *argv, int argn) {
  (void)argn;
  return soundml_amd_to_db(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5]);
}


/* ---- Stft.Kernel (stft.ml:366-622) and Stft.power_stage's body (stft.ml:1364-1409): the streaming face --------------
 * The device-resident state machine of the library: carry, prelude and tail live in device memory; a step uploads one
 * chunk and downloads the frames that completed.  Mutable and single-owner like the reference's kernels
 * (stft.mli:436-437): the OCaml value owns it, the finaliser destroys it.  power < 0 = the complex face. */
#define Kernel_val(v) (*((smx_stft_kernel **)Data_custom_val(v)))
static void kernel_finalize(value v) { smx_stft_kernel_destroy(Kernel_val(v)); }
static struct custom_operations kernel_ops = {"soundml.amd.stft_kernel", kernel_finalize, custom_compare_default,
                                              custom_hash_default, custom_serialize_default,
                                              custom_deserialize_default, custom_compare_ext_default,
                                              custom_fixed_length_default};

CAMLprim value soundml_amd_kernel_prepare(value v_cfg, value v_wide, value v_channels, value v_max_block, value v_power) {
  CAMLparam5(v_cfg, v_wide, v_channels, v_max_block, v_power);
  CAMLlocal1(v_handle);
  const smx_stft_config *c = Stft_val(v_cfg);
  const int dtype_bytes = Bool_val(v_wide) ? 8 : 4;
  const double power = Double_val(v_power);
  smx_stft_kernel *k = NULL;
  if (power < 0.0)
    smx_raise(smx_stft_kernel_prepare(c, dtype_bytes, Long_val(v_channels), Long_val(v_max_block), &k));
  else
    smx_raise(smx_stft_kernel_prepare_power(c, dtype_bytes, Long_val(v_channels), Long_val(v_max_block), power, &k));
  v_handle = caml_alloc_custom_mem(&kernel_ops, sizeof(smx_stft_kernel *),
                                   (mlsize_t)(Long_val(v_channels) * 4 * smx_stft_config_fft_size(c) * dtype_bytes));
  Kernel_val(v_handle) = k;
  CAMLreturn(v_handle);
}

CAMLprim value soundml_amd_kernel_frame_bound(value v_k) {
  CAMLparam1(v_k);
  int64_t bound = 0;
  smx_raise(smx_stft_kernel_frame_bound(Kernel_val(v_k), &bound));
  CAMLreturn(Val_long(bound));
}

/* Kernel.step / Kernel.flush: chunk [channels; m] (m = 0 and is_flush = true for the drain) -> frames written into
 * out [channels; bins; capacity] (complex or real by the kernel's face); returns the number of frames emitted, which the
 * OCaml side slices off (0 = the reference's None). */
CAMLprim value soundml_amd_kernel_step(value v_k, value v_chunk, value v_out, value v_channels, value v_m,
                                       value v_capacity, value v_is_flush) {
  CAMLparam5(v_k, v_chunk, v_out, v_channels, v_m);
  CAMLxparam2(v_capacity, v_is_flush);
  smx_stft_kernel *k = Kernel_val(v_k);
  const int64_t channels = Long_val(v_channels), m = Long_val(v_m), capacity = Long_val(v_capacity);
  const int is_flush = Bool_val(v_is_flush);
  if (channels < 1 || m < 0 || capacity < 0) caml_failwith("soundml_amd: invalid geometry");
  /* The library reads and writes the rows of the channel count the KERNEL holds, whatever the caller passes: follow the
   * first chunk of a stream as the reference's state does (stft.ml:521-559), refuse a later disagreement (the reference
   * fails in its concatenation), and only then trust the extents. */
  int64_t have = 0;
  smx_raise(smx_stft_kernel_channels(k, &have));
  if (have != channels) {
    if (is_flush) caml_invalid_argument("flush: the kernel's stream has another channel count");
    smx_raise(smx_stft_kernel_set_channels(k, channels));   /* SMX_INVALID_ARGUMENT -> Invalid_argument once samples were fed */
  }
  if (!is_flush && ba_dim(v_chunk) < channels * m) caml_failwith("soundml_amd: buffer extents disagree with geometry");
  int64_t bound = 0;
  smx_raise(smx_stft_kernel_frame_bound(k, &bound));
  if (capacity < bound) caml_failwith("soundml_amd: output capacity below the kernel's frame bound");
  if (ba_dim(v_out) < channels * smx_stft_config_bins(smx_stft_kernel_config(k)) * capacity)
    caml_failwith("soundml_amd: output extents disagree with geometry");
  /* out is complex (2 components per element) or real; either way a Bigarray element is one spectrum value */
  void *chunk = is_flush ? NULL : Caml_ba_data_val(v_chunk);
  void *out = Caml_ba_data_val(v_out);
  int64_t emitted = 0;
  int status;
  caml_release_runtime_system();
  status = is_flush ? smx_stft_kernel_flush(k, out, capacity, &emitted) : smx_stft_kernel_step(k, chunk, m, out, capacity, &emitted);
  caml_acquire_runtime_system();
  smx_raise(status);
  CAMLreturn(Val_long(emitted));
}
CAMLprim value soundml_amd_kernel_step_bc(value *argv, int argn) {
  (void)argn;
  return soundml_amd_kernel_step(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5], argv[6]);
}

CAMLprim value soundml_amd_kernel_reset(value v_k) {
  CAMLparam1(v_k);
  smx_raise(smx_stft_kernel_reset(Kernel_val(v_k)));
  CAMLreturn(Val_unit);
}

/* ---- Stft.Synthesis (stft.ml:1271-1298) and the body of synthesis_stage (stft.ml:1417-1442): incremental synthesis -------
 * State (the last ceil(fft/hop) - 1 spectra, the held samples) lives in device memory; a step uploads one chunk of frames
 * [channels; bins; k] and downloads the samples they settle.  Mutable and single-owner like the reference's kernel
 * (stft.mli:520-522): the OCaml value owns it, the finaliser destroys it. */
#define Synthesis_val(v) (*((smx_stft_synthesis **)Data_custom_val(v)))
static void synthesis_finalize(value v) { smx_stft_synthesis_destroy(Synthesis_val(v)); }
static struct custom_operations synthesis_ops = {"soundml.amd.stft_synthesis", synthesis_finalize, custom_compare_default,
                                                 custom_hash_default, custom_serialize_default,
                                                 custom_deserialize_default, custom_compare_ext_default,
                                                 custom_fixed_length_default};

CAMLprim value soundml_amd_synthesis_prepare(value v_cfg, value v_wide, value v_channels, value v_max_block) {
  CAMLparam4(v_cfg, v_wide, v_channels, v_max_block);
  CAMLlocal1(v_handle);
  const smx_stft_config *c = Stft_val(v_cfg);
  smx_stft_synthesis *s = NULL;
  smx_raise(smx_stft_synthesis_prepare(c, Bool_val(v_wide) ? 8 : 4, Long_val(v_channels), Long_val(v_max_block), &s));
  v_handle = caml_alloc_custom_mem(&synthesis_ops, sizeof(smx_stft_synthesis *),
                                   (mlsize_t)(Long_val(v_channels) * 8 * smx_stft_config_fft_size(c) * (Bool_val(v_wide) ? 8 : 4)));
  Synthesis_val(v_handle) = s;
  CAMLreturn(v_handle);
}

CAMLprim value soundml_amd_synthesis_numbers(value v_cfg, value v_s) {   /* (Config.synthesis_latency, sample bound of a step) */
  CAMLparam2(v_cfg, v_s);
  CAMLlocal1(v_pair);
  int64_t bound = 0;
  smx_raise(smx_stft_synthesis_sample_bound(Synthesis_val(v_s), &bound));
  v_pair = caml_alloc_tuple(2);
  Store_field(v_pair, 0, Val_long(smx_stft_synthesis_latency(Stft_val(v_cfg))));
  Store_field(v_pair, 1, Val_long(bound));
  CAMLreturn(v_pair);
}

/* Synthesis.step / flush: frames z [channels; bins; k] (k = 0 and is_flush = true for the drain) -> samples written into
 * out [channels; capacity]; returns how many per channel (0 = the reference's None).  The prepared channel count is the
 * extent both buffers are checked against (the library reads and writes exactly that many rows). */
CAMLprim value soundml_amd_synthesis_step(value v_s, value v_z, value v_out, value v_channels, value v_bins, value v_k,
                                          value v_capacity, value v_is_flush, value v_wide) {
  CAMLparam5(v_s, v_z, v_out, v_channels, v_bins);
  CAMLxparam4(v_k, v_capacity, v_is_flush, v_wide);
  smx_stft_synthesis *s = Synthesis_val(v_s);
  const int64_t channels = Long_val(v_channels), bins = Long_val(v_bins), k = Long_val(v_k), capacity = Long_val(v_capacity);
  const int is_flush = Bool_val(v_is_flush), wide = Bool_val(v_wide);
  if (channels < 1 || bins < 0 || k < 0 || capacity < 0) caml_failwith("soundml_amd: invalid geometry");
  /* the element sizes the kernel was prepared for (as soundml_amd_stft_invert checks them): a complex64 chunk given to a
     float64 kernel would be read past its end, a complex128 one given to a float32 kernel reinterpreted */
  if (!is_flush && ba_kind(v_z) != (wide ? CAML_BA_COMPLEX64 : CAML_BA_COMPLEX32)) caml_failwith("soundml_amd: unsupported or mixed dtypes");
  if (ba_kind(v_out) != (wide ? CAML_BA_FLOAT64 : CAML_BA_FLOAT32)) caml_failwith("soundml_amd: unsupported or mixed dtypes");
  if (!is_flush && ba_dim(v_z) < channels * bins * k) caml_failwith("soundml_amd: buffer extents disagree with geometry");
  if (ba_dim(v_out) < channels * capacity) caml_failwith("soundml_amd: output extents disagree with geometry");
  void *z = is_flush ? NULL : Caml_ba_data_val(v_z);
  void *out = Caml_ba_data_val(v_out);
  int64_t emitted = 0;
  int status;
  caml_release_runtime_system();
  status = is_flush ? smx_stft_synthesis_flush(s, out, capacity, &emitted)
                    : smx_stft_synthesis_step(s, z, bins, k, out, capacity, &emitted);
  caml_acquire_runtime_system();
  smx_raise(status);
  CAMLreturn(Val_long(emitted));
}
CAMLprim value soundml_amd_synthesis_step_bc(value *argv, int argn) {
  (void)argn;
  return soundml_amd_synthesis_step(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5], argv[6], argv[7], argv[8]);
}

CAMLprim value soundml_amd_synthesis_reset(value v_s) {
  CAMLparam1(v_s);
  smx_raise(smx_stft_synthesis_reset(Synthesis_val(v_s)));
  CAMLreturn(Val_unit);
}

CAMLprim value soundml_amd_stage_numbers(value v_cfg, value v_max_items) {   /* (stage_latency, frame_bound max_items) */
  CAMLparam2(v_cfg, v_max_items);
  CAMLlocal1(v_pair);
  const smx_stft_config *c = Stft_val(v_cfg);
  v_pair = caml_alloc_tuple(2);
  Store_field(v_pair, 0, Val_long(smx_stft_stage_latency(c)));
  Store_field(v_pair, 1, Val_long(smx_stft_frame_bound(c, Long_val(v_max_items))));
  CAMLreturn(v_pair);
}

/* Stft.griffin_lim (stft.ml:941-1017): magnitudes s [lead; bins; frames] -> signal [lead; out_len]; the whole loop on the
 * device.  init: an empty Bigarray = the reference's default random phase; length < 0 = not given. */
CAMLprim value soundml_amd_griffin_lim(value v_cfg, value v_s, value v_init, value v_out, value v_lead, value v_bins,
                                       value v_frames, value v_n_iter, value v_momentum, value v_length) {
  CAMLparam5(v_cfg, v_s, v_init, v_out, v_lead);
  CAMLxparam5(v_bins, v_frames, v_n_iter, v_momentum, v_length);
  const smx_stft_config *c = Stft_val(v_cfg);
  const int64_t lead = Long_val(v_lead), bins = Long_val(v_bins), frames = Long_val(v_frames);
  const int64_t n_iter = Long_val(v_n_iter), length = Long_val(v_length);
  const double momentum = Double_val(v_momentum);
  const int has_length = length >= 0;
  const int kind = ba_kind(v_s);
  if ((kind != CAML_BA_FLOAT32 && kind != CAML_BA_FLOAT64) || ba_kind(v_out) != kind)
    caml_failwith("soundml_amd: unsupported or mixed dtypes");
  int64_t out_len = length;
  if (!has_length) smx_raise(smx_stft_output_length(c, frames < 0 ? 0 : frames, &out_len));
  const int has_init = ba_dim(v_init) > 0;
  if (lead < 0 || bins < 0 || frames < 0 || ba_dim(v_s) < lead * bins * frames || ba_dim(v_out) < lead * out_len ||
      (has_init && (ba_kind(v_init) != kind || ba_dim(v_init) < lead * bins * frames)))
    caml_failwith("soundml_amd: buffer extents disagree with geometry");
  void *s = Caml_ba_data_val(v_s), *out = Caml_ba_data_val(v_out);
  void *init = has_init ? Caml_ba_data_val(v_init) : NULL;
  int status;
  caml_release_runtime_system();
  status = kind == CAML_BA_FLOAT32
               ? smx_stft_griffin_lim_f32(c, (const float *)s, lead, bins, frames, n_iter, momentum, (const float *)init,
                                          has_length, length, (float *)out)
               : smx_stft_griffin_lim_f64(c, (const double *)s, lead, bins, frames, n_iter, momentum, (const double *)init,
                                          has_length, length, (double *)out);
  caml_acquire_runtime_system();
  smx_raise(status);
  CAMLreturn(Val_unit);
}
CAMLprim value soundml_amd_griffin_lim_bc(value *argv, int argn) {
  (void)argn;
  return soundml_amd_griffin_lim(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5], argv[6], argv[7], argv[8], argv[9]);
}

/* Soundml.chroma_stft (soundml.ml:97-107), fused power spectrogram + projection + norm */
CAMLprim value soundml_amd_chroma_stft(value v_stft, value v_chroma, value v_x, value v_out, value v_lead, value v_n,
                                       value v_power, value v_norm, value v_p) {
  CAMLparam5(v_stft, v_chroma, v_x, v_out, v_lead);
  CAMLxparam4(v_n, v_power, v_norm, v_p);
  const smx_stft_config *sc = Stft_val(v_stft);
  const smx_chroma_config *cc = Chroma_val(v_chroma);
  const int64_t lead = Long_val(v_lead), n = Long_val(v_n);
  const int kind = ba_kind(v_x);
  if ((kind != CAML_BA_FLOAT32 && kind != CAML_BA_FLOAT64) || ba_kind(v_out) != kind)
    caml_failwith("soundml_amd: unsupported or mixed dtypes");
  int64_t frames = 0;
  smx_raise(smx_stft_frames(sc, n, &frames));
  if (lead < 0 || ba_dim(v_x) < lead * n || ba_dim(v_out) < lead * smx_chroma_config_n_chroma(cc) * frames)
    caml_failwith("soundml_amd: buffer extents disagree with geometry");
  void *x = Caml_ba_data_val(v_x), *out = Caml_ba_data_val(v_out);
  const double power = Double_val(v_power), p = Double_val(v_p);
  const int norm = Int_val(v_norm);
  int status;
  caml_release_runtime_system();
  status = kind == CAML_BA_FLOAT32 ? smx_chroma_stft_f32(sc, cc, (const float *)x, lead, n, power, norm, p, (float *)out)
                                   : smx_chroma_stft_f64(sc, cc, (const double *)x, lead, n, power, norm, p, (double *)out);
  caml_acquire_runtime_system();
  smx_raise(status);
  CAMLreturn(Val_unit);
}
CAMLprim value soundml_amd_chroma_stft_bc(value *argv, int argn) {
  (void)argn;
  return soundml_amd_chroma_stft(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5], argv[6], argv[7], argv[8]);
}

/* ---- FIR block convolution (BASELINE config 4; the reference lists the module as planned) ---------------------------- */
#define Fir_val(v) (*((smx_fir_plan **)Data_custom_val(v)))
static void fir_finalize(value v) { smx_fir_plan_destroy(Fir_val(v)); }
static struct custom_operations fir_ops = {"soundml.amd.fir_plan", fir_finalize, custom_compare_default,
                                           custom_hash_default, custom_serialize_default,
                                           custom_deserialize_default, custom_compare_ext_default,
                                           custom_fixed_length_default};
CAMLprim value soundml_amd_fir_plan(value v_taps) {   /* taps: float64 Bigarray */
  CAMLparam1(v_taps);
  CAMLlocal1(v_handle);
  if (ba_kind(v_taps) != CAML_BA_FLOAT64) caml_failwith("soundml_amd: taps must be float64");
  smx_fir_plan *p = NULL;
  smx_raise(smx_fir_plan_create((const double *)Caml_ba_data_val(v_taps), ba_dim(v_taps), &p));
  v_handle = caml_alloc_custom_mem(&fir_ops, sizeof(smx_fir_plan *), (mlsize_t)(16 * smx_fir_plan_block(p)));
  Fir_val(v_handle) = p;
  CAMLreturn(v_handle);
}
CAMLprim value soundml_amd_fir_apply(value v_plan, value v_x, value v_y, value v_channels, value v_n) {
  CAMLparam5(v_plan, v_x, v_y, v_channels, v_n);
  const smx_fir_plan *p = Fir_val(v_plan);
  const int64_t channels = Long_val(v_channels), n = Long_val(v_n);
  if (ba_kind(v_x) != CAML_BA_FLOAT32 || ba_kind(v_y) != CAML_BA_FLOAT32) caml_failwith("soundml_amd: unsupported dtype");
  if (channels < 0 || n < 0 || ba_dim(v_x) < channels * n || ba_dim(v_y) < channels * n)
    caml_failwith("soundml_amd: buffer extents disagree with geometry");
  const float *x = (const float *)Caml_ba_data_val(v_x);
  float *y = (float *)Caml_ba_data_val(v_y);
  int status;
  caml_release_runtime_system();
  status = smx_fir_apply_f32(p, x, channels, n, y);
  caml_acquire_runtime_system();
  smx_raise(status);
  CAMLreturn(Val_unit);
}

/* ---- Resample: the executor's two numeric pieces (resample.ml:1186-1196, 1456-1599) ----------------------------------
 * soundml_amd_resample_shape has EXACTLY the signature of the reference's soundml_resample_shape (resample_stubs.c:377-
 * 415): spectra, plan spectrum, destination (complex64-kind Bigarrays = interleaved complex128), lines, N, L, M -- so
 *     external resample_shape_c : ... = "soundml_amd_resample_shape_bc" "soundml_amd_resample_shape"
 * is the whole change in resample.ml.  Same validation, same messages, same bits. */
CAMLprim value soundml_amd_resample_shape(value v_x, value v_h, value v_y, value v_lines, value v_n, value v_sl, value v_sm) {
  CAMLparam5(v_x, v_h, v_y, v_lines, v_n);
  CAMLxparam2(v_sl, v_sm);
  const int64_t lines = Long_val(v_lines), n = Long_val(v_n), sl = Long_val(v_sl), sm = Long_val(v_sm);
  if (lines < 0 || n < 2 || (n % 2) != 0 || sl < 1 || sm < 1 || (sl > 1 && sm > 1))
    caml_failwith("soundml_resample_shape: invalid geometry");
  const int64_t w = sl > 1 ? n * sl : (sm > 1 ? n / sm : n);
  if (w < 2 || (sm > 1 && (n % sm) != 0)) caml_failwith("soundml_resample_shape: invalid geometry");
  if (ba_kind(v_x) != CAML_BA_COMPLEX64 || ba_kind(v_h) != CAML_BA_COMPLEX64 || ba_kind(v_y) != CAML_BA_COMPLEX64)
    caml_failwith("soundml_resample_shape: unsupported dtype");
  const int64_t bins = (n / 2) + 1, obins = (w / 2) + 1;
  if (ba_dim(v_x) < lines * bins || ba_dim(v_h) < (sl > 1 ? obins : bins) || ba_dim(v_y) < lines * obins)
    caml_failwith("soundml_resample_shape: buffer extents disagree");
  const double *x = (const double *)Caml_ba_data_val(v_x), *h = (const double *)Caml_ba_data_val(v_h);
  double *y = (double *)Caml_ba_data_val(v_y);
  int status;
  caml_release_runtime_system();
  status = smx_resample_shape_c128(x, h, y, lines, n, sl, sm);
  caml_acquire_runtime_system();
  smx_raise(status);
  CAMLreturn(Val_unit);
}
CAMLprim value soundml_amd_resample_shape_bc(value *argv, int argn) {
  (void)argn;
  return soundml_amd_resample_shape(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5], argv[6]);
}

#define Stage_val(v) (*((smx_resample_stage **)Data_custom_val(v)))
static void stage_finalize(value v) { smx_resample_stage_destroy(Stage_val(v)); }
static struct custom_operations stage_ops = {"soundml.amd.resample_stage", stage_finalize, custom_compare_default,
                                             custom_hash_default, custom_serialize_default,
                                             custom_deserialize_default, custom_compare_ext_default,
                                             custom_fixed_length_default};
CAMLprim value soundml_amd_resample_stage(value v_proto, value v_l, value v_m, value v_k) {   /* proto: float64, 2 K L + 1 */
  CAMLparam4(v_proto, v_l, v_m, v_k);
  CAMLlocal1(v_handle);
  const int64_t l = Long_val(v_l), m = Long_val(v_m), k = Long_val(v_k);
  if (ba_kind(v_proto) != CAML_BA_FLOAT64 || l < 1 || k < 0 || ba_dim(v_proto) != 2 * k * l + 1)
    caml_failwith("soundml_amd: prototype disagrees with the stage");
  smx_resample_stage *s = NULL;
  smx_raise(smx_resample_stage_create((const double *)Caml_ba_data_val(v_proto), l, m, k, &s));
  v_handle = caml_alloc_custom_mem(&stage_ops, sizeof(smx_resample_stage *), (mlsize_t)(32 * (2 * k * l + 1)));
  Stage_val(v_handle) = s;
  CAMLreturn(v_handle);
}
/* one whole stage over planar [channels; n] -> [channels; ceil(n L / M)] (what ols_run + drain emit for an offline apply) */
CAMLprim value soundml_amd_resample_stage_apply(value v_stage, value v_x, value v_y, value v_channels, value v_n) {
  CAMLparam5(v_stage, v_x, v_y, v_channels, v_n);
  const smx_resample_stage *s = Stage_val(v_stage);
  const int64_t channels = Long_val(v_channels), n = Long_val(v_n);
  if (ba_kind(v_x) != CAML_BA_FLOAT32 || ba_kind(v_y) != CAML_BA_FLOAT32) caml_failwith("soundml_amd: unsupported dtype");
  const int64_t n_out = smx_resample_stage_out_length(s, n < 0 ? 0 : n);
  if (channels < 0 || n < 0 || ba_dim(v_x) < channels * n || ba_dim(v_y) < channels * n_out)
    caml_failwith("soundml_amd: buffer extents disagree with geometry");
  const float *x = (const float *)Caml_ba_data_val(v_x);
  float *y = (float *)Caml_ba_data_val(v_y);
  int status;
  caml_release_runtime_system();
  status = smx_resample_stage_apply_f32(s, x, channels, n, y);
  caml_acquire_runtime_system();
  smx_raise(status);
  CAMLreturn(Val_unit);
}

/* Resample.Kernel of one overlap-save stage (resample.mli:270-319): the block carry lives on the device.  The kernel value keeps
 * its stage alive (field 1 of the pair the OCaml side holds). */
#define Rkernel_val(v) (*((smx_resample_kernel **)Data_custom_val(v)))
static void rkernel_finalize(value v) { smx_resample_kernel_destroy(Rkernel_val(v)); }
static struct custom_operations rkernel_ops = {"soundml.amd.resample_kernel", rkernel_finalize, custom_compare_default,
                                               custom_hash_default, custom_serialize_default,
                                               custom_deserialize_default, custom_compare_ext_default,
                                               custom_fixed_length_default};
CAMLprim value soundml_amd_resample_kernel_prepare(value v_stage, value v_channels, value v_max_block) {
  CAMLparam3(v_stage, v_channels, v_max_block);
  CAMLlocal1(v_handle);
  smx_resample_kernel *k = NULL;
  smx_raise(smx_resample_kernel_prepare(Stage_val(v_stage), Long_val(v_channels), Long_val(v_max_block), &k));
  v_handle = caml_alloc_custom_mem(&rkernel_ops, sizeof(smx_resample_kernel *), (mlsize_t)4096);
  Rkernel_val(v_handle) = k;
  CAMLreturn(v_handle);
}
/* (out_bound n, pending): the capacities the OCaml side allocates for a step of n samples and for the flush */
CAMLprim value soundml_amd_resample_kernel_bounds(value v_k, value v_n) {
  CAMLparam2(v_k, v_n);
  CAMLlocal1(v_pair);
  v_pair = caml_alloc_tuple(2);
  Store_field(v_pair, 0, Val_long(smx_resample_kernel_out_bound(Rkernel_val(v_k), Long_val(v_n))));
  Store_field(v_pair, 1, Val_long(smx_resample_kernel_pending(Rkernel_val(v_k))));
  CAMLreturn(v_pair);
}
/* step (is_flush = false): x [channels; n] -> out [channels; capacity]; returns the samples emitted per channel */
CAMLprim value soundml_amd_resample_kernel_step(value v_k, value v_x, value v_out, value v_channels, value v_n, value v_capacity,
                                                value v_is_flush) {
  CAMLparam5(v_k, v_x, v_out, v_channels, v_n);
  CAMLxparam2(v_capacity, v_is_flush);
  smx_resample_kernel *k = Rkernel_val(v_k);
  const int64_t channels = Long_val(v_channels), n = Long_val(v_n), capacity = Long_val(v_capacity);
  const int is_flush = Bool_val(v_is_flush);
  if (ba_kind(v_out) != CAML_BA_FLOAT32 || (!is_flush && ba_kind(v_x) != CAML_BA_FLOAT32)) caml_failwith("soundml_amd: unsupported dtype");
  if (channels < 1 || n < 0 || capacity < 0) caml_failwith("soundml_amd: invalid geometry");
  if (!is_flush && ba_dim(v_x) < channels * n) caml_failwith("soundml_amd: buffer extents disagree with geometry");
  if (ba_dim(v_out) < channels * capacity) caml_failwith("soundml_amd: output extents disagree with geometry");
  const float *x = is_flush ? NULL : (const float *)Caml_ba_data_val(v_x);
  float *out = (float *)Caml_ba_data_val(v_out);
  int64_t emitted = 0;
  int status;
  caml_release_runtime_system();
  status = is_flush ? smx_resample_kernel_flush_f32(k, out, capacity, &emitted)
                    : smx_resample_kernel_step_f32(k, x, n, n > 0 ? n : 1, out, capacity, &emitted);
  caml_acquire_runtime_system();
  smx_raise(status);
  CAMLreturn(Val_long(emitted));
}
CAMLprim value soundml_amd_resample_kernel_step_bc(value *argv, int argn) {
  (void)argn;
  return soundml_amd_resample_kernel_step(argv[0], argv[1], argv[2], argv[3], argv[4], argv[5], argv[6]);
}
CAMLprim value soundml_amd_resample_kernel_reset(value v_k) {
  CAMLparam1(v_k);
  smx_raise(smx_resample_kernel_reset(Rkernel_val(v_k)));
  CAMLreturn(Val_unit);
}
