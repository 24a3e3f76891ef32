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
  v_handle = caml_alloc_custom(&stft_ops, sizeof(smx_stft_config *), 0, 1);
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
  else   /* the host entry points analyse every frame; [p0, p1) must be the whole grid here */
    status = kind == CAML_BA_FLOAT32
                 ? smx_stft_power_spectrum_f32(c, (const float *)x, lead, n, power, (float *)out)
                 : smx_stft_power_spectrum_f64(c, (const double *)x, lead, n, power, (double *)out);
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
  v_handle = caml_alloc_custom(&mel_ops, sizeof(smx_mel_config *), 0, 1);
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
  v_handle = caml_alloc_custom(&chroma_ops, sizeof(smx_chroma_config *), 0, 1);
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
