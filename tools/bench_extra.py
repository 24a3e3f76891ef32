#!/usr/bin/env python3
"""Secondary measurements (not the bench.py contract): BASELINE configs C1 (plumbing), C3 (mel), C4 (FIR).
Prints one JSON line per config; parity of each against the oracle is checked on a subset in the same run."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import soundml_amd as S
from soundml_amd import Stft, Mel, Fir
from oracle import soundml_oracle as O


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def rel_err(got, want):
    kind = np.complex128 if np.iscomplexobj(want) or np.iscomplexobj(got) else np.float64
    want = np.asarray(want, kind)
    return float(np.max(np.abs(np.asarray(got, kind) - want)) / np.max(np.abs(want)))


# C1: one 10 s mono 44.1 kHz 440 Hz sine, fft 1024 hop 256 (the reference's example / plumbing config)
n = 441000
x1 = np.sin(2 * np.pi * 440.0 * np.arange(n) / 44100.0).astype(np.float32)
c1 = Stft.Config.create(fft_size=1024, hop=256)
t0 = time.perf_counter(); p1 = Stft.power_spectrum(c1, x1); host_ms = (time.perf_counter() - t0) * 1e3
want = O.power_spectrum(O.stft_config(1024, hop=256), x1)
x1d = torch.from_numpy(x1).cuda()
med, mn = timeit(lambda: Stft.power_spectrum(c1, x1d))
print(json.dumps({"config": "C1", "shape": list(p1.shape), "host_call_ms": round(host_ms, 3), "device_ms": round(med, 4),
                  "peak_bin": int(np.argmax(p1[:, 800])), "max_rel_err_vs_oracle": rel_err(p1, want)}))

# C3: mel spectrogram (128 mels) on the C2 batch
torch.manual_seed(42)
x = torch.rand(256, 480000, device="cuda") * 2 - 1
sc = Stft.Config.create(fft_size=2048, hop=512)
mc = Mel.Config.create(n_mels=128, sample_rate=48000, fft_size=2048)
frames = Stft.frames(sc, 480000)
# timed through the C ABI on preallocated device buffers (the Python mirror allocates its result per call)
import ctypes
from soundml_amd._lib import lib, check
vp = ctypes.c_void_p
out_mel = torch.empty(256, 128, frames, device="cuda")
med, mn = timeit(lambda: check(lib.smx_mel_spectrogram_f32_dev(sc._h, mc._h, vp(x.data_ptr()), 256, 480000, 480000, 2.0,
                                                               vp(out_mel.data_ptr()), None)), reps=20)
out_c = torch.empty(256, 1025, frames, 2, device="cuda")
med_c, _ = timeit(lambda: check(lib.smx_stft_transform_range_f32_dev(sc._h, vp(x.data_ptr()), 256, 480000, 480000, 0, frames,
                                                                    vp(out_c.data_ptr()), None)), reps=20)
zc = torch.view_as_complex(out_c[:2]).cpu().numpy()
wz = O.transform(O.stft_config(2048, hop=512), x[:2].cpu().numpy())
print(json.dumps({"config": "C2 complex (Stft.transform)", "transform_ms": round(med_c, 4), "Mframes_per_s": round(256 * frames / med_c / 1e3, 1),
                  "GBs_algorithmic": round(256 * frames * 10248 / med_c / 1e6, 1), "max_rel_err_vs_oracle": rel_err(zc, wz)}))
out_x = torch.empty(256, 480000, device="cuda")
med_i, _ = timeit(lambda: check(lib.smx_stft_invert_f32_dev(sc._h, vp(out_c.data_ptr()), 256, 1025, frames, 1, 480000,
                                                            vp(out_x.data_ptr()), None)), reps=10)
print(json.dumps({"config": "C2 inverse (Stft.invert of the transform)", "invert_ms": round(med_i, 4),
                  "Mframes_per_s": round(256 * frames / med_i / 1e3, 1), "GBs_algorithmic": round(256 * frames * 10248 / med_i / 1e6, 1),
                  "max_abs_round_trip_err": float((out_x[:, 4096:-4096] - x[:, 4096:-4096]).abs().max())}))
mag = torch.view_as_complex(out_c).abs().contiguous()
med_g, _ = timeit(lambda: check(lib.smx_stft_griffin_lim_f32_dev(sc._h, vp(mag.data_ptr()), 256, 1025, frames, 32, 0.99, None, 1,
                                                                 480000, vp(out_x.data_ptr()), None)), reps=3, warm=1)
print(json.dumps({"config": "C2 Griffin-Lim (32 iterations, 256 clips)", "griffin_lim_ms": round(med_g, 2),
                  "ms_per_clip": round(med_g / 256, 3), "analysis_synthesis_pairs_per_s": round(32 * 256 / med_g * 1e3, 1)}))
del out_c, out_x, mag
p = Stft.power_spectrum(sc, x)
med_apply, _ = timeit(lambda: Mel.apply(mc, p), reps=10)
m = S.mel_spectrogram(sc, mc, x[:2])
wm = O.mel_spectrogram(O.stft_config(2048, hop=512), O.mel_config(128, 48000, 2048), x[:2].cpu().numpy())
flop = 2.0 * 128 * 1025 * 256 * frames
print(json.dumps({"config": "C3", "mel_spectrogram_ms": round(med, 4), "mel_apply_only_ms": round(med_apply, 4),
                  "Mframes_per_s": round(256 * frames / med / 1e3, 1), "GBs_algorithmic_fused": round(256 * frames * 2560 / med / 1e6, 1), "dense_equiv_TFLOPs_apply": round(flop / med_apply / 1e9, 2),
                  "apply_GBs": round((256 * frames * (4100 + 512)) / med_apply / 1e6, 1), "max_rel_err_vs_oracle": rel_err(m.cpu().numpy(), wm)}))
# log-mel tail: Soundml.mfcc from audio (20 coefficients) and Convert.power_to_db of the C2 power spectrogram
out_cep = torch.empty(256, 20, frames, device="cuda")
med_mfcc, _ = timeit(lambda: check(lib.smx_mfcc_f32_dev(sc._h, mc._h, vp(x.data_ptr()), 256, 480000, 480000, 20, 0, 0.0,
                                                        vp(out_cep.data_ptr()), None)), reps=10)
out_db = torch.empty_like(p)
med_db, _ = timeit(lambda: check(lib.smx_power_to_db_f32_dev(vp(p.data_ptr()), p.numel(), 1.0, 1e-10, 1, 80.0, vp(out_db.data_ptr()), None)), reps=10)
wdb = O.power_to_db(p[:1].cpu().numpy(), top_db=None)
gdb = S.power_to_db(p[:1])
print(json.dumps({"config": "C2 log-mel tail", "mfcc20_from_audio_ms": round(med_mfcc, 4), "mfcc_Mframes_per_s": round(256 * frames / med_mfcc / 1e3, 1),
                  "power_to_db_top80_ms": round(med_db, 4), "power_to_db_GBs": round(2 * p.numel() * 4 / med_db / 1e6, 1),
                  "power_to_db_max_abs_err_dB": float(np.max(np.abs(gdb.cpu().numpy() - wdb)))}))
del out_db
# spectral-shape features and chroma on the C2 spectrogram (256 x 1025 x 938 float32, 984 MB read once per pass)
spec_bytes = 256 * frames * 1025 * 4
feat_out = torch.empty(256, 1, frames, device="cuda")
rows = {}
for name, call in (
        ("centroid", lambda: lib.smx_spectral_centroid_f32_dev(vp(p.data_ptr()), 256, 1025, frames, None, 0, 48000, vp(feat_out.data_ptr()), None)),
        ("bandwidth", lambda: lib.smx_spectral_bandwidth_f32_dev(vp(p.data_ptr()), 256, 1025, frames, 2.0, None, 0, None, 0, 0, 48000, vp(feat_out.data_ptr()), None)),
        ("rolloff", lambda: lib.smx_spectral_rolloff_f32_dev(vp(p.data_ptr()), 256, 1025, frames, 0.85, None, 0, 48000, vp(feat_out.data_ptr()), None)),
        ("flatness", lambda: lib.smx_spectral_flatness_f32_dev(vp(p.data_ptr()), 256, 1025, frames, 1e-10, 2.0, vp(feat_out.data_ptr()), None))):
    ms, _ = timeit(lambda: check(call()), reps=10)
    rows[name + "_ms"] = round(ms, 4)
    rows[name + "_GBs_one_pass"] = round(spec_bytes / ms / 1e6, 1)
got_c = S.spectral_centroid(p[:2], sample_rate=48000).cpu().numpy()
rows["centroid_max_rel_err_vs_oracle"] = rel_err(got_c, O.spectral_centroid(p[:2].cpu().numpy(), sample_rate=48000))
print(json.dumps({"config": "C2 spectral features (device spectrogram -> [256; 1; 938], includes the validity readback)", **rows}))
cc = S.Chroma.Config.create(48000, 2048)
out_ch = torch.empty(256, 12, frames, device="cuda")
med_ca, _ = timeit(lambda: check(lib.smx_chroma_apply_f32_dev(cc._h, vp(p.data_ptr()), 256, 1025, frames, 1, 0.0, vp(out_ch.data_ptr()), None)), reps=10)
med_cs, _ = timeit(lambda: check(lib.smx_chroma_stft_f32_dev(sc._h, cc._h, vp(x.data_ptr()), 256, 480000, 480000, 2.0, 1, 0.0,
                                                             vp(out_ch.data_ptr()), None)), reps=10)
wc = O.chroma_stft(O.stft_config(2048, hop=512), O.chroma_config(48000, 2048), x[:2].cpu().numpy())
print(json.dumps({"config": "C2 chroma (12 bands, inf norm)", "chroma_apply_ms": round(med_ca, 4), "apply_GBs": round(spec_bytes / med_ca / 1e6, 1),
                  "chroma_stft_ms": round(med_cs, 4), "Mframes_per_s": round(256 * frames / med_cs / 1e3, 1),
                  "max_abs_err_vs_oracle": float(np.max(np.abs(out_ch[:2].cpu().numpy() - wc)))}))
del p

# other transform sizes (generic path): the usual vocoder front end (fft 1024 / hop 256, 80 mels, 22.05 kHz), fft 512,
# whisper's fft 400 / hop 160, and the float64 interior at C2
del x
for fft, hop, n_mels, sr, n in ((1024, 256, 80, 22050, 441000), (512, 128, 80, 16000, 441000), (400, 160, 80, 16000, 160000)):
    cs = Stft.Config.create(fft_size=fft, hop=hop)
    ms_ = Mel.Config.create(n_mels=n_mels, sample_rate=sr, fft_size=fft)
    fr = Stft.frames(cs, n)
    xs_ = torch.rand(256, n, device="cuda") * 2 - 1
    po = torch.empty(256, fft // 2 + 1, fr, device="cuda")
    mo = torch.empty(256, n_mels, fr, device="cuda")
    t_p, _ = timeit(lambda: check(lib.smx_stft_power_range_f32_dev(cs._h, vp(xs_.data_ptr()), 256, n, n, 0, fr, 2.0, vp(po.data_ptr()), None)), reps=10)
    t_m, _ = timeit(lambda: check(lib.smx_mel_spectrogram_f32_dev(cs._h, ms_._h, vp(xs_.data_ptr()), 256, n, n, 2.0, vp(mo.data_ptr()), None)), reps=10)
    wp = O.power_spectrum(O.stft_config(fft, hop=hop), xs_[:1, :20000].cpu().numpy())
    gp = Stft.power_spectrum(cs, xs_[:1, :20000]).cpu().numpy()
    print(json.dumps({"config": "fft %d hop %d, 256 clips x %d frames" % (fft, hop, fr), "power_ms": round(t_p, 4),
                      "power_Mframes_per_s": round(256 * fr / t_p / 1e3, 1), "power_GBs_algorithmic": round(256 * fr * (hop + fft // 2 + 1) * 4 / t_p / 1e6, 1),
                      "mel%d_ms" % n_mels: round(t_m, 4), "mel_Mframes_per_s": round(256 * fr / t_m / 1e3, 1), "max_rel_err_vs_oracle": rel_err(gp, wp)}))
    del xs_, po, mo
x = torch.rand(256, 480000, device="cuda") * 2 - 1
out_p = torch.empty(256, 1025, frames, device="cuda")
S.set_interior("float64")
t_s, _ = timeit(lambda: check(lib.smx_stft_power_range_f32_dev(sc._h, vp(x.data_ptr()), 256, 480000, 480000, 0, frames, 2.0, vp(out_p.data_ptr()), None)), reps=5)
S.set_interior("float32")
print(json.dumps({"config": "C2 with the float64 interior (set_interior float64)", "power_ms": round(t_s, 4), "Mframes_per_s": round(256 * frames / t_s / 1e3, 1)}))
del x, out_p

# C4: 8192-tap lowpass on 8 ch x 60 s 48 kHz
h = Fir.design_lowpass(8192, 0.25, 100.0)
plan = Fir.Plan.create(h)
xs = torch.rand(8, 2880000, device="cuda") * 2 - 1
med, mn = timeit(lambda: Fir.apply(plan, xs), reps=10)
y = Fir.apply(plan, xs[:1, :100000]).cpu().numpy()
wy = O.fir_filter(h, xs[:1, :100000].cpu().numpy())
print(json.dumps({"config": "C4", "fir_ms": round(med, 4), "Msamples_per_s": round(8 * 2880000 / med / 1e3, 1),
                  "GBs_algorithmic": round(8 * 2880000 * 8 / med / 1e6, 1), "block": plan.block,
                  "max_abs_err_vs_oracle": float(np.max(np.abs(y.astype(np.float64) - wy)))}))
