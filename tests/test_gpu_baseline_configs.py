"""BASELINE.json configurations at FULL size on the GPU, every output value against the CPU oracle
(`-m gpu`; the C restatement runs on all host cores, so a whole batch is seconds of oracle):

  C2  256 x 10 s x 48 kHz, STFT 2048 / 512            every one of the 240 128 frames
  C3  128-mel spectrogram of the same batch            the whole [256; 128; 938] result
  C4  8192-tap FIR on 8 ch x 60 s                      all 23 040 000 samples
  C5  4096 x 30 s in one call (71 GB resident)         64 random clips + the first and last, all frames
  N>1 bench.py --gpus 2 as a subprocess                rank plumbing, shard == slice bit for bit

Gates: north_star's float32 contract |a - e| <= 1e-5 (peak + |e|), evaluated per clip; and a REGRESSION gate
at 2e-6 of the clip's peak for the fft-2048 / 1024 / 512 kernels (measured error 2-8e-7 of the peak: a change that
costs a decimal digit fails here long before it reaches the contract).

Reference laws these follow: stft_grid.ml:58-73 (every frame of a range equals the offline transform),
stft_grid.ml:180-205 (leading axes are independent slices).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import c_oracle, soundml_oracle as O

import soundml_amd as S
from soundml_amd import Fir, Mel, Stft

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORES = c_oracle.effective_cpus()
CONTRACT = 1e-5        # north_star
REGRESSION = 2e-6      # of the peak; measured 2-8e-7


def clip_batch(torch, lo, hi, n, dev="cuda"):
    """bench.py's generator: clip g is uniform[-1, 1) seeded 42 + g"""
    x = torch.empty(hi - lo, n, device=dev, dtype=torch.float32)
    gen = torch.Generator(device=dev)
    for g in range(lo, hi):
        gen.manual_seed(42 + g)
        x[g - lo].uniform_(-1.0, 1.0, generator=gen)
    return x


def check_clips(got, want, what, regression=REGRESSION):
    """per clip: the contract elementwise, the regression gate on the worst element"""
    assert got.shape == want.shape, (what, got.shape, want.shape)
    worst = 0.0
    for i in range(got.shape[0]):
        e = want[i].astype(np.float64)
        err = np.abs(got[i].astype(np.float64) - e)
        peak = float(np.max(np.abs(e)))
        assert not (err > CONTRACT * (peak + np.abs(e))).any(), "%s clip %d: outside 1e-5 (max err %.3g, peak %.3g)" % (
            what, i, float(err.max()), peak)
        worst = max(worst, float(err.max()) / peak)
    assert worst <= regression, "%s: max error %.3g of the peak, regression gate %.1g" % (what, worst, regression)
    return worst


@pytest.fixture(scope="module")
def c2():
    import torch
    x = clip_batch(torch, 0, 256, 480000)
    cfg = Stft.Config.create(fft_size=2048, hop=512)
    ocfg = O.stft_config(2048, hop=512)
    xh = x.cpu().numpy()
    want = c_oracle.stft(ocfg, xh, 2.0, threads=CORES)        # float64 interior, rounded once: [256; 1025; 938] f32
    yield {"x": x, "xh": xh, "cfg": cfg, "ocfg": ocfg, "want": want}


@pytest.mark.timeout(900)
def test_c2_every_frame(c2):
    import torch
    p = Stft.power_spectrum(c2["cfg"], c2["x"])
    assert tuple(p.shape) == (256, 1025, 938) and p.dtype == torch.float32
    got = p.cpu().numpy()
    del p
    worst = check_clips(got, c2["want"], "C2 power")
    print("C2: 240128 frames, max error %.3g of the peak" % worst)
    # the numpy form of the oracle agrees with the C form on a clip (two independent restatements)
    w0 = O.power_spectrum(c2["ocfg"], c2["xh"][17])
    np.testing.assert_allclose(c2["want"][17], w0, rtol=2e-6, atol=1e-6 * float(w0.max()))


@pytest.mark.timeout(900)
def test_c2_complex_every_frame(c2):
    """Stft.transform of the whole batch (the same pipeline, complex tile): real and imaginary parts"""
    import torch
    z = Stft.transform(c2["cfg"], c2["x"][:64])
    got = torch.view_as_real(z).cpu().numpy()
    del z
    want = c_oracle.stft(c2["ocfg"], c2["xh"][:64], complex_out=True, threads=CORES)
    want = np.stack([want.real, want.imag], axis=-1)
    check_clips(got, want, "C2 transform")


@pytest.mark.timeout(900)
def test_c3_whole_batch(c2):
    mc = Mel.Config.create(n_mels=128, sample_rate=48000, fft_size=2048)
    omc = O.mel_config(128, 48000, 2048)
    m = S.mel_spectrogram(c2["cfg"], mc, c2["x"])
    assert tuple(m.shape) == (256, 128, 938)
    got = m.cpu().numpy()
    want = c_oracle.mel_apply(omc, c2["want"], threads=CORES)     # mel.ml:231 on the oracle's own spectrogram
    worst = check_clips(got, want, "C3 mel")
    print("C3: max error %.3g of the peak" % worst)
    # and the unfused composition Mel.apply (Stft.power_spectrum x) on a quarter of the batch
    p = Stft.power_spectrum(c2["cfg"], c2["x"][:64])
    check_clips(Mel.apply(mc, p).cpu().numpy(), want[:64], "C3 Mel.apply")


@pytest.mark.timeout(900)
def test_c4_all_samples():
    """8192 taps on 8 x 2 880 000 samples against the oracle's float64 FFT-form convolution.  The error of a
    float32 FFT convolution scales with the filter's L1 gain times the input peak, not with the local output."""
    import torch
    h = Fir.design_lowpass(8192, 0.25, 100.0)
    plan = Fir.Plan.create(h)
    x = clip_batch(torch, 10000, 10008, 2880000)
    y = Fir.apply(plan, x).cpu().numpy()
    xh = x.cpu().numpy()
    bound = 1e-5 * float(np.sum(np.abs(h)))      # max|x| <= 1
    worst = 0.0
    for ch in range(8):
        want = O.fir_filter(h, xh[ch].astype(np.float64))
        worst = max(worst, float(np.max(np.abs(y[ch].astype(np.float64) - want))))
    assert worst <= bound, (worst, bound)
    assert worst <= 0.2 * bound, "regression gate: %.3g vs %.3g" % (worst, 0.2 * bound)
    print("C4: max abs error %.3g (bound %.3g)" % (worst, bound))


@pytest.mark.timeout(1800)
def test_c5_one_call():
    """4096 x 30 s in ONE call: above 2 GB the launcher deals tiles per XCD chunk (another order than C2's)."""
    import torch
    free_b, _ = torch.cuda.mem_get_info()
    clips, n = 4096, 1440000
    cfg = Stft.Config.create(fft_size=2048, hop=512)
    frames = Stft.frames(cfg, n)
    assert frames == 2813
    if free_b < clips * (n + 1025 * frames) * 4 + (8 << 30):
        pytest.skip("needs 71 GB of free device memory")
    x = clip_batch(torch, 0, clips, n)
    p = Stft.power_spectrum(cfg, x)
    assert tuple(p.shape) == (clips, 1025, frames)
    rng = np.random.default_rng(5)
    pick = sorted(set([0, clips - 1] + [int(c) for c in rng.choice(clips, size=64, replace=False)]))
    ocfg = O.stft_config(2048, hop=512)
    idx = torch.tensor(pick, device="cuda")
    xs = x[idx].cpu().numpy()
    got = p[idx].cpu().numpy()
    want = c_oracle.stft(ocfg, xs, 2.0, threads=CORES)
    worst = check_clips(got, want, "C5 power")
    # every clip took part: a cheap whole-tensor property (finite, and Parseval-sized energy per clip)
    assert bool(torch.isfinite(p).all())
    e = p.sum(dim=(1, 2))
    assert float(e.min()) > 0.5 * float(e.max())
    print("C5: %d clips x %d frames checked, max error %.3g of the peak" % (len(pick), frames, worst))


@pytest.mark.timeout(900)
def test_bench_two_ranks_one_box():
    """bench.py --gpus 2 as the driver starts it: the launcher spawns two rank processes (both on device 0 when
    only one is visible; the clock then reduces over gloo because RCCL refuses two ranks on one device), each
    computes its clip range of a C5-shaped job through the HIP path, and shard == slice of the whole batch bit
    for bit on every rank."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--clips", "8", "--seconds", "3", "--steps", "2",
           "--warmup", "1", "--verify-shards", "--no-extras", "--workload", "c5"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["world_size_seen"] == 2
    assert line["ranks_counted"] == 2   # an all-reduce SUM of 1: the collective spanned both ranks
    assert 0 < line["rank_kernel_ms_avg_min"] <= line["rank_kernel_ms_avg_max"]
    assert line["shard_check"] == {"ranks_bit_exact": 2, "ranks": 2, "world_size_seen": 2,
                                   "backend": line["shard_check"]["backend"]}
    assert line["roofline"]["launches_per_step"] == 1
    assert line["config"]["frames_per_gpu"] == 4 * 282
    assert line["value"] > 0


@pytest.mark.parametrize("fft,hop", [(1024, 256), (512, 128), (2048, 512)])
def test_regression_gate_fast_kernels(fft, hop):
    """the float32 kernels of the three common sizes at 2e-6 of the peak (power and complex faces)"""
    import torch
    x = clip_batch(torch, 500, 516, 120000)
    cfg = Stft.Config.create(fft_size=fft, hop=hop)
    ocfg = O.stft_config(fft, hop=hop)
    xh = x.cpu().numpy()
    check_clips(Stft.power_spectrum(cfg, x).cpu().numpy(), c_oracle.stft(ocfg, xh, 2.0, threads=CORES), "power %d" % fft)
    z = torch.view_as_real(Stft.transform(cfg, x)).cpu().numpy()
    w = c_oracle.stft(ocfg, xh, complex_out=True, threads=CORES)
    check_clips(z, np.stack([w.real, w.imag], axis=-1), "transform %d" % fft)


def test_aligned_block_flush_is_bit_identical_to_the_plain_flush():
    """The power and the complex spectrogram at fft 2048 leave LDS in whole aligned 64- / 128-byte blocks (a row's values are carried in
    registers until they complete a block: stft_fast_p32.hpp, SKEW; the fft 1024 / 512 kernels of stft_fast_p16.hpp have no such form and
    ride along as a control: the switches must not touch them).  SMX_POWER_SKEW=0 / SMX_COMPLEX_SKEW=0 select
    the plain per-tile flush: same frame code, so the values must agree bit for bit -- over ranges that start mid-clip,
    partial last tiles, ranges whose workgroups change clip, odd and even row pitches, every origin alignment, general powers."""
    code = """
import json, sys, numpy as np
sys.path.insert(0, %r)
import torch
from soundml_amd import Stft
torch.manual_seed(3)
out = []
for fft in (2048, 1024, 512):
    hop = fft // 4
    c = Stft.Config.create(fft_size=fft, hop=hop)
    for clips, n in ((3, 16 * hop * 3 + 100), (5, 40000), (2, 64 * hop * 9), (7, 30001), (300, 64 * hop * 2 + 7)):
        x = (torch.rand(clips, n, device="cuda") * 2 - 1).float()
        frames = Stft.frames(c, n)
        for a, b in ((0, frames), (3, frames - 2), (5, 6), (1, min(frames, 70))):
            for p in (2.0, 1.0, 0.7):
                pw = Stft.power_range(c, x, a, b, p).contiguous()
                out.append(int(pw.view(torch.int32).to(torch.int64).sum()))
            z = torch.view_as_real(Stft.transform_range(c, x, a, b)).contiguous()
            out.append(int(z.view(torch.int32).to(torch.int64).sum()))
print(json.dumps(out))
""" % ROOT
    res = []
    for skew in ("1", "0"):
        env = dict(os.environ)
        if skew == "0":
            env.update(SMX_POWER_SKEW="0", SMX_COMPLEX_SKEW="0")
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=500, env=env, cwd=ROOT)
        assert out.returncode == 0, out.stdout + out.stderr
        res.append(json.loads(out.stdout.strip().splitlines()[-1]))
    assert res[0] == res[1]


def test_border_tiles_in_the_tile_sequence_equal_the_epilogue():
    """Round 5: at fft 2048 / 1024 / 512 / 256 the frames that reach past the signal ride in the tile sequence of the power, complex and fused mel
    kernels (a tile with such a frame loads through the padding rule, stft_fast_p32.hpp load_frame32_padded) instead of an epilogue
    after the interior tiles / gathered strips.
    SMX_BORDER_INLINE=0 selects the epilogue / strips: same frame code on the same samples, so every value agrees bit for bit --
    every pad mode and alignment, odd hops (unaligned loads), ranges that begin or end inside the border, clips barely longer
    than a frame, many short clips, general powers."""
    code = """
import json, sys, numpy as np
sys.path.insert(0, %r)
import torch
import soundml_amd as S
from soundml_amd import Mel, Stft
torch.manual_seed(5)
out = []
for alignment in ("centered", "left", "right"):
    for pad in ("reflect", "edge", ("constant", 0.37)):
        for fft, hop in ((2048, 512), (2048, 300), (2048, 77), (1024, 256), (1024, 77), (512, 128), (512, 33), (256, 64), (256, 33)):
            c = Stft.Config.create(fft_size=fft, hop=hop, alignment=alignment, pad=pad)
            mc = Mel.Config.create(n_mels=80, sample_rate=16000, fft_size=fft)
            for clips, n in ((3, fft), (2, fft + 1), (5, 40000), (260, 9001), (1, 2 * fft + 511)):
                x = (torch.rand(clips, n, device="cuda") * 2 - 1).float()
                frames = Stft.frames(c, n)
                for a, b in ((0, frames), (1, frames - 1), (0, min(frames, 3)), (max(frames - 2, 0), frames)):
                    if b <= a:
                        continue
                    for p in (2.0, 0.7):
                        pw = Stft.power_range(c, x, a, b, p).contiguous()
                        out.append(int(pw.view(torch.int32).to(torch.int64).sum()))
                    z = torch.view_as_real(Stft.transform_range(c, x, a, b)).contiguous()
                    out.append(int(z.view(torch.int32).to(torch.int64).sum()))
                if hop * 4 == fft and fft >= 512:
                    m = S.mel_spectrogram(c, mc, x).contiguous()
                    out.append(int(m.view(torch.int32).to(torch.int64).sum()))
print(json.dumps(out))
""" % ROOT
    res = []
    for inline in ("1", "0"):
        env = dict(os.environ)
        env.pop("SMX_BORDER_INLINE", None)
        if inline == "0":
            env.update(SMX_BORDER_INLINE="0")
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert out.returncode == 0, out.stdout + out.stderr
        res.append(json.loads(out.stdout.strip().splitlines()[-1]))
    assert len(res[0]) > 1500 and res[0] == res[1]


@pytest.mark.timeout(600)
def test_many_short_clips_take_the_strip_path_and_agree():
    """6000 clips of 4000 samples: 8 frames each, 4 of them touching a border -- 24 000 border frames, above the launcher's
    threshold, so the power kernel reads them from gathered strips (one launch over every frame) instead of its epilogue.
    The same clips in a batch of 50 take the epilogue: identical frame code, so the values agree bit for bit; and three clips
    are checked against the oracle.  Mel and complex outputs of the big batch against small batches too."""
    import torch
    rng = np.random.default_rng(11)
    x = torch.from_numpy(rng.uniform(-1, 1, size=(6000, 4000)).astype(np.float32)).cuda()
    cfg = Stft.Config.create(fft_size=2048, hop=512)
    p = Stft.power_spectrum(cfg, x)
    assert tuple(p.shape) == (6000, 1025, 8)
    for lo in (0, 2950, 5950):
        assert torch.equal(p[lo:lo + 50], Stft.power_spectrum(cfg, x[lo:lo + 50])), lo
    ocfg = O.stft_config(2048, hop=512)
    for clip in (0, 3333, 5999):
        want = O.power_spectrum(ocfg, x[clip].cpu().numpy())
        got = p[clip].cpu().numpy()
        assert np.max(np.abs(got - want)) <= REGRESSION * float(np.max(want)), clip
    z = Stft.transform(cfg, x)
    assert torch.equal(z[100:150], Stft.transform(cfg, x[100:150]))
    mc = Mel.Config.create(n_mels=80, sample_rate=16000, fft_size=2048)
    m = S.mel_spectrogram(cfg, mc, x)
    assert torch.equal(m[4000:4050], S.mel_spectrogram(cfg, mc, x[4000:4050]))
