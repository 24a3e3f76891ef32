"""Power and clocks while one kernel runs back to back (rocm-smi sampled from a thread every 100 ms): the C2 power spectrogram, then the
frame-major analysis of Griffin-Lim, then idle.  Is the headline kernel running into the board's power cap?
  python tools/power_clock_sample.py"""
import ctypes, os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundml_amd import Stft
from soundml_amd._lib import lib, check
vp = ctypes.c_void_p
c = Stft.Config.create(fft_size=2048, hop=512)
x = torch.rand(256, 480000, device="cuda") * 2 - 1
frames = Stft.frames(c, 480000)
out = torch.empty(256, 1025, frames, device="cuda")
z = torch.empty(256, 1025, frames, 2, device="cuda")
samples, stop = [], False
def sampler():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--csv"], capture_output=True, text=True, timeout=5)
            samples.append((time.time(), r.stdout.strip().splitlines()))
        except Exception as e:   # noqa: BLE001
            samples.append((time.time(), ["error: %r" % e]))
        time.sleep(0.1)
th = threading.Thread(target=sampler); th.start()
marks = []
def phase(name, fn, seconds):
    marks.append((time.time(), name))
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        for _ in range(50): fn()
        torch.cuda.synchronize(); n += 50
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): fn()
    b.record(); torch.cuda.synchronize()
    print("%-28s %.4f ms per launch after %d launches" % (name, a.elapsed_time(b) / 50, n), flush=True)
phase("power spectrogram (C2)", lambda: check(lib.smx_stft_power_range_f32_dev(c._h, vp(x.data_ptr()), 256, 480000, 480000, 0, frames, 2.0, vp(out.data_ptr()), None)), 4.0)
phase("Stft.transform (C2)", lambda: check(lib.smx_stft_transform_range_f32_dev(c._h, vp(x.data_ptr()), 256, 480000, 480000, 0, frames, vp(z.data_ptr()), None)), 3.0)
marks.append((time.time(), "idle")); time.sleep(1.5)
stop = True; th.join()
header = None
for t, lines in samples:
    if not lines or lines[0].startswith("error"):
        print("sample:", lines[:1]); break
    if header is None:
        header = lines[0]; print(header)
    name = [m for tm, m in marks if tm <= t]
    print("%6.2f s  %-26s %s" % (t - marks[0][0], name[-1] if name else "start", " | ".join(lines[1:2])))
