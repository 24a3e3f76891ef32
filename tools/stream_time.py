"""Diagnostic: per-step cost of the streaming faces (Stft.Kernel / power_stage) on host chunks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from soundml_amd import Stft
rng = np.random.default_rng(0)
for fft, hop, channels, block in ((2048, 512, 2, 4096), (1024, 256, 1, 1024), (2048, 512, 64, 48000)):
    c = Stft.Config.create(fft_size=fft, hop=hop)
    x = rng.uniform(-1, 1, size=(channels, block * 200)).astype(np.float32)
    for name, factory in (("stage", Stft.stage(c)), ("power_stage", Stft.power_stage(c))):
        st = factory.prepare(max_items=block)
        st.step(x[:, :block])            # first step: allocations
        t0 = time.perf_counter()
        frames = 0
        for i in range(1, 200):
            out = st.step(x[:, i * block:(i + 1) * block])
            frames += 0 if out is None else out.shape[-1]
        dt = time.perf_counter() - t0
        print("fft %d hop %d, %d channels, chunks of %d samples, %-11s: %.1f us per step, %.2f Msamples/s per channel, %d frames"
              % (fft, hop, channels, block, name, dt / 199 * 1e6, 199 * block / dt / 1e6, frames))
