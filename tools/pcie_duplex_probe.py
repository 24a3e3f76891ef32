"""Does the link carry both directions at once?  0.49 GB up and 0.98 GB down (C2's host call) between page-locked host memory and
the device: each alone, both at once on two streams, and both at once with the upload done by a copy KERNEL reading mapped host
memory (shader engines) while the DMA engine carries the download."""
import time
import torch
up_n, down_n = 256 * 480000, 256 * 1025 * 938
hx = torch.empty(up_n, dtype=torch.float32, pin_memory=True).normal_()
hy = torch.empty(down_n, dtype=torch.float32, pin_memory=True)
dx = torch.empty(up_n, device="cuda", dtype=torch.float32)
dy = torch.empty(down_n, device="cuda", dtype=torch.float32).normal_()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(up, down, reps=5):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e1 = e2 = None
        if up:
            with torch.cuda.stream(s1):
                dx.copy_(hx, non_blocking=True)
        if down:
            with torch.cuda.stream(s2):
                hy.copy_(dy, non_blocking=True)
        tu = td = None
        if up:
            s1.synchronize(); tu = (time.perf_counter() - t0) * 1e3
        if down:
            s2.synchronize(); td = (time.perf_counter() - t0) * 1e3
        out.append((tu, td))
    return out
fmt = lambda r: "  ".join("%s/%s" % ("%.1f" % a if a else "-", "%.1f" % b if b else "-") for a, b in r)
print("upload alone (ms):            ", fmt(run(True, False)))
print("download alone:               ", fmt(run(False, True)))
print("both at once (up / down done):", fmt(run(True, True)))
print("upload %.2f GB, download %.2f GB" % (up_n * 4 / 1e9, down_n * 4 / 1e9))
