"""How fast does the DMA engine fill a large page-locked block?  (round 5: a result array from smx_host_alloc was SLOWER than the
staged path.)  Device -> host copies of 0.98 GB: into torch's pinned tensor, into a block of smx_host_alloc in one hipMemcpy and in
16 MB asynchronous pieces, and into the same block with the pages touched first."""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundml_amd._lib import lib, check
hip = ctypes.CDLL("libamdhip64.so")
vp = ctypes.c_void_p
hip.hipMemcpy.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int]
hip.hipMemcpyAsync.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int, vp]
hip.hipStreamSynchronize.argtypes = [vp]
hip.hipHostMalloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t, ctypes.c_uint]
n = 256 * 1025 * 938 * 4
d = torch.empty(n // 4, device="cuda", dtype=torch.float32).normal_()
torch.cuda.synchronize()
def t(fn, reps=6):
    out = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); out.append((time.perf_counter() - t0) * 1e3)
    return " ".join("%.1f" % v for v in out)
hp = torch.empty(n // 4, dtype=torch.float32, pin_memory=True)
print("torch pinned tensor, copy_ + sync:        ", t(lambda: (hp.copy_(d, non_blocking=True), torch.cuda.synchronize())))
p = vp()
check(lib.smx_host_alloc(n, ctypes.byref(p)))
print("smx_host_alloc block, one hipMemcpy:      ", t(lambda: hip.hipMemcpy(p, vp(d.data_ptr()), n, 2)))
s = torch.cuda.Stream()
def pieces(dst):
    for off in range(0, n, 16 << 20):
        hip.hipMemcpyAsync(vp(dst + off), vp(d.data_ptr() + off), min(16 << 20, n - off), 2, vp(s.cuda_stream))
    hip.hipStreamSynchronize(vp(s.cuda_stream))
print("smx_host_alloc block, 16 MB async pieces: ", t(lambda: pieces(p.value)))
for flags, name in ((0, "default"), (0x1, "portable"), (0x40000000, "non-coherent"), (0x80000000, "coherent"), (0x20000000 | 0x1, "numa-user|portable")):
    q = vp()
    if hip.hipHostMalloc(ctypes.byref(q), n, flags) != 0:
        print("hipHostMalloc flags %#x failed" % flags); continue
    print("hipHostMalloc %-20s one hipMemcpy: " % name, t(lambda: hip.hipMemcpy(q, vp(d.data_ptr()), n, 2), 4))
    hip.hipHostFree(q)
a = np.empty(n, dtype=np.uint8)
print("pageable numpy array (fresh), hipMemcpy:  ", t(lambda: hip.hipMemcpy(vp(a.ctypes.data), vp(d.data_ptr()), n, 2), 3))
