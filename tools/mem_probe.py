"""Diagnostic: what the memory system of this box sustains for plain streaming traffic of the C2 sizes."""
import torch
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
x = torch.rand(256, 480000, device="cuda")
out = torch.empty(256, 1025, 938, device="cuda")
y = torch.empty_like(x)
src = torch.rand(256, 1025, 938, device="cuda")
ms = t(lambda: out.fill_(1.0)); print("fill 0.98 GB: %.3f ms  %.2f TB/s" % (ms, out.numel() * 4 / ms / 1e9))
ms = t(lambda: y.copy_(x)); print("copy 0.49 GB: %.3f ms  %.2f TB/s (r+w)" % (ms, 2 * x.numel() * 4 / ms / 1e9))
ms = t(lambda: out.copy_(src)); print("copy 0.98 GB: %.3f ms  %.2f TB/s (r+w)" % (ms, 2 * out.numel() * 4 / ms / 1e9))
ms = t(lambda: x.sum()); print("read 0.49 GB (sum): %.3f ms  %.2f TB/s" % (ms, x.numel() * 4 / ms / 1e9))
# transposed write: [256,938,1025] -> [256,1025,938] (the same scatter shape as the spectrogram write, done by torch)
st = torch.rand(256, 938, 1025, device="cuda")
ms = t(lambda: out.copy_(st.transpose(1, 2))); print("transpose copy 0.98 GB: %.3f ms  %.2f TB/s (r+w)" % (ms, 2 * out.numel() * 4 / ms / 1e9))
