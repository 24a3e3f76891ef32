"""numpy model of the FIR kernel's FFT: in-place Stockham passes (radix 16/4/2) with every element
held in registers between the read and write phases, XOR-swizzled LDS addresses; checks the result
against numpy.fft and counts LDS bank conflicts of each pass (float2 accesses, 32 slots)."""
import numpy as np

def radices(N):
    n = int(np.log2(N)); r = []
    while n >= 4: r.append(16); n -= 4
    if n == 3: r += [4, 2]
    elif n == 2: r.append(4)
    elif n == 1: r.append(2)
    return r

def swz(a):
    return a ^ ((a >> 5) & 31)

def conflicts(addrs):   # addrs: per-lane float2 indices of one wave instruction (64 lanes, 2 halves of 32)
    worst = 1
    for h in range(0, len(addrs), 32):
        slots = swz(np.asarray(addrs[h:h + 32])) % 32
        worst = max(worst, np.bincount(slots, minlength=32).max())
    return worst

def fft_stockham(x):
    N = len(x); T = N // 16
    lds = np.zeros(N, complex); lds[swz(np.arange(N))] = x
    Ns = 1
    report = []
    for R in radices(N):
        G = 16 // R            # groups of R elements per thread
        regs = np.zeros((T, 16), complex)
        rd_conf = wr_conf = 1
        for i in range(G):
            t = np.arange(T) + T * i
            for j in range(R):
                a = t + j * (N // R)
                regs[:, i * R + j] = lds[swz(a)]
                rd_conf = max(rd_conf, max(conflicts(a[w:w + 64]) for w in range(0, T, 64)))
        out = np.zeros(N, complex)
        for i in range(G):
            t = np.arange(T) + T * i
            k = t % Ns
            v = regs[:, i * R:(i + 1) * R] * np.exp(-2j * np.pi * np.outer(k, np.arange(R)) / (Ns * R))
            V = np.fft.fft(v, axis=1)
            j0 = (t // Ns) * Ns * R + k
            for j in range(R):
                a = j0 + j * Ns
                out[swz(a)] = V[:, j]
                wr_conf = max(wr_conf, max(conflicts(a[w:w + 64]) for w in range(0, T, 64)))
        lds = out
        report.append((R, Ns, rd_conf, wr_conf))
        Ns *= R
    return lds[swz(np.arange(N))], report

rng = np.random.default_rng(0)
for N in (1024, 2048, 4096, 8192, 16384):
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    y, rep = fft_stockham(x)
    err = np.max(np.abs(y - np.fft.fft(x))) / np.max(np.abs(y))
    print(N, radices(N), "err %.1e" % err, "  (radix, Ns, read-conflict, write-conflict):", rep)
    assert err < 1e-12
