"""Lane-level numpy model of stft_fast.hip (index math only, float64): checks the
radix-16 / LDS-transpose / radix-16 / quad-DPP radix-4 / bpermute post-pass /
tile-row permutation against numpy's rfft.  Development aid, not shipped logic."""
import numpy as np

N, M = 2048, 1024
rng = np.random.default_rng(0)
x = rng.standard_normal(N)
win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)
ref = np.fft.rfft(x * win)

w_m = np.exp(-2j * np.pi * np.arange(M) / M)
w_n = np.exp(-2j * np.pi * np.arange(M + 1) / N)
hwin = 0.5 * win

def fft4(a, b, c, d):
    t0, t1, t2, t3 = a + c, a - c, b + d, (b - d) * (-1j)
    return t0 + t2, t1 + t3, t0 - t2, t1 - t3

def fft16(v):  # v: [64][16]
    v = v.copy()
    for n0 in range(4):
        v[:, n0], v[:, 4 + n0], v[:, 8 + n0], v[:, 12 + n0] = fft4(v[:, n0], v[:, 4 + n0], v[:, 8 + n0], v[:, 12 + n0])
    W = lambda m: np.exp(-2j * np.pi * m / 16)
    for k0 in range(4):
        for n0 in range(4):
            v[:, 4 * k0 + n0] *= W(n0 * k0)
    for k0 in range(4):
        v[:, 4 * k0], v[:, 4 * k0 + 1], v[:, 4 * k0 + 2], v[:, 4 * k0 + 3] = fft4(v[:, 4 * k0], v[:, 4 * k0 + 1], v[:, 4 * k0 + 2], v[:, 4 * k0 + 3])
    t = np.empty_like(v)
    for k0 in range(4):
        for k1 in range(4):
            t[:, k0 + 4 * k1] = v[:, 4 * k0 + k1]
    return t

lane = np.arange(64)
k1 = lane >> 2; qa = lane & 3
r = ((qa & 1) << 1) | (qa >> 1)
v = np.empty((64, 16), complex)
for j in range(16):
    s = 2 * lane + 128 * j
    v[:, j] = x[s] * hwin[s] + 1j * x[s + 1] * hwin[s + 1]
    # check fft16 against numpy on first use
assert np.allclose(fft16(v), np.fft.fft(v, axis=1))
v = fft16(v)
for k in range(1, 16):
    v[:, k] *= w_m[lane * k]
XROW = 68
exch = np.zeros(16 * XROW, complex)
for k in range(16):
    exch[k * XROW + lane] = v[:, k]
for i in range(16):
    v[:, i] = exch[k1 * XROW + qa + 4 * i]
v = fft16(v)
for q in range(1, 16):
    v[:, q] *= w_m[16 * qa * q]
s1 = np.where(qa < 2, 1.0, -1.0); s2 = np.where(qa & 1, -1.0, 1.0)
def dpp(val, perm):
    return val[(lane & ~3) | np.array(perm)[lane & 3]]
for q in range(16):
    t = v[:, q]
    u = t * s1 + dpp(t, [2, 3, 0, 1])
    w = np.where(qa == 3, u * (-1j), u)
    v[:, q] = w * s2 + dpp(w, [1, 0, 3, 2])
# check Z
z = np.fft.fft((x * hwin)[0::2] + 1j * (x * hwin)[1::2])
for q in range(16):
    assert np.allclose(v[:, q], z[k1 + 16 * q + 256 * r]), q
low4 = lane < 4
prov = np.empty_like(v)
for m in range(16):
    prov[:, m] = np.where(low4, v[:, (m + 1) & 15], v[:, m])
addr_g = np.where(lane >= 4, 67 - lane, 3 - lane)
r0 = (4 - r) & 3
addr_0 = np.where(lane >= 4, 67 - lane, ((r0 & 1) << 1) | (r0 >> 1))
tile = np.full(1025, np.nan)
X = np.full(1025, np.nan, complex)
for q in range(16):
    addr = addr_0 if q == 0 else addr_g
    p = prov[addr, 15 - q]
    e = v[:, q] + np.conj(p); d = v[:, q] - np.conj(p)
    w = w_n[k1 + 256 * r + 16 * q]
    tr = e.real + w.real * d.imag + w.imag * d.real
    ti = e.imag - w.real * d.real + w.imag * d.imag
    rows = 4 * k1 + r + 64 * q
    assert len(set(rows)) == 64
    tile[rows] = tr * tr + ti * ti
    X[rows] = tr + 1j * ti
tile[1024] = (2 * (v[0, 0].real - v[0, 0].imag)) ** 2
# flush mapping
out = np.full(1025, np.nan)
seen = set()
for wave in range(8):
    hsel = lane >> 5; jj = (lane & 31) >> 2
    rloc = (jj & 3) + 16 * (jj >> 2) + 4 * hsel
    for it in range(8):
        row = 32 * (wave + 8 * (it >> 1)) + 8 * (it & 1) + rloc
        b = (row & 3) * 256 + (row >> 2)
        out[b] = tile[row]
        seen.update(row.tolist())
assert seen == set(range(1024))
out[1024] = tile[1024]
err = np.max(np.abs(out - np.abs(ref) ** 2)) / np.max(np.abs(ref) ** 2)
print("max rel err vs numpy rfft power:", err)
assert err < 1e-12
# LDS bank checks
for half in range(2):
    ln = lane[32 * half:32 * half + 32]
    for i in range(16):   # exchange read, b64: 64 dword banks
        banks = (2 * (k1[ln] * XROW + qa[ln] + 4 * i)) % 64
        assert len(set(banks)) == 32
    for f in range(16):   # tile write b32: 32 banks
        for q in range(16):
            banks = ((4 * k1[ln] + r[ln] + 64 * q) * 17 + f) % 32
            assert len(set(banks)) == 32
    jj = (ln & 31) >> 2; g = ln & 3
    rloc = (jj & 3) + 16 * (jj >> 2) + 4 * half
    for c in range(4):
        banks = (rloc * 17 + 4 * g + c) % 32
        assert len(set(banks)) == 32
print("index math + bank-conflict model OK")

# ---- pair-sharing post-pass: lane holding Z[k] (q < 8) also produces X[M-k] ------------------
tile2 = np.full(1024, np.nan); nyq2 = None
srow0 = np.where(lane >= 4, 67 - (4 * k1 + r), 3 - (4 * k1 + r) + 64)
writes = collections = 0
import collections as _c
cnt = _c.Counter()
for q in range(8):
    addr = addr_0 if q == 0 else addr_g
    p = prov[addr, 15 - q]
    A = v[:, q]
    E = A + np.conj(p); D = A - np.conj(p)
    w = w_n[k1 + 256 * r + 16 * q]
    P = w * D
    Xk = (E.real + P.imag) + 1j * (E.imag - P.real)
    Xm = (E.real - P.imag) + 1j * (-E.imag - P.real)
    rows_p = 4 * k1 + r + 64 * q
    for l in range(64):
        tile2[rows_p[l]] = abs(Xk[l]) ** 2; cnt[rows_p[l]] += 1
        if not (l < 4 and q == 0):
            rs = srow0[l] + 64 * (15 - q)
            tile2[rs] = abs(Xm[l]) ** 2; cnt[rs] += 1
# lanes 0..3, q = 8: primary only, partner register 8 of lane 3 - rr
p8 = v[addr_g, 8]
A = v[:, 8]; E = A + np.conj(p8); D = A - np.conj(p8); w = w_n[k1 + 256 * r + 16 * 8]; P = w * D
Xk = (E.real + P.imag) + 1j * (E.imag - P.real)
for l in range(4):
    row = 4 * k1[l] + r[l] + 64 * 8
    tile2[row] = abs(Xk[l]) ** 2; cnt[row] += 1
assert set(cnt) == set(range(1024)) and max(cnt.values()) == 1, (len(cnt), max(cnt.values()))
out2 = np.empty(1025)
for row in range(1024):
    out2[(row & 3) * 256 + (row >> 2)] = tile2[row]
out2[1024] = (2 * (v[0, 0].real - v[0, 0].imag)) ** 2
err2 = np.max(np.abs(out2 - np.abs(ref) ** 2)) / np.max(np.abs(ref) ** 2)
print("pair-sharing post-pass max rel err:", err2)
assert err2 < 1e-12
for half in range(2):   # bank check of the secondary writes
    ln = lane[32 * half:32 * half + 32]
    for q in range(8):
        banks = ((srow0[ln] + 64 * (15 - q)) * 17) % 32
        assert len(set(banks)) == 32
print("pair-sharing: every row written exactly once, secondary writes conflict-free")
