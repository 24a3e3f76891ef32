"""Generates soundml_amd/csrc/stft_pk_fft.inc: the register FFT stages of the 32-lane frame pipeline as blocks of packed
float32 instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32), one inline-assembly statement per stage.

Why generated assembly: a complex value is an aligned register pair, and a packed instruction does the two IEEE operations of
a complex add / half a complex product in ONE issue slot (a wave issues one instruction per ~4.2 cycles whatever it is).
hipcc forms the plain packed adds from vector code but not the swapped / negated operand selects that a product by -i, a
conjugate or a complex product needs (it emits v_xor + v_mov), and between two dependent inline-assembly statements it pads
a wait state it cannot prove unnecessary (the dst_sel forwarding hazard of gfx940+): so a whole stage is ONE statement, and
the register plan inside it is made here.

The operations and their order are those of p32_cmul / p32_fft4 / p32_fft16 / fft32 (stft_fast_p32.hpp), which remain in
the tree as the readable statement of the arithmetic: every packed operation is the IEEE operation of its halves, a product
by -i is folded into the operand selects of its consumer (negation is exact), so the results are the same bits.

    python tools/gen/gen_pk_fft.py > soundml_amd/csrc/stft_pk_fft.inc
"""
import sys

C16 = (0.92387953251128674, 0.38268343236508977)       # cos, sin of 2 pi / 16
HH = 0.70710678118654752
C32 = {1: (0.98078528040323043, 0.19509032201612825), 2: C16, 3: (0.83146961230254524, 0.55557023301960218)}


class Block:
    """One asm statement: vector slots (aligned pairs), scalar constant pairs, a list of instructions."""

    def __init__(self, name):
        self.name = name
        self.slots = []          # (kind, cname) kind in {"io", "tmp", "in"}
        self.consts = []         # (cname, (lo, hi))
        self.ins = []

    def slot(self, kind, cname):
        self.slots.append((kind, cname))
        return len(self.slots) - 1

    def const(self, cname, val):
        for i, (n, v) in enumerate(self.consts):
            if n == cname:
                return ("k", i)
        self.consts.append((cname, val))
        return ("k", len(self.consts) - 1)

    def ref(self, s):
        if isinstance(s, tuple):
            return "%%%d" % (len(self.slots) + s[1])
        return "%%%d" % s

    def emit(self, op, dst, srcs, mods=""):
        self.ins.append("%s %s, %s%s" % (op, self.ref(dst), ", ".join(self.ref(s) for s in srcs), (" " + mods) if mods else ""))

    # --- packed forms -------------------------------------------------------------------------------------------
    def add(self, d, a, b): self.emit("v_pk_add_f32", d, [a, b])
    def sub(self, d, a, b): self.emit("v_pk_add_f32", d, [a, b], "neg_lo:[0,1] neg_hi:[0,1]")
    def add_mi(self, d, a, b): self.emit("v_pk_add_f32", d, [a, b], "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")   # a + (-i) b
    def sub_mi(self, d, a, b): self.emit("v_pk_add_f32", d, [a, b], "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")   # a - (-i) b

    def cmul_const(self, t, s, k, ix, sx, iy, sy):
        """t = s * (sx K[ix] + i sy K[iy]) as p32_cmul does it: products of s.x, then fused multiply-adds of s.y."""
        self.emit("v_pk_mul_f32", t, [s, k], "op_sel:[0,%d] op_sel_hi:[0,%d] neg_lo:[0,%d] neg_hi:[0,%d]" % (ix, iy, sx < 0, sy < 0))
        return lambda: self.emit("v_pk_fma_f32", t, [s, k, t], "op_sel:[1,%d,0] op_sel_hi:[1,%d,1] neg_lo:[0,%d,0] neg_hi:[0,%d,0]" % (iy, ix, sy > 0, sx < 0))

    def cmul_reg(self, t, s, w):
        self.emit("v_pk_mul_f32", t, [s, w], "op_sel_hi:[0,1]")
        return lambda: self.emit("v_pk_fma_f32", t, [s, w, t], "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]")

    def text(self, indent="  "):
        outs, ins = [], []
        for kind, cname in self.slots:
            if kind == "io":
                outs.append('"+v"(%s)' % cname)
        for kind, cname in self.slots:
            if kind == "tmp":
                outs.append('"=&v"(%s)' % cname)
        # operand numbering: outputs first (io then tmp), then inputs: remap
        order = [i for i, (k, _) in enumerate(self.slots) if k == "io"] + [i for i, (k, _) in enumerate(self.slots) if k == "tmp"] + \
                [i for i, (k, _) in enumerate(self.slots) if k == "in"]
        remap = {old: new for new, old in enumerate(order)}
        for kind, cname in self.slots:
            if kind == "in":
                ins.append('"v"(%s)' % cname)
        for cname, _ in self.consts:
            ins.append('"s"(%s)' % cname)
        import re
        def fix(line):
            return re.sub(r"%(\d+)", lambda m: "%%%d" % (remap[int(m.group(1))] if int(m.group(1)) < len(self.slots) else int(m.group(1))), line)
        body = "\n".join('%s    "%s\\n\\t"' % (indent, fix(l)) for l in self.ins)
        if not ins:
            return "%sasm(\n%s\n%s    : %s);\n" % (indent, body, indent, ", ".join(outs))
        return "%sasm(\n%s\n%s    : %s\n%s    : %s);\n" % (indent, body, indent, ", ".join(outs), indent, ", ".join(ins))


def twiddle_sel(wx, wy, table):
    """(constant pair name, value, ix, sx, iy, sy) with |wx| = K[ix], |wy| = K[iy]"""
    for name, (lo, hi) in table.items():
        for ix in (0, 1):
            for iy in (0, 1):
                if abs(abs(wx) - (lo, hi)[ix]) < 1e-15 and abs(abs(wy) - (lo, hi)[iy]) < 1e-15:
                    return name, (lo, hi), ix, (1 if wx > 0 else -1), iy, (1 if wy > 0 else -1)
    raise ValueError((wx, wy))


def fft4_pair(b, L, free, quads):
    """Two (or one) 4-point transforms interleaved, in place on logical values L (index -> slot).  quads: [(ia, ib, ic, id, crot)]"""
    st = []
    for (ia, ib, ic, id_, crot) in quads:
        st.append(dict(a=L[ia], b=L[ib], c=L[ic], d=L[id_], x0=free.pop(), x1=free.pop(), crot=crot, ic=ic))
    for q in st: (b.add_mi if q["crot"] else b.add)(q["x0"], q["a"], q["c"])      # t0
    for q in st: b.add(q["x1"], q["b"], q["d"])                                      # t2
    for q in st: (b.sub_mi if q["crot"] else b.sub)(q["c"], q["a"], q["c"])      # t1 -> slot c
    for q in st: b.sub(q["d"], q["b"], q["d"])                                      # u  -> slot d
    for q in st: b.add(q["a"], q["x0"], q["x1"])                                    # a'
    for q in st: b.sub(q["x0"], q["x0"], q["x1"])                                   # c' -> x0
    for q in st: b.add_mi(q["b"], q["c"], q["d"])                                   # b' = t1 + (-i) u
    for q in st: b.sub_mi(q["d"], q["c"], q["d"])                                   # d' = t1 - (-i) u
    for q in st:
        L[q["ic"]] = q["x0"]
        free.append(q["c"])
        free.append(q["x1"])


def cmul_group(b, L, free, items, table):
    """items: [(index, wx, wy)]: L[index] *= (wx + i wy), two at a time"""
    for g in range(0, len(items), 2):
        fin = []
        for (i, wx, wy) in items[g:g + 2]:
            name, val, ix, sx, iy, sy = twiddle_sel(wx, wy, table)
            k = b.const(name, val)
            t = free.pop()
            fin.append((b.cmul_const(t, L[i], k, ix, sx, iy, sy), i, t))
        for f, i, t in fin:
            f()
            free.append(L[i])
            L[i] = t


def gen_fft16():
    b = Block("pk_fft16")
    L = {i: b.slot("io", "s%d" % i) for i in range(16)}
    free = [b.slot("tmp", "s%d" % (16 + i)) for i in range(4)]
    table = {"kB": C16, "kH": (HH, HH)}
    c1, s1 = C16
    for n0 in (0, 2):
        fft4_pair(b, L, free, [(n0, 4 + n0, 8 + n0, 12 + n0, False), (n0 + 1, 5 + n0, 9 + n0, 13 + n0, False)])
    cmul_group(b, L, free, [(5, c1, -s1), (6, HH, -HH), (7, s1, -c1), (9, HH, -HH), (11, -HH, -HH), (13, s1, -c1), (14, -HH, -HH),
                            (15, -c1, s1)], table)
    fft4_pair(b, L, free, [(0, 1, 2, 3, False), (4, 5, 6, 7, False)])
    fft4_pair(b, L, free, [(8, 9, 10, 11, True), (12, 13, 14, 15, False)])      # v[10] stands for (-i) v[10]: W16^4
    out = {}
    for k0 in range(4):
        for k1 in range(4):
            out[k0 + 4 * k1] = L[4 * k0 + k1]
    lines = ["// 16-point forward DFT, natural order in and out: p32_fft16's operations on packed pairs, one statement (80 instructions)",
             "__device__ __forceinline__ void pk_fft16(f2 (&v)[16]) {",
             "  const f2 kB = {(float)%r, (float)%r}, kH = {(float)%r, (float)%r};" % (C16[0], C16[1], HH, HH),
             "  f2 " + ", ".join("s%d = v[%d]" % (i, i) for i in range(16)) + ", s16, s17, s18, s19;"]
    lines.append(b.text().rstrip("\n"))
    for i in range(16):
        lines.append("  v[%d] = s%d;" % (i, out[i]))
    lines.append("}")
    return "\n".join(lines)


def gen_fft8():
    """p32_fft8 (stft_fast_p16.hpp): two 4-point transforms over the even / odd inputs, W8 twiddles, combining pass"""
    b = Block("pk_fft8")
    L = {i: b.slot("io", "s%d" % i) for i in range(8)}
    free = [b.slot("tmp", "s%d" % (8 + i)) for i in range(4)]
    table = {"kH": (HH, HH)}
    fft4_pair(b, L, free, [(0, 2, 4, 6, False), (1, 3, 5, 7, False)])
    cmul_group(b, L, free, [(3, HH, -HH), (7, -HH, -HH)], table)       # o1 = W8^1 v3, o3 = W8^3 v7; o2 = -i v5 is folded below
    # v[k] = e_k + o_k, v[k + 4] = e_k - o_k with e = (v0, v2, v4, v6), o = (v1, o1, -i v5, o3)
    out = {}
    for g in ((0, 1), (2, 3)):
        st = [(k, free.pop()) for k in g]
        for k, t in st:
            (b.add_mi if k == 2 else b.add)(t, L[2 * k], L[2 * k + 1])
        for k, t in st:
            (b.sub_mi if k == 2 else b.sub)(L[2 * k + 1], L[2 * k], L[2 * k + 1])
        for k, t in st:
            out[k] = t
            out[k + 4] = L[2 * k + 1]
            free.append(L[2 * k])
    lines = ["// 8-point forward DFT, natural order in and out: p32_fft8's operations on packed pairs, one statement",
             "__device__ __forceinline__ void pk_fft8(f2 (&v)[8]) {",
             "  const f2 kH = {(float)%r, (float)%r};" % (HH, HH),
             "  f2 " + ", ".join("s%d = v[%d]" % (i, i) for i in range(8)) + ", s8, s9, s10, s11;"]
    lines.append(b.text().rstrip("\n"))
    for i in range(8):
        lines.append("  v[%d] = s%d;" % (i, out[i]))
    lines.append("}")
    return "\n".join(lines)


def gen_fft4x2():
    """Two 4-point transforms in one statement (the radix-4 stage of the 4-lane pipeline at fft 256, stft_fast_p16.hpp): p32_fft4's operations"""
    b = Block("pk_fft4x2")
    L = {i: b.slot("io", "s%d" % i) for i in range(8)}
    free = [b.slot("tmp", "s%d" % (8 + i)) for i in range(4)]
    fft4_pair(b, L, free, [(0, 1, 2, 3, False), (4, 5, 6, 7, False)])
    lines = ["// two 4-point forward DFTs (u and v), natural order in and out: p32_fft4's operations on packed pairs, one statement",
             "__device__ __forceinline__ void pk_fft4x2(f2 (&u)[4], f2 (&v)[4]) {",
             "  f2 " + ", ".join("s%d = %s[%d]" % (i, "uv"[i // 4], i % 4) for i in range(8)) + ", s8, s9, s10, s11;"]
    lines.append(b.text().rstrip("\n"))
    for i in range(8):
        lines.append("  %s[%d] = s%d;" % ("uv"[i // 4], i % 4, L[i]))
    lines.append("}")
    return "\n".join(lines)


def gen_combine(half):
    """v[k] = e[k] + W32^k o[k], v[k + 16] = e[k] - W32^k o[k] for k = 8 half .. 8 half + 7"""
    import math
    b = Block("pk_comb%d" % half)
    E = {k: b.slot("io", "e%d" % k) for k in range(8)}
    O = {k: b.slot("io", "o%d" % k) for k in range(8)}
    free = [b.slot("tmp", "x%d" % i) for i in range(2)]
    table = {"kA": C32[1], "kB": C32[2], "kC": C32[3], "kH": (HH, HH)}
    c1, s1 = C32[1]; c2, s2 = C32[2]; c3, s3 = C32[3]
    tw = {1: (c1, -s1), 2: (c2, -s2), 3: (c3, -s3), 4: (HH, -HH), 5: (s3, -c3), 6: (s2, -c2), 7: (s1, -c1),
          9: (-s1, -c1), 10: (-s2, -c2), 11: (-s3, -c3), 12: (-HH, -HH), 13: (-c3, -s3), 14: (-c2, -s2), 15: (-c1, -s1)}
    for k, (wx, wy) in tw.items():   # the constants are those of fft32
        assert abs(wx - math.cos(2 * math.pi * k / 32)) < 1e-15 and abs(wy + math.sin(2 * math.pi * k / 32)) < 1e-15
    items = [(kk, ) + tw[8 * half + kk] for kk in range(8) if (8 * half + kk) in tw]
    cmul_group(b, O, free, items, table)
    lo, hi = {}, {}
    for g in range(0, 8, 2):
        st = []
        for kk in (g, g + 1):
            st.append((kk, free.pop()))
        for kk, t in st:
            (b.add_mi if 8 * half + kk == 8 else b.add)(t, E[kk], O[kk])
        for kk, t in st:
            (b.sub_mi if 8 * half + kk == 8 else b.sub)(O[kk], E[kk], O[kk])
        for kk, t in st:
            lo[kk] = t
            hi[kk] = O[kk]
            free.append(E[kk])
    names = {}
    for kind_i, (kind, cname) in enumerate(b.slots):
        names[kind_i] = cname
    used = sorted(set(c[0] for c in b.consts))
    vals = dict(b.consts)
    lines = ["// fft32's combining pass for k = %d..%d: v[k] = e[k] + W32^k o[k], v[k + 16] = e[k] - W32^k o[k]" % (8 * half, 8 * half + 7),
             "__device__ __forceinline__ void pk_fft32_combine%d(f2 (&v)[32], const f2 (&e)[16], const f2 (&o)[16]) {" % half,
             "  const f2 " + ", ".join("%s = {(float)%r, (float)%r}" % (n, vals[n][0], vals[n][1]) for n in used) + ";",
             "  f2 " + ", ".join("e%d = e[%d]" % (k, 8 * half + k) for k in range(8)) + ";",
             "  f2 " + ", ".join("o%d = o[%d]" % (k, 8 * half + k) for k in range(8)) + ", x0, x1;"]
    lines.append(b.text().rstrip("\n"))
    for kk in range(8):
        lines.append("  v[%d] = %s; v[%d] = %s;" % (8 * half + kk, names[lo[kk]], 8 * half + kk + 16, names[hi[kk]]))
    lines.append("}")
    return "\n".join(lines)


def gen_twiddle(n):
    """v_i *= w_i (register twiddles), n values"""
    b = Block("pk_twiddle%d" % n)
    V = {i: b.slot("io", "a%d" % i) for i in range(n)}
    free = [b.slot("tmp", "x%d" % i) for i in range(2)]
    W = {i: b.slot("in", "w%d" % i) for i in range(n)}
    for g in range(0, n, 2):
        fin = []
        for i in range(g, min(g + 2, n)):
            t = free.pop()
            fin.append((b.cmul_reg(t, V[i], W[i]), i, t))
        for f, i, t in fin:
            f()
            free.append(V[i])
            V[i] = t
    names = {i: c for i, (_, c) in enumerate(b.slots)}
    args = ", ".join("f2 &v%d" % i for i in range(n)) + ", " + ", ".join("f2 w%d" % i for i in range(n))
    lines = ["// v_i *= w_i, %d complex products (p32_cmul's operations)" % n,
             "__device__ __forceinline__ void pk_twiddle%d(%s) {" % (n, args),
             "  f2 " + ", ".join("a%d = v%d" % (i, i) for i in range(n)) + ", x0, x1;"]
    lines.append(b.text().rstrip("\n"))
    for i in range(n):
        lines.append("  v%d = %s;" % (i, names[V[i]]))
    lines.append("}")
    return "\n".join(lines)


def gen_post(n, cplx):
    """n slots of the real-FFT post-pass: E = Z + conj P, D = Z - conj P, T = -i w D, planes re = (E.x + T.x, E.x - T.x),
    im = (E.y + T.y, +-(E.y - T.y)); power: fma(re, re, im im) per half."""
    b = Block("pk_post")
    Z = {i: b.slot("io", "z%d" % i) for i in range(n)}
    P = {i: b.slot("io", "p%d" % i) for i in range(n)}
    free = [b.slot("tmp", "x%d" % i) for i in range(2)]
    W = {i: b.slot("in", "w%d" % i) for i in range(n)}
    res = {}
    for g in range(0, n, 2):
        grp = list(range(g, min(g + 2, n)))
        X = {i: free.pop() for i in grp}
        for i in grp: b.emit("v_pk_add_f32", X[i], [Z[i], P[i]], "neg_hi:[0,1]")                              # E
        for i in grp: b.emit("v_pk_add_f32", P[i], [Z[i], P[i]], "neg_lo:[0,1]")                              # D -> slot p
        for i in grp: b.emit("v_pk_mul_f32", Z[i], [W[i], P[i]], "op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]")  # (w.y d.x, -(w.x d.x)) -> slot z
        for i in grp: b.emit("v_pk_fma_f32", Z[i], [W[i], P[i], Z[i]], "op_sel:[0,1,0] op_sel_hi:[1,1,1]")    # T -> slot z
        for i in grp: b.emit("v_pk_add_f32", P[i], [X[i], Z[i]], "op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]") # re plane -> slot p
        if cplx:
            for i in grp: b.emit("v_pk_add_f32", X[i], [X[i], Z[i]], "op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[1,0]")   # (E.y + T.y, T.y - E.y)
            for i in grp: res[i] = (P[i], X[i])
            for i in grp: free.append(Z[i])
            # X stays as a result: take new temporaries from the freed z slots
        else:
            for i in grp: b.emit("v_pk_add_f32", X[i], [X[i], Z[i]], "op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]")   # im plane
            for i in grp: b.emit("v_pk_mul_f32", X[i], [X[i], X[i]])
            for i in grp: b.emit("v_pk_fma_f32", P[i], [P[i], P[i], X[i]])                                         # |.|^2 of both bins
            for i in grp: res[i] = (P[i], None)
            for i in grp: free.append(X[i])
    names = {i: c for i, (_, c) in enumerate(b.slots)}
    nm = "pk_post%s%d" % ("_cplx" if cplx else "_power", n)
    args = ", ".join("f2 z%d_, f2 p%d_, f2 w%d" % (i, i, i) for i in range(n))
    outs = ", ".join(("f2 &re%d, f2 &im%d" % (i, i)) if cplx else ("f2 &pw%d" % i) for i in range(n))
    lines = ["// %d slots of the post-pass -> %s" % (n, "planes (re_k, re_(M-k)), (im_k, im_(M-k))" if cplx else "(|X_k|^2, |X_(M-k)|^2)"),
             "__device__ __forceinline__ void %s(%s, %s) {" % (nm, args, outs),
             "  f2 " + ", ".join("z%d = z%d_, p%d = p%d_" % (i, i, i, i) for i in range(n)) + ", x0, x1;"]
    lines.append(b.text().rstrip("\n"))
    for i in range(n):
        if cplx:
            lines.append("  re%d = %s; im%d = %s;" % (i, names[res[i][0]], i, names[res[i][1]]))
        else:
            lines.append("  pw%d = %s;" % (i, names[res[i][0]]))
    lines.append("}")
    return "\n".join(lines)


def gen_powers8():
    """w^2 .. w^7 of a unit twiddle (6 complex products, depth 3): the radix-8 form of gen_powers16 (N = 16384 blocks of the FIR kernel)"""
    b = Block("pk_powers8")
    W = {1: b.slot("in", "w1")}
    for j in range(2, 8):
        W[j] = b.slot("tmp", "w%d" % j)
    def group(items):
        fin = [b.cmul_reg(W[d], W[x], W[y]) for d, x, y in items]
        for f in fin:
            f()
    group([(2, 1, 1)])
    group([(3, 2, 1), (4, 2, 2)])
    group([(5, 4, 1), (6, 4, 2), (7, 4, 3)])
    lines = ["// w[2] .. w[7] = powers of the unit twiddle w[1] (6 complex products, depth 3), one statement",
             "__device__ __forceinline__ void pk_powers8(f2 (&w)[16]) {",
             "  const f2 w1 = w[1];",
             "  f2 " + ", ".join("w%d" % j for j in range(2, 8)) + ";"]
    lines.append(b.text().rstrip("\n"))
    for j in range(2, 8):
        lines.append("  w[%d] = w%d;" % (j, j))
    lines.append("}")
    return "\n".join(lines)


def gen_powers16():
    """w^2 .. w^15 of a unit twiddle w by binary multiplication (depth 4), the powers of fftdev::twiddle_powers<16> on packed pairs:
    the cross-block radix-16 passes of the FIR kernel (fir.hip: fir_ols_pk32_kernel) multiply column n' by W_M^(n' r), r = 1 .. 15."""
    b = Block("pk_powers16")
    W = {1: b.slot("in", "w1")}
    for j in range(2, 16):
        W[j] = b.slot("tmp", "w%d" % j)
    def group(items):   # [(dst, a, b)]: W[dst] = W[a] * W[b]; products first, then the fused multiply-adds
        fin = [b.cmul_reg(W[d], W[x], W[y]) for d, x, y in items]
        for f in fin:
            f()
    group([(2, 1, 1)])
    group([(3, 2, 1), (4, 2, 2)])
    group([(5, 4, 1), (6, 4, 2), (7, 4, 3), (8, 4, 4)])
    group([(8 + j, 8, j) for j in range(1, 8)])
    lines = ["// w[2] .. w[15] = powers of the unit twiddle w[1] (14 complex products, depth 4), one statement",
             "__device__ __forceinline__ void pk_powers16(f2 (&w)[16]) {",
             "  const f2 w1 = w[1];",
             "  f2 " + ", ".join("w%d" % j for j in range(2, 16)) + ";"]
    lines.append(b.text().rstrip("\n"))
    for j in range(2, 16):
        lines.append("  w[%d] = w%d;" % (j, j))
    lines.append("}")
    return "\n".join(lines)


def gen_ols_pairs(n):
    """n pairs (k, M - k) of the overlap-save pointwise stage (fir.hip): real-FFT post-pass, product with H, inverse pre-pass, in place:
       E = A + conj B, D = A - conj B, t = w D, X_k = E - i t, X_(M-k) = conj(E + i t), Y = X H,
       P = Y_k + conj Y_(M-k), Q = Y_k - conj Y_(M-k), u = conj(w) Q,
       A' = conj(P + i u) = (P.x - u.y, -(P.y + u.x)),  B' = P - i u = (P.x + u.y, P.y - u.x)
    (the inverse transform runs as conj(FFT(conj .)), so the conjugates of Z'[k] are stored: fir_ols_split_kernel's `pair`)."""
    b = Block("pk_ols_pairs")
    A = {i: b.slot("io", "a%d" % i) for i in range(n)}
    B = {i: b.slot("io", "b%d" % i) for i in range(n)}
    T = {i: [b.slot("tmp", "x%d_%d" % (i, k)) for k in range(3)] for i in range(n)}
    W = {i: b.slot("in", "w%d" % i) for i in range(n)}
    HK = {i: b.slot("in", "hk%d" % i) for i in range(n)}
    HP = {i: b.slot("in", "hp%d" % i) for i in range(n)}
    R = range(n)
    E, D, t = {i: T[i][0] for i in R}, {i: T[i][1] for i in R}, {i: T[i][2] for i in R}
    for i in R: b.emit("v_pk_add_f32", E[i], [A[i], B[i]], "neg_hi:[0,1]")
    for i in R: b.emit("v_pk_add_f32", D[i], [A[i], B[i]], "neg_lo:[0,1]")
    fin = [b.cmul_reg(t[i], D[i], W[i]) for i in R]                        # t = w D
    for f in fin: f()
    for i in R: b.add_mi(A[i], E[i], t[i])                                  # X_k = E + (-i) t            -> slot a
    for i in R: b.sub_mi(B[i], E[i], t[i])                                  # conj X_(M-k) = E - (-i) t   -> slot b
    fin = [b.cmul_reg(E[i], A[i], HK[i]) for i in R]                        # Y_k -> slot E
    for f in fin: f()
    for i in R: b.emit("v_pk_mul_f32", D[i], [B[i], HP[i]], "op_sel_hi:[0,1]")                                  # Y_(M-k) = conj(slot b) * hp -> slot D
    for i in R: b.emit("v_pk_fma_f32", D[i], [B[i], HP[i], D[i]], "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]")
    for i in R: b.emit("v_pk_add_f32", A[i], [E[i], D[i]], "neg_hi:[0,1]")  # P -> slot a
    for i in R: b.emit("v_pk_add_f32", B[i], [E[i], D[i]], "neg_lo:[0,1]")  # Q -> slot b
    for i in R: b.emit("v_pk_mul_f32", t[i], [B[i], W[i]], "op_sel_hi:[0,1] neg_hi:[0,1]")                      # u = Q conj(w) -> slot t
    for i in R: b.emit("v_pk_fma_f32", t[i], [B[i], W[i], t[i]], "op_sel:[1,1,0] op_sel_hi:[1,0,1]")
    for i in R: b.add_mi(B[i], A[i], t[i])                                  # B' = P + (-i) u = (P.x + u.y, P.y - u.x)
    for i in R: b.emit("v_pk_add_f32", A[i], [A[i], t[i]], "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,1]")   # A' = (P.x - u.y, -(P.y + u.x))
    args = ", ".join("f2 &A%d, f2 &B%d, f2 w%d, f2 hk%d, f2 hp%d" % (i, i, i, i, i) for i in R)
    lines = ["// %d pair(s) (k, M - k) of the overlap-save pointwise stage, in place (18 packed instructions per pair), one statement" % n,
             "__device__ __forceinline__ void pk_ols_pairs%d(%s) {" % (n, args),
             "  f2 " + ", ".join("a%d = A%d, b%d = B%d, x%d_0, x%d_1, x%d_2" % (i, i, i, i, i, i, i) for i in R) + ";"]
    lines.append(b.text().rstrip("\n"))
    for i in R:
        lines.append("  A%d = a%d; B%d = b%d;" % (i, i, i, i))
    lines.append("}")
    return "\n".join(lines)


if __name__ == "__main__":
    print("// GENERATED by tools/gen/gen_pk_fft.py -- do not edit; see that file for the why and the register plans.")
    print("// Included by stft_fast_p32.hpp (inside namespace smx::<anon>); f2 = float ext_vector_type(2), an aligned register pair.")
    print(gen_fft16())
    print(gen_fft8())
    print(gen_fft4x2())
    print(gen_combine(0))
    print(gen_combine(1))
    for n in (8, 7):
        print(gen_twiddle(n))
    for n in (2, 4, 5):
        print(gen_post(n, False))
        print(gen_post(n, True))
    print(gen_powers16())
    print(gen_powers8())
    for n in (1, 2):
        print(gen_ols_pairs(n))
