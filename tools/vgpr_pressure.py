"""Approximate VGPR pressure along a kernel's assembly (linear backward liveness over a line range; branches ignored):
  python tools/vgpr_pressure.py <file.s> <first line> <last line>
prints the live-register count every N instructions and the maximum.  No GPU needed."""
import re, sys
path, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lines = open(path).read().split("\n")[a:b]
ins = []
for l in lines:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    t = t.split(";")[0].strip()
    if not t or t.endswith(":"):
        continue
    ins.append(t)
def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out
defs_uses = []
for t in ins:
    op, _, rest = t.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    d, u = set(), set()
    if op.startswith(("v_", "ds_read", "global_load", "scratch_load", "buffer_load")) and not op.startswith(("v_cmp", "v_readfirstlane", "v_readlane")):
        d = regs(ops[0]) if ops else set()
        for o in ops[1:]:
            u |= regs(o)
        if op.startswith(("v_fmac", "v_mac")) or "dpp" in op or "permlane" in op:
            u |= d
    else:
        for o in ops:
            u |= regs(o)
    defs_uses.append((op, d, u))
live = set()
counts = [0] * len(ins)
for i in range(len(ins) - 1, -1, -1):
    op, d, u = defs_uses[i]
    live -= d
    live |= u
    counts[i] = len(live)
mx = max(counts)
print("instructions", len(ins), "max live", mx, "at instruction", counts.index(mx), ins[counts.index(mx)])
step = max(1, len(ins) // 60)
for i in range(0, len(ins), step):
    print(i, counts[i], ins[i][:60])
