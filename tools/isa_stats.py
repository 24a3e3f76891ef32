"""Per-kernel ISA statistics of stft_fast.hip (VGPRs, scratch, instruction mix) from hipcc -S; no GPU needed."""
import collections, re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else "stft_fast.hip"
pat = sys.argv[2] if len(sys.argv) > 2 else "ILb1ELb1ELb0"
out = "/tmp/isa_%s.s" % os.path.basename(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-S",
                       "--cuda-device-only", "-o", out, os.path.join(ROOT, "soundml_amd", "csrc", src)] + sys.argv[3:],
                      stderr=subprocess.DEVNULL, cwd="/tmp")
text = open(out).read()
for m in re.finditer(r"^(_Z\S+):.*?\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    c = collections.Counter()
    for line in body.split("\n"):
        if line.startswith("\t") and not line.startswith("\t.") and line.strip():
            t = line.strip().split()[0]
            if re.match(r"(v_|s_|ds_|global_|buffer_|scratch_)", t):
                c[t] += 1
    vg = re.search(re.escape(name) + r"\.num_vgpr, (\d+)", text)
    sc = re.search(re.escape(name) + r"\.private_seg_size, (\d+)", text)
    groups = collections.Counter()
    for k, v in c.items():
        g = ("dpp" if "dpp" in k else "permlane" if "permlane" in k else "cndmask" if "cndmask" in k else
             "mov" if k.startswith("v_mov") or k.startswith("v_accvgpr") else "valu" if k.startswith("v_") else
             "lds" if k.startswith("ds_") else "scratch" if k.startswith("scratch") else
             "vmem" if k.startswith(("global", "buffer")) else "s_nop" if k == "s_nop" else
             "s_waitcnt" if k == "s_waitcnt" else "salu")
        groups[g] += v
    print(name[:90], "vgpr", vg and vg.group(1), "scratch", sc and sc.group(1), "instrs", sum(c.values()))
    print("   groups:", dict(groups.most_common()))
    print("   top:", dict(c.most_common(30)))
