"""Registers, scratch and instruction mix of ONE instantiation of a fft-2048 kernel from hipcc -S of stft_fast.hip with
-DSMX_ISA_ONE=<V>: seconds instead of the minutes the whole file takes; no GPU.  Default: stft2048_power32_kernel<true, 2, false, V>
(V = the flush form: 1 pairs, 2 frames, 0 plain); -DSMX_ISA_KERNEL=1: stft2048_complex32_kernel<true, V != 0>; =2: stft2048_mel32_kernel<true, 2, V>; =3 / 4 / 5: the lanes kernels (V = 16 / 8 / 4 lanes) power / mel / complex; =6: stft2048_complex_fm_kernel<V != 0>.
  python tools/isa_one.py [V=1] [extra -D flags ...]        (assembly left in /tmp/isa_one_<V>.s)"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
skew = sys.argv[1] if len(sys.argv) > 1 else "1"
out = "/tmp/isa_one_%s.s" % skew
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                       "-DSMX_ISA_ONE=" + skew, "-o", out, os.path.join(ROOT, "soundml_amd", "csrc", "stft_fast.hip")] + sys.argv[2:], stderr=subprocess.DEVNULL, cwd="/tmp")
text = open(out).read()
for m in re.finditer(r"^(_Z\S+):.*?\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    c = collections.Counter(l.strip().split()[0] for l in body.split("\n") if l.startswith("\t") and not l.startswith("\t.") and l.strip())
    vg = re.search(re.escape(name) + r"\.num_vgpr, (\d+)", text)
    sc = re.search(re.escape(name) + r"\.private_seg_size, (\d+)", text)
    print(name[-70:], "vgpr", vg and vg.group(1), "scratch", sc and sc.group(1), "instrs", sum(c.values()))
    print("   ", {k: v for k, v in c.most_common(400) if k.startswith(("scratch", "v_writelane", "v_readlane", "v_mov_b32", "v_cndmask", "ds_", "global_", "s_waitcnt", "s_nop", "v_accvgpr"))})
