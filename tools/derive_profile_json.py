#!/usr/bin/env python3
"""From profiles/<round>/pmc_hot_kernels.json (tools/profile_round.sh) to the two small files bench.py quotes:
  profiles/hbm_traffic.json   HBM bytes per launch of the headline kernel (roofline.traffic), counters corrected as
                              MI355X_MICROARCH.md prescribes (FETCH_SIZE x 2 on gfx950 for wide coalesced reads; WRITE_SIZE as is)
  profiles/mfma_util.json     MFMA-pipe utilisation of the two mel kernels (north_star: "MFMA utilisation for the mel GEMM")
Usage: python tools/derive_profile_json.py r04"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r07"
src = os.path.join(ROOT, "profiles", rnd, "pmc_hot_kernels.json")
k = json.load(open(src))["kernels"]
hkey = next(n for n in ("stft2048_power32_kernel<true, 2, false, 1>", "stft2048_power32_kernel<true, 2, false, 2>", "stft2048_power32_kernel<true, 2, false, 0>",
                        "stft2048_power32_kernel<true, 2, false>") if n in k)
head = k[hkey]
c = head["counters"]
read_b, write_b = int(c["FETCH_SIZE"] * 1024 * 2), int(c["WRITE_SIZE"] * 1024)
algo = 256 * 938 * 6148
json.dump({"bytes_per_launch": read_b + write_b, "read_bytes": read_b, "write_bytes": write_b, "algorithmic_bytes": algo,
           "fetch_factor": 2, "fetch_factor_why": "TCC_EA0_RDREQ %.2f M requests for 491.5 MB of audio = %.0f B per request: 128-byte requests, which FETCH_SIZE tallies at 64" % (c.get("TCC_EA0_RDREQ_sum", 0) / 1e6, 256 * 480000 * 4 / max(1.0, c.get("TCC_EA0_RDREQ_sum", 1.0))),
           "write_amplification": round(write_b / (256 * 938 * 4100), 3),
           "note": "one launch of the C2 workload (938 frames per clip); FETCH_SIZE doubled per the gfx950 correction (calibrated for "
                   "16 B/lane streams; these loads are 8 B/lane, so the read side is an upper estimate between 1x and 2x FETCH_SIZE); "
                   "WRITE_SIZE as reported",
           "profile": "profiles/%s/pmc_hot_kernels.json" % rnd,
           # what the figure was taken at: bench.py quotes it beside the run it did NOT measure it in
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (tools/profile_round.sh) -> profiles/%s/pmc_hot_kernels.json" % rnd,
           "kernel": hkey, "kernel_us_in_profile": round(head["duration"]["avg_us"], 1) if head.get("duration") else None,
           "commit": os.popen("git -C %s rev-parse --short HEAD 2>/dev/null" % ROOT).read().strip() or None,
           "box": "one MI355X gpurun box (fresh lease; boxes differ by +-10 % in kernel time)",
           "per_kernel": None}, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
# every hot kernel: counter traffic against its algorithmic bytes (the launches of tools/pmc_driver.py)
FR2048, FR1K, FR512, FR256 = 938, 1723, 3446, 6891
# (bytes read, bytes written) that the kernel must move per launch of tools/pmc_driver.py
ALGO = {"stft2048_power32": (256 * FR2048 * 2048, 256 * FR2048 * 4100), "stft2048_complex32": (256 * FR2048 * 2048, 256 * FR2048 * 8200),
        "stft2048_complex_fm": (256 * FR2048 * 2048, 256 * FR2048 * 8200),
        "istft2048_pipe_kernel<true, true>": (256 * FR2048 * (8200 * 2 + 4100), 256 * FR2048 * 2048), "stft2048_mel32": (256 * FR2048 * 2048, 256 * FR2048 * 512),
        "istft2048": (256 * FR2048 * 8200, 256 * FR2048 * 2048), "mel_apply_mfma": (256 * FR2048 * 4100, 256 * FR2048 * 512), "fir_ols": (8 * 2880000 * 4, 8 * 2880000 * 4),
        "stft_power_lanes_kernel<16": (256 * FR1K * 1024, 256 * FR1K * 2052), "stft_power_lanes_kernel<8": (256 * FR512 * 512, 256 * FR512 * 1028),
        "stft_power_lanes_kernel<4": (256 * FR256 * 256, 256 * FR256 * 516),
        "wide64::stft2048_power_wide": (256 * FR2048 * 2048, 256 * FR2048 * 4100),
        "stft4096_power64": (256 * 431 * 4096, 256 * 431 * 8196)}


def fetch_factor(cc, algo_read):
    """FETCH_SIZE = TCC_EA0_RDREQ x 64 B on gfx950 whatever the request's size (MI355X_MICROARCH.md: a 128-byte request is tallied
    at 64).  Which size a kernel's requests have is read off the request counter against the bytes the kernel must read: the
    power spectrogram's 4.00 M requests fetch 491.5 MB = 123 B each (128-byte requests: x 2), the inverse pipeline's 28.9 M fetch
    1.969 GB = 68 B each (64-byte requests: x 1; x 2 would claim 1.7x the spectra it reads).  x 1 where the requests come out
    below 96 bytes each AND FETCH_SIZE x 1 already covers (>= 0.9 of) the algorithmic reads; x 2 otherwise (an upper bound where the sizes mix)."""
    req = cc.get("TCC_EA0_RDREQ_sum")
    if not req or not algo_read:
        return 2, None, "no request counter: the guide's x 2"
    per = algo_read / req
    x1 = cc["FETCH_SIZE"] * 1024
    if per < 96 and x1 >= 0.90 * algo_read:
        return 1, per, "%.0f B of algorithmic reads per TCC_EA0_RDREQ request: 64-byte requests, FETCH_SIZE as reported" % per
    return 2, per, ("%.0f B of algorithmic reads per TCC_EA0_RDREQ request: 128-byte requests, FETCH_SIZE x 2" % per) if per >= 96 else \
        ("%.0f B of algorithmic reads per request but FETCH_SIZE x 1 is below the algorithmic reads: mixed request sizes, x 2 is an upper bound" % per)


table = {}
for name, v in k.items():
    cc = v["counters"]
    if "FETCH_SIZE" not in cc or "WRITE_SIZE" not in cc:
        continue
    algo_rw = next((b for key, b in ALGO.items() if name.startswith(key)), None)
    algo_b = sum(algo_rw) if algo_rw else None
    factor, per_req, why = fetch_factor(cc, algo_rw[0] if algo_rw else None)
    rb, wb = int(cc["FETCH_SIZE"] * 1024 * factor), int(cc["WRITE_SIZE"] * 1024)
    table[name] = {"read_bytes": rb, "fetch_factor": factor, "fetch_factor_why": why, "write_bytes": wb, "algorithmic_bytes": algo_b,
                   "algorithmic_read_bytes": algo_rw[0] if algo_rw else None, "algorithmic_write_bytes": algo_rw[1] if algo_rw else None,
                   "traffic_over_algorithmic": round((rb + wb) / algo_b, 3) if algo_b else None,
                   "read_over_algorithmic": round(rb / algo_rw[0], 3) if algo_rw else None, "write_over_algorithmic": round(wb / algo_rw[1], 3) if algo_rw else None,
                   # the other choice of the factor, for the record
                   "traffic_over_algorithmic_other_factor": round((int(cc["FETCH_SIZE"] * 1024 * (3 - factor)) + wb) / algo_b, 3) if algo_b else None,
                   # what the CU's memory pipe carried: vector-memory read instructions x 512 B (64 lanes x 8 B) against the algorithmic reads
                   "vmem_read_request_bytes_over_algorithmic_reads": round(cc["SQ_INSTS_VMEM_RD"] * 512 / algo_rw[0], 3) if (algo_rw and cc.get("SQ_INSTS_VMEM_RD")) else None,
                   "avg_us": round(v["duration"]["avg_us"], 1) if v.get("duration") else None,
                   "lds_bank_conflict_over_active": round(cc["SQ_LDS_BANK_CONFLICT"] / cc["SQ_LDS_IDX_ACTIVE"], 4) if cc.get("SQ_LDS_IDX_ACTIVE") else None}
# the bench line of the traced run itself (profiles/<round>/bench_n1.json): every kernel's duration in that run's trace beside the
# `ms` the line reports for it -- tests/test_host_logic.py holds them within 5 %
bench_path = os.path.join(ROOT, "profiles", rnd, "bench_n1.json")   # the line of the TRACED run the durations come from
if os.path.exists(bench_path):
    line = json.loads(open(bench_path).read().strip().splitlines()[-1])
    ex = line.get("extra", {})
    pairs = {"istft2048_pipe_kernel<true, true>": None, "stft2048_complex_fm": None,   # Griffin-Lim's kernels: the line has the whole loop only (extra.c2_griffin_lim)
             "stft2048_power32": line["roofline"].get("kernel_ms_avg"), "stft2048_complex32": ex.get("c2_complex", {}).get("ms"),
             "istft2048": ex.get("c2_invert", {}).get("ms"), "stft2048_mel32": ex.get("c3_mel", {}).get("ms"),
             "stft_power_lanes_kernel<16": ex.get("c1_batch", {}).get("ms"), "stft_power_lanes_kernel<4": ex.get("fft256_batch", {}).get("ms"), "fir_ols": ex.get("c4_fir", {}).get("ms"),
             "wide64::stft2048_power_wide": ex.get("c2_float64_interior", {}).get("ms"), "stft4096_power64": ex.get("fft4096_batch", {}).get("ms")}
    # the fft-4096 step is TWO launches (launches_per_step in the line): the gather of the clips' border strips, then the pipeline; the
    # event pair around the step holds both, so the gather's average in the same trace (kernel_stats.csv) stands beside the pipeline's
    gather_us = None
    stats_path = os.path.join(ROOT, "profiles", rnd, "kernel_stats.csv")
    if os.path.exists(stats_path):
        import csv
        for r in csv.DictReader(open(stats_path)):
            if "gather_padded_kernel" in r["Name"]:
                gather_us = round(float(r["AverageNs"]) / 1e3, 1)
    for name, row in table.items():
        ms = next((v for key, v in pairs.items() if name.startswith(key)), None)
        row["bench_ms"] = ms
        if name.startswith("stft4096_power64") and ms is not None and gather_us is not None:
            row["companion_us"] = gather_us
            row["companion"] = "gather_padded_kernel (the border strips), launched before the pipeline inside the same timed step"
        row["bench_source"] = "profiles/%s/bench_n1.json (the traced run itself)" % rnd if ms is not None else None
tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
tj["per_kernel"] = table
json.dump(tj, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
out = {"profile": "profiles/%s/pmc_hot_kernels.json" % rnd,
       "definition": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); fp32 MFMA peak 157.3 TFLOP/s (MI355X_MICROARCH.md)"}
# the fused kernel's product: v_mfma_f32_4x4x1_16B_f32 (16 blocks x 4 x 4 x 1 x 2 = 512 flop) where the banded plan is taken
# (kernel<.., 2, 2> / <.., 2, 1>), v_mfma_f32_16x16x4_f32 (2048 flop) for the dense one; Mel.apply: v_mfma_f32_32x32x2_f32 (4096)
mel_key = next(n for n in ("stft2048_mel32_kernel<true, 2, 2>", "stft2048_mel32_kernel<true, 2, 1>", "stft2048_mel32_kernel<true, 2, 0>", "stft2048_mel32_kernel<true, 2>") if n in k)
for name, key, flop_per_mfma in (("fused_audio_to_mel", mel_key, 2048 if mel_key.endswith(("2, 0>", "true, 2>")) else 512), ("mel_apply", "mel_apply_mfma_kernel<true>", 4096)):
    cc, dur = k[key]["counters"], k[key]["duration"]
    util = cc["SQ_VALU_MFMA_BUSY_CYCLES"] / (cc["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    flops = cc["SQ_INSTS_VALU_MFMA_F32"] * flop_per_mfma
    out[name] = {"kernel": key, "mfma_pipe_busy": round(util, 4), "executed_tflops": round(flops / (dur["avg_us"] * 1e-6) / 1e12, 1),
                 "avg_us": round(dur["avg_us"], 1)}
json.dump(out, open(os.path.join(ROOT, "profiles", "mfma_util.json"), "w"), indent=1)
print(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")).read())
print(json.dumps(out, indent=1))
