#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the round's profile set.
#   tools/profile_round.sh <tag>     ->  gpurun_out/<tag>/{kernel_stats.csv, bench_n1.json, pmc.json, hot_kernel_stats.csv}
# 1. rocprofv3 --kernel-trace --stats of the bench command (the roofline's kernel duration must agree with it)
# 2. PMC passes (one counter group per pass, --pmc alone) over tools/pmc_driver.py: every hot kernel of C2 (power, complex, inverse) / C3 / C4
set -u
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# The bench command itself under the tracer, every `extra` included (C5's one-GPU point runs the frame-per-lane instantiation of the
# headline kernel, <true, 2, false, 2>, so it does not pollute the average of C2's <true, 2, false, 1>): the line it prints and the
# per-dispatch trace of the SAME run -- profiles/hbm_traffic.json takes every kernel's duration from this trace (median over the
# kernel's launches in the run: preconditioning + timed steps) and holds it against the `ms` this line reports.
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ROOT/bench.py --no-cpu-baseline > "$OUT/bench_n1.json" 2> "$OUT/trace.err"
find "$OUT/trace" -name '*kernel_stats.csv' | head -1 | xargs -r -I{} cp {} "$OUT/kernel_stats.csv"
find "$OUT/trace" -name '*kernel_trace.csv' | head -1 | xargs -r -I{} cp {} "$OUT/bench_kernel_trace.csv"
# the plain bench line of the same box (every `extra`): what profiles/hbm_traffic.json's per-kernel durations are held against
(cd "$ROOT" && timeout 600 python3 bench.py > "$OUT/bench_n1_full.json" 2> "$OUT/bench_full.err")
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_hot" -- python3 $ROOT/tools/pmc_driver.py > "$OUT/trace_hot.log" 2>&1
find "$OUT/trace_hot" -name '*kernel_stats.csv' | head -1 | xargs -r -I{} cp {} "$OUT/hot_kernel_stats.csv"
find "$OUT/trace_hot" -name '*kernel_trace.csv' | head -1 | xargs -r -I{} cp {} "$OUT/hot_kernel_trace.csv"
i=0
while read -r group; do
  i=$((i+1))
  SUSTAIN_MS=0 timeout 300 rocprofv3 --pmc $group --output-format csv -d "$OUT/pmc_g$i" -- python3 $ROOT/tools/pmc_driver.py > "$OUT/pmc_g$i.log" 2>&1 || echo "group $i failed"
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SMEM
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
GRBM_GUI_ACTIVE GRBM_COUNT
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum
GROUPS
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json, re
out_dir = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out_dir + '/pmc_g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not re.search(r'stft2048|stft4096|stft_power_lanes|mel_apply|fir_ols|istft2048', k):
            continue
        k = re.sub(r'\(smx::.*', '', k.replace('void smx::(anonymous namespace)::', ''))
        rows[k][r['Counter_Name']].append(float(r['Counter_Value']))
# durations: the traced bench run (bench_kernel_trace.csv): median over a kernel's launches in that run -- the same program, the
# same preconditioning as the line in bench_n1.json; beside them the driver's own sustained figure (last 10 of >= 600 ms of launches)
import statistics
dur = {}
def by_kernel(path):
    per = collections.defaultdict(list)
    for f in glob.glob(path):
        for r in csv.DictReader(open(f)):
            k = re.sub(r'\(smx::.*', '', r['Kernel_Name'].replace('void smx::(anonymous namespace)::', ''))
            per[k].append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    for v in per.values():
        v.sort()
    return per
bench_per, drv_per = by_kernel(out_dir + '/bench_kernel_trace.csv'), by_kernel(out_dir + '/hot_kernel_trace.csv')
for k, v in drv_per.items():
    last = [d for _, d in v[-10:]]
    dur[k] = {"driver_calls": len(v), "driver_last10_avg_us": sum(last) / len(last) / 1e3}
    if k in bench_per:
        b = [d for _, d in bench_per[k]]
        dur[k].update({"calls": len(b), "avg_us": statistics.median(b) / 1e3, "min_us": min(b) / 1e3, "mean_us": sum(b) / len(b) / 1e3,
                       "of": "median over the kernel's %d launches in the traced bench.py run (bench_n1.json)" % len(b)})
    else:
        dur[k].update({"calls": len(v), "avg_us": sum(last) / len(last) / 1e3, "min_us": min(last) / 1e3, "of": "the last 10 of %d back-to-back launches of tools/pmc_driver.py (no bench line for this kernel)" % len(v)})
res = {"source": "tools/profile_round.sh: rocprofv3 --pmc, one counter group per pass over tools/pmc_driver.py (C2 256 x 480000 fft 2048 / hop 512; "
                 "C3 128 mels; C4 8192 taps on 8 x 2880000); counters per launch, average over each kernel's last 10 launches; durations from the --kernel-trace of bench.py "
                 "itself (median over the kernel's launches in that run), the driver's own sustained figure beside them; FETCH_SIZE / WRITE_SIZE in KB as reported (FETCH_SIZE counts half the bytes "
                 "of wide coalesced reads on gfx950: MI355X_MICROARCH.md)", "kernels": {}}
for k, cs in rows.items():
    res["kernels"][k] = {"duration": dur.get(k), "counters": {c: sum(v[-10:]) / len(v[-10:]) for c, v in sorted(cs.items())}}
json.dump(res, open(out_dir + '/pmc.json', 'w'), indent=1)
for k, v in res["kernels"].items():
    c = v["counters"]
    print(k, v["duration"])
    print("   ", {n: round(c[n]) for n in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_LDS_BANK_CONFLICT", "FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_WRREQ_sum", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE") if n in c})
PY
rm -rf "$OUT"/pmc_g*/ "$OUT"/trace "$OUT"/trace_hot 2>/dev/null
echo done
