#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): SQ counters of the two fft-2048 power kernels -> gpurun_out/<tag>/pmc_power.json
set -u
TAG=${1:-pmc_power}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ROOT/tools/pmc_power_driver.py > "$OUT/trace.log" 2>&1
find "$OUT/trace" -name '*kernel_stats.csv' | head -1 | xargs -r -I{} cp {} "$OUT/kernel_stats.csv"
i=0
while read -r group; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $group --output-format csv -d "$OUT/pmc_g$i" -- python3 $ROOT/tools/pmc_power_driver.py > "$OUT/pmc_g$i.log" 2>&1 || echo "group $i failed"
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_ADDR_CONFLICT
GRBM_GUI_ACTIVE GRBM_COUNT
WRITE_SIZE
GROUPS
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json, re
out_dir = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out_dir + '/pmc_g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'stft2048' not in k: continue
        k = re.sub(r'\(smx::.*', '', k.replace('void smx::(anonymous namespace)::', ''))
        rows[k][r['Counter_Name']].append(float(r['Counter_Value']))
dur = {}
for f in glob.glob(out_dir + '/kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(smx::.*', '', r['Name'].replace('void smx::(anonymous namespace)::', ''))
        dur[k] = {"calls": int(r['Calls']), "avg_us": float(r['AverageNs']) / 1e3, "min_us": float(r['MinNs']) / 1e3}
res = {"source": "tools/pmc_power.sh: rocprofv3 --pmc, one counter group per pass over tools/pmc_power_driver.py (C2 256 x 480000, fft 2048 / hop 512); per launch, average over the launches after the first", "kernels": {}}
for k, cs in rows.items():
    res["kernels"][k] = {"duration": dur.get(k), "counters": {c: (sum(v[1:]) / max(1, len(v) - 1) if len(v) > 1 else v[0]) for c, v in sorted(cs.items())}}
json.dump(res, open(out_dir + '/pmc_power.json', 'w'), indent=1)
for k, v in res["kernels"].items():
    print(k, v["duration"])
    print("   ", {n: round(x) for n, x in v["counters"].items()})
PY
rm -rf "$OUT"/pmc_g*/ "$OUT"/trace 2>/dev/null
echo done
