#!/bin/bash
# Runs ON THE GPU BOX: memory-path counters (TCC / TCP only) of bench.py's kernels, one pass per group.
TAG=${1:-pmcmem}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  i=$((i+1))
  timeout 60 rocprofv3 --pmc $group --output-format csv -d "$OUT/g$i" -- python3 $ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 > "$OUT/g$i.log" 2>&1 || echo "group $i failed"
done <<'GROUPS'
TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_CYCLE_sum
TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum
TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum
GROUPS
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'power_kernel' in k and 'true, false' in k.split('(')[0][-30:]:
            rows['interior'][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in rows.items():
    for c, v in sorted(cs.items()):
        print("%-44s %16.0f" % (c, max(v)))
PY
