#!/bin/bash
# Runs ON THE GPU BOX: list counters once, then TLB / texture-path stall counters for bench.py
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-pmcx}
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters.txt" 2>&1
grep -o "Name:[[:space:]]*[A-Za-z0-9_]*" "$OUT/counters.txt" | awk '{print $2}' | sort -u > "$OUT/counter_names.txt"
wc -l "$OUT/counter_names.txt"
grep -i "utcl\|tlb\|stall" "$OUT/counter_names.txt" | tr '\n' ' '
echo
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1"
pmc() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- $BENCH > "$OUT/pmc_$name.log" 2>&1
  python3 - "$OUT/pmc_$name" <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r['Kernel_Name'][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in rows.items():
    if 'stft' in k:
        print({c: round(max(v), 1) for c, v in cs.items()})
PY
}
pmc t1 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum
pmc t2 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum
pmc t3 TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum
pmc t4 TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum
pmc t5 TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum TCC_WRREQ_STALL_max
pmc t6 SQ_INSTS_VALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES
