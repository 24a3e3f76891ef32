#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats and PMC passes for bench.py.
# Usage: tools/gpu_profile.sh <tag> [bench args...]
# Summaries land in gpurun_out/<tag>/ ; copy the ones to be judged into profiles/.
set -u
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline $*"
echo "== kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH --steps 20 --warmup 3 > "$OUT/trace.log" 2>&1
find "$OUT/trace" -name '*kernel_stats.csv' | head -1 | xargs -r head -8
pmc() {  # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_$name.log" 2>&1
  python3 - "$OUT/pmc_$name" <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in rows.items():
    if 'stft' in k or 'mel' in k or 'fir' in k:
        print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, 'launches', max(len(v) for v in cs.values()))
PY
}
echo "== pmc passes"
pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
pmc sq2 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc l2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
echo "== done"
