#!/bin/bash
# Runs ON THE GPU BOX: MFMA counters of the two mel kernels at C3 (north_star: "MFMA utilisation for the mel GEMM").
# One pass per group; MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs... see DESIGN 5).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-pmcmel}
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
pmc() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 $ROOT/tools/mel_pmc_driver.py > "$OUT/pmc_$name.log" 2>&1
  python3 - "$OUT/pmc_$name" "$OUT/$name.json" <<'PY'
import csv, glob, sys, collections, json
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: sum(v[1:]) / max(1, len(v) - 1) for c, v in cs.items()} for k, cs in rows.items() if 'mel' in k}
print(json.dumps(out, indent=1))
json.dump(out, open(sys.argv[2], 'w'), indent=1)
PY
}
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES
pmc act GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ROOT/tools/mel_pmc_driver.py > "$OUT/trace.log" 2>&1
find "$OUT/trace" -name '*kernel_stats.csv' | head -1 | xargs -r head -6
find "$OUT/trace" -name '*kernel_stats.csv' | head -1 | xargs -r -I{} cp {} "$OUT/mel_kernel_stats.csv"
