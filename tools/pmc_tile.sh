#!/bin/bash
# SQ counter passes for a bench workload, summarised per kernel (instruction mix, waits, MFMA).  Usage: tools/pmc_tile.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc$i" -o bench -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/bench$i.log" 2>&1 || { tail -5 "$OUT/bench$i.log"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(sys.argv[1], "pmc*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, v in agg.items():
    if "h2_column" not in k: continue
    print(k)
    for c, x in sorted(v.items()): print(f"   {c:30s} {x / max(cnt[k][c], 1):14.5g} per dispatch ({cnt[k][c]} dispatches)")
PY
