#!/bin/bash
# SQ counter pass (issue vs wait) for a bench workload.  Usage: tools/pmc_sq.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sq_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d "$OUT/pmc" -o bench -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1 || { tail -5 "$OUT/bench.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "pmc/**/*counter_collection.csv"), recursive=True)[0]
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:48]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, v in agg.items():
    if "h2_column" not in k: continue
    n = max(cnt[k], 1)
    print(k, "dispatches", n)
    for c, x in sorted(v.items()): print(f"   {c:22s} {x / n:14.4g} per dispatch")
    wc = v.get("SQ_WAVE_CYCLES", 1)
    print(f"   wait_any/wave_cycles = {v.get('SQ_WAIT_ANY', 0) / wc:.3f}   active_inst/wave_cycles = {v.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}   wait_inst/wave_cycles = {v.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}")
PY
