#!/bin/bash
# rocprofv3 evidence for the closed-loop simulator (chain-4096, 250 steps, 1 scenario): kernel stats, then FETCH_SIZE in its own pass
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_closed_loop; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o cl -- python3 "$ROOT/tools/closed_loop_bench.py" chain4096 250 1 > "$OUT/run_trace.log" 2>&1 || { tail -5 "$OUT/run_trace.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o cl -- python3 "$ROOT/tools/closed_loop_bench.py" chain4096 250 1 > "$OUT/run_fetch.log" 2>&1 || { tail -5 "$OUT/run_fetch.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
f = glob.glob(os.path.join(out, "trace/**/*kernel_stats.csv"), recursive=True)[0]
print("== rocprofv3 --kernel-trace --stats (tools/closed_loop_bench.py chain4096 250 1) ==")
for r in list(csv.DictReader(open(f)))[:6]:
    print({k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")})
f = glob.glob(os.path.join(out, "pmc_fetch/**/*counter_collection.csv"), recursive=True)[0]
agg = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") == "FETCH_SIZE":
        a = agg[r["Kernel_Name"][:60]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print("== rocprofv3 --pmc FETCH_SIZE (KB; double it on gfx950 for streamed reads) ==")
for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:4]:
    print(f"{k:60s} dispatches={n:6d} per_dispatch={v / max(n, 1):.6g} KB")
for name in ("run_trace.log", "run_fetch.log"):
    print("==", name, open(os.path.join(out, name)).read().strip().splitlines()[-1])
PY
