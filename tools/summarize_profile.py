"""Condenses a tools/profile_bench.sh output directory into the text summary committed under profiles/."""
import csv, glob, json, os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
from collections import defaultdict
out = sys.argv[1]
def find(pattern):
    r = glob.glob(os.path.join(out, pattern), recursive=True)
    return r[0] if r else None
print("== rocprofv3 --kernel-trace --stats: kernel_stats ==")
f = find("trace/**/*kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = find(f"{tag}/**/*counter_collection.csv")
    print(f"== rocprofv3 --pmc {counter} ==")
    if not f:
        print("no counter file"); continue
    agg = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == counter:
            a = agg[r["Kernel_Name"][:60]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]:
        print(f"{k:60s} dispatches={n:6d} sum={v:.6g} per_dispatch={v / max(n, 1):.6g}")
for name in ("bench_trace.log", "bench_fetch.log", "bench_write.log"):
    p = os.path.join(out, name)
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines:
            d = json.loads(lines[-1])
            print(f"== {name}: value={d['value']} {d['unit']} ms_per_step={d['ms_per_step']} kernel_avg_ms={d['roofline']['kernel_avg_ms']} frac={d['roofline']['frac']}")
