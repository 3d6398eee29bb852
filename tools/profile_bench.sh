#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box through gpurun).  Usage: tools/profile_bench.sh <tag> [bench args...]
# Pass 1: --kernel-trace --stats (per-kernel time).  Passes 2,3: PMC FETCH_SIZE / WRITE_SIZE, each in its own run
# (TCC slots: FETCH_SIZE 3 + WRITE_SIZE 2 > 4; and never combined with sys/hip traces).
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/bench_trace.log" 2>&1 || { tail -5 "$OUT/bench_trace.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o bench -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/bench_fetch.log" 2>&1 || { tail -5 "$OUT/bench_fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o bench -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/bench_write.log" 2>&1 || { tail -5 "$OUT/bench_write.log"; exit 1; }
python3 "$ROOT/tools/summarize_profile.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
