#!/bin/bash
# One call on the GPU box: the rocprofv3 passes and bench lines behind profiles/r02_* (tools/update_profiles.py copies them in).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
timeout -k 10 300 bash tools/profile_bench.sh readme --steps 60 --warmup 5 > gpurun_out/prof_readme.log 2>&1 &&
timeout -k 10 300 bash tools/profile_bench.sh chain4096 --workload chain4096 --steps 15 --warmup 2 > gpurun_out/prof_chain4096.log 2>&1 &&
timeout -k 10 400 bash tools/profile_bench.sh grid32 --workload grid32 --steps 15 --warmup 2 > gpurun_out/prof_grid32.log 2>&1 &&
timeout -k 10 300 bash tools/pmc_tile.sh grid32 --workload grid32 --steps 6 --warmup 2 > gpurun_out/pmc_grid32.txt 2>&1 &&
timeout -k 10 500 bash tools/profile_bench.sh random10000 --workload random10000_d2 --steps 4 --warmup 1 > gpurun_out/prof_random10000.log 2>&1 &&
timeout -k 10 400 bash tools/pmc_tile.sh random10000 --workload random10000_d2 --steps 2 --warmup 1 > gpurun_out/pmc_random10000.txt 2>&1
mkdir -p gpurun_out/prof_son
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_son/trace" -o bench -- python3 "$ROOT/bench.py" --workload chain4096 --objective sum_of_norms --steps 3 --warmup 1 > "$ROOT/gpurun_out/prof_son/bench_trace.log" 2>&1)
python3 tools/summarize_profile.py gpurun_out/prof_son > gpurun_out/prof_son/summary.txt
echo profiles done
