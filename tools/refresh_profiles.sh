#!/bin/bash
# One call on the GPU box: the rocprofv3 passes behind profiles/r03_* (tools/update_profiles.py copies them in afterwards).
# Every workload: pass 1 --kernel-trace --stats, passes 2/3 --pmc FETCH_SIZE / WRITE_SIZE (own runs); the tile-kernel workloads
# also the SQ counter passes of tools/pmc_tile.sh.  The sum-of-norms launch gets the same three passes + SQ waits.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
timeout -k 10 300 bash tools/profile_bench.sh readme --steps 60 --warmup 5 > gpurun_out/prof_readme.log 2>&1 &&
timeout -k 10 300 bash tools/profile_bench.sh chain4096 --workload chain4096 --steps 15 --warmup 2 > gpurun_out/prof_chain4096.log 2>&1 &&
timeout -k 10 400 bash tools/profile_bench.sh grid32 --workload grid32 --steps 15 --warmup 2 > gpurun_out/prof_grid32.log 2>&1 &&
timeout -k 10 300 bash tools/pmc_tile.sh grid32 --workload grid32 --steps 6 --warmup 2 > gpurun_out/pmc_grid32.txt 2>&1 &&
timeout -k 10 500 bash tools/profile_bench.sh random10000 --workload random10000_d2 --steps 4 --warmup 1 > gpurun_out/prof_random10000.log 2>&1 &&
timeout -k 10 400 bash tools/pmc_tile.sh random10000 --workload random10000_d2 --steps 2 --warmup 1 > gpurun_out/pmc_random10000.txt 2>&1 &&
timeout -k 10 500 bash tools/profile_bench.sh random10000_act1 --workload random10000_d2_act1 --steps 4 --warmup 1 > gpurun_out/prof_random10000_act1.log 2>&1 &&
timeout -k 10 400 bash tools/pmc_tile.sh random10000_act1 --workload random10000_d2_act1 --steps 2 --warmup 1 > gpurun_out/pmc_random10000_act1.txt 2>&1 &&
timeout -k 10 600 bash tools/profile_bench.sh son --workload chain4096 --objective sum_of_norms --steps 3 --warmup 1 > gpurun_out/prof_son.log 2>&1 &&
timeout -k 10 400 bash tools/pmc_tile.sh son --workload chain4096 --objective sum_of_norms --steps 2 --warmup 1 > gpurun_out/pmc_son.txt 2>&1
echo profiles done rc=$?
