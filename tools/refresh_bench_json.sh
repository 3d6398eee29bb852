#!/bin/bash
# One call on the GPU box: the bench lines behind profiles/r03_bench_*.json at the current build (copy gpurun_out/benchjson/* into profiles/).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/benchjson
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > gpurun_out/benchjson/$name.log 2>&1 && tail -1 gpurun_out/benchjson/$name.log > gpurun_out/benchjson/r03_bench_$name.json; }
run readme &&
run chain4096 --workload chain4096 &&
run grid32 --workload grid32 &&
run random10000 --workload random10000_d2 --steps 10 --warmup 2 &&
run random10000_act1 --workload random10000_d2_act1 --steps 10 --warmup 2 &&
run chain4096_sum_of_norms --workload chain4096 --objective sum_of_norms --steps 3 --warmup 1
for f in gpurun_out/benchjson/r03_bench_*.json; do python3 -c "import json,sys; d=json.loads(open('$f').read()); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
