#!/bin/bash
export SLS_LAB=1      # diagnostic knobs are honoured in lab mode only
# A/B of two prebuilt libraries on the throughput workloads: $1 = alternative .so (the default library is "A")
D=systemlevelcontrol.jl_amd
cp $D/libsls_mi355x.so /tmp/lib_A.so
run() {
  for w in chain4096 chain1024; do
    python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  $w', j['value'], j['ms_per_step'], j['roofline']['kernel_avg_ms'], j['config']['max_residual_rank0'])"
  done
  SLS_NO_TWISTED=1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  readme one-wave', j['value'], j['ms_per_step'], j['roofline']['kernel_avg_ms'])"
}
echo "A (default build)"; run
cp $1 $D/libsls_mi355x.so; echo "B ($1)"; run
cp /tmp/lib_A.so $D/libsls_mi355x.so; echo "A again"; run
