#!/bin/bash
export SLS_LAB=1      # diagnostic knobs are honoured in lab mode only
# scan of the regularisation δ_rel at stopping tolerance 1e-10: GPU suite pass/fail + bench lines
for d in 1e-12 1e-13 1e-14 1e-15; do
  export SLS_DELTA_REL=$d SLS_TOL=1e-10
  echo "=== delta_rel $d"
  timeout -k 10 300 python -m pytest tests -m gpu -q 2>&1 | tail -6
  python bench.py --steps 50 --warmup 5 2>&1 | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('readme', j['value'], j['config']['max_residual_rank0'], j['config']['max_refinement_passes'])"
  python bench.py --workload chain4096 --steps 10 --warmup 2 2>&1 | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain4096', j['value'], j['config']['max_residual_rank0'], j['config']['max_refinement_passes'])"
done
