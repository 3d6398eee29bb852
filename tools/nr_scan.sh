#!/bin/bash
export SLS_LAB=1      # diagnostic knobs are honoured in lab mode only
# Newton steps per pivot reciprocal (SLS_GJ_NR): pass counts / residuals / bench for the prebuilt variants _nr1.so, _nr0.so
D=systemlevelcontrol.jl_amd
cp $D/libsls_mi355x.so /tmp/lib_nr2.so
for v in 2 1 0; do
  if [ $v = 2 ]; then cp /tmp/lib_nr2.so $D/libsls_mi355x.so; else cp $D/_nr$v.so $D/libsls_mi355x.so; fi
  echo "=== NR=$v"
  timeout 100 python tools/iters_hist.py readme_chain | grep -v "^h2_\|^  iters"
  timeout 100 python tools/debug_weighted.py | grep "iters\|err/col" | cut -c1-200
  timeout 100 python tools/grid_cols.py | tail -1
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', j['value'], j['ms_per_step'], j['roofline']['kernel_avg_ms'])"
done
cp /tmp/lib_nr2.so $D/libsls_mi355x.so
