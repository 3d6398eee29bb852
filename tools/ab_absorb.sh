# A/B of the sliver absorption (SLS_ABSORB=0 restores one launch per one-wave class): bench.py per workload, value / ms / launch list
export SLS_LAB=1      # diagnostic knobs are honoured in lab mode only
for w in ${@:-chain4096}; do
for v in 1 0; do
  SLS_ABSORB=$v timeout -k 10 300 python bench.py --workload $w > gpurun_out/ab_absorb_${w}_$v.log 2>&1 || exit 1
  tail -1 gpurun_out/ab_absorb_${w}_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w absorb=$v', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['config'].get('unsolved_rank0'), d['config'].get('max_residual_rank0'))"
done
done
