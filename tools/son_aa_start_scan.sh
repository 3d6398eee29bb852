# sum-of-norms pass on chain-4096 for several Anderson start steps (SLS_SON_AA_START): value, ms per pass, columns not converged
export SLS_LAB=1      # diagnostic knobs are honoured in lab mode only
for st in ${@:-20 10 5 2}; do
  SLS_SON_AA_START=$st timeout -k 10 300 python bench.py --workload chain4096 --objective sum_of_norms --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/son_start_$st.log 2>&1 || exit 1
  tail -1 gpurun_out/son_start_$st.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('aa_start=$st', d['value'], d['ms_per_step'], 'unsolved', c.get('unsolved_rank0'), 'max passes', c.get('max_refinement_passes'))"
done
