mkdir -p gpurun_out
for o in "bneck=1" "pwx_bn=128" "pair_single=1" "pair_single=2" "pair_single=0" "pwx=2" "hconv2=2" "c3v=0" "bneck=1" "wgrad_big=0" "stem_pool=0"; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 --opt $o 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-16s' % '$o', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if k in ('conv_igemm','conv_wgrad')})" | tee -a gpurun_out/r05_option_sweep.txt
done
