for v in 0 8 0 8; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 --opt pwx_dbg=$v 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pwx_dbg=$v', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})" | tee -a gpurun_out/r05_ab_pwx_dbg8.txt
done
