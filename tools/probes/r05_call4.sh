set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "param or finalize or reduce or batched" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "parity_fp32 or set_trainable" 2>&1 | tail -5
for v in 8 4 2 8 4 2; do
  URSO_WGRAD_GROUP=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad_group=$v', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})" | tee -a gpurun_out/r05_ab_wgrad_group.txt
done
