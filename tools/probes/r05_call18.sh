mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "gradient_norm_from or forced_collectives or graph_replay or parity_fp32 or adam or frozen or set_trainable" 2>&1 | tail -8
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "param or finalize or batched or sgd or adam" 2>&1 | tail -4
for v in 1 0 1 0; do
  URSO_FUSE_SQNORM=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse_sqnorm=$v', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if k in ('param_grad_finalize','optimizer')})" | tee -a gpurun_out/r05_ab_fuse_sqnorm.txt
done
