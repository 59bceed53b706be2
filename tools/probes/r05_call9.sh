set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "bottleneck or split_k or tfsame or dense_head" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "forced_collectives" 2>&1 | tail -40
for v in "bneck=1" "bneck=0" "bneck=1" "bneck=0"; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 --opt $v 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05_ab_bneck.txt
done
for v in 8 16 12 8 16 12; do
  URSO_WGRAD_GROUP=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad_group=$v', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})" | tee -a gpurun_out/r05_ab_wgrad_group.txt
done
timeout 300 python tools/layer_profile.py 2>/dev/null | grep -E "bottleneck|dense|final" 
