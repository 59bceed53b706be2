set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "bottleneck or dense_head" 2>&1 | tail -15
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_layerwise_gpu.py tests/test_net_api_gpu.py -x -q 2>&1 | tail -25
for v in 1 0 1 0; do
  URSO_DENSE_MULTI=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dense_multi=$v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05_ab_dense_multi.txt
done
timeout 300 python tools/layer_profile.py 2>/dev/null | grep -E "bottleneck|dense|final|loss" | tee gpurun_out/r05_heads.txt
