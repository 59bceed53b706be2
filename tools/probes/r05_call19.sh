mkdir -p gpurun_out
for v in 1 0 1 0 1 0; do
  URSO_SIDE_STREAM=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side_stream=$v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05_ab_side_stream.txt
done
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_layerwise_gpu.py -x -q 2>&1 | tail -6
