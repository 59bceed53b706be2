mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee gpurun_out/r05_pytest_b.txt
for v in 1 0 1 0; do
  URSO_DENSE_MULTI=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dense_multi=$v', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})" | tee -a gpurun_out/r05_ab_dense_multi2.txt
done
timeout 300 python tools/layer_profile.py 2>/dev/null | grep -E "bottleneck|dense|final|loss|reduce|finalize" | tee gpurun_out/r05_heads2.txt
