mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "pair or wide_pointwise or sampled or shortcut" 2>&1 | tail -8
for v in "" oldepi "" oldepi; do
  URSO_LIB_VARIANT=$v timeout 200 python tools/probes/pair_probe.py 2>/dev/null | tee -a gpurun_out/r05_pair_probe.txt
done
for v in "" oldepi "" oldepi; do
  URSO_LIB_VARIANT=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant=$v', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})" | tee -a gpurun_out/r05_ab_pair_epi.txt
done
