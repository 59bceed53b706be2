timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "bottleneck or split_k or tfsame" 2>&1 | tail -4
for v in 3 1 3 1; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 --opt bneck=$v 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bneck=$v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05_ab_bneck_fwd.txt
done
timeout 300 python tools/layer_profile.py 2>/dev/null | grep -E "bottleneck" 
timeout 300 python tools/layer_profile.py --opt bneck=1 2>/dev/null | grep -E "bottleneck" 
