set -x
mkdir -p gpurun_out
for v in 1 0 1 0; do
  URSO_ZERO_EVERY_STEP=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('zero_every_step=$v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05_ab_zero_once.txt
done
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee gpurun_out/r05_call8_pytest.txt
