# round 5, first GPU call: bare L2 probe, A/B of the forked weight-gradient stream, then the GPU test suite
set -x
mkdir -p gpurun_out
timeout 300 tools/probes/bin/l2_bw_probe > gpurun_out/r05_l2_probe.txt 2>&1
tail -5 gpurun_out/r05_l2_probe.txt
for v in 0 1 0 1; do
  URSO_WGRAD_STREAM=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad_stream=$v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05_ab_wgrad_stream.txt
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r05_call1_pytest.txt
