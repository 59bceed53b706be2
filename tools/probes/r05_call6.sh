set -x
mkdir -p gpurun_out
for v in "" pf5 pf8; do URSO_LIB_VARIANT=$v timeout 300 python tools/probes/c3v_probe.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05_c3v_probe.txt; done
for v in 3 1 2 0 3 1 2 0; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 --opt pair_single=$v 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair_single=$v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05_ab_pair_single.txt
done
