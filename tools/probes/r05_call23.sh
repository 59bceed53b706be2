mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "stem" 2>&1 | tail -3
for v in "" nocmpx "" nocmpx; do
  URSO_LIB_VARIANT=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant=$v', d['value'], d['ms_per_step'], d['kernels']['conv_wgrad']['ms_per_step'])" | tee -a gpurun_out/r05_ab_stemw_cmpx.txt
done
bash tools/probes/prof_stats.sh "stemw|mold" 2>&1 | tail -4
bash tools/probes/prof_stats.sh "stemw|mold" nocmpx 2>&1 | tail -4
