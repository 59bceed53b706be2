mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "3x3_64_channel or dense_heads_in_one" 2>&1 | tail -6
for v in 1 0 1 0; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 --opt c3_deep=$v 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3_deep=$v', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})" | tee -a gpurun_out/r05_ab_c3_deep.txt
done
bash tools/probes/prof_stats.sh "c3d_kernel|c3_kernel|dense_multi|finalize|reduce_part" 2>&1 | tail -12
