set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "3x3_128_channel" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "register_filter_3x3_everywhere or cfg2_width" 2>&1 | tail -5
for v in 0 1 0 1; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 --opt c3v=$v 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3v=$v', d['value'], d['ms_per_step'], [(t['kernel'][3:16], t['launches'], round(t['avg_launch_ms']*1e3,1)) for t in d['roofline']['top5']])" | tee -a gpurun_out/r05_ab_c3v.txt
done
cd /tmp && export TMPDIR=/tmp
bash $GRAFT_REPO_ROOT/tools/probes/prof_stats.sh "c3v_kernel|c3w_kernel" 2>&1 | tail -6 | tee -a $GRAFT_REPO_ROOT/gpurun_out/r05_ab_c3v.txt
