for o in "hconv2_shape=0" "hconv2_shape=31" "hconv2_shape=32" "hconv2_shape=22" "hconv2_shape=21" "hconv2_shape=42" "hconv2_shape=0" "hconv2_shape=31"; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 --opt $o 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-18s' % '$o', d['value'], d['ms_per_step'], d['kernels']['conv_igemm']['ms_per_step'], [(t['kernel'][3:20], t['launches'], round(t['avg_launch_ms']*1e3,1)) for t in d['roofline']['top5'][:3]])" | tee -a gpurun_out/r05_hconv2_shapes.txt
done
