for v in "" ns2 "" ns2; do
  URSO_LIB_VARIANT=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant=$v', d['value'], d['ms_per_step'], [(t['kernel'][3:22], t['launches'], round(t['avg_launch_ms']*1e3,1), t['frac']) for t in d['roofline']['top5']])"
done
