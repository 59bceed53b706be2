#!/bin/bash
# complementary fork (URSO_WGRAD_STREAM=2) against the single chain: bits, then the step, alternating in one call
python tools/probes/fork2_check.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do for v in 0 2 1; do
  URSO_WGRAD_STREAM=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('URSO_WGRAD_STREAM=$v  %.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))"
done; done
