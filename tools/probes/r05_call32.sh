#!/bin/bash
# complementary fork with one edge per fork: bits at two sizes (three steps), then the step, alternating
export URSO_OPT_HCONV_STREAMK=0
CHK_MODES=0,1,2 CHK_STEPS=3 python tools/probes/fork2_check.py 2>&1 | grep "mode"
CHK_MODES=0,2 CHK_H=512 CHK_W=640 CHK_B=32 python tools/probes/fork2_check.py 2>&1 | grep "mode"
unset URSO_OPT_HCONV_STREAMK
for i in 1 2 3; do for v in 0 2; do
  URSO_WGRAD_STREAM=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('URSO_WGRAD_STREAM=$v  %.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))"
done; done
