#!/bin/bash
for i in 1 2 3; do for v in 0 1; do
  URSO_CAPTURE_PRIO=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('URSO_CAPTURE_PRIO=$v  %.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))"
done; done
