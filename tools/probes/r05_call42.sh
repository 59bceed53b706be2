#!/bin/bash
for i in 1 2 3 4 5; do for g in 8 16; do
  URSO_WGRAD_GROUP=$g timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-fork-check --pcie-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('URSO_WGRAD_GROUP=$g  %.3f ms' % d['ms_per_step'])"
done; done
