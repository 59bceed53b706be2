#!/bin/bash
for i in 1 2; do for v in "0 0" "2 0" "2 224" "2 192" "2 160" "2 128"; do
  set -- $v
  URSO_WGRAD_STREAM=$1 URSO_FORK_MAIN_CUS=$2 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('URSO_WGRAD_STREAM=$1 main cus $2  %.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))"
done; done
