#!/bin/bash
run() { URSO_WGRAD_GROUP=$1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('URSO_WGRAD_GROUP=$1  %.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))"; }
for i in 1 2 3; do run 8; run 16; run 24; run 32; done
