#!/bin/bash
run() { URSO_FORK_EARLY=$1 URSO_FORK_LATE=$2 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('early >= $1  late >= $2  %.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))"; }
for i in 1 2; do
  run 150 400; run 150 1e9; run 150 250; run 300 400; run 300 1e9; run 100 400; run 50 400
done
