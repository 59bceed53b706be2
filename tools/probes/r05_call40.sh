#!/bin/bash
URSO_FORK_SIDE_CUS=128 URSO_FORK_MAIN_CUS=128 python tools/probes/fork2_check.py 2>&1 | grep -v amdgpu
run() { URSO_FORK_SIDE_CUS=$1 URSO_FORK_MAIN_CUS=$2 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fork-check --pcie-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('side cus $1  chain cus $2  %.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))"; }
for i in 1 2; do
  run 0 0; run 160 0; run 128 0; run 96 0; run 64 0; run 128 128; run 160 96; run 96 160; run 128 192; run 64 192
done
