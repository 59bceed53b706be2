#!/bin/bash
URSO_FORK_LATE_AT=1 python tools/probes/fork2_check.py 2>&1 | grep -v amdgpu | tail -2
for i in 1 2 3; do for v in 0 1; do
  URSO_FORK_LATE_AT=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fork-check --pcie-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('late deferred $v  %.3f ms' % d['ms_per_step'])"
done; done
