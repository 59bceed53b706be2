#!/bin/bash
run() { URSO_WGRAD_STREAM=$1 URSO_FORK_AT="$2" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pcie-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('URSO_WGRAD_STREAM=$1 at $2  %.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))"; }
for i in 1 2; do
  run 0 "^dgrad:res3"; run 2 "^dgrad:res3"; run 2 "^dgrad:res4c"; run 2 "^dgrad:res4a"; run 2 "^dgrad:res3c"; run 2 "^dgrad:res3a"; run 2 "^dgrad:res2"; run 2 "^dgrad:res2b"
done
