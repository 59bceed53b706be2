#!/bin/bash
for i in 1 2; do
  echo "chain"; URSO_WGRAD_STREAM=0 python tools/config_sweep.py cfg4_r101_n24_bf16 cfg2_r50_bf16 2>&1 | grep cfg | cut -c1-60
  for r in 0.4 0.8 1.2 2.0; do echo "room $r"; URSO_FORK_ROOM=$r python tools/config_sweep.py cfg4_r101_n24_bf16 cfg2_r50_bf16 2>&1 | grep cfg | cut -c1-60; done
done
