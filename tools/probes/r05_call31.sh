#!/bin/bash
export URSO_OPT_HCONV_STREAMK=0
echo "--- all deferred"; python tools/probes/fork2_check.py 2>&1 | grep "mode"
for re_ in "wgrad_heads" "bottleneck" "res5" "res4.*branch2b|branch2b.*res4" "res4\w_branch2[ac]" "res4a_branch1|res5a_branch1"; do
  echo "--- only $re_"; URSO_WGRAD_DEFER_RE="$re_" python tools/probes/fork2_check.py 2>&1 | grep "mode"
done
