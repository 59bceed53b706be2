mkdir -p gpurun_out
for v in "" pd1 pd2 pd4 pd8 pd6 pd15 ""; do
  URSO_LIB_VARIANT=$v timeout 200 python tools/probes/pair_probe.py 2>/dev/null | tee -a gpurun_out/r05_pair_probe.txt
done
