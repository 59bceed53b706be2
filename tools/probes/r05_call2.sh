# round 5, call 2: access-pattern probe, the new parity tests (numbers into the log), the two fixed tests, pwx time breakdown
set -x
mkdir -p gpurun_out
timeout 300 tools/probes/bin/l2_bw_probe rowseg > gpurun_out/r05_rowseg_probe.txt 2>&1
cat gpurun_out/r05_rowseg_probe.txt
export URSO_PARITY_LOG=$PWD/gpurun_out/r05_parity.txt
rm -f $URSO_PARITY_LOG
timeout 1500 python -m pytest tests/test_layerwise_gpu.py -x -q -s 2>&1 | tail -40 | tee gpurun_out/r05_call2_layerwise.txt
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -k "five_seeds or full_benchmark_batch or fused_pointwise_pairs_change or weight_gradient_folded or refuses or same_rounding_points or cfg2_width" 2>&1 | tail -15 | tee gpurun_out/r05_call2_model.txt
cat $URSO_PARITY_LOG
timeout 600 python tools/pwx_probe.py 2>&1 | tee gpurun_out/r05_pwx_probe.txt | tail -20
