set -x
mkdir -p gpurun_out
export URSO_PARITY_LOG=$PWD/gpurun_out/r05_parity.txt
rm -f $URSO_PARITY_LOG
timeout 1500 python -m pytest tests/test_layerwise_gpu.py -q -s 2>&1 | tail -40 | tee gpurun_out/r05_call3_layerwise.txt
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -x -q -k "five_seeds or full_benchmark_batch or bucket_round" 2>&1 | tail -15 | tee gpurun_out/r05_call3_model.txt
cat $URSO_PARITY_LOG
unset URSO_PARITY_LOG
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --pcie-steps 0 --profile-steps 0 > /tmp/kt_bench.txt 2>&1
tail -1 /tmp/kt_bench.txt | cut -c1-200
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/probes/kernel_gaps.py $f --steps 10 | tee $GRAFT_REPO_ROOT/gpurun_out/r05_kernel_gaps.txt
