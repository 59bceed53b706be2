set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "3x3_128_channel" 2>&1 | tail -3
for v in "" seq nosb ""; do URSO_LIB_VARIANT=$v timeout 300 python tools/probes/c3v_probe.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05_c3v_probe2.txt; done
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "register_filter_3x3_everywhere or cfg2_width" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
bash $GRAFT_REPO_ROOT/tools/probes/prof_stats.sh "c3v_kernel|c3w_kernel" 2>&1 | tail -4 | tee -a $GRAFT_REPO_ROOT/gpurun_out/r05_c3v_probe2.txt
