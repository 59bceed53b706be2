R=$(pwd)
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/mf; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 -M --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
        --kernel-trace --output-format csv -d "$O/pmcM" -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 1 > /dev/null 2> "$O/pmcM.err"
python "$R/tools/mfma_util.py" "$(find "$O/pmcM" -name '*counter_collection.csv')" "$R/gpurun_out/r03_mfma_util.json" > "$O/mfma.log" 2>&1
rm -rf $O/pmcM
cd $R
python tools/kernels_md.py > gpurun_out/KERNELS.md 2> gpurun_out/kernels_md.err
for g in 0 8; do echo "cfg5 group $g"; URSO_WGRAD_GROUP=$g python tools/config_sweep.py 2>/dev/null | grep "cfg5\|cfg4"; done > gpurun_out/cfg45_ab.txt
