# two rocprofv3 --pmc passes (kernel-trace only) over a probe script: bash tools/probes/pmc_one.sh tools/probes/stemw_one.py stemw
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/p1 /tmp/p2
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p1 -- python $R/$1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d /tmp/p2 -- python $R/$1 > /dev/null 2>&1
python $R/tools/pmc_kernel.py $(find /tmp/p1 /tmp/p2 -name '*counter_collection.csv') --match $2
k=$(find /tmp/p1 -name '*kernel_trace.csv'); python - <<PY
import csv
for r in csv.DictReader(open("$k")):
    if '$2' in r['Kernel_Name']:
        print('trace dur us', (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
PY
