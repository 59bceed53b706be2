#!/bin/bash
# Regenerates the artefacts kept under profiles/ on a GPU box (run through gpurun from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/collect_profiles.sh r01'
# rocprofv3 --pmc runs are separate passes (no trace domains besides --kernel-trace), bounded by `timeout`.
set -u
TAG=${1:-r06}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/$TAG; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 -M --kernel-trace --stats -d "$O/stats" -- python "$R/bench.py" --steps 8 --warmup 2 --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/stats.err"
DB=$(find "$O/stats" -name "*.db" | head -1)
python "$R/tools/rocprof_summary.py" "$DB" "$O/${TAG}_bench" > "$O/sum.log" 2>&1
# the same command with every launch on ONE chain (URSO_WGRAD_STREAM=0): the forked graph's branches overlap, so its per-kernel durations are not
# per-kernel costs and sum to more than the step; this CSV is the one whose AverageNs agrees with bench.py's roofline.avg_launch_ms
URSO_WGRAD_STREAM=0 timeout 400 rocprofv3 -M --kernel-trace --stats -d "$O/stats1" -- python "$R/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --no-fork-check --pcie-steps 0 > "$O/bench_under_rocprof_single_chain.json" 2> "$O/stats1.err"
DB1=$(find "$O/stats1" -name "*.db" | head -1)
python "$R/tools/rocprof_summary.py" "$DB1" "$O/${TAG}_bench_single_chain" >> "$O/sum.log" 2>&1
mv "$O/${TAG}_bench_single_chain_kernel_stats.csv" "$O/${TAG}_bench_kernel_stats_single_chain.csv"
timeout 300 rocprofv3 -M --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmcF" -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 1 > /dev/null 2>&1
timeout 300 rocprofv3 -M --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmcW" -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 1 > /dev/null 2>&1
python "$R/tools/pmc_traffic.py" "$(find "$O/pmcF" -name '*counter_collection.csv')" "$(find "$O/pmcW" -name '*counter_collection.csv')" \
       "$O/${TAG}_pmc_traffic.json" resnet50 32 512 640 bfloat16 > "$O/pmc.log" 2>&1
cp "$O/${TAG}_pmc_traffic.json" "$R/profiles/${TAG}_pmc_traffic.json"      # bench.py reads roofline.traffic from here
# MFMA-pipe utilisation / instruction mix / LDS conflicts per kernel family (8 SQ slots + GRBM: one pass)
timeout 300 rocprofv3 -M --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
        --kernel-trace --output-format csv -d "$O/pmcM" -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 1 > /dev/null 2> "$O/pmcM.err"
python "$R/tools/mfma_util.py" "$(find "$O/pmcM" -name '*counter_collection.csv')" "$O/${TAG}_mfma_util.json" > "$O/mfma.log" 2>&1
cd "$R"
timeout 300 python tools/layer_profile.py > "$O/${TAG}_layer_profile.txt" 2>&1
timeout 900 python tools/config_sweep.py 2>&1 | grep -v amdgpu.ids > "$O/${TAG}_config_sweep.txt"
timeout 600 python bench.py > "$O/${TAG}_bench.json" 2> "$O/bench.err"
timeout 600 python tools/dp_dryrun.py 2>/dev/null | grep "^{" | tail -1 > "$O/${TAG}_dp_dryrun.json"
timeout 300 python tools/kernels_md.py > "$O/KERNELS.md" 2> "$O/kernels_md.err"      # -> KERNELS.md at the repo root
rm -rf "$O/stats" "$O/stats1" "$O/pmcF" "$O/pmcW" "$O/pmcM"
cut -c1-300 "$O/${TAG}_bench.json"
echo "copy $O/${TAG}_* into profiles/ and commit"
