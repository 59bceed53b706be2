#!/usr/bin/env python3
"""MFMA-pipe utilisation and instruction mix per kernel family from a rocprofv3 --pmc pass over bench.py (CSV output):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
              SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -- python bench.py ...
    python tools/mfma_util.py <counter_collection.csv> <out.json>
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): the share of CU-busy SIMD cycles in which the matrix pipe
was executing (SQ_VALU_MFMA_BUSY_CYCLES counts pipe cycles, 32 per v_mfma_f32_32x32x16 and 16 per 16x16x32 -- MI355X_MICROARCH.md);
gpu_frac uses GRBM_GUI_ACTIVE x 256 CUs x 4 instead (includes idle CUs at the tail of a launch)."""
import collections, csv, json, sys

FAMILIES = [("stemw_kernel", "conv_stemw (stem weight gradient, un-pool in LDS)"), ("stem_pool_kernel", "conv_stem (7x7 stem + ReLU + 3x3/s2 max-pool in one kernel)"), ("stem_kernel", "conv_stem (7x7 stem, im2col on the LDS read side)"), ("c3g_kernel", "conv_c3g (3x3 weight gradient, 64 channels, gradient in registers)"), ("c3_kernel", "conv_c3 (3x3, 64 channels, filter in registers)"), ("c3v_kernel", "conv_c3 (3x3, 128 channels, 16 filters per wave)"), ("c3w_kernel", "conv_c3 (3x3, 128 channels, filter in registers)"), ("pairw_kernel", "conv_pairw (backward pair + weight gradient)"), ("pairx_kernel", "conv_pairx (backward launch behind a stage's first block)"), ("pairs_kernel", "conv_pairs (forward pair + projection shortcut)"), ("pair_kernel", "conv_pair (fused pointwise pairs)"), ("hconv2_kernel", "conv_halo2 (3x3 halo tile, whole tiles of a per-layer shape)"), ("hconv_kernel", "conv_halo (3x3 halo tile)"), ("pwx_kernel", "conv_pwx (8-wave big-tile pointwise GEMM)"), ("pw_kernel", "conv_pw (DMA implicit GEMM)"), ("igemm_kernel", "conv_igemm (general)"),
            ("wgrad_group_big_kernel", "wgrad 256x256, several layers per launch"), ("wgrad_group_kernel", "wgrad 128x128, several layers per launch"), ("hwgrad2_kernel", "conv_hwgrad (3x3 weight gradient, two layers per launch)"), ("hwgrad_kernel", "conv_hwgrad (3x3 weight gradient, gradient in registers)"), ("wgrad_tr64_kernel", "wgrad 128x64"), ("wgrad_tr_kernel", "wgrad 128x128"), ("wgrad_kernel", "wgrad fp32")]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    fam = next((f for f, _ in FAMILIES if f in name), None)
    if fam is None:
        continue
    acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[fam].add(r.get("Dispatch_Id", r.get("Correlation_Id", len(launches[fam]))))
out = {"counters_are": "sums over all profiled launches of the kernel family", "kernels": {}}
for fam, label in FAMILIES:
    if fam not in acc:
        continue
    c = acc[fam]
    d = {"label": label, "launches": len(launches[fam])}
    d.update({k: v for k, v in sorted(c.items())})
    if c.get("SQ_BUSY_CU_CYCLES"):
        d["mfma_busy_frac"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * c["SQ_BUSY_CU_CYCLES"])
    if c.get("GRBM_GUI_ACTIVE"):
        d["mfma_busy_frac_of_gpu_time"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * 256 * c["GRBM_GUI_ACTIVE"])
    if c.get("SQ_INSTS_MFMA"):
        d["valu_per_mfma"] = (c.get("SQ_INSTS_VALU", 0.0) - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"]
        d["lds_insts_per_mfma"] = c.get("SQ_INSTS_LDS", 0.0) / c["SQ_INSTS_MFMA"]
    if c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    out["kernels"][fam] = d
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, d in out["kernels"].items():
    print(k, {x: (round(v, 4) if isinstance(v, float) and v < 100 else v) for x, v in d.items() if x in
              ("launches", "mfma_busy_frac", "mfma_busy_frac_of_gpu_time", "valu_per_mfma", "lds_insts_per_mfma", "lds_bank_conflict_frac")})
