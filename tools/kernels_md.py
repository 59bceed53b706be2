#!/usr/bin/env python3
"""Generates KERNELS.md: which kernel every layer class of the cfg2 step gets, its tile shape / occupancy, the roof that binds it and the
measured time and fraction of that roof (one eager, single-chain training step under the library's launch profiler; HIP events).
    python tools/kernels_md.py > gpurun_out/KERNELS.md      (on a GPU box; copy to the repo root)"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import make_config, synthetic_batch
from ursonet_amd.engine import Engine

PEAK_TF, PEAK_GBS = 2500.0, 8000.0
# kernel family -> (source file, block tile / waves / LDS / resident blocks per CU, what it is used for)
FAMILIES = [
    ("pwx_kernel", "conv_pwx.hip", "160 px x {256,128} filters, 8 waves (80x64 wave tiles), 3-4 stage LDS-DMA ring 156/148 KiB, 1 block/CU", "reduction-heavy pointwise layers (K >= 512) of stages 4-5"),
    ("pw_kernel", "conv_pw.hip", "128 px x {128,64}, 4 waves, 2 LDS-DMA stages 64/48 KiB, 2-3 blocks/CU", "general 16-bit DMA implicit GEMM: pointwise, strided / sampled layers, 3x3 on small grids"),
    ("hconv2_kernel", "conv_halo2.hip", "128 MI virtual px x 64 NJ filters per tile, (MI, NJ) per layer (cfg2: 384 x 128 in stage 4, 384 x 64 in stage 5: one WHOLE tile per CU), 8 waves (32 MI x 32 NJ, 32x32x16 MFMA), halo double buffer + 2- / 4-slot filter ring 160 KiB, one LDS address register per tap", "3x3 / stride-1 layers with >= 128 channels (stages 4-5)"),
    ("hconv_kernel", "conv_halo.hip", "256 virtual px x 128 filters, 8 waves (64x64, 32x32x16 MFMA), halo tile + 3-slot filter ring 160 KiB, 1 block/CU, (tile, chunk) stream-K", "3x3 / stride-1 layers with >= 128 channels (stages 4-5)"),
    ("c3v_kernel", "conv_c3.hip", "8x16 px x 128 filters, 8 waves x 16 filters over the whole reduction (no channel-half exchange), deferred output tile: one barrier per tile, 1 block/CU", "3x3 layers with 128 channels (stage 3)"),
    ("c3w_kernel", "conv_c3.hip", "8x16 / 4x32 px x 128 filters, wave = 32 filters x one channel half, exchange through LDS", "3x3 layers with 128 channels where the 4x32 geometry fits better (option c3v = 0: everywhere)"),
    ("c3_kernel", "conv_c3.hip", "4x32 px x 64 filters, 4 waves, filter (72 KiB) in registers, 72 KiB LDS, 2 blocks/CU", "3x3 layers with 64 channels (stage 2)"),
    ("c3g_kernel", "conv_c3g.hip", "whole 576 x 64 gradient in registers, one partial per block", "3x3 weight gradient, 64 channels"),
    ("pairs_kernel", "conv_pairs.hip", "32-px halves, 4 waves, W1|Ws + W2 in registers, 80 KiB LDS, 2 blocks/CU", "first stage-2 forward pair with the projection shortcut inside"),
    ("pairw_kernel", "conv_pairw.hip", "64-px tiles, 8 waves, 3 x 48 KiB stages, 1 block/CU", "stage-2 backward pair + weight gradient of the block-closing layer"),
    ("pairx_kernel", "conv_pairx.hip", "all 160 KiB of LDS in two rings", "backward launch behind stage 2's first block (pair + 2 weight gradients + shortcut)"),
    ("pair_kernel", "conv_pair.hip", "64 / 32-px tiles, 4 / 8 waves, both filter matrices in registers, 80-144 KiB LDS", "fused pointwise pairs of stages 2-3; single c -> 4c layers of stages 2-5"),
    ("stemw_kernel", "conv_stemw.hip", "8x32 output px, 4 waves, un-pool in LDS", "7x7 stem weight gradient (+ max-pool backward)"),
    ("stem_pool_kernel", "conv_stem.hip", "17x32 conv px -> 8x15 pooled px x 64 filters per tile (stride 16x30), 4 waves, filters in registers, pooling on integer keys (value over tap priority): row max in registers, column max over DPP wave_shl", "7x7 / stride-2 stem + ReLU + 3x3/s2 max-pool (conv1 output never written)"),
    ("stem_kernel", "conv_stem.hip", "8x32 output px x 64 filters, 4 waves, filters in registers", "7x7 / stride-2 stem"),
    ("wgrad_group_big_kernel", "conv_wgrad.hip", "256 (k) x 256 (n) tile, 8 waves (128 x 64 wave tiles), 4-stage ring of 32-pixel steps 128 KiB, 1 block/CU, over a device table of layers: up to 8 layers per launch, 1/n of the splits each", "weight gradients of consecutive wide layers (>= 256 channels and filters) of a bucket, grouped"),
    ("wgrad_group_kernel", "conv_wgrad.hip", "the 128 x 128 body of wgrad_tr_kernel over a device table of layers: up to 8 layers per launch, 512 blocks shared, 1/n of the splits each", "weight gradients of consecutive 1x1 / strided layers of a bucket, grouped"),
    ("hwgrad2_kernel", "conv_hwgrad.hip", "hwgrad_kernel for two layers, the CUs shared in proportion to their work", "3x3 weight gradients (>= 128 channels), two layers per launch"),
    ("hwgrad_kernel", "conv_hwgrad.hip", "one (64 ch, 64 filters) x 9 taps gradient group per block in registers, 8 waves, 128-virtual-pixel tiles, halo run + offset tables 160 KiB, 1 block/CU", "3x3 weight gradients (>= 128 channels)"),
    ("wgrad_tr64_kernel", "conv_wgrad.hip", "128 (k) x 64 (n) tile, 4 waves, 48 KiB LDS, 3 blocks/CU, split over pixels", "weight gradient, <= 64 filters"),
    ("wgrad_tr_kernel", "conv_wgrad.hip", "128 (k) x 128 (n) tile, 4 waves, 64 KiB LDS, 2 blocks/CU, transposing LDS reads, split over pixels to 512 blocks", "weight gradient (1x1 and 3x3)"),
    ("bneck_fwd_kernel", "conv_bneck.hip", "16 px x all (<= 32) filters per block, 16 waves split K = 9 C, wave-order sum through LDS", "bottleneck_layer forward (3x3 / stride 2), one launch, no split-K workspace"),
    ("bneck_dgrad_kernel", "conv_bneck.hip", "parity class x 256 channels x pixel chunk per block, 4 waves x 64 channels, class taps in registers", "bottleneck_layer data gradient: only the 4 / 2 / 2 / 1 real taps of a pixel's parity class"),
    ("dense_wgrad_multi_kernel", "conv_dense.hip", "64 k x 64 n per block, one 32-deep MFMA step per 16 x 16 outputs, operands gathered transposed from L2", "Dense heads: all four weight gradients in one launch"),
    ("dense_multi_kernel", "conv_dense.hip", "dense_kernel's body, a block belongs to one of <= 4 layers; optional second reduction segment", "Dense heads: the layers of one depth in one launch; the two gradients into the bottleneck features as one layer"),
    ("igemm_kernel", "conv_igemm.hip", "128 x {128,64}, register-staged, split-K + finish", "fp32, Dense heads, bottleneck_layer (small grids)"),
    ("reduce_partials", "conv_wgrad.hip", "batched over a gradient bucket", "fixed-order sum of the weight-gradient partials"),
    ("finalize_", "prep.hip", "batched over a gradient bucket", "dW scale, BN gamma/beta gradients, L2 term"),
    ("weight_prep", "prep.hip", "batched over all layers", "BN fold + layouts + cast, every step"),
    ("stem_pack", "prep.hip", "elementwise", "7x7x3 stem filter -> packed 7x4x8 layout"),
    ("expand2", "pool_loss_optim.hip", "elementwise", "compact -> dense gradient rows (stages 4-5 residual side)"),
    ("subsample2", "pool_loss_optim.hip", "elementwise", "even rows / columns gather"),
    ("softmax_xent", "pool_loss_optim.hip", "one block per row", "soft-label cross-entropy + gradient"),
    ("rel_l2", "pool_loss_optim.hip", "one block", "relative L2 loss + gradient"),
    ("maxpool", "pool_loss_optim.hip", "4x8 output tiles", "3x3/s2 max-pool"),
    ("mold_kernel", "pool_loss_optim.hip", "elementwise", "uint8 -> mean-subtracted packed input"),
    ("sgd_kernel", "pool_loss_optim.hip", "elementwise", "clip + momentum SGD"),
    ("sqnorm", "pool_loss_optim.hip", "two-level reduction", "global gradient norm"),
]


def family(sym):
    for f in FAMILIES:
        if f[0] in sym:
            return f
    return (sym[:24], "", "", "")


def layer_class(label):
    m = re.match(r"(fwd|dgrad|wgrad)(?:_heads)?(?:\+wgrad)?:(.*)", label)
    if not m:
        return label.split(":")[0], label
    kind, rest = m.group(1), m.group(2)
    first = rest.split("+")[0]
    st = re.match(r"res(\d)([a-z])_(branch\w+)", first)
    if st:
        stage, blk, br = st.group(1), st.group(2), st.group(3)
        pos = "entry block" if blk == "a" else "blocks b.."
        fused = (" (+%d fused)" % rest.count("+")) if "+" in rest else ""
        samp = " @sampled" if "@sampled" in rest else ""
        return "%s stage %s %s %s%s%s" % (kind, stage, pos, br.split("@")[0], fused, samp), label
    if re.match(r"(loc|ori|k\d)_(dense_\d|final)", first):
        return "%s Dense heads" % kind, label
    return "%s %s" % (kind, first), label


cfg = make_config(backbone="resnet50", h=512, w=640, batch=32, regress_ori=False, ori_bins=16, dtype="bfloat16")
eng = Engine(cfg, "training", seed=1234, randomize_bn=True)
img, loc, ori, _ = synthetic_batch(cfg, 32, seed=1)
eng.load_batch(img, loc, ori)
eng.step_eager(); eng.step_eager()
recs = eng.profile_step()
tot = sum(r[2] for r in recs)
groups = {}
for label, kid, ms, fl, by, nl, sym in recs:
    cls, _ = layer_class(label)
    fam = family(sym)
    g = groups.setdefault((cls, fam[0]), dict(n=0, ms=0.0, fl=0.0, by=0.0, fam=fam, first=len(groups)))
    g["n"] += 1; g["ms"] += ms; g["fl"] += fl; g["by"] += by
print("# KERNELS -- which kernel every layer class of the benchmarked step gets, and how close it runs to the roof that binds it\n")
print("Generated by `tools/kernels_md.py` on an MI355X (cfg2: ResNet-50, bottleneck 32, ori_resolution 16, batch 32 x 512 x 640, bf16; one eager")
print("single-chain step under the library's launch profiler, %.2f ms summed over %d launches; the hipGraph replay of the same step is what" % (tot, len(recs)))
print("`bench.py` times).  Roof of a launch = max(algorithmic FLOPs / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s) (MI355X_MICROARCH.md); `frac` =")
print("that time / measured time.  The selection policy lives in `urso_conv_igemm_ex` (conv_igemm.hip: register-filter pair kernel -> c3 -> halo ->")
print("stem -> pwx -> pw -> general igemm, each with a `*_fits` predicate) and in the engine's plan rewrites (fused pairs, sampled block outputs).\n")
print("| layer class | launches | kernel | avg us | TFLOP/s | GB/s (alg.) | bound | frac of roof |")
print("|---|---|---|---|---|---|---|---|")
for (cls, famname), g in sorted(groups.items(), key=lambda kv: kv[1]["first"]):
    ms = g["ms"]
    t_m, t_h = g["fl"] / (PEAK_TF * 1e12), g["by"] / (PEAK_GBS * 1e9)
    bound = "mfma" if t_m >= t_h else "hbm"
    frac = max(t_m, t_h) / (ms * 1e-3) if ms > 0 else 0.0
    print("| %s | %d | `%s` | %.1f | %s | %s | %s | %.2f |" % (cls, g["n"], famname, ms / g["n"] * 1e3, ("%.0f" % (g["fl"] / (ms * 1e9))) if g["fl"] else "-",
                                                     ("%.0f" % (g["by"] / (ms * 1e6))) if g["by"] else "-", bound if (g["fl"] or g["by"]) else "-", frac))
print("\n## Kernel families\n")
print("| kernel | file | tile / waves / LDS / residency | used for | launches | ms per step | share |")
print("|---|---|---|---|---|---|---|")
byfam = {}
for (cls, famname), g in groups.items():
    b = byfam.setdefault(famname, dict(n=0, ms=0.0, fam=g["fam"]))
    b["n"] += g["n"]; b["ms"] += g["ms"]
for famname, b in sorted(byfam.items(), key=lambda kv: -kv[1]["ms"]):
    print("| `%s` | %s | %s | %s | %d | %.3f | %.1f %% |" % (famname, b["fam"][1], b["fam"][2], b["fam"][3], b["n"], b["ms"], 100 * b["ms"] / tot))
