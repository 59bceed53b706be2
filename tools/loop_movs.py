#!/usr/bin/env python3
"""Register copies inside loops, per kernel: hipcc can give the arms of a branch inside a loop different registers for loop-carried MFMA
accumulators and copy them back every iteration (conv_hwgrad.hip, round 4).  python tools/loop_movs.py [file.hip ...]"""
import collections, glob, os, re, subprocess, sys
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ursonet_amd", "csrc")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(root, "*.hip")))
for f in files:
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only", f, "-o", "-"],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    kern, inloop, cnt = None, False, None
    depth, bydepth = 0, None
    out = []
    alines = asm.split("\n")
    for li, l in enumerate(alines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern, cnt, inloop = m.group(1), collections.Counter(), False
            depth, bydepth = 0, collections.defaultdict(collections.Counter)
            continue
        if kern is None:
            continue
        if l.startswith(".LBB") or l.startswith("; %bb"):
            head = " ".join(alines[li:li + 4]) if "Parent Loop" in l else l       # a nested header's comment runs over several lines
            inloop = "in Loop" in head or "Loop Header" in head
            md = re.findall(r"Depth=(\d+)", head)
            depth = max(int(x) for x in md) if md else 0
        t = l.strip().split()
        if inloop and t and t[0][0] in "vsdb":
            cnt[t[0]] += 1
            bydepth[depth][t[0]] += 1
        if "s_endpgm" in l:
            mf = sum(v for k, v in cnt.items() if k.startswith("v_mfma"))
            mv = cnt["v_mov_b32_e32"] + 2 * cnt["v_mov_b64_e32"] + cnt["v_accvgpr_read_b32"] + cnt["v_accvgpr_write_b32"]
            valu = sum(v for k, v in cnt.items() if k.startswith("v_") and not k.startswith("v_mfma"))
            # the deepest loop level that holds MFMAs: copies THERE are paid per MFMA step
            deep = max([d for d, c in bydepth.items() if any(k.startswith("v_mfma") for k in c)] or [0])
            c = bydepth[deep]
            dmf = sum(v for k, v in c.items() if k.startswith("v_mfma"))
            dmv = c["v_mov_b32_e32"] + 2 * c["v_mov_b64_e32"] + c["v_accvgpr_read_b32"] + c["v_accvgpr_write_b32"]
            if mf:
                out.append((kern, mf, mv, valu, deep, dmf, dmv))
            kern = None
    for kern, mf, mv, valu, deep, dmf, dmv in out:
        flag = "  <-- copies in the MFMA loop" if dmv > 0.25 * dmf else ""
        print("%-18s %-60s all loops: mfma %4d moves %4d VALU %5d | depth %d: mfma %4d moves %4d%s" % (os.path.basename(f), kern[:60], mf, mv, valu, deep, dmf, dmv, flag))
