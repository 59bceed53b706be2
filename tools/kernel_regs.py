#!/usr/bin/env python3
"""Register / LDS budget of every kernel of a source file (hipcc -Rpass-analysis=kernel-resource-usage).  python tools/kernel_regs.py conv_c3 [filter]"""
import subprocess, sys, re, os
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ursonet_amd", "csrc", sys.argv[1] + ".hip")
flt = sys.argv[2] if len(sys.argv) > 2 else "DF16b"
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage",
                      "-c", src, "-o", "/dev/null"] + os.environ.get("URSO_VARIANT_FLAGS", "").split(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
cur = None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        if "error" in line: print(line)
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); vals = {}
    elif ":" in t and cur:
        k, v = t.split(":", 1); vals[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size") and flt in cur:
            print("%-70s VGPR %3s AGPR %3s spillV %3s scratch %4s occ %s LDS %s" % (cur[:70], vals.get("VGPRs"), vals.get("AGPRs"), vals.get("VGPRs Spill"), vals.get("ScratchSize [bytes/lane]"), vals.get("Occupancy [waves/SIMD]"), v.strip()))
