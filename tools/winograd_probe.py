#!/usr/bin/env python3
"""Winograd F(2x2,3x3) for the 3x3 / stride-1 layers, decided by measurement.
A non-fused evaluation (input transform -> 16 frequency GEMMs -> output transform) is timed piece by piece on the cfg2 shapes
against the direct kernels of the library.  The 16 GEMMs [T x C] . [C x N] are timed as ONE pointwise conv of 16 T pixels through the
library's own MFMA kernel (same FLOPs and activation bytes, one shared filter: a LOWER bound of a real batched launch); the transforms
are the HBM-bound kernels of tools/probes/winograd_probe.hip.  Numerical error of a real Winograd evaluation (16 launches, bf16 and
fp32 frequency-domain products) is measured against a CPU fp32 conv on a small shape.  Output: a table + JSON (profiles/r02_winograd.json).
"""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from ursonet_amd import hip

src = os.path.join(ROOT, "tools", "probes", "winograd_probe.hip")
so = os.path.join(ROOT, "tools", "probes", "winograd_probe.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", src, "-o", so])
lib = C.CDLL(so)
dt, tdt = hip.BF16, torch.bfloat16
st = lambda: torch.cuda.current_stream().cuda_stream
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
Gm = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)


def timeit(fn, iters=20, rounds=3):
    best = 1e9
    for _ in range(rounds):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def layer(B, H, W, Cc, N):
    T = B * ((H + 1) // 2) * ((W + 1) // 2)
    x = torch.randn(B, H, W, Cc, device="cuda").to(tdt)
    wf = (torch.randn(N, 3, 3, Cc, device="cuda") / (9 * Cc) ** 0.5).to(tdt)
    bias = torch.randn(N, device="cuda")
    y = torch.empty(B, H, W, N, device="cuda", dtype=tdt)
    V = torch.empty(16, T, Cc, device="cuda", dtype=tdt)
    Mb = torch.empty(16, T, N, device="cuda", dtype=tdt)
    Mf = torch.empty(16, T, N, device="cuda", dtype=torch.float32)
    u = (torch.randn(N, Cc, device="cuda") / Cc ** 0.5).to(tdt)
    g3 = hip.geom(B, H, W, Cc, H, W, N, 3, 3, 1, 1, 1, 1)
    gg = hip.geom(1, 1, 16 * T, Cc, 1, 16 * T, N, 1, 1)
    r = {"shape": [B, H, W, Cc, N], "tiles": T}
    with hip.options(hconv=0):
        r["direct_dma_us"] = timeit(lambda: hip.conv_igemm(g3, dt, hip.EPI_RELU, x, wf, bias, None, None, y))
    with hip.options(hconv=2):
        r["direct_halo_us"] = timeit(lambda: hip.conv_igemm(g3, dt, hip.EPI_RELU, x, wf, bias, None, None, y))
    r["in_transform_us"] = timeit(lambda: lib.wino_in_launch(C.c_void_p(x.data_ptr()), C.c_void_p(V.data_ptr()), B, H, W, Cc, C.c_void_p(st())))
    r["gemm16_bf16_us"] = timeit(lambda: hip.conv_igemm(gg, dt, 0, V, u, None, None, None, Mb))
    r["gemm16_f32out_us"] = timeit(lambda: hip.conv_igemm(gg, dt, hip.EPI_OUT_F32, V, u, None, None, None, Mf))
    r["out_transform_bf16_us"] = timeit(lambda: lib.wino_out_launch(C.c_void_p(Mb.data_ptr()), 0, C.c_void_p(bias.data_ptr()), C.c_void_p(y.data_ptr()), B, H, W, N, C.c_void_p(st())))
    r["out_transform_f32_us"] = timeit(lambda: lib.wino_out_launch(C.c_void_p(Mf.data_ptr()), 1, C.c_void_p(bias.data_ptr()), C.c_void_p(y.data_ptr()), B, H, W, N, C.c_void_p(st())))
    r["winograd_bf16_total_us"] = r["in_transform_us"] + r["gemm16_bf16_us"] + r["out_transform_bf16_us"]
    r["winograd_f32_total_us"] = r["in_transform_us"] + r["gemm16_f32out_us"] + r["out_transform_f32_us"]
    r["direct_tflops"] = 2.0 * B * H * W * N * 9 * Cc / min(r["direct_dma_us"], r["direct_halo_us"]) / 1e6
    return r


def accuracy(B=2, H=16, W=20, Cc=128, N=128):
    """A REAL Winograd evaluation (16 frequency GEMM launches with the transformed filters) vs CPU fp32."""
    torch.manual_seed(0)
    T = B * (H // 2) * (W // 2)
    x = torch.randn(B, H, W, Cc, device="cuda").to(tdt)
    w = torch.randn(N, 3, 3, Cc, device="cuda") / (9 * Cc) ** 0.5
    wf = w.to(tdt)
    bias = torch.randn(N, device="cuda")
    U = torch.einsum("ia,nabc,jb->ijnc", Gm.cuda(), wf.float(), Gm.cuda()).reshape(16, N, Cc).to(tdt).contiguous()      # G g G^T per (n, c)
    V = torch.empty(16, T, Cc, device="cuda", dtype=tdt)
    lib.wino_in_launch(C.c_void_p(x.data_ptr()), C.c_void_p(V.data_ptr()), B, H, W, Cc, C.c_void_p(st()))
    ref = F.relu(F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wf.float().cpu().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1) + bias.cpu())
    out = {}
    gk = hip.geom(1, 1, T, Cc, 1, T, N, 1, 1)
    for name, f32 in (("bf16_products", 0), ("fp32_products", 1)):
        M = torch.empty(16, T, N, device="cuda", dtype=torch.float32 if f32 else tdt)
        for k in range(16):
            hip.conv_igemm(gk, dt, hip.EPI_OUT_F32 if f32 else 0, V[k], U[k], None, None, None, M[k])
        y = torch.empty(B, H, W, N, device="cuda", dtype=tdt)
        lib.wino_out_launch(C.c_void_p(M.data_ptr()), f32, C.c_void_p(bias.data_ptr()), C.c_void_p(y.data_ptr()), B, H, W, N, C.c_void_p(st()))
        torch.cuda.synchronize()
        out[name] = float((y.float().cpu() - ref).abs().max() / ref.abs().max())
    yd = torch.empty(B, H, W, N, device="cuda", dtype=tdt)
    hip.conv_igemm(hip.geom(B, H, W, Cc, H, W, N, 3, 3, 1, 1, 1, 1), dt, hip.EPI_RELU, x, wf, bias, None, None, yd)
    torch.cuda.synchronize()
    out["direct"] = float((yd.float().cpu() - ref).abs().max() / ref.abs().max())
    return out


res = {"layers": [layer(32, 64, 80, 128, 128), layer(32, 32, 40, 256, 256), layer(32, 16, 20, 512, 512)], "max_rel_err_vs_cpu_fp32": accuracy()}
for r in res["layers"]:
    print("B%d %dx%d C%d N%d: direct %.1f (dma) / %.1f (halo) us | winograd bf16: in %.1f + gemm>= %.1f + out %.1f = %.1f us | fp32 products: %.1f us"
          % (*r["shape"], r["direct_dma_us"], r["direct_halo_us"], r["in_transform_us"], r["gemm16_bf16_us"], r["out_transform_bf16_us"],
             r["winograd_bf16_total_us"], r["winograd_f32_total_us"]))
print("max rel err vs CPU fp32:", res["max_rel_err_vs_cpu_fp32"])
print(json.dumps(res))
