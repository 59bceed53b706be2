"""GPU parity tests of every HIP kernel, called THROUGH THE C ABI (ursonet_amd.hip ctypes
bindings) and checked against the CPU oracle (oracle/graph_ref.py, oracle/pose_math.py; torch
autograd on CPU for gradients).  Tolerances: fp32 kernels 2e-5 of max|ref| (exact-fp32 MFMA,
different summation order); bf16 kernels 1.5e-2 of max|ref| with inputs pre-rounded to bf16 so
that only accumulation order and the output rounding differ."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {0: 2e-5, 1: 1.5e-2, 2: 2e-3}


def _hip():
    import ursonet_amd.hip as hip
    return hip


def rnd(t, dt):
    hip = _hip()
    return t.to(hip.TORCH_DT[dt]).to(torch.float32)


def dev(t, dt=None):
    hip = _hip()
    t = t.contiguous()
    if dt is not None:
        t = t.to(hip.TORCH_DT[dt])
    return t.cuda()


def relerr(got, ref):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all(), "non-finite values in kernel output"
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


def prep_weights(w_hwio, dt, bias=None, bn=None, npad=None, want_wd=True):
    """Runs urso_conv_weight_prep; returns (wf, wd, biasf, scale) device tensors."""
    hip = _hip()
    KH, KW, Ci, N = w_hwio.shape
    npad = npad or N
    tdt = hip.TORCH_DT[dt]
    wf = torch.empty(npad * KH * KW * Ci, dtype=tdt, device="cuda")
    wd = torch.empty(Ci * KH * KW * npad, dtype=tdt, device="cuda") if want_wd else None
    biasf = torch.empty(npad, dtype=torch.float32, device="cuda")
    scale = torch.empty(npad, dtype=torch.float32, device="cuda")
    g = [dev(t) if t is not None else None for t in (bn or (None, None, None, None))]
    hip.conv_weight_prep(KH, KW, Ci, N, npad, dt, dev(w_hwio), dev(bias) if bias is not None else None,
                         g[0], g[1], g[2], g[3], 1e-3, wf, wd, biasf, scale)
    return wf, wd, biasf, scale


CONV_CASES = [
    # B, H,  W,  C,  N, k, s, (pt, pl), name
    (2, 16, 20, 64, 64, 1, 1, (0, 0), "1x1"),
    (2, 16, 20, 64, 256, 1, 1, (0, 0), "1x1_wideN"),
    (2, 16, 24, 128, 64, 1, 2, (0, 0), "1x1_s2"),
    (2, 12, 20, 64, 64, 3, 1, (1, 1), "3x3_same"),
    (3, 9, 11, 32, 160, 3, 1, (1, 1), "3x3_ragged"),
    (2, 16, 20, 64, 32, 3, 2, (0, 0), "3x3_s2_tfsame"),      # bottleneck_layer: pad (0,1)
    (2, 16, 16, 64, 128, 3, 2, (1, 1), "3x3_s2_pad1"),       # shallow trunk conv1
    (4, 1, 1, 256, 1024, 1, 1, (0, 0), "dense"),
    (2, 8, 8, 8, 24, 3, 1, (1, 1), "tinyC"),
]


def _out_hw(H, W, k, s, pad, case):
    if case in ("3x3_s2_tfsame",):
        return -(-H // s), -(-W // s)
    return (H + 2 * pad[0] - k) // s + 1, (W + 2 * pad[1] - k) // s + 1


def _ref_conv(x_nhwc, w_hwio, s, pad, OH, OW):
    """fp32 CPU reference with explicit zero padding (bottom/right as needed to reach OH x OW)."""
    k = w_hwio.shape[0]
    H, W = x_nhwc.shape[1:3]
    pb = max((OH - 1) * s + k - H - pad[0], 0)
    pr = max((OW - 1) * s + k - W - pad[1], 0)
    x = F.pad(x_nhwc.permute(0, 3, 1, 2), (pad[1], pr, pad[0], pb))
    return F.conv2d(x, w_hwio.permute(3, 2, 0, 1), stride=s).permute(0, 2, 3, 1)


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[-1] for c in CONV_CASES])
def test_conv_forward_and_gradients(case, dt):
    hip = _hip()
    B, H, W, Ci, N, k, s, pad, name = case
    torch.manual_seed(sum(map(ord, name)))
    OH, OW = _out_hw(H, W, k, s, pad, name)
    x = rnd(torch.randn(B, H, W, Ci), dt)
    w = torch.randn(k, k, Ci, N) / np.sqrt(k * k * Ci)
    bias = torch.randn(N) * 0.1
    gamma, beta = torch.rand(N) + 0.5, torch.randn(N) * 0.1
    mean, var = torch.randn(N) * 0.1, torch.rand(N) + 0.5
    res = rnd(torch.randn(B, OH, OW, N), dt)
    wf, wd, biasf, scale = prep_weights(w, dt, bias, (gamma, beta, mean, var))
    # reference uses the SAME folded+rounded weights so that only the kernel arithmetic is compared
    sc = gamma / torch.sqrt(var + 1e-3)
    w_fold = rnd(w * sc, dt)
    b_fold = sc * bias + beta - mean * sc
    x_r = x.clone().requires_grad_(True)
    w_r = w_fold.clone().requires_grad_(True)
    z = _ref_conv(x_r, w_r, s, pad, OH, OW) + b_fold + res
    y_ref = F.relu(z)
    # ---- forward: conv + bias + residual + relu
    g = hip.geom(B, H, W, Ci, OH, OW, N, k, k, s, s, pad[0], pad[1])
    y = torch.empty(B, OH, OW, N, dtype=hip.TORCH_DT[dt], device="cuda")
    hip.conv_igemm(g, dt, hip.EPI_RELU, dev(x, dt), wf, biasf, dev(res, dt), None, y)
    torch.cuda.synchronize()
    e = relerr(y, y_ref)
    assert e < TOL[dt], "forward %s dt=%d rel err %.3e" % (name, dt, e)
    y2 = torch.empty_like(y)                               # conv + bias + relu (no residual): the form of every branch2a / branch2b layer
    hip.conv_igemm(g, dt, hip.EPI_RELU, dev(x, dt), wf, biasf, None, None, y2)
    torch.cuda.synchronize()
    e = relerr(y2, F.relu(z - res))
    assert e < TOL[dt], "forward (no residual) %s dt=%d rel err %.3e" % (name, dt, e)
    # ---- backward reference
    dy = rnd(torch.randn(B, OH, OW, N), dt)
    dz_ref = (dy * (y_ref > 0)).detach()
    dz_ref = rnd(dz_ref, dt)
    (z * dz_ref).sum().backward()
    # ---- data gradient: gather-form implicit GEMM with flipped taps, + add tensor, + relu mask of x
    addt = rnd(torch.randn(B, H, W, Ci), dt)
    gd = hip.geom(B, OH, OW, N, H, W, Ci, k, k, 1, 1, k - 1 - pad[0], k - 1 - pad[1], s, s)
    dx = torch.empty(B, H, W, Ci, dtype=hip.TORCH_DT[dt], device="cuda")
    hip.conv_igemm(gd, dt, 0, dev(dz_ref, dt), wd, None, dev(addt, dt), dev(x, dt), dx)
    torch.cuda.synchronize()
    dx_ref = (x_r.grad + addt) * (x > 0)
    e = relerr(dx, dx_ref)
    assert e < TOL[dt], "dgrad %s dt=%d rel err %.3e" % (name, dt, e)
    # the same pass as res{3,4,5}x_branch2b runs it: ReLU mask of x, no residual gradient (conv_halo.hip takes this form)
    hip.conv_igemm(gd, dt, 0, dev(dz_ref, dt), wd, None, None, dev(x, dt), dx)
    torch.cuda.synchronize()
    e = relerr(dx, x_r.grad * (x > 0))
    assert e < TOL[dt], "dgrad (mask only) %s dt=%d rel err %.3e" % (name, dt, e)
    if k == 1 and s == 2:
        # compact form used for res{3,4,5}a_branch{2a,1}: GEMM over the output pixels, scattered to every 2nd pixel
        gs = hip.geom(B, OH, OW, N, OH, OW, Ci, 1, 1, FH=H, FW=W, OSH=s, OSW=s)
        dx2 = torch.zeros(B, H, W, Ci, dtype=hip.TORCH_DT[dt], device="cuda")
        hip.conv_igemm(gs, dt, 0, dev(dz_ref, dt), wd, None, None, dev(x, dt), dx2)
        torch.cuda.synchronize()
        first = x_r.grad * (x > 0)
        assert relerr(dx2, first) < TOL[dt], "compact dgrad"
        hip.conv_igemm(gs, dt, 0, dev(dz_ref, dt), wd, None, dx2, dev(x, dt), dx2)      # in-place accumulate
        torch.cuda.synchronize()
        assert relerr(dx2, 2 * rnd(first, dt)) < TOL[dt] * 2, "compact dgrad accumulate"
    # ---- weight gradient (raw, w.r.t. the folded filter) + column sums
    ws = torch.empty(hip.conv_wgrad_ws_bytes(g, dt) // 4 + 16, dtype=torch.float32, device="cuda")
    dw = torch.empty(k, k, Ci, N, dtype=torch.float32, device="cuda")
    cs = torch.empty(N, dtype=torch.float32, device="cuda")
    hip.conv_wgrad(g, dt, dev(x, dt), dev(dz_ref, dt), ws, dw, cs)
    torch.cuda.synchronize()
    e = relerr(dw, w_r.grad)
    assert e < TOL[dt] * (1 if dt == 0 else 0.5), "wgrad %s dt=%d rel err %.3e" % (name, dt, e)
    e = relerr(cs, dz_ref.sum(dim=(0, 1, 2)))
    assert e < 1e-4, "colsum %s dt=%d rel err %.3e" % (name, dt, e)


# The persistent conv kernels (conv_pw.hip pw_kernel, conv_igemm.hip igemm_kernel) walk several tiles per block only when a
# launch has more tiles than resident blocks (cap = 512 / 768): the cross-tile machinery -- the next tile's first K-tile copied
# under the epilogue, the rolling residual / mask prefetch, the hand-counted vmcnt waits -- is what every full-size layer of
# the benchmark runs.  (a) shapes with thousands of tiles at their production grid, (b) small shapes with the grid capped
# through urso_set_option("grid_cap") so that every block still walks >= 5 tiles.  Same CPU reference as the small cases.
BIG_CASES = [
    (4, 256, 320, 64, 256, 1, 1, (0, 0), "big_1x1_64_256"),       # res2x_branch2c / branch1 (+ residual, ReLU): 5120 tiles
    (4, 256, 320, 256, 64, 1, 1, (0, 0), "big_1x1_256_64"),       # res2x_branch2a; its dgrad is the 64 -> 256 add+mask layer
    (4, 256, 320, 64, 64, 3, 1, (1, 1), "big_3x3_64"),            # res2x_branch2b
    (4, 128, 160, 256, 128, 1, 2, (0, 0), "big_1x1_s2"),          # res3a_branch2a: strided forward, scattered compact dgrad
    (8, 64, 80, 128, 128, 3, 1, (1, 1), "big_3x3_128"),           # res3x_branch2b
    (16, 32, 40, 256, 256, 3, 1, (1, 1), "big_3x3_256"),          # res4x_branch2b
    (16, 32, 40, 1024, 256, 1, 1, (0, 0), "big_1x1_1024_256"),    # res4x_branch2a / dgrad of branch2c
    (16, 64, 80, 128, 512, 1, 1, (0, 0), "big_1x1_128_512"),      # res3x_branch2c: stage-3 shapes of conv_pair.hip / conv_pwgrad.hip
    (16, 64, 80, 512, 128, 1, 1, (0, 0), "big_1x1_512_128"),      # res3x_branch2a
]
CAP_CASES = [
    (2, 32, 40, 64, 256, 1, 1, (0, 0), "cap_1x1_64_256"),
    (2, 32, 40, 256, 64, 1, 1, (0, 0), "cap_1x1_256_64"),
    (2, 32, 40, 64, 64, 3, 1, (1, 1), "cap_3x3_64"),
    (2, 32, 40, 128, 128, 3, 1, (1, 1), "cap_3x3_128"),
    (2, 32, 48, 128, 64, 1, 2, (0, 0), "cap_1x1_s2"),
    (3, 19, 23, 32, 160, 3, 1, (1, 1), "cap_3x3_ragged"),
]


@pytest.mark.parametrize("dt", [0, 1, 2])
@pytest.mark.parametrize("case", BIG_CASES, ids=[c[-1] for c in BIG_CASES])
def test_conv_full_size_tile_streams(case, dt):
    test_conv_forward_and_gradients(case, dt)


@pytest.mark.parametrize("dt", [0, 1, 2])
@pytest.mark.parametrize("case", CAP_CASES, ids=[c[-1] for c in CAP_CASES])
def test_conv_capped_grid_tile_streams(case, dt):
    hip = _hip()
    with hip.options(grid_cap=8):
        test_conv_forward_and_gradients(case, dt)


HALO_CASES = [c for c in BIG_CASES + CAP_CASES if c[5] == 3 and c[3] % 128 == 0 and c[4] % 128 == 0]


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("cap", [0, 8, 24])
@pytest.mark.parametrize("case", HALO_CASES, ids=[c[-1] for c in HALO_CASES])
def test_halo_conv_kernel_forced(case, dt, cap):
    """conv_halo.hip on every shape it accepts (urso_set_option hconv = 2 overrides the tile-count policy), with the grid at its
    production size and capped to 8 / 24 blocks so that a block's contiguous run holds several tiles: continuous halo / filter
    stream across tile seams, several filter tiles per pixel tile (N = 256), transposed epilogue, mask-only data gradient."""
    hip = _hip()
    with hip.options(hconv=2, hconv2=0, grid_cap=cap):
        test_conv_forward_and_gradients(case, dt)


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("cap", [0, 8])
@pytest.mark.parametrize("shape", [32, 31, 22, 21, 12, 11])
@pytest.mark.parametrize("case", HALO_CASES, ids=[c[-1] for c in HALO_CASES])
def test_halo2_conv_kernel_forced(case, shape, dt, cap):
    """conv_halo2.hip (whole tiles of 128 MI x 64 NJ, accumulators in AccVGPRs, two-slot filter ring, halo row offsets from an LDS table)
    on every tile shape that fits the layer, with the grid at its production size and capped to 8 blocks so that a block walks several
    tiles: the continuous halo / filter stream across tile seams, the next tile's offset table filled while this one runs, the epilogue's
    stores in flight across the seam; forward with and without ReLU / residual-free, masked data gradient (the forms res{4,5}x_branch2b use)."""
    hip = _hip()
    B, H, W, Ci, N = case[:5]
    with hip.options(hconv=2, hconv2=2, hconv2_shape=shape, grid_cap=cap):
        g = hip.geom(B, H, W, Ci, H, W, N, 3, 3, 1, 1, 1, 1)
        if hip.conv_igemm_halo2_shape(g, dt, hip.EPI_RELU) != shape:
            pytest.skip("tile shape %d does not fit this layer" % shape)
        test_conv_forward_and_gradients(case, dt)


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("shape", [(32, 32, 40, 256, 256), (32, 16, 20, 512, 512), (16, 32, 40, 256, 256), (3, 19, 23, 128, 384)])
def test_halo2_conv_equals_whole_tile_halo_kernel_bit_for_bit(shape, dt):
    """Same MFMA, same k order per output element: conv_halo2.hip on every tile shape that fits (incl. the one its cost model picks for the
    cfg2 stage-4 / stage-5 layers) equals conv_halo.hip's whole-tile schedule bit for bit -- forward with ReLU and the masked data-gradient
    form -- and is reproducible from launch to launch."""
    hip = _hip()
    B, H, W, Ci, N = shape
    torch.manual_seed(B + H + Ci)
    tdt = hip.TORCH_DT[dt]
    x = torch.randn(B, H, W, Ci, device="cuda").to(tdt)
    wf = (torch.randn(N, 3, 3, Ci, device="cuda") / np.sqrt(9 * Ci)).to(tdt)
    bias = torch.randn(N, device="cuda")
    msk = torch.randn(B, H, W, N, device="cuda").to(tdt)
    g = hip.geom(B, H, W, Ci, H, W, N, 3, 3, 1, 1, 1, 1)

    def run():
        y = torch.full((B, H, W, N), 3.0, dtype=tdt, device="cuda"); ym = torch.full((B, H, W, N), 3.0, dtype=tdt, device="cuda")
        hip.conv_igemm(g, dt, hip.EPI_RELU, x, wf, bias, None, None, y)
        hip.conv_igemm(g, dt, 0, x, wf, bias, None, msk, ym)
        torch.cuda.synchronize()
        return y, ym
    with hip.options(hconv=2, hconv2=0, c3=0):
        y0, ym0 = run()
    taken = 0
    with hip.options(hconv=2, c3=0):
        picked = hip.conv_igemm_halo2_shape(g, dt, hip.EPI_RELU)
    for shp in (32, 31, 22, 21, 12, 11):
        with hip.options(hconv=2, hconv2=2, hconv2_shape=shp, c3=0):
            if hip.conv_igemm_halo2_shape(g, dt, hip.EPI_RELU) != shp:
                continue
            y, ym = run(); y2, ym2 = run()
        taken += 1
        assert torch.equal(y, y0) and torch.equal(ym, ym0), "tile shape %d differs from conv_halo.hip" % shp
        assert torch.equal(y, y2) and torch.equal(ym, ym2), "tile shape %d is not reproducible" % shp
    assert taken >= 2
    if B >= 16:
        assert picked in (32, 31), "cost model: cfg2 / cfg4 stage-4 and stage-5 layers run on 384-pixel tiles (picked %d)" % picked


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("shape", [(8, 64, 80, 128, 128, 0), (16, 32, 40, 256, 256, 0), (32, 16, 20, 512, 512, 0), (2, 32, 40, 128, 256, 8),
                                   (3, 19, 23, 128, 128, 24), (32, 32, 40, 256, 256, 0), (4, 16, 20, 512, 128, 40)])
def test_halo_conv_stream_k_schedule(shape, dt):
    """conv_halo.hip with its hand-over workspace (urso_conv_igemm_ws + urso_conv_igemm_halo_ws_bytes): the layer's (tile, 64-channel
    chunk) units are dealt to the blocks in equal contiguous runs, tiles cut by a run boundary are completed from fp32 partial
    accumulators in a fixed order.  Against the CPU fp32 conv at the kernel tolerance, against the whole-tile schedule of the same
    kernel (which differs only in where the fp32 sum is split), twice in a row on the same workspace (the flags must come back zero),
    forward with ReLU and data-gradient form with a mask.  Shapes: tiles spanning 2 and 3 runs, runs spanning several tiles, several
    filter tiles per pixel tile, ragged sizes, capped grids; hconv_dbg = 8 engages the schedule also where the policy would keep whole
    tiles, the stage-4 / stage-5 shapes of cfg2 (16 / 32 x ... x 256 / 512) engage it by policy."""
    hip = _hip()
    B, H, W, Ci, N, cap = shape
    torch.manual_seed(B * 7 + H + Ci)
    tdt = hip.TORCH_DT[dt]
    x = rnd(torch.randn(B, H, W, Ci), dt)
    w = torch.randn(3, 3, Ci, N) / np.sqrt(9 * Ci)
    bias = torch.randn(N) * 0.1
    msk = rnd(torch.randn(B, H, W, N), dt)
    wf, wd, biasf, _ = prep_weights(w, dt, bias, None)
    g = hip.geom(B, H, W, Ci, H, W, N, 3, 3, 1, 1, 1, 1)
    z = _ref_conv(x, rnd(w, dt), 1, (1, 1), H, W) + bias
    ws = torch.zeros(hip.conv_igemm_halo_ws_bytes() // 4 + 16, dtype=torch.float32, device="cuda")
    xd, md = dev(x, dt), dev(msk, dt)
    with hip.options(hconv=2, grid_cap=cap, c3=0):
        assert hip.conv_igemm_halo_ok(g, dt, hip.EPI_RELU)
        y0 = torch.empty(B, H, W, N, dtype=tdt, device="cuda")
        hip.conv_igemm(g, dt, hip.EPI_RELU, xd, wf, biasf, None, None, y0)                       # whole tiles per block
        outs = []
        for force in (8, 8, 0):
            with hip.options(hconv_dbg=force):
                y = torch.full((B, H, W, N), 3.0, dtype=tdt, device="cuda")
                hip.conv_igemm_ws(g, dt, hip.EPI_RELU, xd, wf, biasf, None, None, y, ws)       # stream-K (forced, forced again, by policy)
                torch.cuda.synchronize()
            assert int(ws[:1024].view(torch.int32).abs().max()) == 0, "hand-over flags not left zero"
            outs.append(y)
        ym = torch.empty_like(y0)
        with hip.options(hconv_dbg=8):
            hip.conv_igemm_ws(g, dt, 0, xd, wf, biasf, None, md, ym, ws)                        # no ReLU, mask
        torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]), "stream-K result is not reproducible"
    assert relerr(outs[0], F.relu(z)) < TOL[dt] and relerr(outs[2], F.relu(z)) < TOL[dt]
    assert relerr(outs[0], y0.float()) < 0.5 * TOL[dt]
    assert relerr(ym, z * (msk > 0)) < TOL[dt]


def test_wgrad_split_plan_reaches_its_resident_block_target():
    """The 16-bit weight-gradient kernels split the pixel dimension over one resident wave of blocks (512, x1.5 for the
    narrow tile): the full-size cases above must actually run with hundreds of splits ('wgrad at splits == 512')."""
    hip = _hip()
    g = hip.geom(4, 256, 320, 64, 256, 320, 64, 1, 1)
    assert hip.conv_wgrad_splits(g, 1) >= 256
    g = hip.geom(32, 128, 160, 64, 128, 160, 64, 3, 3, 1, 1, 1, 1)
    assert hip.conv_wgrad_splits(g, 1) * 5 >= 512


@pytest.mark.parametrize("dt", [0, 1])
def test_igemm_out_f32_and_padded_head(dt):
    """loc_final-style head: N=3 padded to 8, fp32 output, no activation (net.py:316)."""
    hip = _hip()
    B, K, N, NP = 4, 1024, 3, 8
    torch.manual_seed(5)
    x = rnd(torch.randn(B, 1, 1, K), dt)
    w = torch.randn(1, 1, K, N) / 32
    bias = torch.randn(N)
    wf, wd, biasf, scale = prep_weights(w, dt, bias, None, npad=NP)
    g = hip.geom(B, 1, 1, K, 1, 1, NP, 1, 1)
    y = torch.full((B, NP), 7.0, dtype=torch.float32, device="cuda")
    hip.conv_igemm(g, dt, hip.EPI_OUT_F32, dev(x, dt), wf, biasf, None, None, y)
    torch.cuda.synchronize()
    ref = x.reshape(B, K) @ rnd(w.reshape(K, N), dt) + bias
    assert relerr(y[:, :N], ref) < TOL[dt]
    assert float(y[:, N:].abs().max()) == 0.0
    # dgrad through the padded head
    dz = torch.zeros(B, NP); dz[:, :N] = torch.randn(B, N); dz = rnd(dz, dt)
    gd = hip.geom(B, 1, 1, NP, 1, 1, K, 1, 1)
    dx = torch.empty(B, K, dtype=hip.TORCH_DT[dt], device="cuda")
    hip.conv_igemm(gd, dt, 0, dev(dz, dt), wd, None, None, None, dx)
    torch.cuda.synchronize()
    assert relerr(dx, dz[:, :N] @ rnd(w.reshape(K, N), dt).T) < TOL[dt]


@pytest.mark.parametrize("dt", [0, 1, 2])
@pytest.mark.parametrize("size", [(2, 32, 48, 0), (3, 36, 44, 0), (2, 128, 160, 8), (4, 512, 640, 0)], ids=["small", "ragged", "capped", "full"])
def test_stem_pixel_pair_conv(dt, size):
    """conv1: ZeroPadding2D(3)+Conv2D 7x7 s2 (net.py:170-171) as a 7x4-tap conv on pixel pairs,
    forward + weight gradient (no data gradient: the image needs none).  'capped' / 'full': every block of the persistent
    kernel walks several tiles (grid cap 8 on 80 tiles; 2560 tiles on the production grid); 'ragged': output sizes that are not multiples
    of the 8 x 32 tile of conv_stem.hip.  16-bit dtypes run conv_stem.hip (option stem, default) AND the DMA kernel's form (stem = 0)."""
    hip = _hip()
    B, H, W, cap = size
    N = 64
    for stem in ((1, 0) if dt else (1,)):
        with hip.options(grid_cap=cap, stem=stem):
            _stem_case(hip, dt, B, H, W, N)


@pytest.mark.parametrize("dt", [0, 1, 2])
@pytest.mark.parametrize("shape", [(2, 32, 48), (1, 5, 7), (3, 36, 44)], ids=["quads", "ragged", "quads2"])
def test_mold_images_float_source(dt, shape):
    """urso_mold_images on a float32 source (what Engine.load_batch hands over), with and without a mean: [B,H,W,3] -> [B,H,W,4] in dt with a
    zero fourth channel.  Pixel counts that are multiples of 4 take the four-pixels-per-thread kernel, the others the scalar one."""
    hip = _hip()
    B, H, W = shape
    torch.manual_seed(5)
    src = (torch.rand(B, H, W, 3) * 255 - 120).cuda()
    for mean in (None, torch.tensor([123.7, 116.8, 103.9]).cuda()):
        out = torch.full((B, H, W, 4), 7.0, dtype=hip.TORCH_DT[dt], device="cuda")
        hip.mold_images(B, H, W, src, mean, dt, out)
        torch.cuda.synchronize()
        ref = src if mean is None else src - mean
        assert torch.equal(out[..., :3].float(), ref.to(hip.TORCH_DT[dt]).float()) and float(out[..., 3].float().abs().max()) == 0.0


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("shape", [(2, 64, 128), (1, 6, 12), (3, 5, 7)], ids=["octets", "octets_small", "ragged"])
def test_mold_images_uint8_source_eight_pixels_per_thread(dt, shape):
    """urso_mold_images on uint8 frames (Engine.load_batch_u8; net.py:1346 mold_image): the 8-pixels-per-thread form (round 6) against the
    one-pixel form (option mold_scalar) and against torch, bit for bit; pixel counts that are no multiple of 8 keep the scalar form."""
    hip = _hip()
    B, H, W = shape
    torch.manual_seed(6)
    src = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8).cuda()
    mean = torch.tensor([123.7, 116.8, 103.9]).cuda()
    outs = []
    for scalar in (0, 1):
        with hip.options(mold_scalar=scalar):
            out = torch.full((B, H, W, 4), 7.0, dtype=hip.TORCH_DT[dt], device="cuda")
            hip.mold_images(B, H, W, src, mean, dt, out)
            torch.cuda.synchronize()
            outs.append(out)
    assert torch.equal(outs[0], outs[1])
    ref = (src.float() - mean).to(hip.TORCH_DT[dt])
    assert torch.equal(outs[0][..., :3], ref) and float(outs[0][..., 3].float().abs().max()) == 0.0


def _stem_case(hip, dt, B, H, W, N):
    torch.manual_seed(7)
    img = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8)
    meanp = torch.tensor([123.7, 116.8, 103.9])
    molded = torch.empty(B, H, W, 4, dtype=hip.TORCH_DT[dt], device="cuda")
    hip.mold_images(B, H, W, img.cuda(), meanp.cuda(), dt, molded)
    torch.cuda.synchronize()
    x = rnd(img.float() - meanp, dt)
    assert relerr(molded[..., :3], x) < 1e-6 and float(molded[..., 3].float().abs().max()) == 0.0
    w = torch.randn(7, 7, 3, N) / 12
    bias = torch.randn(N) * 0.1
    bn = (torch.rand(N) + 0.5, torch.randn(N) * 0.1, torch.randn(N) * 0.1, torch.rand(N) + 0.5)
    wf = torch.empty(N * 7 * 4 * 8, dtype=hip.TORCH_DT[dt], device="cuda")
    biasf = torch.empty(N, dtype=torch.float32, device="cuda"); scale = torch.empty_like(biasf)
    hip.stem_weight_pack(N, dt, dev(w), dev(bias), dev(bn[0]), dev(bn[1]), dev(bn[2]), dev(bn[3]), 1e-3, wf, biasf, scale)
    OH, OW = H // 2, W // 2
    g = hip.geom(B, H, W // 2, 8, OH, OW, N, 7, 4, 2, 1, 3, 2)
    y = torch.empty(B, OH, OW, N, dtype=hip.TORCH_DT[dt], device="cuda")
    hip.conv_igemm(g, dt, hip.EPI_RELU, molded, wf, biasf, None, None, y)
    torch.cuda.synchronize()
    sc = bn[0] / torch.sqrt(bn[3] + 1e-3)
    w_fold = rnd(w * sc, dt).requires_grad_(True)
    z = _ref_conv(x, w_fold, 2, (3, 3), OH, OW) + (sc * bias + bn[1] - bn[2] * sc)
    assert relerr(y, F.relu(z)) < TOL[dt]
    dz = rnd(torch.randn(B, OH, OW, N), dt)
    (z * dz).sum().backward()
    ws = torch.empty(hip.conv_wgrad_ws_bytes(g, dt) // 4 + 16, dtype=torch.float32, device="cuda")
    dwp = torch.empty(7 * 4 * 8 * N, dtype=torch.float32, device="cuda")
    cs = torch.empty(N, dtype=torch.float32, device="cuda")
    hip.conv_wgrad(g, dt, molded, dev(dz, dt), ws, dwp, cs)
    dw = torch.empty(7, 7, 3, N, dtype=torch.float32, device="cuda")
    hip.stem_wgrad_unpack(N, dwp, dw)
    torch.cuda.synchronize()
    assert relerr(dw, w_fold.grad) < TOL[dt]
    assert relerr(cs, dz.sum((0, 1, 2))) < max(TOL[dt] * 1e-2, 2e-5)          # column sums of the (already rounded) gradient: fp32 sums
    if dt and hip.get_option("stem"):
        # the same weight gradient from the gradient of a max-pool behind the conv (urso_stem_wgrad_pooled) against
        # urso_maxpool3x3s2_bwd + urso_conv_wgrad
        PH, PW = OH // 2, OW // 2
        pooled = torch.empty(B, PH, PW, N, dtype=hip.TORCH_DT[dt], device="cuda")
        am = torch.empty(B, PH, PW, N, dtype=torch.uint8, device="cuda")
        hip.maxpool_fwd(B, OH, OW, N, dt, y, pooled, am)
        if hip.stem_conv_pool_ok(g, dt):
            # conv1 + ReLU + max-pool in one kernel (urso_stem_conv_pool): the pooled tensor and its arg-max bytes, bit for bit
            pooled2 = torch.full_like(pooled, float("nan")); am2 = torch.full_like(am, 255)
            hip.stem_conv_pool(g, dt, molded, wf, biasf, pooled2, am2)
            torch.cuda.synchronize()
            assert torch.equal(pooled2.float(), pooled.float()), "fused stem + pool: %d values differ" % int((pooled2.float() != pooled.float()).sum())
            assert torch.equal(am2, am), "fused stem + pool: %d arg-max bytes differ" % int((am2 != am).sum())
        else:
            assert (OH | OW) & 1, "urso_stem_conv_pool_ok refused an even conv grid"
        dpool = dev(rnd(torch.randn(B, PH, PW, N), dt), dt)
        dzp = torch.empty(B, OH, OW, N, dtype=hip.TORCH_DT[dt], device="cuda")
        hip.maxpool_bwd(B, OH, OW, N, dt, pooled, dpool, am, 1, dzp)
        dw1, cs1 = torch.empty_like(dwp), torch.empty_like(cs)
        hip.conv_wgrad(g, dt, molded, dzp, ws, dw1, cs1)
        torch.cuda.synchronize()
        dw2, cs2 = torch.full_like(dwp, float("nan")), torch.full_like(cs, float("nan"))
        hip.stem_wgrad_pooled(g, dt, molded, dpool, am, ws, dw2, cs2)
        torch.cuda.synchronize()
        assert float((dw2 - dw1).abs().max()) <= 2e-5 * float(dw1.abs().max()) and float((cs2 - cs1).abs().max()) <= 2e-5 * float(cs1.abs().max()) + 1e-5
        assert float(dw1.abs().max()) > 0 and 0.02 < float((dzp.float() != 0).float().mean()) < 0.3


@pytest.mark.parametrize("dt", [0, 1])
def test_split_k_small_grid_layers(dt):
    """bottleneck_layer-like (K = 3*3*512 on a handful of tiles) and Dense-like shapes through the split-K path
    (urso_conv_igemm_ws) must equal the unsplit kernel's math: bias + residual + ReLU + mask after the reduction."""
    hip = _hip()
    torch.manual_seed(21)
    for (B, H, W, Ci, N, k, s, pad, relu) in [(2, 8, 10, 512, 32, 3, 2, (0, 0), False), (8, 1, 1, 2560, 256, 1, 1, (0, 0), True)]:
        OH, OW = (-(-H // s), -(-W // s)) if k == 3 else (1, 1)
        x = rnd(torch.randn(B, H, W, Ci), dt)
        w = torch.randn(k, k, Ci, N) / np.sqrt(k * k * Ci)
        bias = torch.randn(N)
        res = rnd(torch.randn(B, OH, OW, N), dt)
        wf, wd, biasf, scale = prep_weights(w, dt, bias, None)
        g = hip.geom(B, H, W, Ci, OH, OW, N, k, k, s, s, pad[0], pad[1])
        nbytes = hip.conv_igemm_ws_bytes(g, dt)
        assert nbytes > 0, "expected the split-K plan for this shape"
        ws = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device="cuda")
        y = torch.empty(B, OH, OW, N, dtype=hip.TORCH_DT[dt], device="cuda")
        hip.conv_igemm_ws(g, dt, hip.EPI_RELU if relu else 0, dev(x, dt), wf, biasf, dev(res, dt), dev(res, dt), y, ws)
        y2 = torch.empty_like(y)
        hip.conv_igemm(g, dt, hip.EPI_RELU if relu else 0, dev(x, dt), wf, biasf, dev(res, dt), dev(res, dt), y2)
        torch.cuda.synchronize()
        z = _ref_conv(x, rnd(w, dt), s, pad, OH, OW) + bias + res
        ref = (torch.relu(z) if relu else z) * (res > 0)
        assert relerr(y, ref) < TOL[dt] and relerr(y2, ref) < TOL[dt]


def test_weight_prep_layouts():
    hip = _hip()
    torch.manual_seed(3)
    KH, KW, Ci, N, NP = 3, 3, 40, 50, 56
    w = torch.randn(KH, KW, Ci, N)
    b = torch.randn(N)
    bn = (torch.rand(N) + 0.5, torch.randn(N), torch.randn(N), torch.rand(N) + 0.5)
    wf, wd, biasf, scale = prep_weights(w, 0, b, bn, npad=NP)
    torch.cuda.synchronize()
    s = bn[0] / torch.sqrt(bn[3] + 1e-3)
    wfr = torch.zeros(NP, KH, KW, Ci); wfr[:N] = (w * s).permute(3, 0, 1, 2)
    wdr = torch.zeros(Ci, KH, KW, NP); wdr[..., :N] = (w * s).flip(0, 1).permute(2, 0, 1, 3)
    assert relerr(wf.view(NP, KH, KW, Ci), wfr) < 1e-6
    assert relerr(wd.view(Ci, KH, KW, NP), wdr) < 1e-6
    assert relerr(biasf[:N], s * b + bn[1] - bn[2] * s) < 1e-6 and relerr(scale[:N], s) < 1e-6
    assert float(biasf[N:].abs().max()) == 0 and float((scale[N:] - 1).abs().max()) == 0


def test_param_grad_finalize_matches_autograd_through_frozen_bn():
    """gamma/beta/bias/kernel gradients recovered from (dw_raw, colsum) == autograd through
    conv -> BatchNorm(training=False) (net.py:60-76) + the L2 term of net.py:1008-1012."""
    from oracle import graph_ref as G
    hip = _hip()
    torch.manual_seed(11)
    B, H, W, Ci, N = 2, 6, 5, 16, 24
    x = torch.randn(B, Ci, H, W)
    P = {"kernel": torch.randn(3, 3, Ci, N).requires_grad_(True), "bias": torch.randn(N).requires_grad_(True)}
    Q = {"gamma": (torch.rand(N) + 0.5).requires_grad_(True), "beta": torch.randn(N).requires_grad_(True),
         "moving_mean": torch.randn(N), "moving_variance": torch.rand(N) + 0.5}
    z = G.batchnorm(G.conv2d(x, P, 1, "same"), Q, False)
    dz = torch.randn_like(z)
    wd_ = 1e-2
    reg = wd_ * (P["kernel"] ** 2).sum() / P["kernel"].numel() + wd_ * (P["bias"] ** 2).sum() / N
    ((z * dz).sum() + reg).backward()
    # raw gradient w.r.t. the FOLDED filter is what urso_conv_wgrad produces: dw_raw = x^T dz
    xp = F.pad(x, (1, 1, 1, 1))
    dw_raw = torch.zeros(3, 3, Ci, N)
    for ky in range(3):
        for kx in range(3):
            dw_raw[ky, kx] = torch.einsum("bchw,bnhw->cn", xp[:, :, ky:ky + H, kx:kx + W], dz)
    colsum = dz.sum(dim=(0, 2, 3))
    K = 9 * Ci
    gw = torch.empty(K, N, device="cuda"); gb = torch.empty(N, device="cuda")
    gg = torch.empty(N, device="cuda"); gbe = torch.empty(N, device="cuda")
    ws = torch.empty(hip.param_grad_finalize_ws_bytes(K, N) // 4 + 4, device="cuda")
    hip.param_grad_finalize(K, N, N, dev(dw_raw.reshape(K, N)), dev(colsum), dev(P["kernel"].detach().reshape(K, N)),
                            dev(P["bias"].detach()), dev(Q["gamma"].detach()), dev(Q["moving_mean"]),
                            dev(Q["moving_variance"]), 1e-3, wd_, 1, 1, gw, gb, gg, gbe, ws)
    torch.cuda.synchronize()
    assert relerr(gw, P["kernel"].grad.reshape(K, N)) < 2e-5
    assert relerr(gb, P["bias"].grad) < 2e-5
    assert relerr(gg, Q["gamma"].grad) < 2e-5
    assert relerr(gbe, Q["beta"].grad) < 2e-5


@pytest.mark.parametrize("shape", [(2, 12, 16, 16), (3, 10, 20, 64), (1, 34, 70, 8)])      # the forward kernel walks 4 x 8 output tiles: whole and ragged
@pytest.mark.parametrize("dt", [0, 1])
def test_maxpool_same_fwd_bwd(dt, shape):
    from oracle import graph_ref as G
    hip = _hip()
    torch.manual_seed(2)
    B, H, W, Cc = shape
    x = F.relu(rnd(torch.randn(B, H, W, Cc), dt))          # post-ReLU input with many zero ties
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    yr = G.maxpool_3x3_s2_same(xr)
    y = torch.empty(B, H // 2, W // 2, Cc, dtype=hip.TORCH_DT[dt], device="cuda")
    am = torch.empty(B, H // 2, W // 2, Cc, dtype=torch.uint8, device="cuda")
    hip.maxpool_fwd(B, H, W, Cc, dt, dev(x, dt), y, am)
    torch.cuda.synchronize()
    assert relerr(y, yr.permute(0, 2, 3, 1)) == 0.0
    dy = rnd(torch.randn(B, H // 2, W // 2, Cc), dt)
    (yr * dy.permute(0, 3, 1, 2)).sum().backward()
    dx = torch.empty(B, H, W, Cc, dtype=hip.TORCH_DT[dt], device="cuda")
    hip.maxpool_bwd(B, H, W, Cc, dt, y, dev(dy, dt), am, 1, dx)
    torch.cuda.synchronize()
    # fused ReLU mask: gradient w.r.t. the pre-ReLU tensor = dx * (x > 0)
    ref = xr.grad.permute(0, 2, 3, 1) * (x > 0)
    assert relerr(dx, ref) < (1e-6 if dt == 0 else 1e-2)


@pytest.mark.parametrize("label_sum", [1.0, 0.6, 1.7], ids=["normalised", "sum_below_1", "sum_above_1"])
@pytest.mark.parametrize("K", [4096, 13824, 64, 1000, 17000])
def test_softmax_xent_soft_labels(K, label_sum):
    """Loss and gradient against the oracle; label rows that do NOT sum to one pin the TF-kernel semantics on both sides: the
    gradient is softmax - p (not softmax * sum(p) - p), the loss logsumexp * sum(p) - sum(p z)."""
    from oracle import graph_ref as G
    hip = _hip()
    torch.manual_seed(K)
    B = 6
    z = F.relu(torch.randn(B, K) * 2).requires_grad_(True)          # logits are post-ReLU (net.py:350)
    p = torch.softmax(torch.randn(B, K) * 3, -1) * label_sum
    loss = G.softmax_loss(p, z) * 0.7
    loss.backward()
    lo = torch.empty(1, device="cuda"); dz = torch.empty(B, K, device="cuda"); rw = torch.empty(B, device="cuda")
    hip.softmax_xent(B, K, dev(z.detach()), dev(p), 0.7, 1, 0, lo, dz, rw)
    torch.cuda.synchronize()
    assert abs(float(lo) - float(loss)) < 1e-5 * abs(float(loss))
    assert relerr(dz, z.grad * (z.detach() > 0)) < 1e-4


def test_regression_losses():
    from oracle import graph_ref as G
    hip = _hip()
    torch.manual_seed(9)
    B, LD = 5, 8
    gt = torch.randn(B, 3) * 5
    pred = torch.zeros(B, LD); pred[:, :3] = torch.randn(B, 3) * 5
    pr = pred[:, :3].clone().requires_grad_(True)
    l = G.rel_loss(gt, pr) * 1.3; l.backward()
    lo = torch.empty(1, device="cuda"); dp = torch.empty(B, LD, device="cuda"); nr = torch.empty(2, device="cuda")
    hip.rel_l2(B, 3, LD, dev(gt), dev(pred), 1.3, 0, lo, dp, nr)
    torch.cuda.synchronize()
    assert abs(float(lo) - float(l)) < 1e-5 * float(l)
    assert relerr(dp[:, :3], pr.grad) < 1e-5 and float(dp[:, 3:].abs().max()) == 0
    # quaternion head: l2-normalize + 1-|dot| (net.py:345-346, 724-733)
    q_gt = F.normalize(torch.randn(B, 4), dim=-1)
    x = torch.zeros(B, LD); x[:, :4] = torch.randn(B, 4)
    xr = x[:, :4].clone().requires_grad_(True)
    qn = xr * torch.rsqrt(torch.clamp((xr * xr).sum(-1, keepdim=True), min=1e-12))
    l2 = G.one_minus_dot_prod(q_gt, qn) * 0.9; l2.backward()
    qo = torch.empty(B, 4, device="cuda"); dx = torch.empty(B, LD, device="cuda")
    hip.absdot(B, 4, LD, 1, dev(q_gt), dev(x), 0.9, 0, qo, lo, dx)
    torch.cuda.synchronize()
    assert relerr(qo, qn) < 1e-6 and abs(float(lo) - float(l2)) < 1e-5
    assert relerr(dx[:, :4], xr.grad) < 1e-5
    # MSE (keypoint mode, net.py:735-748)
    pm = pred[:, :3].clone().requires_grad_(True)
    l3 = G.mse_loss(gt, pm); l3.backward()
    hip.mse(B, 3, LD, dev(gt), dev(pred), 1.0, 0, lo, dp)
    torch.cuda.synchronize()
    assert abs(float(lo) - float(l3)) < 1e-5 * float(l3) and relerr(dp[:, :3], pm.grad) < 1e-5


@pytest.mark.parametrize("gscale", [0.01, 30.0])
def test_sgd_momentum_global_clipnorm(gscale):
    """keras SGD(momentum, clipnorm) with the GLOBAL norm [A10] vs oracle.sgd_step."""
    from oracle import graph_ref as G
    hip = _hip()
    torch.manual_seed(1)
    n = 1_000_003
    w = torch.randn(n); g = torch.randn(n) * gscale / np.sqrt(n); v = torch.randn(n) * 0.01
    P = {"l": {"w": w.clone()}}; vel = {"l": {"w": v.clone()}}
    norm = G.sgd_step(P, {"l": {"w": g}}, vel, 0.05, 0.9, 5.0)
    wd_, gd, vd = dev(w), dev(g), dev(v)
    ws = torch.empty(hip.sqnorm_ws_bytes(n) // 4, device="cuda"); nsq = torch.empty(1, device="cuda")
    hyper = torch.tensor([0.05, 0.9, 5.0], device="cuda")
    hip.sqnorm(n, gd, ws, nsq)
    hip.sgd_momentum_clip(n, wd_, gd, vd, hyper, nsq)
    torch.cuda.synchronize()
    assert abs(float(nsq) ** 0.5 - norm) < 1e-4 * norm
    assert (norm >= 5.0) == (gscale > 1)
    assert relerr(wd_, P["l"]["w"]) < 1e-6 and relerr(vd, vel["l"]["w"]) < 1e-5


def test_quat_weighted_average_decode_against_golden():
    """GPU soft-argmax decode vs the reference's se3lib.quat_weighted_avg outputs (golden)."""
    import os
    hip = _hip()
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "ori_codec.npz"))
    for n in (8, 16):
        Hq = gold["Hquat_%d" % n]; logits = gold["logits_%d" % n]; qref = gold["wavg_q_%d" % n][4:]
        Aref = gold["wavg_A_%d" % n][4:]
        B, K = logits.shape
        q = torch.empty(B, 4, device="cuda"); A = torch.empty(B, 16, device="cuda")
        hip.quat_wavg_decode(B, K, torch.tensor(logits).cuda(), torch.tensor(Hq).cuda(), q, A)
        torch.cuda.synchronize()
        dots = np.abs((q.cpu().numpy() * qref).sum(-1))
        assert dots.min() > 1 - 1e-5, dots
        assert np.abs(A.cpu().numpy().reshape(B, 4, 4) - Aref).max() < 2e-5


@pytest.mark.parametrize("dt", [0, 1])
def test_batched_param_phases_equal_per_layer_entry_points(dt):
    """urso_param_batch_run (prep / split reduction / finalisation over several layers in ONE launch each) is
    bit-identical to the per-layer entry points, including a layer whose wgrad is not split."""
    hip = _hip()
    torch.manual_seed(5)
    tdt = hip.TORCH_DT[dt]
    wd_ = 1e-3
    layers = [dict(KH=3, KW=3, C=32, N=40, npad=40, B=4, H=48, W=40, bn=True, bias=False),
              dict(KH=1, KW=1, C=64, N=24, npad=24, B=2, H=40, W=36, bn=True, bias=True),
              dict(KH=1, KW=1, C=16, N=13, npad=16, B=3, H=1, W=1, bn=False, bias=True)]         # dense head, single split
    descs, keep, ref = [], [], []
    for L in layers:
        KH, KW, Ci, N, NP = L["KH"], L["KW"], L["C"], L["N"], L["npad"]
        K = KH * KW * Ci
        w = dev(torch.randn(K, N)); b = dev(torch.randn(N)) if L["bias"] else None
        bn = [dev(t) for t in (torch.rand(N) + 0.5, torch.randn(N), torch.randn(N), torch.rand(N) + 0.5)] if L["bn"] else [None] * 4
        g = hip.geom(L["B"], L["H"], L["W"], Ci, L["H"], L["W"], NP, KH, KW, 1, 1, KH // 2, KW // 2)
        x = dev(torch.randn(L["B"], L["H"], L["W"], Ci)).to(tdt)
        dz = dev(torch.randn(L["B"], L["H"], L["W"], NP)).to(tdt)
        if NP > N:
            dz[..., N:] = 0
        # ---- per-layer reference path
        wf = torch.empty(NP * K, dtype=tdt, device="cuda"); wdl = torch.empty(NP * K, dtype=tdt, device="cuda")
        biasf = torch.empty(NP, device="cuda"); scale = torch.empty(NP, device="cuda")
        hip.conv_weight_prep(KH, KW, Ci, N, NP, dt, w, b, bn[0], bn[1], bn[2], bn[3], 1e-3, wf, wdl, biasf, scale)
        ws = torch.empty(hip.conv_wgrad_ws_bytes(g, dt) // 4 + 64, device="cuda")
        dwr = torch.empty(K * NP, device="cuda"); cs = torch.empty(NP, device="cuda")
        hip.conv_wgrad(g, dt, x, dz, ws, dwr, cs)
        gw = torch.empty(K * N, device="cuda"); gb = torch.empty(N, device="cuda") if L["bias"] else None
        gg = torch.empty(N, device="cuda") if L["bn"] else None; gbe = torch.empty(N, device="cuda") if L["bn"] else None
        fws = torch.empty(hip.param_grad_finalize_ws_bytes(K, N) // 4 + 64, device="cuda")
        hip.param_grad_finalize(K, N, NP, dwr, cs, w, b, bn[0], bn[2], bn[3], 1e-3, wd_, 1, 1, gw, gb, gg, gbe, fws)
        ref.append((wf, wdl, biasf, scale, gw, gb, gg, gbe))
        # ---- batched path
        d = hip.ParamDesc()
        splits = hip.conv_wgrad_splits(g, dt)
        hip.param_desc_init(d, KH, KW, Ci, N, NP, splits, 1e-3, wd_)
        t = dict(wf=torch.empty_like(wf), wd=torch.empty_like(wdl), biasf=torch.empty_like(biasf), scale=torch.empty_like(scale),
                 ws=torch.empty_like(ws), dw_raw=torch.empty_like(dwr), colsum=torch.empty_like(cs),
                 dotpart=torch.empty(d.ks * N + 16, device="cuda"), gw=torch.empty_like(gw),
                 gb=torch.empty_like(gb) if gb is not None else None, gg=torch.empty_like(gg) if gg is not None else None,
                 gbe=torch.empty_like(gbe) if gbe is not None else None, x=x, dz=dz, g=g, w=w, b=b, bn=bn)
        keep.append(t)
        for f, v in (("w", w), ("b", b), ("gamma", bn[0]), ("beta", bn[1]), ("mean", bn[2]), ("var", bn[3]), ("wf", t["wf"]),
                     ("wd", t["wd"]), ("biasf", t["biasf"]), ("scale", t["scale"]), ("dw_raw", t["dw_raw"]), ("colsum", t["colsum"]),
                     ("dotpart", t["dotpart"]), ("gw", t["gw"]), ("gb", t["gb"]), ("ggamma", t["gg"]), ("gbeta", t["gbe"])):
            setattr(d, f, hip.ptr(v))
        d.part = t["ws"].data_ptr(); d.colpart = t["ws"].data_ptr() + 4 * splits * (K * NP + hip.WGRAD_PART_PAD)
        d.trainable, d.bn_trainable = 1, 1
        descs.append(d)
    assert descs[0].splits > 1 and descs[2].splits == 1
    pb = hip.ParamBatch(descs, torch.device("cuda"))
    ids = list(range(len(descs)))
    for ph in (hip.PB_PREP, hip.PB_REDUCE, hip.PB_FINALIZE_MAT, hip.PB_FINALIZE_VEC):
        nb = pb.plan(ph, "t", ids)
        # a layer with 2 .. 16 split partials has no reduction blocks: the finalisation sums its partials itself, in the reduction's order
        # (the bit-equality with urso_conv_wgrad + urso_param_grad_finalize below is that claim's test)
        assert (nb > 0) if (ph != hip.PB_REDUCE or any(d.splits > 16 for d in descs)) else (nb == 0), (ph, nb, [d.splits for d in descs])
    pb.run(hip.PB_PREP, "t", dt)
    for t in keep:
        hip.conv_wgrad_partial(t["g"], dt, t["x"], t["dz"], t["ws"])
    for ph in (hip.PB_REDUCE, hip.PB_FINALIZE_MAT, hip.PB_FINALIZE_VEC):
        pb.run(ph, "t", dt)
    torch.cuda.synchronize()
    for t, r in zip(keep, ref):
        for got, exp in zip((t["wf"], t["wd"], t["biasf"], t["scale"], t["gw"], t["gb"], t["gg"], t["gbe"]), r):
            if exp is not None:
                assert torch.equal(got, exp)


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("shape", [(2, 20, 24, 64, 128, 1, 1, 0), (2, 17, 19, 32, 40, 3, 1, 0), (3, 16, 16, 64, 256, 1, 2, 0),
                                   (2, 32, 40, 64, 256, 1, 1, 8), (2, 32, 48, 256, 128, 1, 2, 8), (4, 128, 160, 64, 256, 1, 1, 0)])
def test_relu_bit_masks_emit_and_consume(dt, shape):
    """urso_conv_igemm_ex: (a) the forward epilogue's bit mask == (stored output > 0) bit for bit; (b) a data-gradient
    pass masked by the bit array is identical to one masked by the activation tensor itself (incl. the scattered
    stride-2 1x1 form and a residual add).  The last three shapes run the multi-tile stream of the persistent kernel (grid
    capped to 8 blocks, or more tiles than resident blocks)."""
    hip = _hip()
    with hip.options(grid_cap=shape[-1]):
        _bitmask_case(hip, dt, shape[:-1])


def _bitmask_case(hip, dt, shape):
    B, H, W, Ci, N, k, s = shape
    torch.manual_seed(sum(shape) + dt)
    tdt = hip.TORCH_DT[dt]
    pad = k // 2
    OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = dev(torch.randn(B, H, W, Ci), dt)
    w = torch.randn(k, k, Ci, N) / (k * k * Ci) ** 0.5
    wf, wd, biasf, _ = prep_weights(w, dt, bias=torch.randn(N) * 0.1)
    res = dev(torch.randn(B, OH, OW, N), dt)
    g = hip.geom(B, H, W, Ci, OH, OW, N, k, k, s, s, pad, pad)
    assert hip.conv_igemm_bits_ok(g, dt, hip.EPI_RELU)
    y = torch.empty(B, OH, OW, N, dtype=tdt, device="cuda")
    bits = torch.full((B * OH * OW * N // 8,), 0xAA, dtype=torch.uint8, device="cuda")
    hip.conv_igemm_ex(g, dt, hip.EPI_RELU | hip.EPI_EMIT_BITS, x, wf, biasf, res, None, y, bits)
    y0 = torch.empty_like(y)
    hip.conv_igemm(g, dt, hip.EPI_RELU, x, wf, biasf, res, None, y0)
    torch.cuda.synchronize()
    assert torch.equal(y, y0)
    pos = (y.float() > 0).reshape(-1, 8).to(torch.int32)
    exp = (pos << torch.arange(8, device="cuda", dtype=torch.int32)).sum(1).to(torch.uint8)
    assert torch.equal(bits, exp)
    assert 0.2 < float(pos.float().mean()) < 0.8
    # ---- consume: gradient w.r.t. an input tensor X that is itself post-ReLU
    xr = torch.relu(x.float()).to(tdt)                      # the forward input as a post-ReLU activation
    xpos = (xr.float() > 0).reshape(-1, 8).to(torch.int32)
    xbits = (xpos << torch.arange(8, device="cuda", dtype=torch.int32)).sum(1).to(torch.uint8)
    dy = dev(torch.randn(B, OH, OW, N), dt)
    add = dev(torch.randn(B, H, W, Ci), dt)
    if k == 1 and s > 1:
        gd = hip.geom(B, OH, OW, N, OH, OW, Ci, 1, 1, FH=H, FW=W, OSH=s, OSW=s)
        d_a = add.clone(); d_b = add.clone()
        hip.conv_igemm_ex(gd, dt, 0, dy, wd, None, d_a, xr, d_a)
        hip.conv_igemm_ex(gd, dt, hip.EPI_MASK_BITS, dy, wd, None, d_b, xbits, d_b)
    else:
        gd = hip.geom(B, OH, OW, N, H, W, Ci, k, k, 1, 1, k - 1 - pad, k - 1 - pad, s, s)
        d_a = torch.empty(B, H, W, Ci, dtype=tdt, device="cuda"); d_b = torch.empty_like(d_a)
        hip.conv_igemm_ex(gd, dt, 0, dy, wd, None, add, xr, d_a)
        hip.conv_igemm_ex(gd, dt, hip.EPI_MASK_BITS, dy, wd, None, add, xbits, d_b)
    assert hip.conv_igemm_bits_ok(gd, dt, 0)
    torch.cuda.synchronize()
    assert torch.equal(d_a, d_b)
    assert float((d_b.float() != 0).float().mean()) > 0.1


def test_bit_masks_refused_where_unsupported():
    hip = _hip()
    g = hip.geom(2, 8, 8, 16, 8, 8, 16, 1, 1)
    assert not hip.conv_igemm_bits_ok(g, 0, hip.EPI_RELU)                          # fp32
    assert not hip.conv_igemm_bits_ok(g, 1, hip.EPI_RELU | hip.EPI_OUT_F32)
    x = torch.randn(2, 8, 8, 16, device="cuda"); w = torch.randn(16, 16, device="cuda"); y = torch.empty_like(x)
    bits = torch.zeros(2 * 8 * 8 * 2, dtype=torch.uint8, device="cuda")
    with pytest.raises(hip.UrsoHipError):
        hip.conv_igemm_ex(g, 0, hip.EPI_RELU | hip.EPI_EMIT_BITS, x, w, None, None, None, y, bits)


def test_adam_amsgrad_clip_against_numpy():
    """urso_adam_amsgrad_clip over three steps (device-side step counter) vs a float64 NumPy restatement of
    Keras' Adam(amsgrad=True, clipnorm) update (net.py:982-983)."""
    hip = _hip()
    rng = np.random.default_rng(5)
    n = 10007
    w = rng.normal(size=n).astype(np.float32)
    lr, b1, b2, eps, clip = 1e-3, 0.9, 0.999, 1e-7, 5.0
    W = dev(torch.tensor(w)); M = torch.zeros(n, device="cuda"); V = torch.zeros(n, device="cuda"); VH = torch.zeros(n, device="cuda")
    hyper = torch.tensor([lr, b1, b2, eps, clip, 0.0, 1.0 - b1, 1.0 - b2], device="cuda")
    ws = torch.empty(hip.sqnorm_ws_bytes(n) // 4 + 4, device="cuda"); nsq = torch.zeros(1, device="cuda")
    wr = w.astype(np.float64); m = np.zeros(n); v = np.zeros(n); vh = np.zeros(n)
    for t in range(1, 4):
        g = (rng.normal(size=n) * (3.0 if t == 2 else 0.01)).astype(np.float32)     # step 2 is clipped
        G = dev(torch.tensor(g))
        hip.sqnorm(n, G, ws, nsq)
        hip.adam_amsgrad_clip(n, W, G, M, V, VH, hyper, nsq)
        norm = np.sqrt((g.astype(np.float64) ** 2).sum())
        gc = g.astype(np.float64) * (clip / norm if norm >= clip else 1.0)
        lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        m = b1 * m + (1 - b1) * gc; v = b2 * v + (1 - b2) * gc * gc; vh = np.maximum(vh, v)
        wr = wr - lr_t * m / (np.sqrt(vh) + eps)
        torch.cuda.synchronize()
        assert float(hyper[5]) == t
        assert np.abs(W.cpu().numpy() - wr).max() < 2e-6
        assert np.abs(VH.cpu().numpy() - vh).max() <= 2e-6 * vh.max()


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("MN", [(600, 64), (130, 40), (96, 2048), (20000, 256), (5000, 24), (3000, 2048)])
def test_batch_stat_bn_kernels_against_autograd(dt, MN):
    """urso_bn_batch_stats / urso_bn_apply / urso_bn_backward (TRAIN_BN=None mode) vs torch autograd through a
    training-mode batch norm + residual + ReLU; moving statistics follow Keras' update (momentum 0.99, variance fed
    var*M/(M-(1+eps)))."""
    hip = _hip()
    M, N = MN
    if dt == 0 and N % 4 or dt == 1 and N % 8:
        pytest.skip("N not a vector multiple")
    torch.manual_seed(M + N + dt)
    tdt = hip.TORCH_DT[dt]
    z = (torch.randn(M, N) * 2 + torch.randn(N)).to(tdt)
    res = torch.randn(M, N).to(tdt)
    gamma, beta = torch.rand(N) + 0.5, torch.randn(N)
    mm, mv = torch.randn(N), torch.rand(N) + 0.5
    eps, mom = 1e-3, 0.99
    zf = z.float().clone().requires_grad_(True)
    gam = gamma.clone().requires_grad_(True); bet = beta.clone().requires_grad_(True)
    mu, var = zf.mean(0), zf.var(0, unbiased=False)
    y = torch.relu((zf - mu) * torch.rsqrt(var + eps) * gam + bet + res.float())
    gy = torch.randn(M, N).to(tdt).float()
    g_masked = gy * (y > 0)                                   # what the consuming layer's dgrad epilogue hands over
    (y * gy).sum().backward()
    Z, R = dev(z), dev(res)
    ws = torch.empty(hip.bn_ws_bytes(M, N) // 8 + 8, dtype=torch.float64, device="cuda")
    mean_d, var_d = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    MMd, MVd = dev(mm.clone()), dev(mv.clone())
    hip.bn_batch_stats(M, N, dt, Z, ws, mean_d, var_d, MMd, MVd, mom, eps)
    Y = torch.empty(M, N, dtype=tdt, device="cuda")
    hip.bn_apply(M, N, dt, Z, mean_d, var_d, dev(gamma), dev(beta), eps, R, 1, Y)
    dbeta, dgamma = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    gbeta, ggamma = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    DZ = torch.empty(M, N, dtype=tdt, device="cuda")
    hip.bn_backward(M, N, dt, dev(g_masked.to(tdt)), Z, mean_d, var_d, dev(gamma), eps, ws, dbeta, dgamma, 1, gbeta, ggamma, DZ)
    torch.cuda.synchronize()
    tol = 2e-5 if dt == 0 else 1.5e-2
    assert relerr(mean_d, mu) < 1e-5 and relerr(var_d, var) < 1e-5
    assert relerr(MMd, mm * mom + mu.detach() * (1 - mom)) < 1e-5
    assert relerr(MVd, mv * mom + var.detach() * (M / (M - (1 + eps))) * (1 - mom)) < 1e-5
    assert relerr(Y, y) < tol
    assert relerr(gbeta, bet.grad) < max(tol, 1e-4) and relerr(ggamma, gam.grad) < max(tol, 1e-4)
    assert relerr(DZ, zf.grad) < (1e-4 if dt == 0 else 3e-2)
    assert torch.equal(gbeta, dbeta) and torch.equal(ggamma, dgamma)


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("masked", [True, False], ids=["relu_bits", "plain"])
@pytest.mark.parametrize("shape", [(2, 32, 40, 1024, 256, 512, 0), (4, 16, 20, 2048, 512, 1024, 0), (2, 64, 80, 512, 128, 256, 0), (3, 20, 24, 256, 320, 128, 16),
                                   (1, 16, 20, 64, 512, 136, 0), (1, 16, 20, 64, 512, 160, 8)],
                         ids=["stage4_entry", "stage5_entry", "stage3_entry", "ragged_capped", "short_first_segment", "short_first_segment_n160"])
def test_pointwise2_two_reduction_segments(dt, masked, shape):
    """urso_conv_pointwise2 (conv_pwx.hip, SEG2): dst = mask(src0 . W0^T + src1 . W1^T) -- the data gradients of a stage's projection shortcut and
    branch2a into the stage's input as ONE launch (net.py:121-126, 138, 148-157) -- against a CPU fp32 reference rounded once, and against the
    two urso_conv_igemm_ex launches it replaces (those round the first product to 16 bits before the second is added: one more rounding
    step).  Shapes: the three entry blocks of ResNet-50 at reduced pixel counts, a ragged one with a capped grid (every block walks several
    tiles across the segment switch), and a first segment of ONE K-tile."""
    hip = _hip()
    B, OH, OW, C0, C1, N, cap = shape
    M = B * OH * OW
    tdt = hip.TORCH_DT[dt]
    torch.manual_seed(M + C0 + dt)
    flags = hip.EPI_MASK_BITS if masked else 0
    with hip.options(pwx=2, grid_cap=cap):
        if masked and N % 32:                               # the bit mask is read a dword (32 channels) at a time
            assert not hip.conv_pointwise2_ok(B, OH, OW, C0, C1, N, dt, flags)
            return
        assert hip.conv_pointwise2_ok(B, OH, OW, C0, C1, N, dt, flags)
        assert not hip.conv_pointwise2_ok(B, OH, OW, C0 + 8, C1, N, dt, flags) and not hip.conv_pointwise2_ok(B, OH, OW, C0, C1, N, 0, flags)
        x0, x1 = dev(torch.randn(M, C0), dt), dev(torch.randn(M, C1), dt)
        w0, w1 = dev(torch.randn(N, C0) / C0 ** 0.5, dt), dev(torch.randn(N, C1) / C1 ** 0.5, dt)
        keep = torch.rand(M, N, device="cuda") > 0.4
        bits = (keep.reshape(-1, 8).to(torch.int32) << torch.arange(8, device="cuda", dtype=torch.int32)).sum(1).to(torch.uint8) if masked else None
        dst = torch.full((M, N), 9.0, device="cuda").to(tdt)
        hip.conv_pointwise2(B, OH, OW, C0, C1, N, dt, flags, x0, w0, x1, w1, None, bits, dst)
        torch.cuda.synchronize()
        f = lambda t: t.float().cpu()
        ref = f(x0) @ f(w0).T + f(x1) @ f(w1).T
        if masked:
            ref = ref * keep.cpu()
        ref = ref.to(tdt).float()
        tol = 8e-3 if dt == 1 else 1e-3                     # one rounding step of values of a few units
        assert float((f(dst) - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
        if masked:
            assert float(f(dst)[~keep.cpu()].abs().max()) == 0.0
        # the two launches it replaces: first product stored (rounded), then added to the second
        g0, g1 = hip.geom(B, OH, OW, C0, OH, OW, N, 1, 1), hip.geom(B, OH, OW, C1, OH, OW, N, 1, 1)
        two = torch.empty_like(dst)
        hip.conv_igemm_ex(g0, dt, flags, x0, w0, None, None, bits, two, None)
        hip.conv_igemm_ex(g1, dt, flags, x1, w1, None, two, bits, two, None)
        torch.cuda.synchronize()
        assert float((dst.float() - two.float()).abs().max()) <= 2 * tol * max(1.0, float(two.float().abs().max()))
        # deterministic: a second launch writes the same bits
        again = torch.empty_like(dst)
        hip.conv_pointwise2(B, OH, OW, C0, C1, N, dt, flags, x0, w0, x1, w1, None, bits, again)
        torch.cuda.synchronize()
        assert torch.equal(dst, again)


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("emit", [True, False], ids=["emit_bits", "no_bits"])
@pytest.mark.parametrize("shape", [(2, 32, 40, 256, 512, 1024, 0), (4, 16, 20, 512, 1024, 2048, 0), (3, 20, 24, 128, 256, 512, 16)],
                         ids=["stage4_entry", "stage5_entry", "ragged_capped"])
def test_pointwise2_forward_form_shortcut_inside(dt, emit, shape):
    """urso_conv_pointwise2, forward form: out = ReLU(b . W2c^T + x . Wbr1^T + bias) with the emitted ReLU bit mask -- branch2c of a stage's
    first block with the projection shortcut as a second reduction segment (net.py:148-157) -- against a CPU fp32 reference rounded once, and
    against the two urso_conv_igemm_ex launches it replaces (shortcut stored in 16 bits, then added: one more rounding step)."""
    hip = _hip()
    B, OH, OW, C0, C1, N, cap = shape
    M = B * OH * OW
    tdt = hip.TORCH_DT[dt]
    torch.manual_seed(M + C0 + dt)
    flags = hip.EPI_RELU | (hip.EPI_EMIT_BITS if emit else 0)
    with hip.options(pwx=2, grid_cap=cap):
        assert hip.conv_pointwise2_ok(B, OH, OW, C0, C1, N, dt, flags)
        x0, x1 = dev(torch.randn(M, C0), dt), dev(torch.randn(M, C1), dt)
        w0, w1 = dev(torch.randn(N, C0) / C0 ** 0.5, dt), dev(torch.randn(N, C1) / C1 ** 0.5, dt)
        b0, b1 = torch.randn(N, device="cuda") * 0.3, torch.randn(N, device="cuda") * 0.3
        dst = torch.full((M, N), 9.0, device="cuda").to(tdt)
        bits = torch.full((M, N // 8), 0xAA, dtype=torch.uint8, device="cuda") if emit else None
        hip.conv_pointwise2(B, OH, OW, C0, C1, N, dt, flags, x0, w0, x1, w1, b0 + b1, None, dst, bits)
        torch.cuda.synchronize()
        f = lambda t: t.float().cpu()
        ref = torch.relu(f(x0) @ f(w0).T + f(x1) @ f(w1).T + (b0 + b1).cpu()).to(tdt).float()
        tol = 8e-3 if dt == 1 else 1e-3
        assert float((f(dst) - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
        if emit:
            pos = (dst.float() > 0).reshape(-1, 8).to(torch.int32)
            exp = (pos << torch.arange(8, device="cuda", dtype=torch.int32)).sum(1).to(torch.uint8)
            assert torch.equal(bits.reshape(-1), exp) and 0.2 < float(pos.float().mean()) < 0.8
        g0, g1 = hip.geom(B, OH, OW, C0, OH, OW, N, 1, 1), hip.geom(B, OH, OW, C1, OH, OW, N, 1, 1)
        sc, two = torch.empty_like(dst), torch.empty_like(dst)
        hip.conv_igemm_ex(g1, dt, 0, x1, w1, b1, None, None, sc, None)                       # the shortcut, stored
        hip.conv_igemm_ex(g0, dt, hip.EPI_RELU, x0, w0, b0, sc, None, two, None)              # branch2c + shortcut + ReLU
        torch.cuda.synchronize()
        assert float((dst.float() - two.float()).abs().max()) <= 2 * tol * max(1.0, float(two.float().abs().max()))


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("c", [64, 128], ids=["stage2", "stage3"])
@pytest.mark.parametrize("shape", [(1, 8, 8, 0), (3, 24, 40, 0), (2, 64, 80, 8), (4, 128, 160, 0), (32, 32, 40, 0)],
                         ids=["one_tile", "small", "capped", "multi_tile", "chip"])
def test_fused_pointwise_pair_forward_and_backward(dt, shape, c):
    """urso_conv_pair (conv_pair.hip) against a CPU fp32 reference with the same two rounding points (the 256-channel tensor and the
    64-channel output are each rounded once to the storage dtype), forward form (bias + residual + ReLU, then bias + ReLU, with the
    emitted ReLU bit mask) and backward form (residual-gradient add + bit mask, then activation mask); and against the two
    urso_conv_igemm_ex launches it replaces.  'capped' / 'multi_tile' make every block walk several tiles, 'chip' gives every resident block of the chip a few; c = 64 / 128 are the
    stage-2 (4 waves, 2 LDS stages) and stage-3 (8 waves, 3 LDS stages) shapes of the kernel."""
    hip = _hip()
    B, H, W, cap = shape
    M = B * H * W
    tdt = hip.TORCH_DT[dt]
    torch.manual_seed(M + dt)
    assert hip.conv_pair_ok(M, dt, c, 4 * c) and not hip.conv_pair_ok(M + 1, dt, c, 4 * c) and not hip.conv_pair_ok(M, 0, c, 4 * c)
    assert not hip.conv_pair_ok(M, dt, 256, 1024) and not hip.conv_pair_ok(M, dt, c, 2 * c)
    src, add, act = dev(torch.randn(M, c), dt), dev(torch.randn(M, 4 * c), dt), dev(torch.randn(M, c), dt)
    w1, w2 = dev(torch.randn(4 * c, c) / c ** 0.5, dt), dev(torch.randn(c, 4 * c) / (2 * c ** 0.5), dt)
    b1, b2 = torch.randn(4 * c, device="cuda") * 0.3, torch.randn(c, device="cuda") * 0.3
    rnd = lambda t: t.to(tdt).float()
    f = lambda t: t.float().cpu()
    tol = 1.2e-2 if dt == 1 else 1.5e-3                     # one output rounding step on values of a few units
    # ---- forward
    mid = torch.full((M, 4 * c), 9.0, device="cuda").to(tdt); dst = torch.full((M, c), 9.0, device="cuda").to(tdt)
    bits = torch.full((M, c // 2), 0xAA, dtype=torch.uint8, device="cuda")
    with hip.options(grid_cap=cap):
        hip.conv_pair(M, c, dt, 0, src, w1, b1, add, bits, mid, w2, b2, None, dst)
        mid_nb = torch.empty_like(mid); dst_nb = torch.empty_like(dst)
        hip.conv_pair(M, c, dt, 0, src, w1, b1, add, None, mid_nb, w2, b2, None, dst_nb)      # variant without the bit mask
    torch.cuda.synchronize()
    assert torch.equal(mid, mid_nb) and torch.equal(dst, dst_nb)
    ref_mid = rnd(torch.relu(f(src) @ f(w1).T + b1.cpu() + f(add)))
    assert float((f(mid) - ref_mid).abs().max()) <= tol * max(1.0, float(ref_mid.abs().max()))
    ref_dst = rnd(torch.relu(f(mid) @ f(w2).T + b2.cpu()))                               # second layer from the STORED first output
    assert float((f(dst) - ref_dst).abs().max()) <= tol * max(1.0, float(ref_dst.abs().max()))
    pos = (mid.float() > 0).reshape(-1, 8).to(torch.int32)
    exp = (pos << torch.arange(8, device="cuda", dtype=torch.int32)).sum(1).to(torch.uint8)
    assert torch.equal(bits.reshape(-1), exp) and 0.2 < float(pos.float().mean()) < 0.8
    g1, g2 = hip.geom(B, H, W, c, H, W, 4 * c, 1, 1), hip.geom(B, H, W, 4 * c, H, W, c, 1, 1)
    m2, d2, bits2 = torch.empty_like(mid), torch.empty_like(dst), torch.empty_like(bits)
    hip.conv_igemm_ex(g1, dt, hip.EPI_RELU | hip.EPI_EMIT_BITS, src, w1, b1, add, None, m2, bits2)
    hip.conv_igemm_ex(g2, dt, hip.EPI_RELU, m2, w2, b2, None, None, d2)
    torch.cuda.synchronize()
    assert float((mid.float() - m2.float()).abs().max()) <= tol * float(m2.float().abs().max())
    assert float((dst.float() - d2.float()).abs().max()) <= 2 * tol * float(d2.float().abs().max())
    # ---- backward (no bias: bit-identical to the separate launches)
    gbits = torch.randint(0, 256, (M, c // 2), dtype=torch.uint8, device="cuda")
    with hip.options(grid_cap=cap):
        hip.conv_pair(M, c, dt, 1, src, w1, None, add, gbits, mid, w2, None, act, dst)
    hip.conv_igemm_ex(g1, dt, hip.EPI_MASK_BITS, src, w1, None, add, gbits, m2)
    hip.conv_igemm_ex(g2, dt, 0, m2, w2, None, None, act, d2)
    torch.cuda.synchronize()
    assert torch.equal(mid, m2) and torch.equal(dst, d2)
    keep = ((gbits.cpu().to(torch.int32).reshape(-1, 1) >> torch.arange(8, dtype=torch.int32)) & 1).reshape(M, 4 * c).float()
    ref_mid = rnd((f(src) @ f(w1).T + f(add)) * keep)
    assert float((f(mid) - ref_mid).abs().max()) <= tol * max(1.0, float(ref_mid.abs().max()))
    ref_dst = rnd((f(mid) @ f(w2).T) * (f(act) > 0))
    assert float((f(dst) - ref_dst).abs().max()) <= tol * max(1.0, float(ref_dst.abs().max()))
    assert float((dst.float() != 0).float().mean()) > 0.1
    with pytest.raises(hip.UrsoHipError):
        hip.conv_pair(M, c, dt, 1, src, w1, None, add, None, mid, w2, None, act, dst)          # backward form needs the bit mask


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("c", [64, 128, 256, 512], ids=["stage2", "stage3", "stage4", "stage5"])
@pytest.mark.parametrize("variant", ["add_relu_bits", "add_relu", "plain", "relu_only", "add_maskbits"])
def test_wide_pointwise_layers_take_the_register_filter_kernel(dt, c, variant):
    """urso_conv_igemm_ex on a c -> 4c pointwise layer (res{2c,3d,4x}_branch2c, the stride-1 shortcut conv; stage 4 also the masked data
    gradient of branch2a: 'add_maskbits') runs the single-layer form of conv_pair.hip (option pair, default on): against the CPU fp32
    reference, against the DMA kernel (pair = 0) and, for an emitted bit mask, bit for bit against (stored output > 0); grid at
    production size and capped to 8 blocks (multi-tile stream; stage 4 / 5: two / eight block groups of 512 / 256 filters)."""
    hip = _hip()
    if variant == "add_maskbits" and c < 256:
        pytest.skip("stages 2-3 run that layer inside the fused backward pair")
    B, H, W = (3, 40, 48) if c < 512 else (2, 16, 24)
    M = B * H * W
    tdt = hip.TORCH_DT[dt]
    torch.manual_seed(c + dt)
    x = dev(torch.randn(B, H, W, c), dt)
    w = torch.randn(1, 1, c, 4 * c) / c ** 0.5
    mb = variant == "add_maskbits"
    wf, _, biasf, _ = prep_weights(w, dt, bias=torch.randn(4 * c) * 0.2)
    if mb:
        biasf = None
    add = dev(torch.randn(B, H, W, 4 * c), dt) if variant.startswith("add") else None
    relu = variant not in ("plain", "add_maskbits")
    bits = torch.full((M * 4 * c // 8,), 0x55, dtype=torch.uint8, device="cuda") if variant.endswith("_bits") else None
    mask = torch.randint(0, 256, (M * 4 * c // 8,), dtype=torch.uint8, device="cuda") if mb else None
    flags = (hip.EPI_RELU if relu else 0) | (hip.EPI_EMIT_BITS if bits is not None else 0) | (hip.EPI_MASK_BITS if mb else 0)
    g = hip.geom(B, H, W, c, H, W, 4 * c, 1, 1)
    ref = x.float().cpu().reshape(M, c) @ wf.float().cpu().reshape(4 * c, c).T
    if biasf is not None:
        ref = ref + biasf.cpu()
    if add is not None:
        ref = ref + add.float().cpu().reshape(M, 4 * c)
    if relu:
        ref = torch.relu(ref)
    if mb:
        keep = ((mask.cpu().to(torch.int32).reshape(-1, 1) >> torch.arange(8, dtype=torch.int32)) & 1).reshape(M, 4 * c).float()
        ref = ref * keep
    outs = {}
    for pair, cap in ((1, 0), (1, 8), (0, 0)):
        y = torch.full((B, H, W, 4 * c), 5.0, device="cuda").to(tdt)
        if bits is not None:
            bits.fill_(0x55)
        with hip.options(pair=pair, grid_cap=cap):
            hip.conv_igemm_ex(g, dt, flags, x, wf, biasf, add, mask, y, bits)
        torch.cuda.synchronize()
        assert relerr(y.reshape(M, 4 * c), ref) < (1.2e-2 if dt == 1 else 1.5e-3)
        if bits is not None:
            pos = (y.float() > 0).reshape(-1, 8).to(torch.int32)
            assert torch.equal(bits, (pos << torch.arange(8, device="cuda", dtype=torch.int32)).sum(1).to(torch.uint8))
        outs[(pair, cap)] = y.float()
    assert torch.equal(outs[(1, 0)], outs[(1, 8)])
    assert float((outs[(1, 0)] - outs[(0, 0)]).abs().max()) <= (1.6e-2 if dt == 1 else 2e-3) * float(outs[(0, 0)].abs().max())
    # the same layer with the sampled second output (urso_conv_pointwise_sampled): dst and bit mask unchanged bit for bit, the second
    # tensor = the even rows / columns of dst; stages 2-4 (the 512-channel shape and the masked form do not offer it)
    if not mb and c < 512:
        for cap in (0, 8):
            y2 = torch.full((B, H, W, 4 * c), 5.0, device="cuda").to(tdt); ys = torch.full((B, H // 2, W // 2, 4 * c), 5.0, device="cuda").to(tdt)
            bits2 = torch.full_like(bits, 0x55) if bits is not None else None
            with hip.options(grid_cap=cap):
                assert hip.conv_pointwise_sampled_ok(g, dt, flags, add is not None)
                hip.conv_pointwise_sampled(g, dt, flags, x, wf, biasf, add, y2, bits2, ys)
            torch.cuda.synchronize()
            assert torch.equal(y2.float(), outs[(1, 0)]) and torch.equal(ys, y2[:, ::2, ::2]) and (bits is None or torch.equal(bits2, bits))
    else:
        assert not hip.conv_pointwise_sampled_ok(g, dt, flags, add is not None)


C3W_CASES = [
    (1, 4, 32, 128, 128, 3, 1, (1, 1), "c3w_one_tile"),
    (3, 17, 45, 128, 128, 3, 1, (1, 1), "c3w_ragged"),
    (2, 64, 96, 128, 128, 3, 1, (1, 1), "c3w_multi_tile"),
    (8, 64, 80, 128, 128, 3, 1, (1, 1), "c3w_stage3_rows"),
    (2, 21, 50, 128, 128, 3, 1, (1, 1), "c3w_ragged_16"),          # 8 x 16 tiles with partial rows and columns on both borders
]


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("cap", [0, 8])
@pytest.mark.parametrize("c3v", [1, 0], ids=["wave16", "halves"])
@pytest.mark.parametrize("case", C3W_CASES, ids=[c[-1] for c in C3W_CASES])
def test_3x3_128_channel_layers_register_filter_kernel(case, dt, cap, c3v):
    """The 128-channel forms of conv_c3.hip, forced on every shape (option c3 = 3: the default policy takes them only where their tiles fit
    the image width): forward, data gradient with mask and weight gradient against the CPU fp32 reference; ragged sizes, capped grid.
    'halves' = c3w_kernel (8 waves, each 32 filters x one 64-channel half of the filter in registers, the two halves' sums exchanged through
    LDS; the only form on 4 x 32 tiles); 'wave16' (option c3v, default) = c3v_kernel on the 8 x 16 tiles (c3w_ragged, c3w_stage3_rows:
    every wave 16 filters over the whole reduction, no exchange)."""
    hip = _hip()
    with hip.options(c3=3, c3v=c3v, grid_cap=cap):
        test_conv_forward_and_gradients(case, dt)


C3_CASES = [
    (1, 4, 32, 64, 64, 3, 1, (1, 1), "c3_one_tile"),
    (2, 12, 20, 64, 64, 3, 1, (1, 1), "c3_narrow_image"),        # W < tile width: partial tiles, right border inside the patch
    (3, 17, 45, 64, 64, 3, 1, (1, 1), "c3_ragged"),              # H % 4 != 0, W % 32 != 0
    (2, 64, 96, 64, 64, 3, 1, (1, 1), "c3_multi_tile"),
    (4, 256, 320, 64, 64, 3, 1, (1, 1), "c3_full_size_rows"),    # more tiles than resident blocks at production grid size
]


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("cap", [0, 8])
@pytest.mark.parametrize("c3", [1, 0], ids=["regfilter", "dma"])
@pytest.mark.parametrize("case", C3_CASES, ids=[c[-1] for c in C3_CASES])
def test_3x3_64_channel_layers_register_filter_kernel(case, dt, cap, c3):
    """The 64-channel 3x3 layers (res2x_branch2b, forward and -- through the flipped filter -- data gradient with its ReLU mask) in
    conv_c3.hip (option c3, default on) and, for reference, in the DMA kernel: forward, data gradient and weight gradient against the
    CPU fp32 reference of test_conv_forward_and_gradients; image sizes that are not multiples of the 4 x 32 tile (zero-filled halo,
    dropped out-of-image stores) and a grid capped to 8 blocks (double-buffered halo stream across several tiles)."""
    hip = _hip()
    with hip.options(c3=c3, grid_cap=cap):
        test_conv_forward_and_gradients(case, dt)


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("c", [64, 128], ids=["stage2", "stage3"])
@pytest.mark.parametrize("shape", [(2, 8, 16, 0), (4, 8, 24, 0), (4, 64, 80, 8), (8, 64, 80, 0)], ids=["small", "rows_straddle_tiles", "capped", "multi_tile"])
def test_backward_pair_with_compact_add_operand(dt, c, shape):
    """urso_conv_pair mode 1 with the residual gradient given in COMPACT form (only the even rows / columns of the pixel grid, the
    pixels a stride-2 stage entry samples) must equal, bit for bit, the same call on the dense tensor with explicit zeros; image
    widths that do not divide the 64- / 32-pixel tiles make tiles straddle image rows.  urso_rows_subsample2 against indexing."""
    hip = _hip()
    B, H, W, cap = shape
    M = B * H * W
    tdt = hip.TORCH_DT[dt]
    torch.manual_seed(M + c)
    src, act = dev(torch.randn(M, c), dt), dev(torch.randn(M, c), dt)
    w1, w2 = dev(torch.randn(4 * c, c) / c ** 0.5, dt), dev(torch.randn(c, 4 * c) / (2 * c ** 0.5), dt)
    compact = dev(torch.randn(B, H // 2, W // 2, 4 * c), dt)
    dense = torch.zeros(B, H, W, 4 * c, dtype=tdt, device="cuda")
    dense[:, ::2, ::2] = compact
    bits = torch.randint(0, 256, (M, c // 2), dtype=torch.uint8, device="cuda")
    outs = []
    for add, hw in ((dense, None), (compact, (H, W))):
        mid = torch.full((M, 4 * c), 3.0, device="cuda").to(tdt); dst = torch.full((M, c), 3.0, device="cuda").to(tdt)
        with hip.options(grid_cap=cap):
            hip.conv_pair(M, c, dt, 1, src, w1, None, add, bits, mid, w2, None, act, dst, add_hw=hw)
        torch.cuda.synchronize()
        outs.append((mid, dst))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float((outs[1][0].float() != 0).float().mean()) > 0.2
    with pytest.raises(hip.UrsoHipError):
        hip.conv_pair(M, c, dt, 0, src, w1, None, compact, None, mid, w2, None, None, dst, add_hw=(H, W))      # forward form has no compact operand
    sub = torch.empty(B, H // 2, W // 2, c // 2, dtype=torch.uint8, device="cuda")
    hip.rows_subsample2(B, H, W, c // 2, bits, sub)
    torch.cuda.synchronize()
    assert torch.equal(sub, bits.reshape(B, H, W, c // 2)[:, ::2, ::2])


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("shape", [(1, 8, 8, 0), (3, 24, 40, 0), (2, 64, 80, 8), (4, 128, 160, 0), (32, 32, 40, 0)],
                         ids=["one_tile", "small", "capped", "multi_tile", "chip"])
def test_forward_pair_with_the_projection_shortcut_inside(dt, shape):
    """urso_conv_pair_shortcut (conv_pairs.hip): mid = relu(src W1^T + b1 + xin Ws^T + bs), dst = relu(mid W2^T + b2) against the CPU fp32
    reference (mid rounded once; dst from the STORED mid), the emitted ReLU bit mask bit for bit against (stored mid > 0), the variant
    without a bit mask, and against the two launches it replaces (shortcut conv, then urso_conv_pair), which round the shortcut's
    output once more.  'capped' / 'multi_tile' / 'chip': several tiles per block through the three-stage input pipeline."""
    hip = _hip()
    B, H, W, cap = shape
    M, c = B * H * W, 64
    tdt = hip.TORCH_DT[dt]
    torch.manual_seed(M + dt + 3)
    src, xin = dev(torch.randn(M, c), dt), dev(torch.randn(M, c), dt)
    w1, ws, w2 = dev(torch.randn(4 * c, c) / c ** 0.5, dt), dev(torch.randn(4 * c, c) / c ** 0.5, dt), dev(torch.randn(c, 4 * c) / (2 * c ** 0.5), dt)
    b1, bs, b2 = torch.randn(4 * c, device="cuda") * 0.3, torch.randn(4 * c, device="cuda") * 0.3, torch.randn(c, device="cuda") * 0.3
    rnd = lambda t: t.to(tdt).float()
    f = lambda t: t.float().cpu()
    tol = 1.2e-2 if dt == 1 else 1.5e-3
    mid = torch.full((M, 4 * c), 9.0, device="cuda").to(tdt); dst = torch.full((M, c), 9.0, device="cuda").to(tdt)
    bits = torch.full((M, c // 2), 0xAA, dtype=torch.uint8, device="cuda")
    mid_nb, dst_nb = torch.empty_like(mid), torch.empty_like(dst)
    with hip.options(grid_cap=cap):
        hip.conv_pair_shortcut(M, dt, src, w1, b1, xin, ws, bs, bits, mid, w2, b2, dst)
        hip.conv_pair_shortcut(M, dt, src, w1, b1, xin, ws, bs, None, mid_nb, w2, b2, dst_nb)
    torch.cuda.synchronize()
    assert torch.equal(mid, mid_nb) and torch.equal(dst, dst_nb)
    ref_mid = rnd(torch.relu(f(src) @ f(w1).T + b1.cpu() + f(xin) @ f(ws).T + bs.cpu()))
    assert float((f(mid) - ref_mid).abs().max()) <= tol * max(1.0, float(ref_mid.abs().max()))
    ref_dst = rnd(torch.relu(f(mid) @ f(w2).T + b2.cpu()))
    assert float((f(dst) - ref_dst).abs().max()) <= tol * max(1.0, float(ref_dst.abs().max()))
    pos = (mid.float() > 0).reshape(-1, 8).to(torch.int32)
    exp = (pos << torch.arange(8, device="cuda", dtype=torch.int32)).sum(1).to(torch.uint8)
    assert torch.equal(bits.reshape(-1), exp) and 0.2 < float(pos.float().mean()) < 0.8
    g1 = hip.geom(B, H, W, c, H, W, 4 * c, 1, 1)
    sc, m2, d2 = torch.empty_like(mid), torch.empty_like(mid), torch.empty_like(dst)
    hip.conv_igemm_ex(g1, dt, 0, xin, ws, bs, None, None, sc, None)
    hip.conv_pair(M, c, dt, 0, src, w1, b1, sc, None, m2, w2, b2, None, d2)
    torch.cuda.synchronize()
    assert float((mid.float() - m2.float()).abs().max()) <= 2 * tol * float(m2.float().abs().max())
    assert float((dst.float() - d2.float()).abs().max()) <= 3 * tol * float(d2.float().abs().max())
    with pytest.raises(hip.UrsoHipError):
        hip.conv_pair_shortcut(M + 1, dt, src, w1, b1, xin, ws, bs, None, mid, w2, b2, dst)


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("compact", [False, True], ids=["dense_add", "compact_add"])
@pytest.mark.parametrize("shape", [(1, 8, 8, 0), (2, 8, 24, 0), (4, 64, 80, 8), (8, 64, 80, 0), (32, 64, 80, 0)],
                         ids=["one_tile", "few_tiles", "capped", "multi_tile", "chip"])
def test_backward_pair_that_also_accumulates_the_weight_gradient(dt, shape, compact):
    """urso_conv_pair_wgrad (conv_pairw.hip): mid / dst bit for bit those of urso_conv_pair mode 1 (dense and compact add operand);
    the per-block fp32 partials summed over blocks against the fp32 product u^T mid of the STORED tensors (what urso_conv_wgrad
    computes from them) and against urso_conv_wgrad itself; blocks without tiles ('one_tile', 'few_tiles') write zero partials;
    'capped' / 'multi_tile' / 'chip' walk the three-stage input pipeline over several tiles per block."""
    hip = _hip()
    B, H, W, cap = shape
    M, c = B * H * W, 64
    tdt = hip.TORCH_DT[dt]
    torch.manual_seed(M + dt + 7)
    src, u = dev(torch.randn(M, c), dt), dev(torch.relu(torch.randn(M, c)), dt)
    w1, w2 = dev(torch.randn(4 * c, c) / c ** 0.5, dt), dev(torch.randn(c, 4 * c) / (2 * c ** 0.5), dt)
    if compact:
        add, hw = dev(torch.randn(B, H // 2, W // 2, 4 * c), dt), (H, W)
    else:
        add, hw = dev(torch.randn(M, 4 * c), dt), None
    bits = torch.randint(0, 256, (M, c // 2), dtype=torch.uint8, device="cuda")
    mid0 = torch.empty(M, 4 * c, dtype=tdt, device="cuda"); dst0 = torch.empty(M, c, dtype=tdt, device="cuda")
    mid = torch.full((M, 4 * c), 5.0, device="cuda").to(tdt); dst = torch.full((M, c), 5.0, device="cuda").to(tdt)
    with hip.options(grid_cap=cap):
        splits = hip.conv_pair_wgrad_splits(M, dt)
        assert splits >= 8 and splits % 8 == 0 and hip.conv_pair_wgrad_splits(M + 1, dt) == 0 and hip.conv_pair_wgrad_splits(M, 0) == 0
        stride = c * 4 * c + hip.WGRAD_PART_PAD
        part = torch.full((splits * stride,), float("nan"), device="cuda"); colpart = torch.full((splits * 4 * c,), float("nan"), device="cuda")
        hip.conv_pair(M, c, dt, 1, src, w1, None, add, bits, mid0, w2, None, u, dst0, add_hw=hw)
        hip.conv_pair_wgrad(M, dt, src, w1, add, bits, mid, w2, u, dst, part, colpart, stride, add_hw=hw)
    torch.cuda.synchronize()
    assert torch.equal(mid, mid0) and torch.equal(dst, dst0)
    dw = part.reshape(splits, stride)[:, :c * 4 * c].double().sum(0).reshape(c, 4 * c)
    cs = colpart.reshape(splits, 4 * c).double().sum(0)
    ref_dw = u.double().T @ mid.double()
    ref_cs = mid.double().sum(0)
    assert float((dw - ref_dw).abs().max()) <= 2e-5 * float(ref_dw.abs().max())
    assert float((cs - ref_cs).abs().max()) <= 2e-5 * float(ref_cs.abs().max()) + 1e-4
    g = hip.geom(B, H, W, c, H, W, 4 * c, 1, 1)
    ws = torch.empty(hip.conv_wgrad_ws_bytes(g, dt) // 4 + 64, dtype=torch.float32, device="cuda")
    dw2 = torch.empty(c * 4 * c, dtype=torch.float32, device="cuda"); cs2 = torch.empty(4 * c, dtype=torch.float32, device="cuda")
    hip.conv_wgrad(g, dt, u, mid, ws, dw2, cs2)
    torch.cuda.synchronize()
    assert float((dw - dw2.double().reshape(c, 4 * c)).abs().max()) <= 2e-5 * float(ref_dw.abs().max())
    assert float((cs - cs2.double()).abs().max()) <= 2e-5 * float(ref_cs.abs().max()) + 1e-4
    with pytest.raises(hip.UrsoHipError):
        hip.conv_pair_wgrad(M, dt, src, w1, add, bits, mid, w2, u, dst, part, colpart, c * 4 * c - 1, add_hw=hw)   # partials would overlap
    if not compact:
        # single-layer form: both gradients of a 64 -> 256 layer from one pass over dz (here dz = mid, x = u), masked and unmasked
        for masked in (1, 0):
            dx = torch.full((M, c), 5.0, device="cuda").to(tdt)
            part.fill_(float("nan")); colpart.fill_(float("nan"))
            with hip.options(grid_cap=cap):
                hip.conv_dgrad_wgrad_pw(M, dt, mid, w2, u, masked, dx, part, colpart, stride)
            g2 = hip.geom(B, H, W, 4 * c, H, W, c, 1, 1)
            dx2 = torch.empty_like(dx)
            hip.conv_igemm_ex(g2, dt, 0, mid, w2, None, None, u if masked else None, dx2)
            torch.cuda.synchronize()
            assert torch.equal(dx, dx2) and (not masked or torch.equal(dx, dst))
            dw3 = part.reshape(splits, stride)[:, :c * 4 * c].double().sum(0).reshape(c, 4 * c)
            cs3 = colpart.reshape(splits, 4 * c).double().sum(0)
            assert float((dw3 - ref_dw).abs().max()) <= 2e-5 * float(ref_dw.abs().max())
            assert float((cs3 - ref_cs).abs().max()) <= 2e-5 * float(ref_cs.abs().max()) + 1e-4


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("shape", [(1, 8, 8, 0), (2, 8, 24, 0), (4, 64, 80, 8), (8, 64, 80, 0), (32, 64, 80, 0)],
                         ids=["one_tile", "few_tiles", "capped", "multi_tile", "chip"])
def test_backward_pair_of_the_stage_entry_block(dt, shape):
    """urso_conv_pair_wgrad_entry (conv_pairx.hip) against the two launches it replaces -- urso_conv_pair_wgrad (which writes mid) and
    urso_conv_dgrad_wgrad_pw on that mid: dst and dxin bit for bit, both layers' summed partials and column sums against the fp64
    products of the stored tensors; blocks without tiles write zero partials; several tiles per block through the 3-stage / 2-stage
    input rings ('capped', 'multi_tile', 'chip')."""
    hip = _hip()
    B, H, W, cap = shape
    M, c = B * H * W, 64
    tdt = hip.TORCH_DT[dt]
    torch.manual_seed(M + dt + 11)
    src, u, xin = dev(torch.randn(M, c), dt), dev(torch.relu(torch.randn(M, c)), dt), dev(torch.relu(torch.randn(M, c)), dt)
    w1, w2, ws = dev(torch.randn(4 * c, c) / c ** 0.5, dt), dev(torch.randn(c, 4 * c) / (2 * c ** 0.5), dt), dev(torch.randn(c, 4 * c) / (2 * c ** 0.5), dt)
    add = dev(torch.randn(M, 4 * c), dt)
    bits = torch.randint(0, 256, (M, c // 2), dtype=torch.uint8, device="cuda")
    mid = torch.empty(M, 4 * c, dtype=tdt, device="cuda")
    dst0, dx0 = torch.empty(M, c, dtype=tdt, device="cuda"), torch.empty(M, c, dtype=tdt, device="cuda")
    dst, dx = torch.full((M, c), 5.0, device="cuda").to(tdt), torch.full((M, c), 5.0, device="cuda").to(tdt)
    with hip.options(grid_cap=cap):
        splits = hip.conv_pair_wgrad_splits(M, dt)
        stride = c * 4 * c + hip.WGRAD_PART_PAD
        mk = lambda n: torch.full((n,), float("nan"), device="cuda")
        part0, col0, ps0, cs0 = mk(splits * stride), mk(splits * 4 * c), mk(splits * stride), mk(splits * 4 * c)
        part, col, ps, cs = mk(splits * stride), mk(splits * 4 * c), mk(splits * stride), mk(splits * 4 * c)
        hip.conv_pair_wgrad(M, dt, src, w1, add, bits, mid, w2, u, dst0, part0, col0, stride)
        for masked in (1, 0):
            hip.conv_dgrad_wgrad_pw(M, dt, mid, ws, xin, masked, dx0, ps0, cs0, stride)
            hip.conv_pair_wgrad_entry(M, dt, src, w1, add, bits, w2, u, dst, ws, xin, masked, dx, part, col, ps, cs, stride)
            torch.cuda.synchronize()
            assert torch.equal(dst, dst0) and torch.equal(dx, dx0)
            assert masked == 0 or float((dx.float() == 0).float().mean()) > 0.3
    summed = lambda t: t.reshape(splits, stride)[:, :c * 4 * c].double().sum(0).reshape(c, 4 * c)
    ref_dw, ref_ds, ref_cs = u.double().T @ mid.double(), xin.double().T @ mid.double(), mid.double().sum(0)
    assert float((summed(part) - ref_dw).abs().max()) <= 2e-5 * float(ref_dw.abs().max())
    assert float((summed(ps) - ref_ds).abs().max()) <= 2e-5 * float(ref_ds.abs().max())
    assert float((summed(part) - summed(part0)).abs().max()) <= 2e-5 * float(ref_dw.abs().max())
    assert float((summed(ps) - summed(ps0)).abs().max()) <= 2e-5 * float(ref_ds.abs().max())
    for cc in (col, cs):
        assert float((cc.reshape(splits, 4 * c).double().sum(0) - ref_cs).abs().max()) <= 2e-5 * float(ref_cs.abs().max()) + 1e-4
    with pytest.raises(hip.UrsoHipError):
        hip.conv_pair_wgrad_entry(M, dt, src, w1, add, bits, w2, u, dst, ws, xin, 0, dx, part, col, ps, cs, c * 4 * c - 1)


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("shape", [(2, 16, 24, 64, 64, 3), (2, 32, 40, 128, 128, 3), (3, 16, 16, 64, 256, 1), (4, 128, 160, 64, 64, 3)],
                         ids=["c3x3_64", "c3x3_128", "pointwise_wide", "stage2_rows"])
def test_weight_gradient_with_dz_on_a_coarser_grid(dt, shape):
    """urso_conv_wgrad with a scattered dz operand (geometry FH / FW / OSH / OSW): the gradient tensor is dense [B, H, W, N] but
    non-zero only at even rows / columns; the stride-2 geometry reading it in place must give exactly the weight gradient computed
    from the gathered compact tensor, and (up to the summation order of a different split) the stride-1 weight gradient of the dense
    tensor with its explicit zeros."""
    hip = _hip()
    B, H, W, Ci, N, k = shape
    pad = k // 2
    tdt = hip.TORCH_DT[dt]
    torch.manual_seed(sum(shape) + dt)
    x = dev(torch.randn(B, H, W, Ci), dt)
    compact = dev(torch.randn(B, H // 2, W // 2, N), dt)
    dense = torch.zeros(B, H, W, N, dtype=tdt, device="cuda")
    dense[:, ::2, ::2] = compact

    def wgrad(g, dz):
        ws = torch.empty(hip.conv_wgrad_ws_bytes(g, dt) // 4 + 64, dtype=torch.float32, device="cuda")
        dw = torch.empty(k * k * Ci * N, dtype=torch.float32, device="cuda"); cs = torch.empty(N, dtype=torch.float32, device="cuda")
        hip.conv_wgrad(g, dt, x, dz, ws, dw, cs)
        torch.cuda.synchronize()
        return dw, cs
    g_compact = hip.geom(B, H, W, Ci, H // 2, W // 2, N, k, k, 2, 2, pad, pad)
    g_scatter = hip.geom(B, H, W, Ci, H // 2, W // 2, N, k, k, 2, 2, pad, pad, FH=H, FW=W, OSH=2, OSW=2)
    g_dense = hip.geom(B, H, W, Ci, H, W, N, k, k, 1, 1, pad, pad)
    a, b, c = wgrad(g_compact, compact), wgrad(g_scatter, dense), wgrad(g_dense, dense)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert relerr(b[0], c[0]) < 2e-5 and relerr(b[1], c[1]) < 2e-5
    assert float(b[0].abs().max()) > 0


PWX_FORMS = ["relu", "add_relu_bits", "mask_tensor", "add_maskbits", "maskbits", "add_relu"]


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("form", PWX_FORMS)
@pytest.mark.parametrize("shape", [(2, 16, 20, 1024, 256, 0), (3, 17, 21, 512, 320, 0), (4, 32, 40, 256, 512, 8), (2, 9, 13, 128, 72, 0),
                                   (32, 16, 20, 2048, 512, 0), (32, 32, 40, 1024, 256, 0), (32, 16, 20, 512, 2048, 0)],
                         ids=["one_round", "ragged_MN", "capped_multi_tile", "tiny_tails", "stage5_2a", "stage4_2a", "stage5_2c"])
def test_big_tile_pointwise_kernel_forced(dt, form, shape):
    """conv_pwx.hip (option pwx = 2: every supported pointwise layer; pwx_bn forces its 256- and 128-filter tiles) through urso_conv_igemm_ex
    for each instantiated epilogue form: against the CPU fp32 reference, and BIT FOR BIT against the DMA kernel of conv_pw.hip (same MFMA,
    same k order, same fp32 epilogue) -- output, emitted ReLU bit mask, masked data-gradient forms.  Shapes: whole tiles, ragged M and N
    tails (zero-filled rows, dropped stores), a grid capped to 8 blocks (the copy stream continues across tile seams and epilogues; counted
    vmcnt of the first mid-step after an epilogue), K = 128 (two K-steps per tile: the stream runs more than a tile ahead), and the three
    cfg2 geometries of VERDICT r02 item 1 at full size."""
    hip = _hip()
    B, H, W, K, N, cap = shape
    if B * H * W * max(K, N) > 20e6 and (dt == 2 or form not in ("relu", "add_relu_bits", "mask_tensor", "add_maskbits")):
        pytest.skip("full-size geometries: bf16 and the four forms the cfg2 plan uses")
    M = B * H * W
    tdt = hip.TORCH_DT[dt]
    torch.manual_seed(K + N + dt)
    x = dev(torch.randn(B, H, W, K), dt)
    w = torch.randn(1, 1, K, N) / K ** 0.5
    wf, _, biasf, _ = prep_weights(w, dt, bias=torch.randn(N) * 0.2)
    has_add = form.startswith("add")
    relu = "relu" in form
    emit = form.endswith("_bits")
    mbits = "maskbits" in form
    mtens = form == "mask_tensor"
    if mbits or mtens:
        biasf = None                                          # data-gradient forms carry no bias
    add = dev(torch.randn(B, H, W, N), dt) if has_add else None
    if N % 32 and (emit or mbits):
        pytest.skip("bit masks need N % 32 == 0")
    mask = None
    if mbits:
        mask = torch.randint(0, 256, (M * N // 8,), dtype=torch.uint8, device="cuda")
    if mtens:
        mask = dev(torch.randn(B, H, W, N), dt)
    flags = (hip.EPI_RELU if relu else 0) | (hip.EPI_EMIT_BITS if emit else 0) | (hip.EPI_MASK_BITS if mbits else 0)
    g = hip.geom(B, H, W, K, H, W, N, 1, 1)
    ref = x.float().cpu().reshape(M, K) @ wf.float().cpu().reshape(N, K).T
    if biasf is not None:
        ref = ref + biasf.cpu()
    if add is not None:
        ref = ref + add.float().cpu().reshape(M, N)
    if relu:
        ref = torch.relu(ref)
    if mbits:
        keep = ((mask.cpu().to(torch.int32).reshape(-1, 1) >> torch.arange(8, dtype=torch.int32)) & 1).reshape(M, N).float()
        ref = ref * keep
    if mtens:
        ref = ref * (mask.float().cpu().reshape(M, N) > 0).float()
    outs, obits = {}, {}
    for pwx, bn in ((2, 256), (2, 128), (0, 0)):
        y = torch.full((B, H, W, N), 5.0, device="cuda").to(tdt)
        bits = torch.full((M * N // 8,), 0x55, dtype=torch.uint8, device="cuda") if emit else None
        with hip.options(pwx=pwx, pwx_bn=bn, pair=0, grid_cap=cap):
            hip.conv_igemm_ex(g, dt, flags, x, wf, biasf, add, mask, y, bits)
        torch.cuda.synchronize()
        assert relerr(y.reshape(M, N), ref) < (1.2e-2 if dt == 1 else 1.5e-3), (pwx, bn)
        outs[(pwx, bn)], obits[(pwx, bn)] = y, bits
    for key in ((2, 256), (2, 128)):
        assert torch.equal(outs[key], outs[(0, 0)]), key
        if emit:
            assert torch.equal(obits[key], obits[(0, 0)]), key


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("shape", [(2, 16, 24, 64, 256, 0), (3, 18, 22, 128, 512, 0), (4, 32, 40, 256, 1024, 8), (32, 128, 160, 64, 256, 0)],
                         ids=["stage2_small", "stage3_ragged_tiles", "stage4_capped", "stage2_full_size"])
def test_block_output_computed_at_the_sampled_pixels_only(dt, shape):
    """urso_conv_igemm_ex with URSO_EPI_ADD_SRCGRID: a block-closing c -> 4c layer (+ identity shortcut, ReLU, emitted ReLU bit mask)
    evaluated ONLY at the pixels the next stage's stride-2 layers read (net.py:121-126) -- a 1x1 / stride-2 conv whose residual operand
    lives on the input grid.  Must equal, bit for bit, the even rows / columns of the dense layer's output and of its bit mask."""
    hip = _hip()
    B, H, W, c, N, cap = shape
    if B * H * W * N > 60e6 and dt == 2:
        pytest.skip("full size: bf16 only")
    tdt = hip.TORCH_DT[dt]
    torch.manual_seed(c + dt)
    x = dev(torch.randn(B, H, W, c), dt)
    w = torch.randn(1, 1, c, N) / c ** 0.5
    wf, _, biasf, _ = prep_weights(w, dt, bias=torch.randn(N) * 0.2)
    res = dev(torch.randn(B, H, W, N), dt)
    gdense = hip.geom(B, H, W, c, H, W, N, 1, 1)
    y = torch.empty(B, H, W, N, dtype=tdt, device="cuda")
    bits = torch.empty(B * H * W * N // 8, dtype=torch.uint8, device="cuda")
    with hip.options(pair=0, pwx=0):
        hip.conv_igemm_ex(gdense, dt, hip.EPI_RELU | hip.EPI_EMIT_BITS, x, wf, biasf, res, None, y, bits)
    gs = hip.geom(B, H, W, c, H // 2, W // 2, N, 1, 1, 2, 2, 0, 0)
    ys = torch.full((B, H // 2, W // 2, N), 7.0, device="cuda").to(tdt)
    bs = torch.full((B * (H // 2) * (W // 2) * N // 8,), 0x55, dtype=torch.uint8, device="cuda")
    with hip.options(grid_cap=cap):
        hip.conv_igemm_ex(gs, dt, hip.EPI_RELU | hip.EPI_EMIT_BITS | hip.EPI_ADD_SRCGRID, x, wf, biasf, res, None, ys, bs)
        ys2 = torch.empty_like(ys)
        hip.conv_igemm_ex(gs, dt, hip.EPI_RELU | hip.EPI_ADD_SRCGRID, x, wf, biasf, res, None, ys2, None)
    torch.cuda.synchronize()
    assert torch.equal(ys, y[:, ::2, ::2]) and torch.equal(ys2, ys)
    assert torch.equal(bs.view(B, H // 2, W // 2, N // 8), bits.view(B, H, W, N // 8)[:, ::2, ::2])
    assert 0.2 < float((ys.float() > 0).float().mean()) < 0.8
    with pytest.raises(hip.UrsoHipError):                     # padded / masked forms are refused
        hip.conv_igemm_ex(gs, dt, hip.EPI_ADD_SRCGRID, x, wf, biasf, res, res, ys, None)


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("shape", [(2, 16, 24, 64, 0), (3, 18, 22, 128, 0), (4, 32, 40, 256, 8), (8, 64, 80, 128, 0)],
                         ids=["c64", "c128_ragged_tiles", "c256_capped", "stage3_rows"])
def test_3x3_layer_computed_at_the_even_pixels_and_scattered(dt, shape):
    """The 3x3 layer below a sampled block output (Engine._sample_layer_below): a 3x3 / stride-2 / pad-1 conv whose results go to the
    even pixels of the dense output buffer (destination scatter).  Even pixels: bit for bit the dense layer on the same kernel
    (conv_pw.hip; c3 = 0, hconv = 0); the other pixels of the buffer are left untouched."""
    hip = _hip()
    B, H, W, c, cap = shape
    tdt = hip.TORCH_DT[dt]
    torch.manual_seed(c + dt)
    x = dev(torch.relu(torch.randn(B, H, W, c)), dt)
    w = torch.randn(3, 3, c, c) / (9 * c) ** 0.5
    wf, _, biasf, _ = prep_weights(w, dt, bias=torch.randn(c) * 0.2)
    gd = hip.geom(B, H, W, c, H, W, c, 3, 3, 1, 1, 1, 1)
    y = torch.empty(B, H, W, c, dtype=tdt, device="cuda")
    gs = hip.geom(B, H, W, c, H // 2, W // 2, c, 3, 3, 2, 2, 1, 1, FH=H, FW=W, OSH=2, OSW=2)
    ys = torch.full((B, H, W, c), 7.0, device="cuda").to(tdt)
    with hip.options(c3=0, hconv=0, grid_cap=cap):
        hip.conv_igemm(gd, dt, hip.EPI_RELU, x, wf, biasf, None, None, y)
        hip.conv_igemm(gs, dt, hip.EPI_RELU, x, wf, biasf, None, None, ys)
    torch.cuda.synchronize()
    assert torch.equal(ys[:, ::2, ::2], y[:, ::2, ::2])
    odd = torch.ones(H, W, dtype=torch.bool, device="cuda"); odd[::2, ::2] = False
    assert bool((ys[:, odd].float() == 7.0).all())
    ref = torch.relu(_ref_conv(x.float().cpu(), wf.float().cpu().reshape(c, 3, 3, c).permute(1, 2, 3, 0), 1, (1, 1), H, W) + biasf.cpu())     # wf: [N][ky][kx][C]
    assert relerr(y, ref) < TOL[dt]


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("case", [(32, 2560, 1024, "relu"), (32, 1024, 4096, "out_f32"), (32, 1024, 8, "out_f32"), (16, 2560, 1024, "relu"), (2, 512, 264, "relu"),
                                  (32, 4096, 1024, "add_mask"), (32, 8, 1024, "mask"), (5, 1024, 2560, "add")],
                         ids=["dense_0", "ori_final_f32", "loc_final_padded", "batch16", "batch2_ragged_N", "dgrad_final_add_mask", "dgrad_K8", "dgrad_add"])
def test_dense_head_skinny_gemm(dt, case):
    """conv_dense.hip through urso_conv_igemm (option dense, default): the Dense layers of the heads and their data gradients (at most 32
    rows) against the CPU fp32 reference and against the general kernel (dense = 0; different summation order: tolerance, not bits)."""
    hip = _hip()
    M, K, N, form = case
    tdt = hip.TORCH_DT[dt]
    torch.manual_seed(K + N + dt)
    x = dev(torch.randn(M, 1, 1, K), dt)
    w = (torch.randn(N, K) / K ** 0.5)
    wf = dev(w, dt)
    bias = torch.randn(N) * 0.3 if form in ("relu", "out_f32") else None
    add = dev(torch.randn(M, 1, 1, N), dt) if "add" in form else None
    mask = dev(torch.randn(M, 1, 1, N), dt) if "mask" in form else None
    flags = (hip.EPI_RELU if form == "relu" else 0) | (hip.EPI_OUT_F32 if form == "out_f32" else 0)
    g = hip.geom(M, 1, 1, K, 1, 1, N, 1, 1)
    ref = x.float().cpu().reshape(M, K) @ wf.float().cpu().T
    if bias is not None:
        ref = ref + bias
    if add is not None:
        ref = ref + add.float().cpu().reshape(M, N)
    if form == "relu":
        ref = torch.relu(ref)
    if mask is not None:
        ref = ref * (mask.float().cpu().reshape(M, N) > 0).float()
    outs = []
    for dense in (1, 0):
        y = torch.full((M, 1, 1, N), 3.0, device="cuda", dtype=torch.float32 if form == "out_f32" else tdt)
        with hip.options(dense=dense):
            hip.conv_igemm(g, dt, flags, x, wf, dev(bias) if bias is not None else None, add, mask, y)
        torch.cuda.synchronize()
        assert relerr(y.reshape(M, N), ref) < (1.2e-2 if dt == 1 else 1.5e-3) * (1 if form != "out_f32" else 0.2), dense
        outs.append(y.float())
    assert float((outs[0] - outs[1]).abs().max()) <= (1.6e-2 if dt == 1 else 2e-3) * float(outs[1].abs().max())


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("M", [32, 5])
def test_dense_head_weight_gradients_in_one_launch(dt, M):
    """urso_dense_wgrad_multi (conv_dense.hip): dW = x^T dz and the column sums of dz for four Dense layers of the heads' shapes (incl. the final
    layer padded to 8 outputs and a ragged 2,600 x 1,000) in one launch, against urso_conv_wgrad_partial's single split for the same layer
    (one 32-deep MFMA step either way) and the fp32 CPU reference."""
    hip = _hip()
    torch.manual_seed(3 * M + dt)
    layers, refs = [], []
    for (K, N) in [(2560, 1024), (1024, 8), (1024, 4096), (2600, 1000)]:
        x, dz = dev(torch.randn(M, K), dt), dev(torch.randn(M, N), dt)
        part, col = torch.full((K * N + 64,), 7.0, device="cuda"), torch.full((N,), 7.0, device="cuda")
        layers.append(dict(x=x, dz=dz, part=part, colpart=col, M=M, K=K, N=N))
        refs.append((x.float().cpu().T @ dz.float().cpu(), dz.float().cpu().sum(0)))
    hip.DenseWgradMulti(layers, dt).run()
    torch.cuda.synchronize()
    for L, (rw, rc) in zip(layers, refs):
        K, N = L["K"], L["N"]
        assert relerr(L["part"][:K * N].reshape(K, N), rw) < 2e-5 and relerr(L["colpart"], rc) < 2e-5
        assert float(L["part"][K * N:].min()) == 7.0                                   # nothing written behind the matrix
        if K % 8 == 0 and N % 8 == 0:
            g = hip.geom(M, 1, 1, K, 1, 1, N, 1, 1)
            assert hip.conv_wgrad_splits(g, dt) == 1
            ws = torch.empty(hip.conv_wgrad_ws_bytes(g, dt) // 4 + 64, device="cuda")
            hip.conv_wgrad_partial(g, dt, L["x"], L["dz"], ws)
            torch.cuda.synchronize()
            assert relerr(L["part"][:K * N], ws[:K * N]) < 1e-6


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("M", [32, 5])
def test_dense_heads_in_one_launch(dt, M):
    """urso_dense_multi (conv_dense.hip): (a) four Dense layers side by side -- loc_dense_0 / ori_dense_0 on the same input, an fp32 final
    layer padded to 8 outputs, a masked data gradient with a residual operand -- each what urso_conv_igemm's Dense kernel computes for it
    alone (bit for bit while every reduction of the launch is shorter than 2048; beyond that 16 waves split it: summation order); (b) a layer with two reduction segments (the data gradient into the bottleneck features: dZ_ori Wd_ori^T +
    dZ_loc Wd_loc^T) against the fp32 CPU reference and against the two-launch form (second launch accumulating in place: one more
    rounding, so tolerance)."""
    hip = _hip()
    tdt = hip.TORCH_DT[dt]
    torch.manual_seed(7 * M + dt)
    feat = dev(torch.randn(M, 2560), dt)
    specs = [(feat, 2560, 1024, hip.EPI_RELU, True, False, False), (feat, 2560, 1024, hip.EPI_RELU, True, False, False),
             (dev(torch.randn(M, 1024), dt), 1024, 8, hip.EPI_OUT_F32, True, False, False), (dev(torch.randn(M, 4096), dt), 4096, 1024, 0, False, True, True)]
    layers, singles = [], []
    for (x, K, N, flags, has_bias, has_add, has_mask) in specs:
        w = dev(torch.randn(N, K) / K ** 0.5, dt)
        bias = dev(torch.randn(N) * 0.3) if has_bias else None
        add = dev(torch.randn(M, N), dt) if has_add else None
        mask = dev(torch.randn(M, N), dt) if has_mask else None
        ot = torch.float32 if flags & hip.EPI_OUT_F32 else tdt
        y = torch.full((M, N), 3.0, dtype=ot, device="cuda")
        y1 = torch.full((M, N), 5.0, dtype=ot, device="cuda")
        layers.append(dict(src0=x, wgt0=w, K0=K, N=N, M=M, bias=bias, add=add, mask=mask, dst=y, flags=flags))
        hip.conv_igemm(hip.geom(M, 1, 1, K, 1, 1, N, 1, 1), dt, flags, x, w, bias, add, mask, y1)
        singles.append(y1)
    hip.DenseMulti(layers, dt).run()
    torch.cuda.synchronize()
    for L, y1 in zip(layers, singles):                      # K >= 2048 in the launch: 16 waves split the reduction instead of 8 (another fp32 summation order)
        assert float((L["dst"].float() - y1.float()).abs().max()) <= (8e-3 if dt == 1 else 1e-3) * float(y1.float().abs().max())
    hip.DenseMulti(layers[2:3] + [dict(layers[2], dst=torch.empty_like(layers[2]["dst"]))], dt).run()      # short reductions only: the very kernel body
    torch.cuda.synchronize()
    assert torch.equal(layers[2]["dst"], singles[2])
    # ---- two segments
    dz0, dz1 = dev(torch.randn(M, 1024), dt), dev(torch.randn(M, 1024), dt)
    w0, w1 = dev(torch.randn(2560, 1024) / 32.0, dt), dev(torch.randn(2560, 1024) / 32.0, dt)
    y = torch.full((M, 2560), 3.0, dtype=tdt, device="cuda")
    hip.DenseMulti([dict(src0=dz0, wgt0=w0, K0=1024, src1=dz1, wgt1=w1, K1=1024, N=2560, M=M, dst=y)], dt).run()
    y2 = torch.empty_like(y)
    g = hip.geom(M, 1, 1, 1024, 1, 1, 2560, 1, 1)
    hip.conv_igemm(g, dt, 0, dz0, w0, None, None, None, y2)
    hip.conv_igemm(g, dt, 0, dz1, w1, None, y2, None, y2)
    torch.cuda.synchronize()
    ref = dz0.float().cpu() @ w0.float().cpu().T + dz1.float().cpu() @ w1.float().cpu().T
    assert relerr(y, ref) < (6e-3 if dt == 1 else 8e-4)
    assert relerr(y2, ref) < (1.2e-2 if dt == 1 else 1.5e-3)
    with pytest.raises(hip.UrsoHipError):
        hip.DenseMulti([dict(src0=dz0, wgt0=w0, K0=1024, N=2560, M=64, dst=y)], dt).run()          # more than 32 rows


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("shape", [(3, 16, 20, 128, 32, 0), (32, 16, 20, 2048, 32, 0), (2, 20, 30, 256, 32, 0), (2, 9, 11, 64, 32, 1), (2, 4, 4, 512, 32, 0), (2, 16, 20, 64, 24, 0)],
                         ids=["tfsame_small", "cfg2_full", "cfg5_grid", "odd_grid_pad1", "cfg1_r18", "n24"])
def test_bottleneck_layer_kernels(dt, shape):
    """conv_bneck.hip through urso_conv_igemm_ex (option bneck: bit 0 = data gradient, default; bit 1 = forward, opt-in): bottleneck_layer (net.py:639-640: 3x3 / stride 2 / SAME, <= 32
    filters) forward in one launch with the reduction split over the block's waves, and its data gradient by parity class (only the real
    taps).  Forward against the CPU fp32 reference and the general split-K kernel (bneck = 0; other summation order: tolerance); the data
    gradient -- same taps in the same order, zero taps skipped -- BIT FOR BIT against the general dilated kernel, with and without the ReLU
    bit mask of the destination."""
    hip = _hip()
    B, H, W, Ci, N, pad = shape
    tdt = hip.TORCH_DT[dt]
    torch.manual_seed(B + H + Ci + dt)
    OH, OW = ((H + 2 * pad - 3) // 2 + 1, (W + 2 * pad - 3) // 2 + 1) if pad else (-(-H // 2), -(-W // 2))
    x = rnd(torch.randn(B, H, W, Ci), dt)
    w = torch.randn(3, 3, Ci, N) / (9 * Ci) ** 0.5
    bias = torch.randn(N) * 0.3
    npad = 32
    wf, wd, biasf, _ = prep_weights(w, dt, bias, None, npad=npad)
    g = hip.geom(B, H, W, Ci, OH, OW, npad, 3, 3, 2, 2, pad, pad)
    ref = _ref_conv(x, rnd(w, dt), 2, (pad, pad), OH, OW) + bias
    xd = dev(x, dt)
    outs = []
    for bneck in (3, 0):
        ws = torch.empty(hip.conv_igemm_ws_bytes(g, dt) // 4 + 4, dtype=torch.float32, device="cuda")
        for relu in (0, hip.EPI_RELU):
            y = torch.full((B, OH, OW, npad), 3.0, dtype=tdt, device="cuda")
            with hip.options(bneck=bneck):
                hip.conv_igemm_ex(g, dt, relu, xd, wf, biasf, None, None, y, None, ws)
            torch.cuda.synchronize()
            r = torch.relu(ref) if relu else ref
            assert relerr(y[..., :N], r) < TOL[dt], (bneck, relu)
            assert float(y[..., N:].float().abs().max()) == 0.0 if N < npad else True       # padded filters: zero rows, zero bias
            outs.append(y.float())
    assert float((outs[0] - outs[2]).abs().max()) <= TOL[dt] * float(outs[2].abs().max())
    # ---- data gradient: dz [B, OH, OW, 32] -> dx [B, H, W, Ci], gather form with dilation 2
    dz = dev(torch.randn(B, OH, OW, npad), dt)
    if N < npad:
        dz[..., N:] = 0
    gd = hip.geom(B, OH, OW, npad, H, W, Ci, 3, 3, 1, 1, 2 - pad, 2 - pad, 2, 2)
    xr = torch.relu(xd.float()).to(tdt)
    xpos = (xr.float() > 0).reshape(-1, 8).to(torch.int32)
    xbits = (xpos << torch.arange(8, device="cuda", dtype=torch.int32)).sum(1).to(torch.uint8)
    got = {}
    for bneck in (1, 0):
        for mb in (0, 1, 2):                                  # no mask, ReLU bit mask, mask tensor (the destination's own activation)
            dx = torch.full((B, H, W, Ci), 3.0, dtype=tdt, device="cuda")
            with hip.options(bneck=bneck):
                hip.conv_igemm_ex(gd, dt, hip.EPI_MASK_BITS if mb == 1 else 0, dz, wd, None, None, (None, xbits, xr)[mb], dx)
            torch.cuda.synchronize()
            got[(bneck, mb)] = dx
    assert torch.equal(got[(1, 0)], got[(0, 0)]) and torch.equal(got[(1, 1)], got[(0, 1)]) and torch.equal(got[(1, 2)], got[(0, 2)])
    assert torch.equal(got[(1, 1)], got[(1, 2)])
    assert float((got[(1, 1)].float() != 0).float().mean()) > 0.1
    # ... and against autograd on the CPU
    x_r = x.clone().requires_grad_(True)
    z = _ref_conv(x_r, rnd(w, dt), 2, (pad, pad), OH, OW)
    (z * dz[..., :N].float().cpu()).sum().backward()
    assert relerr(got[(1, 0)], x_r.grad) < TOL[dt]


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("shape", [(2, 16, 20, 128, 128), (3, 17, 23, 256, 128), (2, 9, 15, 64, 192), (4, 32, 40, 256, 256), (8, 64, 80, 128, 128),
                                   (2, 16, 20, 512, 512), (1, 7, 94, 128, 64)],
                         ids=["c128", "ragged_c256_n128", "c64_n192", "stage4_rows", "stage3_rows", "stage5_64_groups", "widest_row"])
def test_3x3_weight_gradient_halo_run_kernel(dt, shape):
    """conv_hwgrad.hip (option hwgrad, default) through urso_conv_wgrad: the 3x3 / stride-1 weight gradient with the gradient held in
    registers per (64-channel, 64-filter) group over virtual-pixel tiles -- against the CPU fp32 reference (torch autograd) and against the
    general kernel (hwgrad = 0), weights and column sums; image sizes that are not multiples of anything (zero column / row of the virtual
    grid, ragged last tile), 4 to 64 groups, splits from 4 to 64 per group."""
    hip = _hip()
    B, H, W, C, N = shape
    torch.manual_seed(C + N + dt)
    x = rnd(torch.relu(torch.randn(B, H, W, C)), dt)
    dz = rnd(torch.randn(B, H, W, N), dt)
    w = torch.zeros(3, 3, C, N, requires_grad=True)
    y = _ref_conv(x, w, 1, (1, 1), H, W)
    (y * dz).sum().backward()
    g = hip.geom(B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 1)
    outs = []
    for opt in (1, 0):
        with hip.options(hwgrad=opt):
            ws = torch.empty(hip.conv_wgrad_ws_bytes(g, dt) // 4 + 16, dtype=torch.float32, device="cuda")
            dw = torch.full((3, 3, C, N), 9.0, dtype=torch.float32, device="cuda")
            cs = torch.full((N,), 9.0, dtype=torch.float32, device="cuda")
            hip.conv_wgrad(g, dt, dev(x, dt), dev(dz, dt), ws, dw, cs)
            splits = hip.conv_wgrad_splits(g, dt)
        torch.cuda.synchronize()
        assert relerr(dw, w.grad) < TOL[dt] * 0.5, (opt, relerr(dw, w.grad))
        assert relerr(cs, dz.sum(dim=(0, 1, 2))) < 1e-4
        outs.append((dw, splits))
    assert outs[0][1] >= 1 and relerr(outs[0][0], outs[1][0].cpu()) < 1e-3       # fp32 accumulation in another order


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("shape", [(2, 16, 20, 64, 64, True), (3, 17, 23, 128, 72, True), (2, 9, 15, 64, 128, False), (4, 32, 40, 256, 256, True)],
                         ids=["c64", "odd_sizes_ragged_N", "no_relu", "stage4_rows"])
def test_winograd_f2x2_3x3_forward(dt, shape):
    """urso_conv_winograd_fwd (conv_winograd.hip): the Winograd F(2x2, 3x3) evaluation of a 3x3 / stride-1 / pad-1 layer -- filter and input
    transforms, sixteen frequency GEMMs on MFMA with fp32 outputs, output transform -- against the CPU fp32 conv and against the direct
    kernel.  Tolerance: the transformed operands V = B^T d B and U = G g G^T are stored in 16 bits (their products and sums are fp32), which
    costs about twice the direct kernel's error at bf16.  Odd image sizes: the second row / column of the last tiles is dropped."""
    hip = _hip()
    B, H, W, C, N, relu = shape
    tdt = hip.TORCH_DT[dt]
    torch.manual_seed(C + N + dt)
    x = dev(torch.relu(torch.randn(B, H, W, C)), dt)
    w = torch.randn(3, 3, C, N) / (9 * C) ** 0.5
    wf, _, biasf, _ = prep_weights(w, dt, bias=torch.randn(N) * 0.2)
    g = hip.geom(B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 1)
    ref = _ref_conv(x.float().cpu(), wf.float().cpu().reshape(N, 3, 3, C).permute(1, 2, 3, 0), 1, (1, 1), H, W) + biasf.cpu()
    if relu:
        ref = torch.relu(ref)
    nbytes = hip.conv_winograd_ws_bytes(g, dt)
    assert nbytes > 0
    ws = torch.empty(nbytes // 4 + 16, dtype=torch.float32, device="cuda")
    y = torch.full((B, H, W, N), 5.0, device="cuda").to(tdt); yd = torch.empty_like(y)
    hip.conv_winograd_fwd(g, dt, hip.EPI_RELU if relu else 0, x, wf, biasf, y, ws)
    hip.conv_igemm(g, dt, hip.EPI_RELU if relu else 0, x, wf, biasf, None, None, yd)
    torch.cuda.synchronize()
    e_w, e_d = relerr(y, ref), relerr(yd, ref)
    assert e_w < (3e-2 if dt == 1 else 4e-3) and e_d < TOL[dt], (e_w, e_d)
    with pytest.raises(hip.UrsoHipError):
        hip.conv_winograd_fwd(hip.geom(B, H, W, C, H // 2, W // 2, N, 3, 3, 2, 2, 1, 1), dt, 0, x, wf, biasf, y, ws)     # stride 2: refused
    assert hip.conv_winograd_ws_bytes(hip.geom(B, H, W, C, H, W, N, 1, 1), dt) == 0


def _pw(M, C, N):
    return (1, 1, M, C, 1, M, N, 1, 1, 1, 1, 0, 0)


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("layers", [[_pw(4096, 256, 128), _pw(4096, 128, 256)],
                                    [_pw(8200, 1024, 256), _pw(8200, 256, 1024), _pw(4160, 136, 72), _pw(16384, 128, 512),
                                     (4, 65, 81, 128, 33, 41, 136, 3, 3, 2, 2, 1, 1), (4, 64, 80, 256, 32, 40, 128, 1, 1, 2, 2, 0, 0),
                                     (3, 40, 9, 128, 40, 9, 192, 3, 3, 1, 1, 1, 1)],
                                    [_pw(40960, 256, 1024)] * 2 + [_pw(10240, 2048, 512), _pw(10240, 512, 2048)],
                                    [_pw(8200, 1024, 264), _pw(4160, 264, 520), (4, 65, 81, 256, 33, 41, 328, 3, 3, 2, 2, 1, 1),
                                     (4, 64, 80, 512, 32, 40, 256, 1, 1, 2, 2, 0, 0), (3, 40, 9, 128, 40, 9, 256, 3, 3, 1, 1, 1, 1)]],
                         ids=["pair", "ragged_mixed_geometries", "stage4_stage5_mix", "wide_ragged_mixed_geometries"])
@pytest.mark.parametrize("ring", [0, 4, 5], ids=["double_buffer", "ring4", "ring5"])
def test_grouped_weight_gradients(dt, layers, ring):
    """urso_wgrad_group_plan / _run (conv_wgrad.hip): several layers' weight gradients in one launch -- each layer's partials land in
    its own workspace with the layout of urso_conv_wgrad_partial, only with fewer splits; reduced in the fixed order they must match the
    per-layer launch (itself checked against autograd above; fp32 sums in another grouping) and, for the pointwise layers, the fp64 product
    of the same 16-bit operands and the column sums.  Ragged pixel counts and channel counts that leave partial tiles; layers of different
    pixel counts in one group (one common pixels-per-block); pointwise, strided 1x1, strided 3x3 and narrow-row 3x3 layers side by side
    (the kernel's three addressing modes in one launch)."""
    hip = _hip()
    torch.manual_seed(len(layers) + dt)
    geoms = [hip.geom(*l) for l in layers]
    xs = [dev(torch.relu(torch.randn(g.B, g.H, g.W, g.C)), dt) for g in geoms]
    dzs = [dev(torch.randn(g.B, g.OH, g.OW, g.N), dt) for g in geoms]
    wide = all(g.KH * g.KW * g.C >= 256 and g.N >= 256 for g in geoms)       # -> 256 x 256 tiles, one 8-wave block per CU (wgrad_group_big_kernel)
    with hip.options(wgrad_big=0 if ring else 1):          # (the ring variants belong to the 128 x 128 kernel: they also cover the wide groups on it)
        grp = hip.WgradGroup(geoms, dt)
    assert bool(grp.host[0].mode & 4) == (wide and not ring)
    assert grp.nblocks > 0 and grp.nblocks <= 512 and all(s >= 1 for s in grp.splits) and 0.0 < grp.fill <= 1.0
    assert grp.fill > 0.6 or layers[0][2] < 40960              # (the small groups cannot fill 512 slots with >= 8 steps per block)
    solo = [hip.conv_wgrad_splits(g, dt) for g in geoms]
    assert all(s <= t for s, t in zip(grp.splits, solo)) and (sum(grp.splits) < sum(solo) or layers[0][2] <= 4096)   # (short layers: 8 steps per block either way)
    KN = [g.KH * g.KW * g.C * g.N for g in geoms]
    wss = [torch.full((s * (kn + hip.WGRAD_PART_PAD) + s * g.N + 64,), 7.0, dtype=torch.float32, device="cuda") for s, kn, g in zip(grp.splits, KN, geoms)]
    with hip.options(wgrad_big=0 if ring else 1):
        grp.bind(xs, dzs, wss, "cuda")
    with hip.options(wgrad_ring=ring):                     # 0: 64-pixel double buffer; 4 / 5: stages of the 32-pixel deep ring
        grp.run()
    torch.cuda.synchronize()
    for i, g in enumerate(geoms):
        s, stride = grp.splits[i], KN[i] + hip.WGRAD_PART_PAD
        part = torch.stack([wss[i][k * stride:k * stride + KN[i]] for k in range(s)]).double().sum(0).reshape(-1, g.N)
        col = wss[i][s * stride:s * stride + s * g.N].reshape(s, g.N).double().sum(0)
        assert relerr(col, dzs[i].double().sum(dim=(0, 1, 2))) < 5e-5
        if g.KH == 1 and g.SH == 1:
            ref = xs[i].double().reshape(-1, g.C).t() @ dzs[i].double().reshape(-1, g.N)
            assert relerr(part, ref) < 5e-5, (i, relerr(part, ref))
        ws = torch.empty(hip.conv_wgrad_ws_bytes(g, dt) // 4 + 16, dtype=torch.float32, device="cuda")
        dw = torch.empty(part.shape, dtype=torch.float32, device="cuda"); cs = torch.empty(g.N, dtype=torch.float32, device="cuda")
        hip.conv_wgrad(g, dt, xs[i], dzs[i], ws, dw, cs)
        torch.cuda.synchronize()
        assert float(dw.abs().max()) > 0 and relerr(part, dw.double()) < 5e-5, (i, relerr(part, dw.double()))
    # the same launch again gives the same bits (fixed work list, no atomics)
    before = [w.clone() for w in wss]
    with hip.options(wgrad_ring=ring):
        grp.run()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(before, wss))


def test_grouped_weight_gradient_plan_refuses_what_cannot_be_resident():
    hip = _hip()
    with hip.options(wgrad_big=0):
        assert hip.WgradGroup([hip.geom(*_pw(4096, 2048, 2048))] * 3, 1).nblocks == 0      # 3 x 256 tiles of 128 x 128 > 512 resident blocks
    assert hip.WgradGroup([hip.geom(*_pw(4096, 2048, 2048))] * 3, 1).nblocks > 0           # 3 x 64 tiles of 256 x 256: one 8-wave block per CU
    assert hip.WgradGroup([hip.geom(*_pw(4096, 2048, 2048))] * 5, 1).nblocks == 0          # 5 x 64 > 256
    g = hip.geom(32, 32, 40, 256, 32, 40, 1024, 1, 1, 1, 1, 0, 0)
    assert hip.wgrad_group_fits(g, 1) and not hip.wgrad_group_fits(g, 0)
    assert not hip.wgrad_group_fits(hip.geom(32, 32, 40, 256, 32, 40, 256, 3, 3, 1, 1, 1, 1), 1)     # conv_hwgrad.hip's layer
    assert hip.wgrad_group_fits(hip.geom(32, 64, 80, 256, 32, 40, 256, 3, 3, 2, 2, 1, 1), 1)         # a strided 3x3 layer
    assert not hip.wgrad_group_fits(hip.geom(32, 32, 40, 256, 32, 40, 64, 1, 1, 1, 1, 0, 0), 1)      # <= 64 filters: the narrow kernel


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("shapes", [((4, 32, 40, 256, 256), (4, 32, 40, 256, 256)), ((4, 33, 41, 128, 128), (3, 17, 23, 256, 128)),
                                    ((8, 64, 80, 128, 128), (8, 32, 40, 256, 256))],
                         ids=["twins", "ragged_unequal", "stage3_with_stage4"])
def test_two_3x3_weight_gradients_in_one_launch(dt, shapes):
    """urso_conv_wgrad_partial2 (conv_hwgrad.hip: hwgrad2_kernel): two register-resident 3x3 weight gradients share a launch, the CUs split in
    proportion to their work; each layer's partials (fewer than alone) reduce to what urso_conv_wgrad gives for the layer alone."""
    hip = _hip()
    torch.manual_seed(dt + shapes[0][3])
    gs = [hip.geom(B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 1) for (B, H, W, C, N) in shapes]
    sp = hip.conv_wgrad_pair_splits(gs[0], gs[1], dt)
    assert sp is not None and all(1 <= s <= hip.conv_wgrad_splits(g, dt) for s, g in zip(sp, gs))
    assert sum(sp) < sum(hip.conv_wgrad_splits(g, dt) for g in gs)
    xs = [dev(torch.relu(torch.randn(g.B, g.H, g.W, g.C)), dt) for g in gs]
    dzs = [dev(torch.randn(g.B, g.H, g.W, g.N), dt) for g in gs]
    KN = [9 * g.C * g.N for g in gs]
    wss = [torch.full((s * (kn + hip.WGRAD_PART_PAD) + s * g.N + 64,), 7.0, dtype=torch.float32, device="cuda") for s, kn, g in zip(sp, KN, gs)]
    hip.conv_wgrad_partial2(gs[0], gs[1], dt, xs[0], dzs[0], wss[0], xs[1], dzs[1], wss[1])
    torch.cuda.synchronize()
    for i, g in enumerate(gs):
        s, stride = sp[i], KN[i] + hip.WGRAD_PART_PAD
        part = torch.stack([wss[i][k * stride:k * stride + KN[i]] for k in range(s)]).double().sum(0).reshape(-1, g.N)
        col = wss[i][s * stride:s * stride + s * g.N].reshape(s, g.N).double().sum(0)
        ws = torch.empty(hip.conv_wgrad_ws_bytes(g, dt) // 4 + 16, dtype=torch.float32, device="cuda")
        dw = torch.empty(part.shape, dtype=torch.float32, device="cuda"); cs = torch.empty(g.N, dtype=torch.float32, device="cuda")
        hip.conv_wgrad(g, dt, xs[i], dzs[i], ws, dw, cs)
        torch.cuda.synchronize()
        assert float(dw.abs().max()) > 0 and relerr(part, dw.double()) < 5e-5, (i, relerr(part, dw.double()))
        assert relerr(col, cs.double()) < 5e-5
    assert hip.conv_wgrad_pair_splits(gs[0], hip.geom(4, 32, 40, 64, 32, 40, 64, 3, 3, 1, 1, 1, 1), dt) is None      # conv_c3g.hip's layer


def test_zero_fill():
    """urso_zero_fill: the library's own fill for the buffer a scattered data gradient lands in (no torch launch inside the captured step)."""
    hip = _hip()
    t = torch.full((3, 37, 16), 5.0, dtype=torch.bfloat16, device="cuda")
    guard = t[2:]                               # the call covers the first two slabs only
    hip.zero_fill(t[:2])
    torch.cuda.synchronize()
    assert float(t[:2].float().abs().max()) == 0.0 and float(guard.float().min()) == 5.0
    with pytest.raises(Exception):
        hip.zero_fill(torch.zeros(3, dtype=torch.bfloat16, device="cuda"))       # 6 bytes: not a multiple of 16


def test_bucket_round_error_feedback_kernel_equals_the_torch_form():
    """urso_bucket_round_ef / urso_bucket_expand_bf16 (ursonet_amd/dp.py GradReducer, compress='bf16') against the three torch passes they
    replace -- t = g + r; c = bf16(t); r = t - float(c) -- bit for bit, on sizes with and without a whole number of 4-element vectors."""
    import ursonet_amd.hip as hip
    gen = torch.Generator(device="cuda").manual_seed(5)
    for n in (4, 1000, 4096 + 3, (1 << 20) + 4):
        g = torch.randn(n + 4, device="cuda", generator=gen)[:n] * 1e-2
        r = torch.randn(n + 4, device="cuda", generator=gen)[:n] * 1e-5
        c = torch.empty(n, dtype=torch.bfloat16, device="cuda")
        g0, r0 = g.clone(), r.clone()
        t = g0 + r0
        c_ref = t.to(torch.bfloat16)
        r_ref = t - c_ref.float()
        hip.bucket_round_ef(g, r, c)
        torch.cuda.synchronize()
        assert torch.equal(g, g0), "the gradient slice is only read"
        assert torch.equal(c.view(torch.int16), c_ref.view(torch.int16)) and torch.equal(r, r_ref)
        back = torch.empty(n, device="cuda")
        hip.bucket_expand_bf16(c, back)
        assert torch.equal(back, c_ref.float())
