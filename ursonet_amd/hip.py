"""ctypes binding of liburso_hip.so (C ABI declared in include/ursonet_hip.h).

The library is the only compute path of this package: if it is missing or fails to
load, importing this module raises -- there is no CPU / PyTorch fallback.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_VARIANT = os.environ.get("URSO_LIB_VARIANT", "")        # kernel experiments: see ursonet_amd/build.py
LIB_PATH = os.path.join(_HERE, "lib", "liburso_hip%s.so" % (("_" + _VARIANT) if _VARIANT else ""))

F32, BF16, F16 = 0, 1, 2
EPI_RELU, EPI_OUT_F32, EPI_MASK_BITS, EPI_EMIT_BITS, EPI_ADD_SRCGRID = 1, 2, 4, 8, 16
K_IGEMM, K_WGRAD, K_PREP, K_FINALIZE, K_POOL, K_LOSS, K_OPTIM, K_DECODE, K_MOLD = range(1, 10)
KERNEL_NAMES = {K_IGEMM: "conv_igemm", K_WGRAD: "conv_wgrad", K_PREP: "weight_prep", K_FINALIZE: "param_grad_finalize",
                K_POOL: "maxpool", K_LOSS: "loss", K_OPTIM: "optimizer", K_DECODE: "quat_decode", K_MOLD: "mold"}

TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}
DT_OF_TORCH = {v: k for k, v in TORCH_DT.items()}
DT_NAME = {F32: "f32", BF16: "bf16", F16: "f16"}


class ConvGeom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("B", "H", "W", "C", "OH", "OW", "N", "KH", "KW", "SH", "SW", "PH", "PW", "DH", "DW",
                 "FH", "FW", "OSH", "OSW")]

    def __repr__(self):
        return "ConvGeom(" + ", ".join("%s=%d" % (n, getattr(self, n)) for n, _ in self._fields_) + ")"


class ProfRecord(C.Structure):
    _fields_ = [("kernel_id", C.c_int32), ("ms", C.c_float), ("flops", C.c_double), ("bytes", C.c_double)]


import threading
# Held while a hipGraph is being captured (Engine.capture, DataParallelEngine.capture) AND by any other thread around its own HIP work
# (ursonet_amd/feeder.py producers: pinned allocations, augmentation kernels, device-to-host copies): HIP refuses such calls from any
# thread while a global-mode capture is in progress and may invalidate the capture.
capture_lock = threading.RLock()


class ProfRecordEx(C.Structure):
    _fields_ = [("kernel_id", C.c_int32), ("ms", C.c_float), ("flops", C.c_double), ("bytes", C.c_double), ("n_launches", C.c_int32),
                ("symbol", C.c_char * 228), ("l2_bytes", C.c_double)]


class UrsoHipError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError("liburso_hip.so not found at %s -- build it with `python -m ursonet_amd.build` "
                      "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
_lib = C.CDLL(LIB_PATH)

_vp, _fp, _i, _f, _sz = C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_size_t
_gp = C.POINTER(ConvGeom)


class ParamDesc(C.Structure):
    """urso_param_desc (include/ursonet_hip.h): one weight layer's parameter-side tensors for the batched phases."""
    _fields_ = ([(n, C.c_int32) for n in ("KH", "KW", "C", "N", "npad", "K", "splits", "ks", "kb", "trainable", "bn_trainable", "bias_from")] +
                [(n, C.c_float) for n in ("eps", "regc", "regb")] +
                [(n, C.c_void_p) for n in ("w", "b", "gamma", "beta", "mean", "var", "wf", "wd", "biasf", "scale", "part", "colpart",
                                           "dw_raw", "colsum", "dotpart", "gw", "gb", "ggamma", "gbeta")])


class DenseLayer(C.Structure):
    """urso_dense_layer (include/ursonet_hip.h): one Dense layer of a urso_dense_multi launch."""
    _fields_ = ([(n, C.c_void_p) for n in ("src0", "wgt0", "src1", "wgt1", "bias", "add", "mask", "dst")] +
                [(n, C.c_int32) for n in ("M", "N", "K0", "K1", "flags")])


class DenseWgradLayer(C.Structure):
    """urso_dense_wgrad_layer (include/ursonet_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("x", "dz", "part", "colpart")] + [(n, C.c_int32) for n in ("M", "K", "N")]


DENSE_MULTI_MAX = 4
PB_PREP, PB_REDUCE, PB_FINALIZE_MAT, PB_FINALIZE_VEC = 0, 1, 2, 3
WGRAD_PART_PAD = 64            # URSO_WGRAD_PART_PAD: floats between consecutive wgrad partial tensors
_dp = C.POINTER(ParamDesc)
_SIGS = {
    "urso_last_error": (C.c_char_p, []),
    "urso_abi_version": (_i, []),
    "urso_set_option": (_i, [C.c_char_p, _i]),
    "urso_get_option": (_i, [C.c_char_p, C.POINTER(C.c_int)]),
    "urso_conv_igemm": (_i, [_gp, _i, _i, _vp, _vp, _fp, _vp, _vp, _vp, _vp]),
    "urso_conv_igemm_ws_bytes": (_sz, [_gp, _i]),
    "urso_conv_igemm_ws": (_i, [_gp, _i, _i, _vp, _vp, _fp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "urso_conv_igemm_bits_ok": (_i, [_gp, _i, _i, _sz]),
    "urso_conv_igemm_algorithmic": (_i, [_gp, _i, _i, _i, _i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "urso_conv_igemm_halo_ok": (_i, [_gp, _i, _i, _i]),
    "urso_conv_igemm_halo2_shape": (_i, [_gp, _i, _i, _i, _i]),
    "urso_conv_igemm_halo_ws_bytes": (_sz, []),
    "urso_conv_igemm_ex": (_i, [_gp, _i, _i, _vp, _vp, _fp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "urso_conv_wgrad_ws_bytes": (_sz, [_gp, _i]),
    "urso_conv_wgrad": (_i, [_gp, _i, _vp, _vp, _vp, _sz, _fp, _fp, _vp]),
    "urso_conv_wgrad_splits": (_i, [_gp, _i]),
    "urso_conv_wgrad_partial": (_i, [_gp, _i, _vp, _vp, _vp, _sz, _vp]),
    "urso_param_desc_init": (_i, [_dp, _i, _i, _i, _i, _i, _i, _f, _f]),
    "urso_param_batch_plan": (_i, [_i, _dp, C.POINTER(C.c_int32), _i, C.POINTER(C.c_int32), _i]),
    "urso_wgrad_group_fits": (_i, [_gp, _i]),
    "urso_conv_wgrad_pair_splits": (_i, [_gp, _gp, _i, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "urso_conv_wgrad_partial2": (_i, [_gp, _gp, _i, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _sz, _vp]),
    "urso_wgrad_group_plan": (_i, [_i, _vp, _i, C.POINTER(C.c_int32), _i]),
    "urso_wgrad_group_run": (_i, [_i, _vp, _vp, _i, _vp, _i, _vp]),
    "urso_param_batch_run": (_i, [_i, _i, _vp, _vp, _i, _vp]),
    "urso_conv_weight_prep": (_i, [_i, _i, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp, _fp, _f, _vp, _vp, _fp, _fp, _vp]),
    "urso_stem_weight_pack": (_i, [_i, _i, _fp, _fp, _fp, _fp, _fp, _fp, _f, _vp, _fp, _fp, _vp]),
    "urso_stem_wgrad_unpack": (_i, [_i, _fp, _fp, _vp]),
    "urso_param_grad_finalize": (_i, [_i, _i, _i, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _f, _f, _i, _i,
                                      _fp, _fp, _fp, _fp, _fp, _sz, _vp]),
    "urso_param_grad_finalize_ws_bytes": (_sz, [_i, _i]),
    "urso_param_grad_finalize_sq": (_i, [_i, _i, _i, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _f, _f, _i, _i,
                                         _fp, _fp, _fp, _fp, _fp, _sz, _fp, _vp]),
    "urso_param_grad_finalize_sq_slots": (_i, [_i, _i]),
    "urso_param_batch_run_sq": (_i, [_i, _i, _vp, _vp, _i, _fp, _vp]),
    "urso_sqnorm_final": (_i, [_i, _fp, _fp, _vp]),
    "urso_bn_ws_bytes": (_sz, [_i, _i]),
    "urso_bn_batch_stats": (_i, [_i, _i, _i, _vp, _vp, _sz, _fp, _fp, _fp, _fp, _f, _f, _vp]),
    "urso_bn_apply": (_i, [_i, _i, _i, _vp, _fp, _fp, _fp, _fp, _f, _vp, _i, _vp, _vp]),
    "urso_bn_backward": (_i, [_i, _i, _i, _vp, _vp, _fp, _fp, _fp, _f, _vp, _sz, _fp, _fp, _i, _fp, _fp, _vp, _vp]),
    "urso_mold_images": (_i, [_i, _i, _i, _i, _vp, _fp, _i, _vp, _vp]),
    "urso_maxpool3x3s2_fwd": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "urso_maxpool3x3s2_bwd": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "urso_softmax_xent_fwd_bwd": (_i, [_i, _i, _fp, _fp, _f, _i, _i, _fp, _vp, _fp, _vp]),
    "urso_rel_l2_fwd_bwd": (_i, [_i, _i, _i, _fp, _fp, _f, _i, _fp, _vp, _fp, _vp]),
    "urso_rel_l2_norms": (_i, [_i, _i, _i, _fp, _fp, _fp, _vp]),
    "urso_rel_l2_from_norms": (_i, [_i, _i, _i, _fp, _fp, _f, _fp, _i, _fp, _fp, _vp, _vp]),
    "urso_absdot_fwd_bwd": (_i, [_i, _i, _i, _i, _fp, _fp, _f, _i, _fp, _fp, _vp, _vp]),
    "urso_mse_fwd_bwd": (_i, [_i, _i, _i, _fp, _fp, _f, _i, _fp, _vp, _vp]),
    "urso_sqnorm_ws_bytes": (_sz, [_sz]),
    "urso_sqnorm": (_i, [_sz, _fp, _vp, _sz, _fp, _vp]),
    "urso_sgd_momentum_clip": (_i, [_sz, _fp, _fp, _fp, _fp, _fp, _vp]),
    "urso_adam_amsgrad_clip": (_i, [_sz, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _vp]),
    "urso_scale_f32": (_i, [_sz, _fp, _f, _vp]),
    "urso_quat_wavg_decode": (_i, [_i, _i, _fp, _fp, _fp, _fp, _vp]),
    "urso_warp_perspective": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "urso_encode_ori": (_i, [_i, _i, _vp, _fp, _vp, C.c_double, _fp, _vp]),
    "urso_encode_loc": (_i, [_i, _i, _vp, _vp, C.c_double, _fp, _vp]),
    "urso_comm_unique_id": (_i, [_vp]),
    "urso_comm_init": (_i, [C.POINTER(_vp), _i, _i, _vp]),
    "urso_comm_allreduce_bucket": (_i, [_vp, _vp, _sz, _i, _vp]),
    "urso_comm_wait": (_i, [_vp, _vp]),
    "urso_comm_destroy": (_i, [_vp]),
    "urso_bucket_round_ef": (_i, [_sz, _fp, _fp, _vp, _vp]),
    "urso_dense_multi": (_i, [_i, C.POINTER(DenseLayer), _i, _vp]),
    "urso_dense_wgrad_multi": (_i, [_i, C.POINTER(DenseWgradLayer), _i, _vp]),
    "urso_bucket_expand_bf16": (_i, [_sz, _vp, _fp, _vp]),
    "urso_conv_pair_ok": (_i, [C.c_longlong, _i, _i, _i]),
    "urso_conv_pair": (_i, [C.c_longlong, _i, _i, _i, _vp, _vp, _fp, _vp, _vp, _vp, _vp, _fp, _vp, _vp, _i, _i, _vp]),
    "urso_conv_pair_shortcut": (_i, [C.c_longlong, _i, _vp, _vp, _fp, _vp, _vp, _fp, _vp, _vp, _vp, _fp, _vp, _vp]),
    "urso_conv_dgrad_wgrad_pw": (_i, [C.c_longlong, _i, _vp, _vp, _vp, _i, _vp, _fp, _fp, _sz, _vp]),
    "urso_conv_pair_wgrad_entry": (_i, [C.c_longlong, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _fp, _fp, _fp, _fp, _sz, _vp]),
    "urso_stem_wgrad_pooled": (_i, [_gp, _i, _vp, _vp, _vp, _vp, _sz, _fp, _fp, _vp]),
    "urso_stem_conv_pool_ok": (_i, [_gp, _i]),
    "urso_stem_conv_pool": (_i, [_gp, _i, _vp, _vp, _fp, _vp, _vp, _vp]),
    "urso_conv_pair_wgrad_splits": (_i, [C.c_longlong, _i]),
    "urso_conv_pair_wgrad": (_i, [C.c_longlong, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _fp, _fp, _sz, _vp]),
    "urso_conv_pointwise_sampled_ok": (_i, [_gp, _i, _i, _i]),
    "urso_conv_pointwise_sampled": (_i, [_gp, _i, _i, _vp, _vp, _fp, _vp, _vp, _vp, _vp, _vp]),
    "urso_rows_subsample2": (_i, [_i, _i, _i, _i, _vp, _vp, _vp]),
    "urso_rows_expand2": (_i, [_i, _i, _i, _i, _vp, _vp, _vp]),
    "urso_rows_scatter2": (_i, [_i, _i, _i, _i, _vp, _vp, _vp]),
    "urso_zero_fill": (_i, [_vp, C.c_size_t, _vp]),
    "urso_rgb_to_grey3": (_i, [_i, _i, _i, _vp, _vp, _vp]),
    "urso_sim2real_op": (_i, [_i, _i, _i, _vp, _vp, _vp, _fp, _vp, _vp, _i, _vp]),
    "urso_pad_images_u8": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "urso_conv_winograd_ws_bytes": (_sz, [_gp, _i]),
    "urso_conv_winograd_fwd": (_i, [_gp, _i, _i, _vp, _vp, _fp, _vp, _vp, _sz, _vp]),
    "urso_prof_enable": (_i, [_i]),
    "urso_prof_collect": (_i, [C.POINTER(ProfRecord), _i]),
    "urso_prof_collect_ex": (_i, [C.POINTER(ProfRecordEx), _i]),
    "urso_conv_pointwise2_ok": (_i, [_i] * 8),
    "urso_conv_pointwise2": (_i, [_i] * 8 + [_vp, _vp, _vp, _vp, _fp, _vp, _vp, _vp, _vp]),
}
EXPORTED_SYMBOLS = sorted(_SIGS)
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(_lib, _name)          # AttributeError here == symbol missing from the .so
    _fn.restype = _res
    _fn.argtypes = _args


def last_error():
    return (_lib.urso_last_error() or b"").decode()


def _chk(rc, what):
    if rc != 0:
        raise UrsoHipError("%s failed (%d): %s" % (what, rc, last_error()))


OPTION_NAMES = ("pw_kernel", "pw_small", "igemm_shortk", "wgrad_narrow", "wgrad_blocks", "wgrad_pipe", "grid_cap", "hconv", "hconv_dbg", "hconv2", "hconv2_shape", "hconv_streamk", "pair", "c3", "c3v", "stem", "stem_pool", "cus", "bneck")


def set_option(name, value):
    """urso_set_option: explicit kernel-policy switch (include/ursonet_hip.h)."""
    _chk(_lib.urso_set_option(name.encode(), int(value)), "urso_set_option")


def get_option(name):
    v = C.c_int(0)
    _chk(_lib.urso_get_option(name.encode(), C.byref(v)), "urso_get_option")
    return int(v.value)


class options(object):
    """Context manager: `with hip.options(grid_cap=16): ...` sets options and restores the previous values."""

    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


# A/B experiments inside one gpurun call: URSO_OPT_<NAME>=<int> in the environment of the PYTHON host is applied once at
# import (the C library itself never reads the environment).
for _o in OPTION_NAMES:
    _e = os.environ.get("URSO_OPT_" + _o.upper())
    if _e is not None:
        _chk(_lib.urso_set_option(_o.encode(), int(_e)), "urso_set_option")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "expected a contiguous device tensor"
    return t.data_ptr()


def stream_ptr(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return s.cuda_stream


def geom(B, H, W, Cin, OH, OW, N, KH, KW, SH=1, SW=1, PH=0, PW=0, DH=1, DW=1, FH=0, FW=0, OSH=0, OSW=0):
    return ConvGeom(B, H, W, Cin, OH, OW, N, KH, KW, SH, SW, PH, PW, DH, DW, FH, FW, OSH, OSW)


# ---------------------------------------------------------------- thin typed wrappers
def conv_igemm(g, dt, flags, src, wgt, bias, add, mask, dst, stream=None):
    _chk(_lib.urso_conv_igemm(C.byref(g), dt, flags, ptr(src), ptr(wgt), ptr(bias), ptr(add), ptr(mask), ptr(dst),
                              stream_ptr(stream)), "urso_conv_igemm")


class DenseMulti(object):
    """Up to DENSE_MULTI_MAX Dense layers of the heads in one launch (urso_dense_multi).  layers: dicts with src0, wgt0, dst, M, N, K0 and
    optionally src1, wgt1, K1 (a second reduction segment), bias, add, mask, flags -- tensors are kept alive by the object."""

    def __init__(self, layers, dt):
        assert 1 <= len(layers) <= DENSE_MULTI_MAX
        self.dt, self.n = dt, len(layers)
        self.host = (DenseLayer * self.n)()
        self.keep = []
        for it, L in zip(self.host, layers):
            for f in ("src0", "wgt0", "src1", "wgt1", "bias", "add", "mask", "dst"):
                t = L.get(f)
                self.keep.append(t)
                setattr(it, f, t.data_ptr() if t is not None else None)
            it.M, it.N, it.K0, it.K1, it.flags = int(L["M"]), int(L["N"]), int(L["K0"]), int(L.get("K1", 0)), int(L.get("flags", 0))

    def run(self, stream=None):
        _chk(_lib.urso_dense_multi(self.n, self.host, self.dt, stream_ptr(stream)), "urso_dense_multi")


class DenseWgradMulti(object):
    """The weight gradients of up to DENSE_MULTI_MAX Dense layers in one launch (urso_dense_wgrad_multi).  layers: dicts with x [M][K], dz [M][N],
    part (fp32, >= K * N) and colpart (fp32, >= N, or None) tensors and M, K, N."""

    def __init__(self, layers, dt):
        assert 1 <= len(layers) <= DENSE_MULTI_MAX
        self.dt, self.n = dt, len(layers)
        self.host = (DenseWgradLayer * self.n)()
        self.keep = []
        for it, L in zip(self.host, layers):
            assert L["part"].dtype == torch.float32 and L["part"].numel() >= L["K"] * L["N"]
            for f in ("x", "dz", "part", "colpart"):
                t = L.get(f)
                self.keep.append(t)
                setattr(it, f, t.data_ptr() if t is not None else None)
            it.M, it.K, it.N = int(L["M"]), int(L["K"]), int(L["N"])

    def run(self, stream=None):
        _chk(_lib.urso_dense_wgrad_multi(self.n, self.host, self.dt, stream_ptr(stream)), "urso_dense_wgrad_multi")


def bucket_round_ef(g, resid, c, stream=None):
    """urso_bucket_round_ef: c = bf16(g + resid), resid = (g + resid) - float(c), one pass (g is only read)."""
    assert g.dtype == torch.float32 and resid.dtype == torch.float32 and c.dtype == torch.bfloat16 and g.numel() == resid.numel() == c.numel()
    _chk(_lib.urso_bucket_round_ef(g.numel(), ptr(g), ptr(resid), ptr(c), stream_ptr(stream)), "urso_bucket_round_ef")


def bucket_expand_bf16(c, g, stream=None):
    """urso_bucket_expand_bf16: g = float(c)."""
    assert g.dtype == torch.float32 and c.dtype == torch.bfloat16 and g.numel() == c.numel()
    _chk(_lib.urso_bucket_expand_bf16(g.numel(), ptr(c), ptr(g), stream_ptr(stream)), "urso_bucket_expand_bf16")


def conv_winograd_ws_bytes(g, dt):
    return int(_lib.urso_conv_winograd_ws_bytes(C.byref(g), dt))


def conv_winograd_fwd(g, dt, flags, src, wgt, bias, dst, ws, stream=None):
    """urso_conv_winograd_fwd: Winograd F(2x2, 3x3) evaluation of a 3x3 / stride-1 / pad-1 forward conv (opt-in; slower than the direct kernels)."""
    _chk(_lib.urso_conv_winograd_fwd(C.byref(g), dt, flags, ptr(src), ptr(wgt), ptr(bias), ptr(dst), ptr(ws), ws.numel() * ws.element_size(),
                                     stream_ptr(stream)), "urso_conv_winograd_fwd")


def conv_igemm_ws_bytes(g, dt):
    return int(_lib.urso_conv_igemm_ws_bytes(C.byref(g), dt))


def conv_igemm_ws(g, dt, flags, src, wgt, bias, add, mask, dst, ws, stream=None):
    _chk(_lib.urso_conv_igemm_ws(C.byref(g), dt, flags, ptr(src), ptr(wgt), ptr(bias), ptr(add), ptr(mask), ptr(dst),
                                 ptr(ws), ws.numel() * ws.element_size() if ws is not None else 0, stream_ptr(stream)),
         "urso_conv_igemm_ws")


def conv_igemm_bits_ok(g, dt, flags, ws_bytes=0):
    return bool(_lib.urso_conv_igemm_bits_ok(C.byref(g), dt, flags, ws_bytes))


def conv_igemm_halo_ws_bytes():
    """Hand-over workspace of the halo-tile kernel's stream-K schedule; its first 4 KiB (flags) must be zero on entry (left zero)."""
    return int(_lib.urso_conv_igemm_halo_ws_bytes())


def conv_igemm_halo_ok(g, dt, flags, has_add=False):
    return bool(_lib.urso_conv_igemm_halo_ok(C.byref(g), dt, flags, int(bool(has_add))))


def conv_igemm_algorithmic(g, dt, flags, has_add=False, has_mask=False):
    """(FLOPs, bytes) the launch profiler records for a urso_conv_igemm_ex launch (host arithmetic)."""
    fl, by = C.c_double(0), C.c_double(0)
    _chk(_lib.urso_conv_igemm_algorithmic(C.byref(g), dt, flags, int(bool(has_add)), int(bool(has_mask)), C.byref(fl), C.byref(by)), "urso_conv_igemm_algorithmic")
    return float(fl.value), float(by.value)


def conv_igemm_halo2_shape(g, dt, flags, has_add=False, has_ws=False):
    """10 * MI + NJ of the whole-tile halo kernel (conv_halo2.hip) for this layer, 0 when conv_halo.hip keeps it."""
    return int(_lib.urso_conv_igemm_halo2_shape(C.byref(g), dt, flags, int(bool(has_add)), int(bool(has_ws))))


def conv_igemm_ex(g, dt, flags, src, wgt, bias, add, mask, dst, bits_out=None, ws=None, stream=None):
    _chk(_lib.urso_conv_igemm_ex(C.byref(g), dt, flags, ptr(src), ptr(wgt), ptr(bias), ptr(add), ptr(mask), ptr(dst), ptr(bits_out),
                                 ptr(ws), (ws.numel() * ws.element_size()) if ws is not None else 0, stream_ptr(stream)), "urso_conv_igemm_ex")


def conv_pointwise2_ok(B, OH, OW, C0, C1, N, dt, flags):
    return bool(_lib.urso_conv_pointwise2_ok(B, OH, OW, C0, C1, N, dt, flags))


def conv_pointwise2(B, OH, OW, C0, C1, N, dt, flags, src0, wgt0, src1, wgt1, bias, mask, dst, bits_out=None, stream=None):
    """dst = epilogue(src0 . wgt0^T + src1 . wgt1^T + bias): two pointwise layers over the same pixels in one launch (include/ursonet_hip.h)."""
    _chk(_lib.urso_conv_pointwise2(B, OH, OW, C0, C1, N, dt, flags, ptr(src0), ptr(wgt0), ptr(src1), ptr(wgt1), ptr(bias), ptr(mask), ptr(dst),
                                   ptr(bits_out), stream_ptr(stream)), "urso_conv_pointwise2")


def conv_wgrad_ws_bytes(g, dt):
    return int(_lib.urso_conv_wgrad_ws_bytes(C.byref(g), dt))


def conv_wgrad(g, dt, x, dz, ws, dw_raw, colsum, stream=None):
    _chk(_lib.urso_conv_wgrad(C.byref(g), dt, ptr(x), ptr(dz), ptr(ws), ws.numel() * ws.element_size(), ptr(dw_raw),
                              ptr(colsum), stream_ptr(stream)), "urso_conv_wgrad")


def conv_wgrad_splits(g, dt):
    return int(_lib.urso_conv_wgrad_splits(C.byref(g), dt))


def conv_wgrad_partial(g, dt, x, dz, ws, stream=None):
    _chk(_lib.urso_conv_wgrad_partial(C.byref(g), dt, ptr(x), ptr(dz), ptr(ws), ws.numel() * ws.element_size(), stream_ptr(stream)),
         "urso_conv_wgrad_partial")


def param_desc_init(d, KH, KW, Cin, N, npad, splits, eps, weight_decay):
    _chk(_lib.urso_param_desc_init(C.byref(d), KH, KW, Cin, N, npad, splits, eps, weight_decay), "urso_param_desc_init")


class ParamBatch(object):
    """A device copy of a ParamDesc array plus per-(phase, layer subset) block maps; run(phase, key) is one launch."""

    def __init__(self, descs, device):
        self.n = len(descs)
        self.host = (ParamDesc * self.n)(*descs)
        raw = torch.frombuffer(bytearray(bytes(self.host)), dtype=torch.uint8)
        self.dev = raw.to(device)
        self.maps = {}

    def plan(self, phase, key, layer_ids):
        ids = (C.c_int32 * len(layer_ids))(*layer_ids)
        nb = _lib.urso_param_batch_plan(phase, self.host, ids, len(layer_ids), None, 0)
        if nb < 0:
            raise UrsoHipError("urso_param_batch_plan: " + last_error())
        buf = (C.c_int32 * (2 * max(nb, 1)))()
        _lib.urso_param_batch_plan(phase, self.host, ids, len(layer_ids), buf, nb)
        t = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.int32).to(self.dev.device)
        self.maps[(phase, key)] = (t, nb)
        return nb

    def run(self, phase, key, dt, stream=None, sqpart=None):
        """sqpart (fp32, >= the phase's block count; FINALIZE phases): every block also leaves the sum of squares of what it stored there."""
        t, nb = self.maps[(phase, key)]
        if nb and sqpart is not None:
            assert sqpart.dtype == torch.float32 and sqpart.numel() >= nb
            _chk(_lib.urso_param_batch_run_sq(phase, dt, ptr(self.dev), ptr(t), nb, ptr(sqpart), stream_ptr(stream)), "urso_param_batch_run_sq")
        elif nb:
            _chk(_lib.urso_param_batch_run(phase, dt, ptr(self.dev), ptr(t), nb, stream_ptr(stream)), "urso_param_batch_run")

    def nblocks(self, phase, key):
        return self.maps[(phase, key)][1]


def conv_wgrad_pair_splits(g0, g1, dt):
    """(splits0, splits1) of urso_conv_wgrad_partial2 for two 3x3 layers of conv_hwgrad.hip, or None when they do not pair."""
    a, b = C.c_int(0), C.c_int(0)
    ok = _lib.urso_conv_wgrad_pair_splits(C.byref(g0), C.byref(g1), dt, C.byref(a), C.byref(b))
    return (a.value, b.value) if ok else None


def conv_wgrad_partial2(g0, g1, dt, x0, dz0, ws0, x1, dz1, ws1, stream=None):
    _chk(_lib.urso_conv_wgrad_partial2(C.byref(g0), C.byref(g1), dt, ptr(x0), ptr(dz0), ptr(ws0), ws0.numel() * ws0.element_size(),
                                       ptr(x1), ptr(dz1), ptr(ws1), ws1.numel() * ws1.element_size(), stream_ptr(stream)), "urso_conv_wgrad_partial2")


class WgradItem(C.Structure):
    """urso_wgrad_item (include/ursonet_hip.h)."""
    _fields_ = [("x", C.c_void_p), ("dz", C.c_void_p), ("part", C.c_void_p), ("colpart", C.c_void_p), ("g", ConvGeom),
                ("M", C.c_int32), ("ktiles", C.c_int32), ("ntiles", C.c_int32), ("splits", C.c_int32), ("m_per_split", C.c_int32),
                ("mode", C.c_int32), ("fill", C.c_int32)]


def wgrad_group_fits(g, dt):
    return bool(_lib.urso_wgrad_group_fits(C.byref(g), dt))


class WgradGroup(object):
    """The weight gradients of several 16-bit layers in one launch (urso_wgrad_group_plan / _run).

    geoms: the layers' forward geometries.  After construction .splits[i] holds each layer's partial count and .fill the fraction of
    the resident block slots the plan keeps busy (.nblocks == 0: the group does not fit one residency); bind() takes the operand /
    workspace tensors (partials laid out as urso_conv_wgrad_partial does)."""

    def __init__(self, geoms, dt):
        self.n, self.dt = len(geoms), dt
        self.host = (WgradItem * self.n)()
        for it, g in zip(self.host, geoms):
            C.memmove(C.byref(it.g), C.byref(g), C.sizeof(ConvGeom))
        nb = _lib.urso_wgrad_group_plan(self.n, self.host, dt, None, 0)
        if nb < 0:
            raise UrsoHipError("urso_wgrad_group_plan: " + last_error())
        self.nblocks = nb
        self.splits = [int(it.splits) for it in self.host] if nb else []
        self.fill = self.host[0].fill / 1000.0 if nb else 0.0     # work / (resident slots x the longest block)
        self.dev = self.map = None

    def bind(self, xs, dzs, wss, device):
        buf = (C.c_int32 * (2 * self.nblocks))()
        if _lib.urso_wgrad_group_plan(self.n, self.host, self.dt, buf, self.nblocks) != self.nblocks:
            raise UrsoHipError("urso_wgrad_group_plan: " + last_error())
        for it, x, dz, ws in zip(self.host, xs, dzs, wss):
            n_part = it.splits * (it.g.KH * it.g.KW * it.g.C * it.g.N + WGRAD_PART_PAD)
            if ws.numel() * ws.element_size() < 4 * (n_part + it.splits * it.g.N):
                raise UrsoHipError("WgradGroup.bind: workspace too small")
            it.x, it.dz, it.part, it.colpart = x.data_ptr(), dz.data_ptr(), ws.data_ptr(), ws.data_ptr() + 4 * n_part
        self.keep = (list(xs), list(dzs), list(wss))
        self.dev = torch.frombuffer(bytearray(bytes(self.host)), dtype=torch.uint8).to(device)
        self.map = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.int32).to(device)

    def run(self, stream=None):
        _chk(_lib.urso_wgrad_group_run(self.dt, ptr(self.dev), C.cast(self.host, C.c_void_p), self.n, ptr(self.map), self.nblocks,
                                       stream_ptr(stream)), "urso_wgrad_group_run")


def bn_ws_bytes(M, N):
    return int(_lib.urso_bn_ws_bytes(M, N))


def bn_batch_stats(M, N, dt, z, ws, mean, var, mmean, mvar, momentum, eps, stream=None):
    _chk(_lib.urso_bn_batch_stats(M, N, dt, ptr(z), ptr(ws), ws.numel() * ws.element_size(), ptr(mean), ptr(var), ptr(mmean), ptr(mvar),
                                  momentum, eps, stream_ptr(stream)), "urso_bn_batch_stats")


def bn_apply(M, N, dt, z, mean, var, gamma, beta, eps, res, relu, y, stream=None):
    _chk(_lib.urso_bn_apply(M, N, dt, ptr(z), ptr(mean), ptr(var), ptr(gamma), ptr(beta), eps, ptr(res), int(relu), ptr(y), stream_ptr(stream)),
         "urso_bn_apply")


def bn_backward(M, N, dt, g, z, mean, var, gamma, eps, ws, dbeta, dgamma, bn_trainable, gbeta, ggamma, dz, stream=None):
    _chk(_lib.urso_bn_backward(M, N, dt, ptr(g), ptr(z), ptr(mean), ptr(var), ptr(gamma), eps, ptr(ws), ws.numel() * ws.element_size(),
                               ptr(dbeta), ptr(dgamma), int(bn_trainable), ptr(gbeta), ptr(ggamma), ptr(dz), stream_ptr(stream)), "urso_bn_backward")


def conv_weight_prep(KH, KW, Cin, N, npad, dt, w, b, gamma, beta, mean, var, eps, wf, wd, biasf, scale, stream=None):
    _chk(_lib.urso_conv_weight_prep(KH, KW, Cin, N, npad, dt, ptr(w), ptr(b), ptr(gamma), ptr(beta), ptr(mean), ptr(var),
                                    eps, ptr(wf), ptr(wd), ptr(biasf), ptr(scale), stream_ptr(stream)),
         "urso_conv_weight_prep")


def stem_weight_pack(N, dt, w, b, gamma, beta, mean, var, eps, wf, biasf, scale, stream=None):
    _chk(_lib.urso_stem_weight_pack(N, dt, ptr(w), ptr(b), ptr(gamma), ptr(beta), ptr(mean), ptr(var), eps, ptr(wf),
                                    ptr(biasf), ptr(scale), stream_ptr(stream)), "urso_stem_weight_pack")


def stem_wgrad_unpack(N, dw_packed, dw_raw, stream=None):
    _chk(_lib.urso_stem_wgrad_unpack(N, ptr(dw_packed), ptr(dw_raw), stream_ptr(stream)), "urso_stem_wgrad_unpack")


def param_grad_finalize_ws_bytes(K, N):
    return int(_lib.urso_param_grad_finalize_ws_bytes(K, N))


def param_grad_finalize(K, N, ldn, dw_raw, colsum, w, b, gamma, mean, var, eps, wd, trainable, bn_trainable,
                        gw, gb, ggamma, gbeta, ws, stream=None):
    _chk(_lib.urso_param_grad_finalize(K, N, ldn, ptr(dw_raw), ptr(colsum), ptr(w), ptr(b), ptr(gamma), ptr(mean),
                                       ptr(var), eps, wd, int(trainable), int(bn_trainable), ptr(gw), ptr(gb),
                                       ptr(ggamma), ptr(gbeta), ptr(ws), ws.numel() * ws.element_size(),
                                       stream_ptr(stream)), "urso_param_grad_finalize")


def param_grad_finalize_sq_slots(K, N):
    return int(_lib.urso_param_grad_finalize_sq_slots(K, N))


def param_grad_finalize_sq(K, N, ldn, dw_raw, colsum, w, b, gamma, mean, var, eps, wd, trainable, bn_trainable,
                           gw, gb, ggamma, gbeta, ws, sqpart, stream=None):
    assert sqpart.dtype == torch.float32 and sqpart.numel() >= param_grad_finalize_sq_slots(K, N)
    _chk(_lib.urso_param_grad_finalize_sq(K, N, ldn, ptr(dw_raw), ptr(colsum), ptr(w), ptr(b), ptr(gamma), ptr(mean),
                                          ptr(var), eps, wd, int(trainable), int(bn_trainable), ptr(gw), ptr(gb),
                                          ptr(ggamma), ptr(gbeta), ptr(ws), ws.numel() * ws.element_size(), ptr(sqpart),
                                          stream_ptr(stream)), "urso_param_grad_finalize_sq")


def sqnorm_final(parts, out, stream=None):
    """urso_sqnorm_final: out[0] = sum of parts (the slots the *_sq finalisation launches wrote), in index order."""
    assert parts.dtype == torch.float32 and out.dtype == torch.float32
    _chk(_lib.urso_sqnorm_final(parts.numel(), ptr(parts), ptr(out), stream_ptr(stream)), "urso_sqnorm_final")


def mold_images(B, H, W, src, mean3, dt, dst, stream=None):
    is_u8 = 1 if src.dtype == torch.uint8 else 0
    assert is_u8 or src.dtype == torch.float32
    _chk(_lib.urso_mold_images(B, H, W, is_u8, ptr(src), ptr(mean3), dt, ptr(dst), stream_ptr(stream)), "urso_mold_images")


def maxpool_fwd(B, H, W, Cc, dt, x, y, argmax, stream=None):
    _chk(_lib.urso_maxpool3x3s2_fwd(B, H, W, Cc, dt, ptr(x), ptr(y), ptr(argmax), stream_ptr(stream)), "urso_maxpool3x3s2_fwd")


def maxpool_bwd(B, H, W, Cc, dt, y, dy, argmax, relu_mask, dx, stream=None):
    _chk(_lib.urso_maxpool3x3s2_bwd(B, H, W, Cc, dt, ptr(y), ptr(dy), ptr(argmax), int(relu_mask), ptr(dx),
                                    stream_ptr(stream)), "urso_maxpool3x3s2_bwd")


def softmax_xent(B, K, logits, labels, weight, relu_mask, dt, loss, dz, row_ws, stream=None):
    _chk(_lib.urso_softmax_xent_fwd_bwd(B, K, ptr(logits), ptr(labels), weight, int(relu_mask), dt, ptr(loss), ptr(dz),
                                        ptr(row_ws), stream_ptr(stream)), "urso_softmax_xent_fwd_bwd")


def rel_l2_norms(B, D, ld, gt, pred, norms, stream=None):
    _chk(_lib.urso_rel_l2_norms(B, D, ld, ptr(gt), ptr(pred), ptr(norms), stream_ptr(stream)), "urso_rel_l2_norms")


def rel_l2_from_norms(B, D, ld, gt, pred, weight, gscale, dt, norms, loss, dpred, stream=None):
    _chk(_lib.urso_rel_l2_from_norms(B, D, ld, ptr(gt), ptr(pred), weight, ptr(gscale), dt, ptr(norms), ptr(loss), ptr(dpred),
                                     stream_ptr(stream)), "urso_rel_l2_from_norms")


def rel_l2(B, D, ld, gt, pred, weight, dt, loss, dpred, norms=None, stream=None):
    _chk(_lib.urso_rel_l2_fwd_bwd(B, D, ld, ptr(gt), ptr(pred), weight, dt, ptr(loss), ptr(dpred), ptr(norms),
                                  stream_ptr(stream)), "urso_rel_l2_fwd_bwd")


def absdot(B, D, ld, normalize, gt, x, weight, dt, q, loss, dx, stream=None):
    _chk(_lib.urso_absdot_fwd_bwd(B, D, ld, int(normalize), ptr(gt), ptr(x), weight, dt, ptr(q), ptr(loss), ptr(dx),
                                  stream_ptr(stream)), "urso_absdot_fwd_bwd")


def mse(B, D, ld, gt, pred, weight, dt, loss, dpred, stream=None):
    _chk(_lib.urso_mse_fwd_bwd(B, D, ld, ptr(gt), ptr(pred), weight, dt, ptr(loss), ptr(dpred), stream_ptr(stream)),
         "urso_mse_fwd_bwd")


def sqnorm_ws_bytes(n):
    return int(_lib.urso_sqnorm_ws_bytes(n))


def sqnorm(n, g, ws, out, stream=None):
    _chk(_lib.urso_sqnorm(n, ptr(g), ptr(ws), ws.numel() * ws.element_size(), ptr(out), stream_ptr(stream)), "urso_sqnorm")


def sgd_momentum_clip(n, w, g, v, hyper, normsq, stream=None):
    _chk(_lib.urso_sgd_momentum_clip(n, ptr(w), ptr(g), ptr(v), ptr(hyper), ptr(normsq), stream_ptr(stream)),
         "urso_sgd_momentum_clip")


def adam_amsgrad_clip(n, w, g, m, v, vhat, hyper, normsq, stream=None):
    _chk(_lib.urso_adam_amsgrad_clip(n, ptr(w), ptr(g), ptr(m), ptr(v), ptr(vhat), ptr(hyper), ptr(normsq), stream_ptr(stream)),
         "urso_adam_amsgrad_clip")


def scale_f32(n, x, s, stream=None):
    _chk(_lib.urso_scale_f32(n, ptr(x), s, stream_ptr(stream)), "urso_scale_f32")


def quat_wavg_decode(B, K, logits, hquat, q, a=None, stream=None):
    _chk(_lib.urso_quat_wavg_decode(B, K, ptr(logits), ptr(hquat), ptr(q), ptr(a), stream_ptr(stream)),
         "urso_quat_wavg_decode")


def warp_perspective(B, H, W, Cc, interp, src, m, dst, stream=None):
    assert src.dtype == torch.uint8 and dst.dtype == torch.uint8 and m.dtype == torch.float64
    _chk(_lib.urso_warp_perspective(B, H, W, Cc, int(interp), ptr(src), ptr(m), ptr(dst), stream_ptr(stream)), "urso_warp_perspective")


def encode_ori(B, K, q, hquat, redundant, var, out, stream=None):
    assert q.dtype == torch.float64 and redundant.dtype == torch.uint8
    _chk(_lib.urso_encode_ori(B, K, ptr(q), ptr(hquat), ptr(redundant), float(var), ptr(out), stream_ptr(stream)), "urso_encode_ori")


def conv_pair_ok(M, dt, c_narrow, c_wide):
    return bool(_lib.urso_conv_pair_ok(int(M), dt, int(c_narrow), int(c_wide)))


def conv_pair(M, c_narrow, dt, mode, src, w1, bias1, add, bits, mid, w2, bias2, mask2, dst, add_hw=None, stream=None):
    """urso_conv_pair: two chained pointwise layers (c -> 4c (+add) -> c, c = 64 or 128) in one pass; mode 0 forward, 1 backward.
    add_hw = (H, W): `add` is the compact [B, H/2, W/2, 4c] gradient of a dense grid that is zero at odd rows / columns."""
    ah, aw = add_hw if add_hw else (0, 0)
    _chk(_lib.urso_conv_pair(int(M), int(c_narrow), dt, int(mode), ptr(src), ptr(w1), ptr(bias1), ptr(add), ptr(bits), ptr(mid), ptr(w2), ptr(bias2),
                             ptr(mask2), ptr(dst), int(ah), int(aw), stream_ptr(stream)), "urso_conv_pair")


def conv_pair_shortcut(M, dt, src, w1, bias1, xin, ws, bias_s, bits, mid, w2, bias2, dst, stream=None):
    """urso_conv_pair_shortcut: mid = relu(src W1^T + bias1 + xin Ws^T + bias_s), dst = relu(mid W2^T + bias2) (64 / 256 channels)."""
    _chk(_lib.urso_conv_pair_shortcut(int(M), dt, ptr(src), ptr(w1), ptr(bias1), ptr(xin), ptr(ws), ptr(bias_s), ptr(bits), ptr(mid), ptr(w2),
                                      ptr(bias2), ptr(dst), stream_ptr(stream)), "urso_conv_pair_shortcut")


def conv_dgrad_wgrad_pw(M, dt, dz, wd, x, mask_by_x, dx, part, colpart, part_stride, stream=None):
    """urso_conv_dgrad_wgrad_pw: dx = dz Wd^T (optionally masked by x > 0) and the split partials of dW = x^T dz, colsum (64 -> 256 layer)."""
    _chk(_lib.urso_conv_dgrad_wgrad_pw(int(M), dt, ptr(dz), ptr(wd), ptr(x), int(bool(mask_by_x)), ptr(dx), ptr(part), ptr(colpart),
                                       int(part_stride), stream_ptr(stream)), "urso_conv_dgrad_wgrad_pw")


def conv_pair_wgrad_entry(M, dt, src, w1, add, bits, w2, u, dst, ws, xin, mask_by_xin, dxin, part, colpart, part_s, colpart_s, part_stride, stream=None):
    """urso_conv_pair_wgrad_entry: backward pair + weight gradient of the block-closing layer + both gradients of the projection
    shortcut (dxin = mid Ws^T, dWs = xin^T mid); the 256-channel gradient mid never leaves the chip."""
    _chk(_lib.urso_conv_pair_wgrad_entry(int(M), dt, ptr(src), ptr(w1), ptr(add), ptr(bits), ptr(w2), ptr(u), ptr(dst), ptr(ws), ptr(xin), int(bool(mask_by_xin)), ptr(dxin),
                                         ptr(part), ptr(colpart), ptr(part_s), ptr(colpart_s), int(part_stride), stream_ptr(stream)),
         "urso_conv_pair_wgrad_entry")


def stem_conv_pool_ok(g, dt):
    """urso_stem_conv_pool_ok: does the packed stem geometry qualify for the fused conv1 + ReLU + max-pool kernel?"""
    return bool(_lib.urso_stem_conv_pool_ok(C.byref(g), dt))


def stem_conv_pool(g, dt, x, wf, bias, y_pooled, argmax, stream=None):
    """urso_stem_conv_pool: conv1 + ReLU + 3x3/s2 max-pool in one kernel; y_pooled [B][OH/2][OW/2][64], argmax its bytes."""
    _chk(_lib.urso_stem_conv_pool(C.byref(g), dt, ptr(x), ptr(wf), ptr(bias), ptr(y_pooled), ptr(argmax), stream_ptr(stream)), "urso_stem_conv_pool")


def stem_wgrad_pooled(g, dt, x, dpool, argmax, ws, dw_raw, colsum, stream=None):
    """urso_stem_wgrad_pooled: the stem's weight gradient from the max-pool output's gradient + arg-max bytes (no conv1-output gradient)."""
    _chk(_lib.urso_stem_wgrad_pooled(C.byref(g), dt, ptr(x), ptr(dpool), ptr(argmax), ptr(ws), ws.numel() * ws.element_size(), ptr(dw_raw),
                                     ptr(colsum), stream_ptr(stream)), "urso_stem_wgrad_pooled")


def conv_pair_wgrad_splits(M, dt):
    return int(_lib.urso_conv_pair_wgrad_splits(int(M), dt))


def conv_pair_wgrad(M, dt, src, w1, add, bits, mid, w2, u, dst, part, colpart, part_stride, add_hw=None, stream=None):
    """urso_conv_pair_wgrad: the stage-2 backward pair + fp32 partials of dW[64][256] / colsum[256] of the block-closing layer
    (x = u, dz = mid), one partial per block: part[s * part_stride + c * 256 + n], colpart[s * 256 + n]."""
    ah, aw = add_hw if add_hw else (0, 0)
    _chk(_lib.urso_conv_pair_wgrad(int(M), dt, ptr(src), ptr(w1), ptr(add), ptr(bits), ptr(mid), ptr(w2), ptr(u), ptr(dst), int(ah), int(aw),
                                   ptr(part), ptr(colpart), int(part_stride), stream_ptr(stream)), "urso_conv_pair_wgrad")


def conv_pointwise_sampled_ok(g, dt, flags, has_add=False):
    return bool(_lib.urso_conv_pointwise_sampled_ok(C.byref(g), dt, flags, int(bool(has_add))))


def conv_pointwise_sampled(g, dt, flags, src, wgt, bias, add, dst, bits_out, dst_sampled, stream=None):
    """urso_conv_pointwise_sampled: a stage-closing c -> 4c pointwise layer that also writes the even-row / even-column pixels of its output."""
    _chk(_lib.urso_conv_pointwise_sampled(C.byref(g), dt, flags, ptr(src), ptr(wgt), ptr(bias), ptr(add), ptr(dst), ptr(bits_out), ptr(dst_sampled),
                                          stream_ptr(stream)), "urso_conv_pointwise_sampled")


def zero_fill(t, stream=None):
    """urso_zero_fill: t[...] = 0 (16-byte aligned tensor whose size is a multiple of 16 bytes)."""
    _chk(_lib.urso_zero_fill(ptr(t), t.numel() * t.element_size(), stream_ptr(stream)), "urso_zero_fill")


def rows_expand2(B, H, W, row_bytes, src, dst, stream=None):
    """urso_rows_expand2: dst[b, y, x, :] = src[b, y/2, x/2, :] at even (y, x), zero elsewhere."""
    _chk(_lib.urso_rows_expand2(B, H, W, row_bytes, ptr(src), ptr(dst), stream_ptr(stream)), "urso_rows_expand2")


def rows_scatter2(B, H, W, row_bytes, src, dst, stream=None):
    """urso_rows_scatter2: dst[b, 2y, 2x, :] = src[b, y, x, :]; every other pixel of dst is left as it is (cleared once by the caller)."""
    _chk(_lib.urso_rows_scatter2(B, H, W, row_bytes, ptr(src), ptr(dst), stream_ptr(stream)), "urso_rows_scatter2")


def rows_subsample2(B, H, W, row_bytes, src, dst, stream=None):
    """urso_rows_subsample2: dst[b, y/2, x/2, :] = src[b, y, x, :] over per-pixel byte rows (ReLU bit masks)."""
    _chk(_lib.urso_rows_subsample2(B, H, W, row_bytes, ptr(src), ptr(dst), stream_ptr(stream)), "urso_rows_subsample2")


def encode_loc(B, K, loc, hmap, sig2, out, stream=None):
    """urso_encode_loc: loc fp64 [B,3], hmap fp64 [K,3] -> out fp32 [B,K] (utils.encode_loc utils.py:349-396)."""
    _chk(_lib.urso_encode_loc(B, K, ptr(loc), ptr(hmap), float(sig2), ptr(out), stream_ptr(stream)), "urso_encode_loc")


def rgb_to_grey3(B, H, W, src, dst, stream=None):
    assert src.dtype == torch.uint8 and dst.dtype == torch.uint8
    _chk(_lib.urso_rgb_to_grey3(B, H, W, ptr(src), ptr(dst), stream_ptr(stream)), "urso_rgb_to_grey3")


def sim2real_op(B, H, W, src, dst, op, par, seed, drop, drop_stride, stream=None):
    assert src.dtype == torch.uint8 and dst.dtype == torch.uint8 and op.dtype == torch.int32 and par.dtype == torch.float32
    assert seed.dtype == torch.int32, "seeds are passed as the bit pattern of uint32 in an int32 tensor"
    _chk(_lib.urso_sim2real_op(B, H, W, ptr(src), ptr(dst), ptr(op), ptr(par), ptr(seed), ptr(drop), int(drop_stride), stream_ptr(stream)),
         "urso_sim2real_op")


def pad_images_u8(B, H, W, Cc, OH, OW, top, left, src, dst, stream=None):
    assert src.dtype == torch.uint8 and dst.dtype == torch.uint8
    _chk(_lib.urso_pad_images_u8(B, H, W, Cc, OH, OW, int(top), int(left), ptr(src), ptr(dst), stream_ptr(stream)), "urso_pad_images_u8")


def prof_enable(on):
    _lib.urso_prof_enable(int(on))


def prof_collect(max_records=65536):
    buf = (ProfRecord * max_records)()
    n = _lib.urso_prof_collect(buf, max_records)
    return [(buf[i].kernel_id, buf[i].ms, buf[i].flops, buf[i].bytes) for i in range(n)]


def prof_collect_ex(max_records=65536):
    """[(kernel_id, ms, flops, bytes, n_launches, symbol)]: symbol = the device symbol of the first kernel the call launched
    (hipKernelNameRefByPtr: what rocprofv3 --kernel-trace prints)."""
    buf = (ProfRecordEx * max_records)()
    n = _lib.urso_prof_collect_ex(buf, max_records)
    return [(buf[i].kernel_id, buf[i].ms, buf[i].flops, buf[i].bytes, buf[i].n_launches, buf[i].symbol.decode(errors="replace")) for i in range(n)]


def prof_collect_l2(max_records=65536):
    """prof_collect_ex with the L2 -> LDS / register bytes of each call appended: [(kernel_id, ms, flops, bytes, n_launches, symbol, l2_bytes)]."""
    buf = (ProfRecordEx * max_records)()
    n = _lib.urso_prof_collect_ex(buf, max_records)
    return [(buf[i].kernel_id, buf[i].ms, buf[i].flops, buf[i].bytes, buf[i].n_launches, buf[i].symbol.decode(errors="replace"), buf[i].l2_bytes)
            for i in range(n)]


COMM_ID_BYTES = 128


def comm_unique_id():
    """urso_comm_unique_id: the 128 bytes rank 0 creates and every rank passes to Comm()."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _chk(_lib.urso_comm_unique_id(C.cast(buf, C.c_void_p)), "urso_comm_unique_id")
    return buf.raw


class Comm(object):
    """urso_comm_*: bucketed gradient averaging over RCCL on the communicator's own stream (include/ursonet_hip.h)."""

    def __init__(self, world, rank, unique_id):
        assert len(unique_id) == COMM_ID_BYTES
        h = C.c_void_p()
        self._id = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        _chk(_lib.urso_comm_init(C.byref(h), int(world), int(rank), C.cast(self._id, C.c_void_p)), "urso_comm_init")
        self.h, self.world, self.rank = h, int(world), int(rank)

    @classmethod
    def from_torch_group(cls, group=None):
        """One communicator over the ranks of an initialised torch.distributed group (the id travels through that group)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.tensor(list(comm_unique_id()) if rank == 0 else [0] * COMM_ID_BYTES, dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=0, group=group)
        return cls(world, rank, bytes(t.cpu().tolist()))

    def allreduce_bucket(self, t, stream=None):
        """In-place average of tensor t (fp32 / bf16 / fp16, contiguous) after the work already on `stream` (default: current)."""
        _chk(_lib.urso_comm_allreduce_bucket(self.h, ptr(t), t.numel(), DT_OF_TORCH[t.dtype], stream_ptr(stream)), "urso_comm_allreduce_bucket")

    def wait(self, stream=None):
        _chk(_lib.urso_comm_wait(self.h, stream_ptr(stream)), "urso_comm_wait")

    def close(self):
        if self.h:
            _lib.urso_comm_destroy(self.h)
            self.h = None
