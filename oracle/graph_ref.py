"""Oracle (test infrastructure): torch-CPU fp32 restatement of the reference's
NN graph, losses and optimizer step.  PARITY UNPINNED for the TF/Keras arithmetic
(see oracle/__init__.py): the reference holds no golden vectors for this path and
TensorFlow 1.x / Keras 2.1.6-2.2.4 (requirements.txt:9-10) are not importable here.
Each function cites the net.py lines it follows; third-party semantics are the
ones listed in SURVEY.md Appendix A (A1..A13).

Parameters are a dict {layer_name: {weight_name: np.ndarray}} in the Keras
layouts (Conv kernel HWIO, Dense kernel [in,out], BN gamma/beta/moving_mean/
moving_variance [C]) -- the "layer/weight API" of SURVEY.md Appendix B.
"""
import math
import re
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3          # Keras BatchNormalization default epsilon            [A3]


# --------------------------------------------------------------------------
# layer inventory (names / shapes), in graph order
# --------------------------------------------------------------------------
def _deep_blocks(arch):
    n4 = {"resnet50": 5, "resnet101": 22}[arch]                         # net.py:188
    return [(2, ["a", "b", "c"], [64, 64, 256]),
            (3, ["a", "b", "c", "d"], [128, 128, 512]),
            (4, ["a"] + [chr(98 + i) for i in range(n4)], [256, 256, 1024]),   # net.py:190
            (5, ["a", "b", "c"], [512, 512, 2048])]


def layer_specs(config):
    """Ordered [(layer_name, kind, {weight_name: shape})] for the whole model."""
    specs = []
    conv = lambda n, kh, kw, ci, co, bias: specs.append(
        (n, "conv", OrderedDict([("kernel", (kh, kw, ci, co))] + ([("bias", (co,))] if bias else []))))
    bn = lambda n, c: specs.append((n, "bn", OrderedDict(
        [("gamma", (c,)), ("beta", (c,)), ("moving_mean", (c,)), ("moving_variance", (c,))])))
    dense = lambda n, i, o: specs.append((n, "dense", OrderedDict([("kernel", (i, o)), ("bias", (o,))])))
    cin = config.NR_IMAGE_CHANNELS
    if config.BACKBONE in ("resnet50", "resnet101"):
        conv("conv1", 7, 7, cin, 64, True); bn("bn_conv1", 64)            # net.py:170-172
        c_in = 64
        for stage, blocks, (f1, f2, f3) in _deep_blocks(config.BACKBONE):
            for b in blocks:
                base, bnb = "res%d%s_branch" % (stage, b), "bn%d%s_branch" % (stage, b)
                conv(base + "2a", 1, 1, c_in, f1, True); bn(bnb + "2a", f1)
                conv(base + "2b", 3, 3, f1, f2, True); bn(bnb + "2b", f2)
                conv(base + "2c", 1, 1, f2, f3, True); bn(bnb + "2c", f3)
                if b == "a":
                    conv(base + "1", 1, 1, c_in, f3, True); bn(bnb + "1", f3)
                c_in = f3
        c5 = 2048
    else:
        assert config.BACKBONE in ("resnet18", "resnet34")                 # net.py:249
        conv("conv0", 7, 7, cin, 64, False); bn("bn_conv0", 64)           # net.py:255-256
        reps = [2, 2, 2, 2] if config.BACKBONE == "resnet18" else [3, 4, 6, 3]
        c_in = 64
        for stage, rep in enumerate(reps):
            f = 64 * 2 ** stage
            for block in range(rep):
                nb = "stage%d_unit%d_" % (stage + 1, block + 1)            # net.py:209
                if block == 0:
                    conv(nb + "sc", 1, 1, c_in, f, False)                  # net.py:225
                conv(nb + "conv1", 3, 3, c_in, f, False); bn(nb + "bn2", f)
                conv(nb + "conv2", 3, 3, f, f, False)
                c_in = f
        c5 = c_in
    bw = config.BOTTLENECK_WIDTH
    conv("bottleneck_layer", 3, 3, c5, bw, True)                           # net.py:639
    h, w = int(config.IMAGE_SHAPE[0]), int(config.IMAGE_SHAPE[1])
    nf = int(bw * h * w / (64 ** 2))                                       # net.py:640
    for br in (("loc",) if config.REGRESS_KEYPOINTS else ("loc", "ori")):   # keypoint model: ori_pred is not a Model output (net.py:675-691), Keras prunes the branch
        x = nf
        for i in range(config.NR_DENSE_LAYERS):
            dense("%s_dense_%d" % (br, i), x, config.BRANCH_SIZE)
            if config.TRAIN_BN:                                            # net.py:304 (truthy only)
                bn("%s_bn_%d" % (br, i), config.BRANCH_SIZE)
            x = config.BRANCH_SIZE
        if br == "loc":
            if config.REGRESS_KEYPOINTS:
                for k in ("k1_final", "k2_final", "k3_final"):
                    dense(k, x, 3)
            elif config.REGRESS_LOC:
                dense("loc_final", x, 3)
            else:
                dense("loc_final", x, config.LOC_BINS_PER_DIM ** 3)
        else:
            if config.REGRESS_ORI:
                if config.ORIENTATION_PARAM == "quaternion":
                    dense("ori_q", x, 4)
                else:
                    dense("ori_final", x, 3)
            else:
                dense("ori_final", x, config.ORI_BINS_PER_DIM ** 3)
    return specs


def init_params(config, seed=1234, randomize_bn=False):
    """Keras default initialisers [A11]: glorot_uniform kernels, zero biases, BN
    gamma=1 beta=0 mean=0 var=1; `randomize_bn` perturbs BN tensors (and biases)
    so the folded-BN path is actually exercised."""
    rng = np.random.default_rng(seed)
    params = OrderedDict()
    for name, kind, ws in layer_specs(config):
        p = OrderedDict()
        for wn, shape in ws.items():
            if wn == "kernel":
                if kind == "conv":
                    rf = shape[0] * shape[1]
                    fan_in, fan_out = rf * shape[2], rf * shape[3]
                else:
                    fan_in, fan_out = shape
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                p[wn] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
            elif wn == "bias":
                p[wn] = (rng.normal(0, 0.05, size=shape) if randomize_bn else np.zeros(shape)).astype(np.float32)
            elif wn == "gamma":
                p[wn] = (rng.uniform(0.5, 1.5, size=shape) if randomize_bn else np.ones(shape)).astype(np.float32)
            elif wn == "beta":
                p[wn] = (rng.normal(0, 0.1, size=shape) if randomize_bn else np.zeros(shape)).astype(np.float32)
            elif wn == "moving_mean":
                p[wn] = (rng.normal(0, 0.1, size=shape) if randomize_bn else np.zeros(shape)).astype(np.float32)
            elif wn == "moving_variance":
                p[wn] = (rng.uniform(0.5, 1.5, size=shape) if randomize_bn else np.ones(shape)).astype(np.float32)
        params[name] = p
    return params


def to_torch(params, requires_grad=True, dtype=torch.float32):
    out = OrderedDict()
    for ln, ws in params.items():
        out[ln] = OrderedDict()
        for wn, a in ws.items():
            t = torch.tensor(np.asarray(a), dtype=dtype)
            trainable = wn not in ("moving_mean", "moving_variance")
            t.requires_grad_(requires_grad and trainable)
            out[ln][wn] = t
    return out


# --------------------------------------------------------------------------
# ops (NCHW inside torch; Keras layouts at the boundary)
# --------------------------------------------------------------------------
def same_pad(n_in, k, s):
    """TF SAME padding [A2]: (before, after)."""
    out = -(-n_in // s)
    total = max((out - 1) * s + k - n_in, 0)
    return total // 2, total - total // 2


def conv2d(x, p, stride=1, padding="valid"):
    """Keras Conv2D [A1]. x NCHW; p['kernel'] HWIO; explicit zero pad for SAME."""
    w = p["kernel"].permute(3, 2, 0, 1)
    kh, kw = w.shape[2], w.shape[3]
    if padding == "same":
        pt, pb = same_pad(x.shape[2], kh, stride)
        pl, pr = same_pad(x.shape[3], kw, stride)
        x = F.pad(x, (pl, pr, pt, pb))
    elif isinstance(padding, int) and padding > 0:                         # ZeroPadding2D
        x = F.pad(x, (padding,) * 4)
    return F.conv2d(x, w, p.get("bias"), stride=stride)


def batchnorm(x, p, training):
    """net.py:60-76 + [A3]. training False -> moving statistics (frozen; gamma/beta still
    receive gradient); None -> batch statistics (biased variance), no sync."""
    shape = (1, -1, 1, 1) if x.dim() == 4 else (1, -1)
    if training is None or training is True:
        dims = (0, 2, 3) if x.dim() == 4 else (0,)
        mean = x.mean(dim=dims)
        var = x.var(dim=dims, unbiased=False)
    else:
        mean, var = p["moving_mean"], p["moving_variance"]
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean.view(shape)) * (inv * p["gamma"]).view(shape) + p["beta"].view(shape)


def maxpool_3x3_s2_same(x):
    """MaxPooling2D((3,3), strides 2, 'same') [A2]: pad with -inf so padding never wins."""
    pt, pb = same_pad(x.shape[2], 3, 2)
    pl, pr = same_pad(x.shape[3], 3, 2)
    x = F.pad(x, (pl, pr, pt, pb), value=float("-inf"))
    return F.max_pool2d(x, 3, 2)


def dense(x, p):
    return x @ p["kernel"] + p["bias"]                                     # [A4]


def relu(x, site=None, hook=None):
    """Activation('relu').  `hook(site, pre_activation) -> bool mask or None` lets a TEST force
    the few ReLU decisions that sit within float rounding of zero to the decision the device
    took (tests/test_model_gpu.py); with hook=None this is plain max(x, 0)."""
    if hook is not None:
        m = hook(site, x)
        if m is not None:
            return x * m.to(x.dtype)
    return F.relu(x)


# --------------------------------------------------------------------------
# 16-bit storage variant of the oracle (test infrastructure for the bf16 / fp16 device paths)
# --------------------------------------------------------------------------
class _RoundAct(torch.autograd.Function):
    """Storage rounding of an activation tensor: the value is rounded to the 16-bit storage type on the way forward and the
    gradient that reaches it (the sum over all consumers, in fp32) is rounded on the way back -- the device keeps both the
    activation and its gradient buffer in the storage type."""
    @staticmethod
    def forward(ctx, x, dtype, fwd):
        ctx.dtype = dtype
        return x.to(dtype).to(x.dtype) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None, None


class StorageRounding(object):
    """`q` argument of forward()/losses()/gradients(): the SAME graph with rounding to `dtype` (torch.bfloat16 / float16)
    inserted exactly where the device stores 16-bit tensors: the molded image, every conv/dense
    output after its fused epilogue, every activation gradient, and the per-step folded filter W*gamma/sqrt(var+eps)
    (straight-through for the gradient w.r.t. W, gamma: the device's fp32 master weights receive the unrounded
    gradient).  In exact arithmetic this is the plain oracle; it exists so that the 16-bit device path can be
    compared at 1e-2 per tensor instead of by cosine similarity.  Frozen BN only (TRAIN_BN False)."""

    def __init__(self, dtype, unstored=()):
        self.dtype = dtype
        # projection shortcuts the device computes INSIDE the launch of the layer that adds them (Engine.shortcut_folded: stage 2's fused pair,
        # stages 4-5's two-segment branch2c): their output never reaches a 16-bit tensor, so it is not rounded here either
        self.unstored = set(unstored)

    def weight(self, w):
        return w + (w.detach().to(self.dtype).to(w.dtype) - w.detach())

    def act(self, x):
        return _RoundAct.apply(x, self.dtype, True)

    def grad(self, x):                      # fp32-stored head outputs: only their gradient buffer is 16-bit
        return _RoundAct.apply(x, self.dtype, False)


def conv_bn(x, Pc, Pb, train_bn, stride=1, padding="valid", q=None):
    """Conv2D followed by the BatchNorm wrapper (net.py:60-76, 101-103 ...).  With q (StorageRounding) and frozen BN
    the pair is evaluated in the device's folded form: conv(x, round(W*s)) + (s*b + beta - mean*s), s = gamma/sqrt(var+eps)
    -- algebraically the same function of (x, W, b, gamma, beta)."""
    if q is None:
        y = conv2d(x, Pc, stride=stride, padding=padding)
        return batchnorm(y, Pb, train_bn) if Pb is not None else y
    assert train_bn is False or Pb is None, "StorageRounding restates the frozen-BN (folded) device path only"
    w = Pc["kernel"]
    if Pb is not None:
        sc = Pb["gamma"] * torch.rsqrt(Pb["moving_variance"] + BN_EPS)
        b0 = Pc["bias"] if "bias" in Pc else torch.zeros_like(sc)
        bias = sc * b0 + Pb["beta"] - Pb["moving_mean"] * sc
        w = w * sc
    else:
        bias = Pc.get("bias")
    return conv2d(x, {"kernel": q.weight(w), **({"bias": bias} if bias is not None else {})}, stride=stride, padding=padding)


def _st(x, q):
    return x if q is None else q.act(x)


# --------------------------------------------------------------------------
# graph builders
# --------------------------------------------------------------------------
def identity_block(x, P, stage, block, train_bn, hook=None, q=None):
    """net.py:85-117."""
    cb, bb = "res%d%s_branch" % (stage, block), "bn%d%s_branch" % (stage, block)
    y = _st(relu(conv_bn(x, P[cb + "2a"], P[bb + "2a"], train_bn, q=q), cb + "2a", hook), q)
    y = _st(relu(conv_bn(y, P[cb + "2b"], P[bb + "2b"], train_bn, padding="same", q=q), cb + "2b", hook), q)
    y = conv_bn(y, P[cb + "2c"], P[bb + "2c"], train_bn, q=q)
    return _st(relu(y + x, cb + "2c", hook), q)


def conv_block(x, P, stage, block, stride, train_bn, hook=None, q=None):
    """net.py:120-158 -- stride sits on 2a and on the shortcut branch1."""
    cb, bb = "res%d%s_branch" % (stage, block), "bn%d%s_branch" % (stage, block)
    y = _st(relu(conv_bn(x, P[cb + "2a"], P[bb + "2a"], train_bn, stride=stride, q=q), cb + "2a", hook), q)
    y = _st(relu(conv_bn(y, P[cb + "2b"], P[bb + "2b"], train_bn, padding="same", q=q), cb + "2b", hook), q)
    y = conv_bn(y, P[cb + "2c"], P[bb + "2c"], train_bn, q=q)
    sc = conv_bn(x, P[cb + "1"], P[bb + "1"], train_bn, stride=stride, q=q)
    if q is None or (cb + "1") not in q.unstored:
        sc = _st(sc, q)
    return _st(relu(y + sc, cb + "2c", hook), q)


def resnet_graph(x, P, arch, train_bn, hook=None, q=None):
    """net.py:161-199 (stage5=True)."""
    x = conv_bn(x, P["conv1"], P["bn_conv1"], train_bn, stride=2, padding=3, q=q)
    x = _st(relu(x, "conv1", hook), q)
    x = maxpool_3x3_s2_same(x)
    for stage, blocks, _ in _deep_blocks(arch):
        for b in blocks:
            if b == "a":
                x = conv_block(x, P, stage, b, 1 if stage == 2 else 2, train_bn, hook, q)
            else:
                x = identity_block(x, P, stage, b, train_bn, hook, q)
    return x


def residual_basic_block(x, P, stage, block, stride, cut, train_bn, hook=None, q=None):
    """net.py:216-240 -- ONE BatchNorm per block (after conv1)."""
    nb = "stage%d_unit%d_" % (stage + 1, block + 1)
    sc = x if cut == "pre" else _st(conv_bn(x, P[nb + "sc"], None, train_bn, stride=stride, q=q), q)
    y = conv_bn(x, P[nb + "conv1"], P[nb + "bn2"], train_bn, stride=stride, padding=1, q=q)
    y = _st(relu(y, nb + "conv1", hook), q)
    y = conv_bn(y, P[nb + "conv2"], None, train_bn, padding=1, q=q)
    return _st(relu(y + sc, nb + "conv2", hook), q)


def resnet_shallow_graph(x, P, arch, train_bn, hook=None, q=None):
    """net.py:242-282."""
    x = conv_bn(x, P["conv0"], P["bn_conv0"], train_bn, stride=2, padding=3, q=q)
    x = _st(relu(x, "conv0", hook), q)
    x = maxpool_3x3_s2_same(x)
    reps = [2, 2, 2, 2] if arch == "resnet18" else [3, 4, 6, 3]
    for stage, rep in enumerate(reps):
        for block in range(rep):
            if block == 0 and stage == 0:
                x = residual_basic_block(x, P, stage, block, 1, "post", train_bn, hook, q)
            elif block == 0:
                x = residual_basic_block(x, P, stage, block, 2, "post", train_bn, hook, q)
            else:
                x = residual_basic_block(x, P, stage, block, 1, "pre", train_bn, hook, q)
    return x


def _qdense(x, p, q):
    return dense(x, p) if q is None else dense(x, {"kernel": q.weight(p["kernel"]), "bias": p["bias"]})


def _branch_trunk(feat, P, config, br, hook=None, q=None):
    x = feat
    for i in range(config.NR_DENSE_LAYERS):
        x = _qdense(x, P["%s_dense_%d" % (br, i)], q)
        if config.TRAIN_BN:
            assert q is None
            x = batchnorm(x, P["%s_bn_%d" % (br, i)], None)               # net.py:306: no training arg
        x = _st(relu(x, "%s_dense_%d" % (br, i), hook), q)
    return x


def _head(x, q):
    return x if q is None else q.grad(x)                                  # fp32 head output, 16-bit gradient buffer


def build_loc_graph(feat, P, config, hook=None, q=None):
    """net.py:288-320."""
    x = _branch_trunk(feat, P, config, "loc", hook, q)
    if config.REGRESS_KEYPOINTS:
        return [_head(_qdense(x, P[k], q), q) for k in ("k1_final", "k2_final", "k3_final")]
    if config.REGRESS_LOC:
        return _head(_qdense(x, P["loc_final"], q), q)
    return relu(_head(_qdense(x, P["loc_final"], q), q), "loc_final", hook)


def build_ori_graph(feat, P, config, hook=None, q=None):
    """net.py:322-352."""
    x = _branch_trunk(feat, P, config, "ori", hook, q)
    if config.REGRESS_ORI:
        if config.ORIENTATION_PARAM == "quaternion":
            qq = _head(_qdense(x, P["ori_q"], q), q)
            return qq * torch.rsqrt(torch.clamp((qq * qq).sum(-1, keepdim=True), min=1e-12))   # [A5]
        return _head(_qdense(x, P["ori_final"], q), q)
    return relu(_head(_qdense(x, P["ori_final"], q), q), "ori_final", hook)


def forward(P, images_nhwc, config, relu_hook=None, q=None):
    """net.py:629-643.  images_nhwc: molded float tensor [B,H,W,C].  Returns (loc, ori)."""
    h, w = images_nhwc.shape[1:3]
    if h / 2 ** 6 != int(h / 2 ** 6) or w / 2 ** 6 != int(w / 2 ** 6):   # net.py:596-600
        raise Exception("Image size must be dividable by 2 at least 6 times "
                        "to avoid fractions when downscaling and upscaling."
                        "For example, use 256, 320, 384, 448, 512, ... etc. ")
    x = images_nhwc.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    if q is not None:
        x = x.to(q.dtype).to(x.dtype)                                     # the molded image is stored in the compute dtype
    if config.BACKBONE in ("resnet50", "resnet101"):
        c5 = resnet_graph(x, P, config.BACKBONE, config.TRAIN_BN, relu_hook, q)
    else:
        c5 = resnet_shallow_graph(x, P, config.BACKBONE, config.TRAIN_BN, relu_hook, q)
    c6 = _st(conv_bn(c5, P["bottleneck_layer"], None, config.TRAIN_BN, stride=2, padding="same", q=q), q)
    feat = c6.permute(0, 2, 3, 1).reshape(c6.shape[0], -1)                # (h,w,c) flatten [A4]
    if config.REGRESS_KEYPOINTS:                                          # outputs [k1, k2, k3] (net.py:678, 689)
        return build_loc_graph(feat, P, config, relu_hook, q), None
    return build_loc_graph(feat, P, config, relu_hook, q), build_ori_graph(feat, P, config, relu_hook, q)


# --------------------------------------------------------------------------
# losses (net.py:705-762) and compile() (net.py:973-1028)
# --------------------------------------------------------------------------
class _SoftmaxXentTF(torch.autograd.Function):
    """tf.nn.softmax_cross_entropy_with_logits as TF's kernel defines it (net.py:710 reaches it through tf.losses.softmax_cross_entropy):
    per-row loss = sum_k p_k (logsumexp(z) - z_k), and the gradient w.r.t. the logits is the kernel's `backprop` output
    softmax(z) - p -- NOT the derivative of the loss expression (softmax(z) * sum_k p_k - p) unless the labels sum to one.  encode_ori's
    soft labels are normalised (utils.py:333-380), so both forms agree on the reference's own data; the restatement follows the kernel
    so that un-normalised labels (sum != 1) are covered too."""

    @staticmethod
    def forward(ctx, z, p):
        lse = torch.logsumexp(z, dim=-1, keepdim=True)
        ctx.save_for_backward(torch.softmax(z, dim=-1) - p)
        return (p * (lse - z)).sum(-1)

    @staticmethod
    def backward(ctx, g):
        (bp,) = ctx.saved_tensors
        return g.unsqueeze(-1) * bp, None


def softmax_loss(y_gt, y_pred):
    """net.py:710 + [A6]: sum_b(-sum_k p log_softmax(z)) / B, gradient softmax - p per row (TF kernel semantics, see above)."""
    return _SoftmaxXentTF.apply(y_pred, y_gt).mean()


def one_minus_dot_prod(y_true, y_pred):
    """net.py:728-731."""
    return (1 - (y_true * y_pred).sum(-1, keepdim=True).abs()).mean()


def mse_loss(y_gt, y_pred):
    """net.py:741-746."""
    return ((y_gt - y_pred) ** 2).mean()


def rel_loss(y_gt, y_pred):
    """net.py:757 + [A7]: Frobenius norms over the WHOLE batch tensor."""
    return torch.linalg.vector_norm((y_gt - y_pred) / torch.linalg.vector_norm(y_gt))


def rel_loss_sharded(y_gt, y_pred, global_norms, world):
    """The data-parallel form of rel_loss for ONE shard of the global batch (exact mode): a surrogate whose
    value is world * (this shard's share of the global loss) and whose gradient w.r.t. y_pred is
    -world (gt - pred) / (||gt-pred||_global ||gt||_global); averaging the shards' gradients gives the global-batch gradient."""
    nd, ng = float(global_norms[0]) ** 0.5, float(global_norms[1]) ** 0.5
    return world * ((y_gt - y_pred) ** 2).sum() / (2.0 * nd * ng)


def rel_norms(y_gt, y_pred):
    return torch.stack([((y_gt - y_pred) ** 2).sum(), (y_gt ** 2).sum()]).detach()


def losses(P, images, gt_loc, gt_ori, config, relu_hook=None, rel_global=None, q=None):
    """net.py:656-669.  Returns (loc_pred, ori_pred, loc_loss, ori_loss).  rel_global = (global_norms, world) switches the
    location loss to its exact data-parallel shard form (rel_loss_sharded)."""
    loc, ori = forward(P, images, config, relu_hook, q)
    if config.REGRESS_KEYPOINTS:
        # net.py:657-659: three MSE losses; gt_loc = k1 target, gt_ori = (k2 target, k3 target).  Returned as
        # (k1, [k2, k3], loc_loss, [k2_loss, k3_loss]).
        k1, k2, k3 = loc
        return k1, [k2, k3], mse_loss(gt_loc, k1), [mse_loss(gt_ori[0], k2), mse_loss(gt_ori[1], k3)]
    if config.REGRESS_LOC and rel_global is not None:
        loc_loss = rel_loss_sharded(gt_loc, loc, rel_global[0], rel_global[1])
    else:
        loc_loss = rel_loss(gt_loc, loc) if config.REGRESS_LOC else softmax_loss(gt_loc, loc)
    ori_loss = one_minus_dot_prod(gt_ori, ori) if config.REGRESS_ORI else softmax_loss(gt_ori, ori)
    return loc, ori, loc_loss, ori_loss


def is_trainable(layer_name, layer_regex=".*"):
    """net.py:1057 -- fullmatch on the layer name."""
    return bool(re.fullmatch(layer_regex, layer_name))


def regularizer(P, config, layer_regex=".*"):
    """net.py:1008-1012 + [A9]: sum_w WD*sum(w^2)/numel(w) over trainable non-gamma/beta weights."""
    reg = 0.0
    for ln, ws in P.items():
        if not is_trainable(ln, layer_regex):
            continue
        for wn, w in ws.items():
            if wn in ("gamma", "beta", "moving_mean", "moving_variance"):
                continue
            reg = reg + config.WEIGHT_DECAY * (w * w).sum() / w.numel()
    return reg


def total_loss(P, images, gt_loc, gt_ori, config, layer_regex=".*", relu_hook=None, rel_global=None, q=None):
    """net.py:993-1012: sum_name LOSS_WEIGHTS[name]*mean(loss) + regulariser."""
    loc, ori, ll, ol = losses(P, images, gt_loc, gt_ori, config, relu_hook, rel_global, q)
    if config.REGRESS_KEYPOINTS:                                          # loss_names of net.py:989-990
        tot = (config.LOSS_WEIGHTS.get("loc_loss", 1.) * ll + config.LOSS_WEIGHTS.get("k2_loss", 1.) * ol[0] +
               config.LOSS_WEIGHTS.get("k3_loss", 1.) * ol[1])
    else:
        tot = config.LOSS_WEIGHTS.get("loc_loss", 1.) * ll + config.LOSS_WEIGHTS.get("ori_loss", 1.) * ol
    return tot + regularizer(P, config, layer_regex), (loc, ori, ll, ol)


def gradients(P, images, gt_loc, gt_ori, config, layer_regex=".*", relu_hook=None, rel_global=None, q=None):
    """Returns (grads {layer:{weight: tensor}}, (loc, ori, loc_loss, ori_loss), total)."""
    leaves = [(ln, wn, w) for ln, ws in P.items() for wn, w in ws.items()
              if w.requires_grad and is_trainable(ln, layer_regex)]
    tot, outs = total_loss(P, images, gt_loc, gt_ori, config, layer_regex, relu_hook, rel_global, q)
    gs = torch.autograd.grad(tot, [w for _, _, w in leaves], allow_unused=True)
    grads = OrderedDict()
    for (ln, wn, w), g in zip(leaves, gs):
        grads.setdefault(ln, OrderedDict())[wn] = g if g is not None else torch.zeros_like(w)
    det = lambda o: o.detach() if torch.is_tensor(o) else ([t.detach() for t in o] if isinstance(o, (list, tuple)) else o)
    return grads, tuple(det(o) for o in outs), tot.detach()


def global_norm(grads):
    s = 0.0
    for ws in grads.values():
        for g in ws.values():
            s = s + (g.double() ** 2).sum()
    return float(torch.sqrt(s))


def sgd_step(P, grads, velocity, lr, momentum, clipnorm):
    """keras.optimizers.SGD(lr, momentum, clipnorm) [A10]: GLOBAL norm clip
    (g <- g*clipnorm/norm if norm >= clipnorm), v <- m*v - lr*g, w <- w + v.  In place.
    Returns the pre-clip global norm."""
    norm = global_norm(grads)
    scale = clipnorm / norm if (clipnorm and clipnorm > 0 and norm >= clipnorm) else 1.0
    with torch.no_grad():
        for ln, ws in grads.items():
            for wn, g in ws.items():
                v = velocity.setdefault(ln, {}).setdefault(wn, torch.zeros_like(g))
                v.mul_(momentum).add_(g * scale, alpha=-lr)
                P[ln][wn].add_(v)
    return norm


def adam_step(P, grads, state, lr, clipnorm, epsilon=1e-7, beta_1=0.9, beta_2=0.999):
    """keras.optimizers.Adam(lr, amsgrad=True, clipnorm) as net.py:982-983 builds it [Keras 2.x Adam.get_updates]:
    global-norm clip; t = iterations + 1; lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
    vhat = max(vhat, v); p -= lr_t * m / (sqrt(vhat) + eps).  In place; `state` carries t, m, v, vhat."""
    norm = global_norm(grads)
    scale = clipnorm / norm if (clipnorm and clipnorm > 0 and norm >= clipnorm) else 1.0
    t = state["t"] = state.get("t", 0) + 1
    lr_t = lr * (math.sqrt(1.0 - beta_2 ** t) / (1.0 - beta_1 ** t))
    with torch.no_grad():
        for ln, ws in grads.items():
            for wn, g in ws.items():
                g = g * scale
                st = state.setdefault((ln, wn), {"m": torch.zeros_like(g), "v": torch.zeros_like(g), "vhat": torch.zeros_like(g)})
                st["m"] = beta_1 * st["m"] + (1.0 - beta_1) * g
                st["v"] = beta_2 * st["v"] + (1.0 - beta_2) * g * g
                st["vhat"] = torch.maximum(st["vhat"], st["v"])
                P[ln][wn].sub_(lr_t * st["m"] / (torch.sqrt(st["vhat"]) + epsilon))
    return norm


def train_step(P, velocity, images, gt_loc, gt_ori, config, lr, layer_regex=".*", relu_hook=None, q=None):
    """One fit_generator step [A13]: fwd, loss, bwd, clip, update.  Returns dict of scalars/outputs."""
    grads, (loc, ori, ll, ol), tot = gradients(P, images, gt_loc, gt_ori, config, layer_regex, relu_hook, q=q)
    if str(getattr(config, "OPTIMIZER", "SGD")).upper() == "SGD":
        norm = sgd_step(P, grads, velocity, lr, config.LEARNING_MOMENTUM, config.GRADIENT_CLIP_NORM)
    else:
        norm = adam_step(P, grads, velocity, lr, config.GRADIENT_CLIP_NORM, 1e-4 if getattr(config, "F16", False) else 1e-7)
    if config.REGRESS_KEYPOINTS:
        return {"loc": loc, "k2": ori[0], "k3": ori[1], "loc_loss": float(ll), "k2_loss": float(ol[0]), "k3_loss": float(ol[1]),
                "total": float(tot), "grad_norm": norm, "grads": grads}
    return {"loc": loc, "ori": ori, "loc_loss": float(ll), "ori_loss": float(ol), "total": float(tot),
            "grad_norm": norm, "grads": grads}


def mold_image(image, config):
    """net.py:1337-1348."""
    dt = np.float16 if config.F16 else np.float32
    if image.shape[-1] == 3:
        return image.astype(dt) - config.MEAN_PIXEL
    return image.astype(dt) - np.mean(config.MEAN_PIXEL)


def algorithmic_flops(config, batch):
    """2*MACs over conv+dense; bwd = dgrad + wgrad = 2x fwd, minus the
    stem's dgrad.  Returns (fwd_flops, fwd_bwd_flops) for `batch` images."""
    h, w = int(config.IMAGE_SHAPE[0]), int(config.IMAGE_SHAPE[1])
    total = 0
    stem = 0
    cur = None
    for name, kind, ws in layer_specs(config):
        if kind == "conv":
            kh, kw, ci, co = ws["kernel"]
            if name in ("conv1", "conv0"):
                m = (h // 2) * (w // 2) * kh * kw * ci * co
                stem = m
                cur = (h // 4, w // 4)                         # after the 3x3/s2 max-pool
            elif name == "bottleneck_layer":
                m = (cur[0] // 2) * (cur[1] // 2) * kh * kw * ci * co
            else:
                # the first strided conv of a block (in spec order) halves the running size
                if re.fullmatch(r"res[345]a_branch2a", name) or re.fullmatch(r"stage[234]_unit1_sc", name):
                    cur = (cur[0] // 2, cur[1] // 2)
                m = cur[0] * cur[1] * kh * kw * ci * co
            total += m
        elif kind == "dense":
            i_, o_ = ws["kernel"]
            total += i_ * o_
    fwd = 2 * total * batch
    return fwd, 3 * fwd - 2 * stem * batch
