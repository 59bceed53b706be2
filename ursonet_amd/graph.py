"""Static description of the UrsoNet graph (topology, Keras layer names, parameter shapes).

Pure Python (no device code) so it is unit-testable on CPU.  Follows the reference's graph
builders: resnet_graph net.py:161-199, conv_block/identity_block net.py:85-158,
resnet_shallow_graph/residual_basic_block net.py:208-282, bottleneck_layer net.py:639-640,
build_loc_graph/build_ori_graph net.py:288-352.  The layer names and weight layouts are the
"Keras-compatible layer/weight API" (conv kernel HWIO + bias, BN gamma/beta/moving_mean/
moving_variance, dense kernel [in,out] + bias) so HDF5/npz weights load by name.
"""
from collections import OrderedDict

BN_EPS = 1e-3        # keras.layers.BatchNormalization default epsilon


class TensorSpec(object):
    def __init__(self, tid, h, w, c, relu):
        self.id, self.h, self.w, self.c, self.relu = tid, h, w, c, relu   # relu: values are post-ReLU (>= 0)

    def __repr__(self):
        return "T%d[%dx%dx%d%s]" % (self.id, self.h, self.w, self.c, ",relu" if self.relu else "")


class ConvSpec(object):
    """One Conv2D/Dense (+ the BatchNorm that follows it, + Add, + ReLU) = one fused kernel launch."""
    op = "conv"

    def __init__(self, name, bn, src, dst, kh, kw, cin, cout, stride, pad, bias, relu, residual=None,
                 dense=False, out_f32=False, stem=False):
        self.name, self.bn, self.src, self.dst = name, bn, src, dst
        self.kh, self.kw, self.cin, self.cout = kh, kw, cin, cout
        self.stride, self.pad = stride, pad            # pad = (top, left); bottom/right follow from the output size
        self.bias, self.relu, self.residual = bias, relu, residual
        self.dense, self.out_f32, self.stem = dense, out_f32, stem


class PoolSpec(object):
    op = "pool"

    def __init__(self, src, dst):
        self.src, self.dst = src, dst


class Graph(object):
    def __init__(self):
        self.tensors, self.nodes = [], []
        self.params = OrderedDict()      # layer name -> OrderedDict(weight name -> shape)
        self.kinds = OrderedDict()       # layer name -> "conv" | "bn" | "dense"
        self.outputs = {}                # "loc" / "ori" / "k1".. -> tensor id
        self.feat = None

    def tensor(self, h, w, c, relu):
        t = TensorSpec(len(self.tensors), h, w, c, relu)
        self.tensors.append(t)
        return t

    def conv(self, name, bn, src, cout, k, stride=1, pad=(0, 0), out_hw=None, bias=True, relu=False, residual=None,
             dense=False, out_f32=False, stem=False, cin=None, kin=None):
        kh = kw = k
        cin = cin if cin is not None else src.c
        if dense:
            oh = ow = 1
        elif out_hw is not None:
            oh, ow = out_hw
        else:
            oh = (src.h + 2 * pad[0] - kh) // stride + 1
            ow = (src.w + 2 * pad[1] - kw) // stride + 1
        dst = self.tensor(oh, ow, cout, relu)
        shape = (kin, cout) if dense else (kh, kw, cin, cout)
        self.params[name] = OrderedDict([("kernel", shape)] + ([("bias", (cout,))] if bias else []))
        self.kinds[name] = "dense" if dense else "conv"
        if bn:
            self.params[bn] = OrderedDict((n, (cout,)) for n in ("gamma", "beta", "moving_mean", "moving_variance"))
            self.kinds[bn] = "bn"
        self.nodes.append(ConvSpec(name, bn, src, dst, kh, kw, kin if dense else cin, cout, stride, pad, bias, relu,
                                   residual, dense, out_f32, stem))
        return dst


def _same_pad_before(n_in, k, s):
    out = -(-n_in // s)
    total = max((out - 1) * s + k - n_in, 0)
    return total // 2


def build_graph(config):
    """Returns the Graph for `config` (H, W from config.IMAGE_SHAPE)."""
    H, W = int(config.IMAGE_SHAPE[0]), int(config.IMAGE_SHAPE[1])
    if H / 2 ** 6 != int(H / 2 ** 6) or W / 2 ** 6 != int(W / 2 ** 6):          # net.py:596-600
        raise Exception("Image size must be dividable by 2 at least 6 times "
                        "to avoid fractions when downscaling and upscaling."
                        "For example, use 256, 320, 384, 448, 512, ... etc. ")
    if config.NR_IMAGE_CHANNELS != 3:
        raise NotImplementedError("only NR_IMAGE_CHANNELS == 3 is supported (config.py:78)")
    if config.TRAIN_BN not in (False, None):
        raise NotImplementedError("TRAIN_BN=%r: the frozen mode (False, config.py:146) and the batch-statistics mode (None) are "
                                  "implemented; True (\"don't use\", net.py:73) adds BN layers to the heads and is not" % (config.TRAIN_BN,))
    g = Graph()
    img = g.tensor(H, W, 3, False)
    deep = config.BACKBONE in ("resnet50", "resnet101")
    if not deep and config.BACKBONE not in ("resnet18", "resnet34"):
        raise AssertionError("unsupported BACKBONE %r" % (config.BACKBONE,))
    # ---- stem: ZeroPadding2D(3) + 7x7/s2 conv + BN + ReLU + MaxPool 3x3/s2 SAME
    stem_name, stem_bn = ("conv1", "bn_conv1") if deep else ("conv0", "bn_conv0")
    x = g.conv(stem_name, stem_bn, img, 64, 7, stride=2, pad=(3, 3), bias=deep, relu=True, stem=True)
    p = g.tensor(x.h // 2, x.w // 2, 64, True)
    g.nodes.append(PoolSpec(x, p))
    x = p
    if deep:
        n4 = {"resnet50": 5, "resnet101": 22}[config.BACKBONE]                     # net.py:188
        stages = [(2, ["a", "b", "c"], (64, 64, 256)), (3, ["a", "b", "c", "d"], (128, 128, 512)),
                  (4, ["a"] + [chr(98 + i) for i in range(n4)], (256, 256, 1024)), (5, ["a", "b", "c"], (512, 512, 2048))]
        for stage, blocks, (f1, f2, f3) in stages:
            for b in blocks:
                cb, bb = "res%d%s_branch" % (stage, b), "bn%d%s_branch" % (stage, b)
                s = 2 if (b == "a" and stage > 2) else 1
                a1 = g.conv(cb + "2a", bb + "2a", x, f1, 1, stride=s, relu=True)
                a2 = g.conv(cb + "2b", bb + "2b", a1, f2, 3, pad=(1, 1), relu=True)          # 'same', stride 1
                if b == "a":                                                                 # conv_block net.py:120-158
                    sc = g.conv(cb + "1", bb + "1", x, f3, 1, stride=s, relu=False)
                    x = g.conv(cb + "2c", bb + "2c", a2, f3, 1, relu=True, residual=sc)
                else:                                                                        # identity_block net.py:85-117
                    x = g.conv(cb + "2c", bb + "2c", a2, f3, 1, relu=True, residual=x)
    else:
        reps = [2, 2, 2, 2] if config.BACKBONE == "resnet18" else [3, 4, 6, 3]
        for stage, rep in enumerate(reps):
            f = 64 * 2 ** stage
            for block in range(rep):
                nb = "stage%d_unit%d_" % (stage + 1, block + 1)                              # net.py:208-214
                s = 2 if (block == 0 and stage > 0) else 1
                if block == 0:                                                               # cut='post' net.py:225
                    sc = g.conv(nb + "sc", None, x, f, 1, stride=s, bias=False, relu=False)
                else:
                    sc = x
                a1 = g.conv(nb + "conv1", nb + "bn2", x, f, 3, stride=s, pad=(1, 1), bias=False, relu=True)
                x = g.conv(nb + "conv2", None, a1, f, 3, pad=(1, 1), bias=False, relu=True, residual=sc)
    # ---- bottleneck_layer: Conv 3x3 s2 padding='SAME' (TF: pad before = 0 for even inputs), linear
    bw = int(config.BOTTLENECK_WIDTH)
    c6 = g.conv("bottleneck_layer", None, x, bw, 3, stride=2,
                pad=(_same_pad_before(x.h, 3, 2), _same_pad_before(x.w, 3, 2)), out_hw=(-(-x.h // 2), -(-x.w // 2)))
    nf = int(bw * H * W / (64 ** 2))                                                          # net.py:640
    assert nf == c6.h * c6.w * c6.c, (nf, c6)
    g.feat = c6
    nd = int(config.NR_DENSE_LAYERS)
    assert nd in range(3)                                                                     # net.py:295

    def trunk(prefix):
        t, kin = c6, nf
        for i in range(nd):
            t = g.conv("%s_dense_%d" % (prefix, i), None, t, int(config.BRANCH_SIZE), 1, relu=True, dense=True, kin=kin)
            kin = int(config.BRANCH_SIZE)
        return t, kin
    t, kin = trunk("loc")
    if config.REGRESS_KEYPOINTS:
        for k in ("k1", "k2", "k3"):
            g.outputs[k] = g.conv(k + "_final", None, t, 3, 1, dense=True, kin=kin, out_f32=True)
        g.outputs["loc"] = g.outputs["k1"]
    elif config.REGRESS_LOC:
        g.outputs["loc"] = g.conv("loc_final", None, t, 3, 1, dense=True, kin=kin, out_f32=True)
    else:
        g.outputs["loc"] = g.conv("loc_final", None, t, int(config.LOC_BINS_PER_DIM) ** 3, 1, relu=True, dense=True,
                                  kin=kin, out_f32=True)
    if config.REGRESS_KEYPOINTS:
        # the keypoint model's outputs are [k1, k2, k3] only (net.py:675-691): ori_pred is built by the reference but is not a
        # Model output, so Keras prunes the whole ori_* branch -- no ori layers, weights or gradients exist in this mode
        return g
    t, kin = trunk("ori")
    if config.REGRESS_ORI:
        if config.ORIENTATION_PARAM == "quaternion":
            g.outputs["ori"] = g.conv("ori_q", None, t, 4, 1, dense=True, kin=kin, out_f32=True)
        else:
            g.outputs["ori"] = g.conv("ori_final", None, t, 3, 1, dense=True, kin=kin, out_f32=True)
    else:
        g.outputs["ori"] = g.conv("ori_final", None, t, int(config.ORI_BINS_PER_DIM) ** 3, 1, relu=True, dense=True,
                                  kin=kin, out_f32=True)
    return g


def layer_regex(layers):
    """Pre-defined trainable-layer selections of UrsoNet.train (net.py:1086-1097)."""
    table = {
        "heads": r"(ori\_.*)|(loc\_.*)|(fpn\_.*)|(bottleneck_layer)",
        "3+": r"(res3.*)|(bn3.*)|(res4.*)|(bn4.*)|(res5.*)|(bn5.*)|(loc\_.*)|(ori\_.*)|(fpn\_.*)|(bottleneck_layer)",
        "4+": r"(res4.*)|(bn4.*)|(res5.*)|(bn5.*)|(loc\_.*)|(ori\_.*)|(fpn\_.*)|(bottleneck_layer)",
        "5+": r"(res5.*)|(bn5.*)|(loc\_.*)|(ori\_.*)|(fpn\_.*)|(bottleneck_layer)",
        "all": ".*",
    }
    return table.get(layers, layers)


def conv_flops(graph, batch):
    """Algorithmic FLOPs (2*MACs over conv + dense layers): (forward, forward+backward) for `batch` images.
    Backward = dgrad + wgrad = 2x forward, minus the stem's data gradient."""
    fwd = 0
    stem = 0
    for n in graph.nodes:
        if n.op != "conv":
            continue
        m = n.dst.h * n.dst.w * n.kh * n.kw * n.cin * n.cout if not n.dense else n.cin * n.cout
        fwd += m
        if n.stem:
            stem = m
    return 2 * fwd * batch, (6 * fwd - 2 * stem) * batch
