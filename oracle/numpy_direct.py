"""Oracle (test infrastructure): second, independent realisation of the graph ops as direct NumPy
loops over NHWC arrays, for TINY shapes only.  It exists so that oracle/graph_ref.py (torch-CPU
functional ops) can be cross-checked before either is trusted as the checker: the
TF/Keras arithmetic itself is not importable here ("parity unpinned", see oracle/__init__.py).
Each function restates the Keras/TF semantics listed in SURVEY.md Appendix A."""
import numpy as np


def same_pad(n_in, k, s):
    """[A2] TF SAME: out = ceil(in/s); pad_total = max((out-1)*s + k - in, 0); before = total//2."""
    out = -(-n_in // s)
    total = max((out - 1) * s + k - n_in, 0)
    return total // 2, total - total // 2


def conv2d(x, w, b=None, stride=1, padding="valid"):
    """[A1] x [B,H,W,C], w HWIO, y[b,oy,ox,n] = sum x[b, oy*s+ky-pt, ox*s+kx-pl, c] * w[ky,kx,c,n] + b[n]."""
    B, H, W, C = x.shape
    kh, kw, _, N = w.shape
    if padding == "same":
        (pt, pb), (pl, pr) = same_pad(H, kh, stride), same_pad(W, kw, stride)
    elif isinstance(padding, int):
        pt = pb = pl = pr = padding
    else:
        pt = pb = pl = pr = 0
    xp = np.zeros((B, H + pt + pb, W + pl + pr, C), dtype=np.float64)
    xp[:, pt:pt + H, pl:pl + W] = x
    OH = (H + pt + pb - kh) // stride + 1
    OW = (W + pl + pr - kw) // stride + 1
    y = np.zeros((B, OH, OW, N), dtype=np.float64)
    for oy in range(OH):
        for ox in range(OW):
            patch = xp[:, oy * stride:oy * stride + kh, ox * stride:ox * stride + kw, :]       # [B,kh,kw,C]
            y[:, oy, ox, :] = np.tensordot(patch, w.astype(np.float64), axes=([1, 2, 3], [0, 1, 2]))
    if b is not None:
        y += b
    return y


def batchnorm_inference(x, gamma, beta, mean, var, eps=1e-3):
    """[A3] (x - moving_mean) / sqrt(moving_var + eps) * gamma + beta."""
    return (x - mean) / np.sqrt(var + eps) * gamma + beta


def relu(x):
    return np.maximum(x, 0)


def maxpool_3x3_s2_same(x):
    """[A2] MaxPooling2D((3,3), strides 2, 'same'): windows clipped at the border (padding never wins)."""
    B, H, W, C = x.shape
    (pt, _), (pl, _) = same_pad(H, 3, 2), same_pad(W, 3, 2)
    OH, OW = -(-H // 2), -(-W // 2)
    y = np.full((B, OH, OW, C), -np.inf)
    for oy in range(OH):
        for ox in range(OW):
            y0, x0 = oy * 2 - pt, ox * 2 - pl
            y[:, oy, ox] = x[:, max(y0, 0):min(y0 + 3, H), max(x0, 0):min(x0 + 3, W)].max(axis=(1, 2))
    return y


def dense(x, w, b):
    return x @ w + b                                                                           # [A4]


def l2_normalize(x):
    return x / np.sqrt(np.maximum((x * x).sum(-1, keepdims=True), 1e-12))                      # [A5]


def softmax_xent(labels, logits):
    """[A6] mean_b( -sum_k labels * log_softmax(logits) )."""
    z = logits - logits.max(-1, keepdims=True)
    lsm = z - np.log(np.exp(z).sum(-1, keepdims=True))
    return float((-(labels * lsm).sum(-1)).mean())


def rel_loss(gt, pred):
    return float(np.sqrt(((gt - pred) ** 2).sum()) / np.sqrt((gt ** 2).sum()))                 # [A7]


def one_minus_dot(gt, pred):
    return float((1 - np.abs((gt * pred).sum(-1))).mean())


def sgd_clipnorm(ws, gs, vs, lr, momentum, clipnorm):
    """[A10] global-norm clip then v = m*v - lr*g; w += v (in place on lists of arrays)."""
    norm = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in gs))
    c = clipnorm / norm if (clipnorm and norm >= clipnorm) else 1.0
    for w, g, v in zip(ws, gs, vs):
        v *= momentum
        v -= lr * c * g
        w += v
    return norm
