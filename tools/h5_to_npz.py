#!/usr/bin/env python3
"""Offline converter for Keras HDF5 weight files: run it on ANY machine that has h5py (the GPU image has none) and carry the .npz to the
MI355X box.  Reads what the reference's load_weights reads (net.py:816-852: `layer_names` / `weight_names` attributes, the nested
`model_weights` group of a full-model file, net.py:830-832) -- the released UrsoNet / ImageNet files of net.py:854-940 included -- and
writes the `.npz` twin UrsoNet.load_weights() of ursonet_amd takes: one array per `<layer>/<weight>` key (kernel, bias, gamma, beta,
moving_mean, moving_variance), values and shapes untouched (HWIO conv kernels, [in, out] dense kernels).

    python tools/h5_to_npz.py weights_soyuz_hard.h5 [out.npz] [--list]

The only imports are numpy and h5py; the reader is the one ursonet_amd/net.py uses when h5py is present (read_keras_h5 below is kept
free of package imports so that this file can be copied alone to the machine that holds the .h5)."""
import sys
from collections import OrderedDict

import numpy as np


def read_keras_h5(path):
    """-> OrderedDict {layer: OrderedDict {weight: array}} in the file's own layer order."""
    try:
        import h5py
    except ImportError:                                                   # no h5py here: the HDF5 C library through ctypes, when this file sits in the repository
        import os
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from ursonet_amd import h5lite
        return h5lite.read_keras_weights(path)
    out = OrderedDict()
    dec = lambda n: n.decode("utf8") if isinstance(n, bytes) else str(n)
    with h5py.File(path, mode="r") as f:
        if "layer_names" not in f.attrs and "model_weights" in f:          # a model.save() file: weights live one group down (net.py:831-832)
            f = f["model_weights"]
        for ln in [dec(n) for n in f.attrs["layer_names"]]:
            grp = f[ln]
            for wn in [dec(n) for n in grp.attrs["weight_names"]]:
                short = wn.split("/")[-1].split(":")[0]                   # 'res2a_branch2a/kernel:0' -> 'kernel'
                out.setdefault(ln, OrderedDict())[short] = np.asarray(grp[wn])
    return out


def main(argv):
    args = [a for a in argv if not a.startswith("--")]
    if not args:
        print(__doc__)
        return 2
    src = args[0]
    dst = args[1] if len(args) > 1 else (src[:-3] if src.endswith(".h5") else src) + ".npz"
    params = read_keras_h5(src)
    if "--list" in argv:
        for ln, ws in params.items():
            print(ln, {wn: tuple(a.shape) for wn, a in ws.items()})
    flat = {"%s/%s" % (ln, wn): a for ln, ws in params.items() for wn, a in ws.items()}
    np.savez(dst, **flat)
    print("%s: %d layers with weights, %d arrays, %.1f M parameters -> %s" % (
        src, len(params), len(flat), sum(a.size for a in flat.values()) / 1e6, dst))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
