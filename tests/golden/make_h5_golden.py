#!/usr/bin/env python3
"""Generates tests/golden/keras_weights.h5, keras_model.h5 and keras_weights_expected.npz WITH THE REAL h5py, laid out the way Keras 2.1.6-2.2.4
writes them (keras/engine/saving.py save_weights_to_hdf5_group / save_model: root attributes `layer_names`, `backend`, `keras_version`; one
group per layer with `weight_names`; datasets named '<layer>/<weight>:0' inside the layer's group; a full-model file keeps the same tree under
`model_weights`).  Run with an interpreter that has h5py -- in the build image: /opt/conda/bin/python3.9 tests/golden/make_h5_golden.py --
ursonet_amd/h5lite.py (ctypes on libhdf5, no h5py) must read these files value for value (tests/test_h5_cpu.py).

The layer set is a miniature of what the reference's files hold (net.py:98-116, 170-172, 302-350): conv kernels HWIO + bias, BatchNormalization
gamma / beta / moving_mean / moving_variance, a bias-free shallow-trunk conv, Dense kernels [in, out] + bias, a layer without weights."""
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20260930)
f32 = lambda *s: rng.standard_normal(s).astype(np.float32)
LAYERS = [
    ("conv1", [("kernel", f32(7, 7, 3, 8)), ("bias", f32(8))]),
    ("bn_conv1", [("gamma", f32(8)), ("beta", f32(8)), ("moving_mean", f32(8)), ("moving_variance", np.abs(f32(8)) + 0.5)]),
    ("activation_1", []),                                   # layers without weights are listed too (empty weight_names)
    ("res2a_branch2a", [("kernel", f32(1, 1, 8, 4)), ("bias", f32(4))]),
    ("stage1_unit1_conv1", [("kernel", f32(3, 3, 4, 4))]),  # shallow trunk: use_bias=False
    ("bottleneck_layer", [("kernel", f32(3, 3, 4, 2)), ("bias", f32(2))]),
    ("loc_dense_0", [("kernel", f32(16, 5)), ("bias", f32(5))]),
    ("ori_final", [("kernel", f32(5, 64)), ("bias", f32(64))]),
]


def save_weights_to_hdf5_group(f, vlen_attrs):
    enc = (lambda names: np.array([n.encode("utf8") for n in names], dtype=h5py.string_dtype()) if vlen_attrs
           else np.array([n.encode("utf8") for n in names], dtype="S"))       # Keras: np.array of bytes -> fixed-length 'S'; newer h5py users: vlen
    f.attrs["layer_names"] = enc([ln for ln, _ in LAYERS])
    f.attrs["backend"] = "tensorflow" if vlen_attrs else b"tensorflow"          # (a str becomes a variable-length string, bytes a fixed-length one)
    f.attrs["keras_version"] = "2.2.4" if vlen_attrs else b"2.2.4"
    for ln, ws in LAYERS:
        g = f.create_group(ln)
        names = ["%s/%s:0" % (ln, wn) for wn, _ in ws]
        g.attrs["weight_names"] = enc(names) if names else np.zeros((0,), dtype="S1")
        for n, (_, a) in zip(names, ws):
            d = g.create_dataset(n, a.shape, dtype=a.dtype)
            d[...] = a


if len(sys.argv) == 1:                                                      # (no argument: regenerate the fixtures)
    with h5py.File(os.path.join(HERE, "keras_weights.h5"), "w") as f:          # model.save_weights(): what the reference's ModelCheckpoint writes
        save_weights_to_hdf5_group(f, vlen_attrs=False)
    with h5py.File(os.path.join(HERE, "keras_model.h5"), "w") as f:            # model.save(): the weights live under model_weights (net.py:830-832)
        f.attrs["model_config"] = '{"class_name": "Model"}'
        save_weights_to_hdf5_group(f.create_group("model_weights"), vlen_attrs=True)
    np.savez(os.path.join(HERE, "keras_weights_expected.npz"), **{"%s/%s" % (ln, wn): a for ln, ws in LAYERS for wn, a in ws})
    print("h5py", h5py.__version__, "hdf5", h5py.version.hdf5_version, "->", [p for p in sorted(os.listdir(HERE)) if p.startswith("keras_")])
if len(sys.argv) > 1:                                                       # `make_h5_golden.py --dump file.h5`: what the real h5py reads out of a file
    with h5py.File(sys.argv[-1], "r") as f:
        g = f["model_weights"] if "layer_names" not in f.attrs else f
        dec = lambda n: n.decode("utf8") if isinstance(n, bytes) else str(n)
        out = {}
        for ln in [dec(n) for n in g.attrs["layer_names"]]:
            for wn in [dec(n) for n in g[ln].attrs["weight_names"]]:
                out["%s/%s" % (ln, wn.split("/")[-1].split(":")[0])] = np.asarray(g[ln][wn])
        np.savez(sys.argv[-1] + ".dump.npz", **out)
        print("backend", dec(g.attrs["backend"]), "keras_version", dec(g.attrs["keras_version"]), len(out), "arrays")
