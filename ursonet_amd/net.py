"""Drop-in for the reference's `net.py` model API (class UrsoNet, net.py:566-1308) on MI355X.

`pose_estimator.py` programs against: UrsoNet(mode, config, model_dir), .train(), .detect(),
.load_weights(), .find_last(), .get_last_checkpoint(), .get_imagenet_weights(), .get_urso_weights(),
.compile(), .set_trainable(), attributes .config .epoch .log_dir .checkpoint_path .keras_model, and
the module functions load_image_gt / data_generator / mold_image / compose_image_meta.  All of that
is provided here with the same names, argument meaning and error behaviour; the numeric work is
done by ursonet_amd.engine.Engine (HIP kernels through the C ABI) -- there is no Keras/TF and no
CPU fallback: building a model without an MI355X raises.

Weights files: Keras HDF5 (`.h5`, by-name, nested `model_weights` groups handled) through h5py or, where this interpreter has none,
through the HDF5 C library itself (ursonet_amd/h5lite.py: ctypes on libhdf5.so, checked against files written by the real h5py);
always also a `.npz` twin with the same layer/weight names and layouts ("<layer>/<weight>" keys).
"""
import datetime
import logging
import os
import re
from collections import OrderedDict

import numpy as np

from . import utils
from .graph import build_graph, layer_regex as _layer_regex


def log(text, array=None):
    """net.py:46-57."""
    if array is not None:
        text = text.ljust(25)
        text += ("shape: {:20}  min: {:10.5f}  max: {:10.5f}  {}".format(
            str(array.shape), array.min() if array.size else "", array.max() if array.size else "", array.dtype))
    print(text)


############################################################
#  Keras-model shim: layer names / weights / trainable flags
############################################################
class _Layer(object):
    def __init__(self, model, name, kind, weight_names):
        self._model, self.name, self.kind = model, name, kind
        self.weight_names = list(weight_names)
        self.trainable = True

    @property
    def weights(self):
        return ["%s/%s:0" % (self.name, w) for w in self.weight_names]

    def get_weights(self):
        eng = self._model._engine
        return [eng.wview(self.name, w).detach().cpu().numpy().copy() for w in self.weight_names]

    def set_weights(self, arrays):
        self._model._engine.set_weights({self.name: dict(zip(self.weight_names, arrays))}, strict=False)


class KerasModelShim(object):
    """What net.py touches on `keras_model`: .layers, .get_layer(name), .predict(), weight access."""

    def __init__(self, owner, graph):
        self._owner = owner
        self.name = "urso_net"
        self.layers = [_Layer(self, n, graph.kinds[n], ws.keys()) for n, ws in graph.params.items()]
        self._by_name = {l.name: l for l in self.layers}
        self.metrics_names = ["loss", "loc_loss", "ori_loss"]

    @property
    def _engine(self):
        return self._owner._engine

    def get_layer(self, name):
        return self._by_name[name]

    def predict(self, molded_images, verbose=0):
        return self._owner._predict(np.asarray(molded_images))


############################################################
#  Data formatting (net.py:1314-1355)
############################################################
def compose_image_meta(image_id, original_image_shape, image_shape, window, scale):
    return np.array([image_id] + list(original_image_shape) + list(image_shape) + list(window) + [scale])


def mold_image(image, config):
    dt = np.float16 if config.F16 else np.float32
    if image.shape[-1] == 3:
        return image.astype(dt) - config.MEAN_PIXEL
    return image.astype(dt) - np.mean(config.MEAN_PIXEL)


def unmold_image(normalized_images, config):
    return (normalized_images + config.MEAN_PIXEL).astype(np.uint8)


############################################################
#  Data generator (net.py:358-559)
############################################################
def load_image_gt(dataset, config, image_id):
    """Image + pose targets for one sample (net.py:363-461): raw sample -> augmentation (sim2real stages and the camera /
    in-plane rotation warps run on the GPU, same NumPy global-RNG draws as the reference) -> resize / pad -> image_meta.
    Returns (image, image_meta, loc, ori), or (image, image_meta, loc, k1, k2) in keypoint mode."""
    from . import feeder
    s = feeder.augment_samples([feeder.load_sample(dataset, config, image_id)], dataset, config)[0]
    image, image_meta = feeder.finish_sample(s, config)
    if config.REGRESS_KEYPOINTS:
        return image, image_meta, s.loc, np.asarray(s.k1).T, np.asarray(s.k2).T
    return image, image_meta, s.loc, s.ori


def data_generator(dataset, config, shuffle=True, batch_size=1):
    """net.py:463-559: yields ([molded images, image_meta, gt_loc, gt_ori], []) forever (keypoint mode: gt_loc, gt_k1, gt_k2);
    tolerates up to 5 failing samples.  Built on ursonet_amd.feeder (pre-allocated per-field batch arrays, batched augmentation)."""
    from . import feeder
    for asm in feeder.batches(dataset, config, shuffle, batch_size, molded=True):
        yield asm.inputs(), []


############################################################
#  Weight files
############################################################
def read_weights_file(path):
    """-> {layer: {weight: array}} from a Keras HDF5 file (h5py needed) or an .npz twin."""
    if path.endswith(".npz"):
        out = OrderedDict()
        with np.load(path) as z:
            for k in z.files:
                ln, wn = k.rsplit("/", 1)
                out.setdefault(ln, OrderedDict())[wn] = z[k]
        return out
    try:
        import h5py
    except ImportError:
        # no h5py for this interpreter: the HDF5 C library itself through ctypes (ursonet_amd/h5lite.py: same files, same library h5py binds)
        from . import h5lite
        if not h5lite.available():
            raise ImportError("`load_weights` of a Keras .h5 file needs h5py or an HDF5 library (libhdf5.so; $URSO_HDF5_LIB): neither found -- "
                              "use the .npz twin (tools/h5_to_npz.py converts on any machine that has one)")
        return h5lite.read_keras_weights(path)
    out = OrderedDict()
    with h5py.File(path, mode='r') as f:
        if 'layer_names' not in f.attrs and 'model_weights' in f:          # net.py:831-832
            f = f['model_weights']
        for ln in [n.decode('utf8') if isinstance(n, bytes) else n for n in f.attrs['layer_names']]:
            grp = f[ln]
            names = [n.decode('utf8') if isinstance(n, bytes) else n for n in grp.attrs['weight_names']]
            for wn in names:
                short = wn.split("/")[-1].split(":")[0]
                out.setdefault(ln, OrderedDict())[short] = np.asarray(grp[wn])
    return out


def write_weights_file(path, params):
    """Always writes `<path without ext>.npz`; additionally a Keras-layout .h5 when h5py exists."""
    base = path[:-3] if path.endswith(".h5") else (path[:-4] if path.endswith(".npz") else path)
    flat = {"%s/%s" % (ln, wn): a for ln, ws in params.items() for wn, a in ws.items()}
    np.savez(base + ".npz", **flat)
    written = [base + ".npz"]
    if path.endswith(".h5"):
        try:
            import h5py
        except ImportError:
            from . import h5lite
            if h5lite.available():                      # the HDF5 library without h5py: the same Keras layout (keras save_weights_to_hdf5_group)
                h5lite.write_keras_weights(path, params)
                written.append(path)
            return written
        with h5py.File(path, "w") as f:
            f.attrs['layer_names'] = [ln.encode('utf8') for ln in params]
            f.attrs['backend'] = b'tensorflow'
            f.attrs['keras_version'] = b'2.2.4'
            for ln, ws in params.items():
                g = f.create_group(ln)
                names = ["%s/%s:0" % (ln, wn) for wn in ws]
                g.attrs['weight_names'] = [n.encode('utf8') for n in names]
                for n, a in zip(names, ws.values()):
                    g.create_dataset(n, data=a)
        written.append(path)
    return written


class BatchLogger(object):
    """net.py:1106-1115 -- per-batch loss history returned by train()."""

    def __init__(self):
        self.ori_loss_acc, self.loc_loss_acc = [], []


############################################################
#  UrsoNet
############################################################
class UrsoNet(object):
    def __init__(self, mode, config, model_dir, build_engine=True):
        assert mode in ['training', 'inference']
        self.mode, self.config, self.model_dir = mode, config, model_dir
        self.set_log_dir()
        self._engine = None
        self._layer_regex = ".*"
        self.keras_model = self.build(mode=mode, config=config, build_engine=build_engine)

    def build(self, mode, config, build_engine=True):
        """net.py:581-699.  Validates the config exactly like the reference (image divisibility
        exception, FC-layer count assert) and lowers the graph onto the GPU engine."""
        assert mode in ['training', 'inference']
        graph = build_graph(config)                   # raises the 'dividable by 2 at least 6 times' Exception
        self._graph = graph
        self._dp, self._rank, self._world = None, 0, 1
        if build_engine:
            from .engine import Engine
            from . import dp
            world = dp.launcher_world()[2]
            if mode == "training" and world > 1:
                # under a launcher (`python -m torch.distributed.run --nproc-per-node N pose_estimator.py train ...`): one process per GPU, this
                # rank's engine takes IMAGES_PER_GPU samples of every global batch of IMAGES_PER_GPU x WORLD_SIZE, the gradient exchange is
                # ursonet_amd.dp.DataParallelEngine (RCCL).  The reference's own knob for this is GPU_COUNT (config.py:20,154) feeding a
                # ParallelModel it left commented out (net.py:694-697); the launcher's world size takes its place
                self._rank, _, self._world = dp.init_from_launcher()
                gc = int(getattr(config, "GPU_COUNT", 1))
                assert gc in (1, self._world), "GPU_COUNT = %d but the launcher started %d ranks" % (gc, self._world)
                self._engine = Engine(config, mode, batch=int(config.IMAGES_PER_GPU))
                self._dp = dp.DataParallelEngine(self._engine, bucket_bytes=int(getattr(config, "DP_BUCKET_BYTES", 32 << 20)),
                                                 compress=getattr(config, "DP_COMPRESS", None))
            else:
                if world > 1:                           # inference under a launcher: N independent replicas, each on its own GPU, no exchange
                    import torch
                    self._rank, local, self._world = dp.launcher_world()
                    torch.cuda.set_device(int(os.environ.get("URSO_DP_DEVICE", local % max(torch.cuda.device_count(), 1))))
                self._engine = Engine(config, mode)
        return KerasModelShim(self, graph)

    # ---------------------------------------------------------------- checkpoints / log dirs
    def get_last_checkpoint(self, model_name):
        """net.py:768-788."""
        dir_names = next(os.walk(self.model_dir))[1]
        assert model_name in dir_names
        model_path = os.path.join(self.model_dir, model_name)
        checkpoints = sorted(f for f in next(os.walk(model_path))[2] if f.startswith("weights"))
        if not checkpoints:
            return model_path, None
        return model_path, os.path.join(model_path, checkpoints[-1])

    def find_last(self):
        """net.py:791-814."""
        dir_names = next(os.walk(self.model_dir))[1]
        key = self.config.NAME.lower()
        dir_names = sorted(f for f in dir_names if f.startswith(key))
        if not dir_names:
            return None, None
        dir_name = os.path.join(self.model_dir, dir_names[-1])
        checkpoints = sorted(f for f in next(os.walk(dir_name))[2] if f.startswith("weights"))
        if not checkpoints:
            return dir_name, None
        return dir_name, os.path.join(dir_name, checkpoints[-1])

    def load_weights(self, weights_in_path, weights_out_path, by_name=False, exclude=None):
        """net.py:816-852: load by name (layers absent from the file keep their values), optional
        exclude list; then point the log dir at weights_out_path."""
        if exclude:
            by_name = True
        params = read_weights_file(weights_in_path)
        if exclude:
            params = OrderedDict((k, v) for k, v in params.items() if k not in exclude)
        if not by_name:
            missing = [ln for ln in self._graph.params if ln not in params]
            if missing:
                raise ValueError("weights file lacks layers %s (use by_name=True)" % missing[:5])
        known = OrderedDict((ln, ws) for ln, ws in params.items() if ln in self._graph.params)
        self._engine.set_weights(known, strict=False)
        self.set_log_dir(weights_out_path)

    def save_weights(self, path):
        return write_weights_file(path, self._engine.get_weights())

    def get_imagenet_weights(self, architecture):
        """net.py:854-884 downloads from GitHub; there is no network here."""
        raise IOError("get_imagenet_weights(%r): pre-trained weights must be provided locally (no network access); "
                      "pass the path of a .h5/.npz file to --weights" % (architecture,))

    def get_urso_weights(self, dataset_name):
        """net.py:886-940."""
        assert dataset_name in ['soyuz_hard', 'dragon_hard', 'speed']
        raise IOError("get_urso_weights(%r): released weights must be provided locally (no network access)" % (dataset_name,))

    def set_log_dir(self, model_path=None):
        """net.py:944-967."""
        if model_path:
            self.log_dir = os.path.dirname(model_path)
            m = re.search(r"(\d{3,4})\.(h5|npz)$", model_path)
            self.epoch = int(m.group(1)[-3:]) if m else int(model_path[-6:-3])
        else:
            self.epoch = 0
            now = datetime.datetime.now()
            self.log_dir = os.path.join(self.model_dir, "{}{:%Y%m%dT%H%M}".format(self.config.NAME.lower(), now))
        self.checkpoint_path = os.path.join(self.log_dir, "weights_{}_*epoch*.h5".format(self.config.NAME.lower()))
        self.checkpoint_path = self.checkpoint_path.replace("*epoch*", "{epoch:04d}")

    # ---------------------------------------------------------------- training
    def compile(self, learning_rate, momentum):
        """net.py:973-1028: (re)creates the optimizer state (momentum buffers reset, as a new Keras
        optimizer would), sets lr/momentum/clipnorm.  Loss assembly and the L2 term live in the engine."""
        eng = self._engine
        eng.hyper[0] = float(learning_rate)
        if eng.adam:                                   # OPTIMIZER != 'SGD': Adam(learning_rate, amsgrad=True, clipnorm), net.py:982-983
            eng.hyper[4] = float(self.config.GRADIENT_CLIP_NORM or 0.0)
        else:
            eng.hyper[1] = float(momentum)
            eng.hyper[2] = float(self.config.GRADIENT_CLIP_NORM or 0.0)
        eng.reset_optimizer()

    def set_trainable(self, layer_regex, keras_model=None, indent=0, verbose=1):
        """net.py:1030-1066."""
        if verbose > 0 and keras_model is None:
            log("Selecting layers to train")
        self._layer_regex = layer_regex
        for layer in self.keras_model.layers:
            layer.trainable = bool(re.fullmatch(layer_regex, layer.name))
            if layer.trainable and verbose > 0:
                log("{}{:20}   ({})".format(" " * indent, layer.name, layer.kind))
        self._engine.set_trainable(layer_regex)

    def train(self, train_dataset, val_dataset, learning_rate, epochs, layers):
        """net.py:1068-1167: STEPS_PER_EPOCH steps per epoch, VALIDATION_STEPS forward-only validation
        batches, a weights checkpoint per epoch, optional CyclicLR; returns the per-batch loss logger."""
        assert self.mode == "training", "Create model in training mode."
        layers = _layer_regex(layers)
        cfg, eng = self.config, self._engine
        # the reference hands Keras two Python generators and `workers = cpu_count` processes (net.py:1100-1163); here a producer
        # thread + loader threads assemble uint8 batches in pinned memory and a side stream uploads batch k+1 while step k runs
        from .feeder import DeviceFeeder
        import torch
        workers = int(getattr(cfg, "LOADER_WORKERS", min(8, os.cpu_count() or 1)))
        rank, world, runner = self._rank, self._world, (self._dp or eng)
        # data-parallel run: every rank walks the SAME shuffled order and takes samples [rank * B, (rank + 1) * B) of each global batch (feeder.batches)
        train_feed = DeviceFeeder(eng, train_dataset, cfg, shuffle=True, workers=workers, rank=rank, world=world)
        val_feed = (DeviceFeeder(eng, val_dataset, cfg, shuffle=True, workers=workers, rank=rank, world=world)
                    if int(cfg.VALIDATION_STEPS) > 0 else None)
        history_full = BatchLogger()
        chief = rank == 0
        if chief:
            log("\nStarting at epoch {}. LR={}\n".format(self.epoch, learning_rate))
            log("Checkpoint Path: {}".format(self.checkpoint_path))
            if world > 1:
                log("Data parallel: {} ranks x {} images (global batch {})".format(world, eng.B, world * eng.B))
        self.set_trainable(layers, verbose=1 if chief else 0)
        self.compile(learning_rate, cfg.LEARNING_MOMENTUM)
        if chief:
            os.makedirs(self.log_dir, exist_ok=True)
        steps, vsteps = int(cfg.STEPS_PER_EPOCH), int(cfg.VALIDATION_STEPS)
        # Keras' BatchLogger reads the batch's losses after every step (net.py:1106-1115): a device -> host round trip per 7 ms step.  Here each
        # step's loss scalars are copied into a device-side history row (same stream, no synchronisation) and read ONCE per epoch -- the same
        # per-batch lists, no bubble; under data parallelism the rows are averaged over the ranks first (the loss of the global batch)
        n_loss = int(eng.loss_buf.numel())
        hist = torch.zeros(max(steps, 1), n_loss, dtype=torch.float32, device=eng.device)
        vhist = torch.zeros(max(vsteps, 1), n_loss, dtype=torch.float32, device=eng.device)
        kp = bool(cfg.REGRESS_KEYPOINTS)
        clr_it = 0
        for epoch in range(self.epoch, epochs):
            for i in range(steps):
                if cfg.CLR:
                    eng.set_lr(utils.clr_triangular(clr_it, cfg.BASE_LEARNING_RATE, cfg.MAX_LEARNING_RATE, cfg.CLR_STEP_SIZE))
                    clr_it += 1
                train_feed.next_into()
                runner.step()
                hist[i].copy_(eng.loss_buf.view(-1), non_blocking=True)
            for i in range(vsteps):
                val_feed.next_into()
                eng.evaluate(read=False)
                vhist[i].copy_(eng.loss_buf.view(-1), non_blocking=True)
            if world == 1 and eng.forked:
                eng.verify_fork()                       # the forked backward pass against the single chain on this epoch's last batch (engine.py)
            if world > 1:
                from . import dp
                dp.average_over_ranks(hist)
                dp.average_over_ranks(vhist)
            h = hist[:steps].cpu().numpy()
            for row in h:
                history_full.ori_loss_acc.append(None if kp else float(row[1]))   # None in keypoint mode, as logs.get('ori_loss') is (net.py:1112)
                history_full.loc_loss_acc.append(float(row[0]))
            val = {}
            if vsteps:
                v = vhist[:vsteps].cpu().numpy().mean(0)
                val = ({"loc_loss": v[0], "k2_loss": v[2], "k3_loss": v[3]} if kp else {"loc_loss": v[0], "ori_loss": v[1]})
            if chief:
                log("epoch %d  loc_loss %.5f  %s" % (
                    epoch + 1, float(np.mean(h[:, 0])) if steps else float("nan"),
                    "  ".join("val_%s %.5f" % (k, float(x)) for k, x in sorted(val.items()))))
                self.save_weights(self.checkpoint_path.format(epoch=epoch + 1))     # replicas are identical: one writer
            if world > 1:
                import torch.distributed as dist
                dist.barrier()                          # nobody runs ahead of the checkpoint (find_last on another rank sees it)
        train_feed.close()
        if val_feed is not None:
            val_feed.close()
        self.epoch = max(self.epoch, epochs)
        return history_full

    # ---------------------------------------------------------------- inference
    def _resized(self, images):
        """resize / pad every frame to the model's input geometry (utils.resize_image) -> [(uint8 frame, window, scale)]."""
        cfg = self.config
        out = []
        for image in images:
            frame, window, scale, _pad, _crop = utils.resize_image(image, min_dim=cfg.IMAGE_MIN_DIM, min_scale=cfg.IMAGE_MIN_SCALE,
                                                                   max_dim=cfg.IMAGE_MAX_DIM, mode=cfg.IMAGE_RESIZE_MODE)
            out.append((frame, window, scale))
        return out

    def mold_inputs(self, images):
        """net.py:1169-1205: (molded images [N,h,w,3], image metas [N,12], windows [N,4])."""
        frames = self._resized(images)
        molded = np.stack([mold_image(f, self.config) for f, _, _ in frames])
        metas = np.stack([compose_image_meta(0, im.shape, f.shape, w, s) for im, (f, w, s) in zip(images, frames)])
        return molded, metas, np.stack([w for _, w, _ in frames])

    def _predict(self, molded_images):
        """keras_model.predict on molded float images -> list of raw output arrays."""
        eng = self._engine
        assert molded_images.shape[0] == eng.B, "batch %d != engine batch %d" % (molded_images.shape[0], eng.B)
        eng.set_input_u8(False)
        eng.load_batch(molded_images.astype(np.float32))
        return self._outputs_after_forward()

    def _outputs_after_forward(self):
        import torch
        eng = self._engine
        eng.forward()
        loc, rest = eng.outputs()
        torch.cuda.synchronize()
        if self.config.REGRESS_KEYPOINTS:
            return [loc.cpu().numpy()] + [k.cpu().numpy() for k in rest]
        return [loc.cpu().numpy(), rest.cpu().numpy()]

    def detect(self, images, verbose=0):
        """net.py:1207-1259: one dict of RAW network outputs per image -- {'loc', 'ori'}, or {'loc', 'k1', 'k2'} in keypoint mode
        (the reference's key names for the three keypoint heads).  uint8 frames take the device path: they are padded on the host,
        uploaded as uint8 and mean-subtracted by the first kernel; anything else goes through mold_inputs / predict."""
        assert self.mode == "inference", "Create model in inference mode."
        assert len(images) == self.config.BATCH_SIZE, "len(images) must be equal to BATCH_SIZE"
        if verbose:
            log("Processing {} images".format(len(images)))
            for image in images:
                log("image", image)
        if all(getattr(im, "dtype", None) == np.uint8 and im.ndim == 3 and im.shape[-1] == 3 for im in images):
            frames = [f for f, _, _ in self._resized(images)]
            assert all(f.shape == frames[0].shape for f in frames), \
                "After resizing, all images must have the same size. Check IMAGE_RESIZE_MODE and image sizes."
            self._engine.load_batch_u8(np.stack(frames))
            outs = self._outputs_after_forward()
        else:
            molded, metas, _ = self.mold_inputs(images)
            assert all(m.shape == molded[0].shape for m in molded), \
                "After resizing, all images must have the same size. Check IMAGE_RESIZE_MODE and image sizes."
            if verbose:
                log("molded_images", molded)
                log("image_metas", metas)
            outs = self.keras_model.predict(molded, verbose=0)
        keys = ("loc", "k1", "k2") if self.config.REGRESS_KEYPOINTS else ("loc", "ori")
        return [dict((k, o[i]) for k, o in zip(keys, outs)) for i in range(len(images))]
