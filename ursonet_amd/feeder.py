"""Input side of the training loop (reference: net.py:358-559 load_image_gt / data_generator and the
fit_generator(workers=cpu_count, max_queue_size=100) call of net.py:1147-1163).

The reference loads, augments, resizes and molds one sample at a time in `cpu_count` worker processes and ships float
images through a queue.  At ~2,800 images/s per GPU that design cannot keep up (32 x 512 x 640 x 3 float32 = 126 MB per step over
PCIe alone is 2.3 ms of an 11.4 ms step).  Here:
  * workers (threads: image decode / synthesis and NumPy release the GIL) only produce the RAW sample -- uint8 frame + pose;
  * everything per-pixel runs batched on the GPU: sim2real stages, camera / in-plane rotation warps (one urso_warp_perspective for
    the minibatch), target re-encoding (urso_encode_ori), zero padding (urso_pad_images_u8); mean subtraction + cast are the
    engine's first kernel (urso_mold_images reads uint8);
  * uint8 frames go host -> device from PINNED memory on a side stream into a double buffer while the previous step computes:
    31 MB instead of 126 MB per step, off the critical path.
`data_generator` keeps the reference's signature and yield format on top of the same pieces.
"""
import logging
import queue
import threading

import numpy as np

from . import utils


def compose_image_meta(image_id, original_image_shape, image_shape, window, scale):
    return np.array([image_id] + list(original_image_shape) + list(image_shape) + list(window) + [scale])


class Sample(object):
    """One raw training sample: uint8 frame + the pose targets the configured heads need."""
    __slots__ = ("image_id", "image", "loc", "ori", "k1", "k2")

    def __init__(self, image_id, image, loc, ori, k1=None, k2=None):
        self.image_id, self.image, self.loc, self.ori, self.k1, self.k2 = image_id, image, loc, ori, k1, k2


def load_sample(dataset, config, image_id):
    """The host-only first third of load_image_gt (net.py:367-388): frame and targets, no augmentation, no resize."""
    image = dataset.load_image(image_id)
    loc = dataset.load_location(image_id) if config.REGRESS_LOC else dataset.load_location_encoded(image_id)
    k1 = k2 = None
    if config.REGRESS_KEYPOINTS:
        k1, k2 = dataset.load_keypoints(image_id)[:2]
    if config.REGRESS_KEYPOINTS or config.REGRESS_ORI:
        getter = {"quaternion": dataset.load_quaternion, "euler_angles": dataset.load_euler_angles,
                  "angle_axis": dataset.load_angle_axis}[config.ORIENTATION_PARAM]
        ori = getter(image_id)
    else:
        ori = dataset.load_orientation_encoded(image_id)
    return Sample(image_id, image, loc, ori, k1, k2)


def _hip_section():
    """The lock that keeps this thread out of HIP while another thread captures a hipGraph (hip.capture_lock; Engine.capture holds it).  Taken
    around the batched augmentation launches with their device-to-host copies and around pin_memory only: disk loads, the CPU draws and the
    resize / padding of a batch run unlocked, so a capture waits for a kernel's worth of work, not for a batch's preparation."""
    from . import hip
    return hip.capture_lock


def augment_samples(samples, dataset, config):
    """The augmentation third (net.py:390-438) for a LIST of samples, in place.  NumPy's GLOBAL generator is consumed exactly as the
    reference consumes it, sample by sample: the sim2real dice (net.py:395), then that sample's rotation dice (net.py:415) and angles
    (utils.py:33 / :62) -- the imgaug stage parameters come from a separate generator, as imgaug's do, and only for the samples the dice
    select.  The samples themselves are loaded ahead of the draws and the pixel work is batched (one sim2real pass set and one warp
    launch for all samples of the list that need it), which changes no draw.  Images come back as uint8 arrays."""
    if not samples:
        return samples
    from . import augment
    rot = bool(config.ROT_AUG or config.ROT_IMAGE_AUG)
    if rot:
        assert config.REGRESS_LOC
        assert config.ORIENTATION_PARAM == 'quaternion'
    n = len(samples)
    h, w = samples[0].image.shape[:2]
    same_size = all(s.image.shape == samples[0].image.shape for s in samples)
    draws, pyr, warp_ids = None, np.zeros((n, 3)), []
    if config.SIM2REAL_AUG:
        draws = []
    for i, s in enumerate(samples):
        if config.SIM2REAL_AUG:
            draws.append(augment.sim2real_draw(1, s.image.shape[0], s.image.shape[1], prng=augment._PIPELINE_RNG))
        if rot:
            dice = np.random.rand(1)
            if config.ROT_AUG and dice > 0.5:
                pyr[i] = (np.random.rand(3) - 0.5) * 20                           # utils.rotate_cam(..., magnitude 20), utils.py:33
                warp_ids.append(i)
            elif config.ROT_IMAGE_AUG and dice <= 0.5:
                pyr[i, 2] = ((np.random.rand(1) - 0.5) * 170)[0]                   # utils.rotate_image, utils.py:62
                warp_ids.append(i)
    groups = [list(range(n))] if same_size else [[i] for i in range(n)]
    for g in groups:
        if config.SIM2REAL_AUG:
            merged = {"apply": np.concatenate([draws[i]["apply"] for i in g]), "order": np.concatenate([draws[i]["order"] for i in g]),
                      "par": np.concatenate([draws[i]["par"] for i in g]), "seeds": np.concatenate([draws[i]["seeds"] for i in g]),
                      "masks": [draws[i]["masks"][0] for i in g]}
            with _hip_section():
                out = augment.sim2real_batch(np.stack([samples[i].image for i in g]), draw=merged).cpu().numpy()
            for k, i in enumerate(g):
                samples[i].image = out[k]
        ids = [i for i in g if i in warp_ids]
        if ids:
            quats = []
            for i in ids:
                if not (config.REGRESS_ORI or config.REGRESS_KEYPOINTS):
                    samples[i].ori = dataset.load_quaternion(samples[i].image_id)   # classification targets are re-encoded from the rotated pose
                quats.append(samples[i].ori)
            with _hip_section():
                warped, t_new, q_new = augment.rotate_cam_batch(np.stack([samples[i].image for i in ids]), np.stack([samples[i].loc for i in ids]),
                                                                np.stack(quats), dataset.camera.K, pyr[ids])
                warped = warped.cpu().numpy()
                enc = None
                if not (config.REGRESS_ORI or config.REGRESS_KEYPOINTS):
                    enc = augment.encode_orientations(q_new, dataset.ori_histogram_map, dataset.ori_output_mask, config.BETA).cpu().numpy()
            for k, i in enumerate(ids):
                s = samples[i]
                s.image, s.loc, s.ori = warped[k], t_new[k], q_new[k]
                if config.REGRESS_KEYPOINTS:
                    s.k1, s.k2 = augment.encode_as_keypoints(s.ori, s.loc)      # net.py:424, 433
                elif enc is not None:
                    s.ori = enc[k]
    return samples


def finish_sample(sample, config):
    """The last third of load_image_gt (net.py:440-456): resize / pad and the image_meta vector."""
    original_shape = sample.image.shape
    image, window, scale, padding, crop = utils.resize_image(
        sample.image, min_dim=config.IMAGE_MIN_DIM, min_scale=config.IMAGE_MIN_SCALE, max_dim=config.IMAGE_MAX_DIM,
        mode=config.IMAGE_RESIZE_MODE)
    meta = compose_image_meta(sample.image_id, original_shape, image.shape, window, scale)
    return image, meta


class BatchAssembler(object):
    """Pre-allocated per-field arrays of one minibatch; `images` is uint8 (device path) or the molded float type (reference format)."""

    def __init__(self, config, batch_size, image_shape, meta_len, image_dtype):
        ft = np.float16 if config.F16 else np.float32
        self.config, self.n = config, batch_size
        self.images = np.zeros((batch_size,) + tuple(image_shape), dtype=image_dtype)
        self.meta = np.zeros((batch_size, meta_len), dtype=np.float64)
        self.loc = np.zeros((batch_size, 3 if config.REGRESS_LOC else config.LOC_BINS_PER_DIM ** 3), dtype=ft)
        if config.REGRESS_KEYPOINTS:
            self.k1, self.k2 = np.zeros((batch_size, 3), dtype=ft), np.zeros((batch_size, 3), dtype=ft)
            self.ori = None
        else:
            width = (4 if config.ORIENTATION_PARAM == 'quaternion' else 3) if config.REGRESS_ORI else config.ORI_BINS_PER_DIM ** 3
            self.ori = np.zeros((batch_size, width), dtype=ft)

    def put(self, b, image, meta, sample):
        self.images[b] = image
        self.meta[b] = meta
        self.loc[b] = sample.loc
        if self.ori is None:
            self.k1[b], self.k2[b] = np.asarray(sample.k1).T, np.asarray(sample.k2).T
        else:
            self.ori[b] = sample.ori

    def inputs(self):
        """[images, image_meta, gt_loc, gt_ori] (keypoints: [.., gt_loc, gt_k1, gt_k2]) -- the model input list of net.py:671-674."""
        if self.ori is None:
            return [self.images, self.meta, self.loc, self.k1, self.k2]
        return [self.images, self.meta, self.loc, self.ori]


DP_SHUFFLE_SEED = 1234


def batches(dataset, config, shuffle, batch_size, molded, workers=0, rank=0, world=1):
    """Endless iterator of BatchAssembler objects.  molded=True: images are mean-subtracted floats (the reference's generator
    format); False: uint8 frames for the device path.  Up to 5 failing samples are logged and skipped, the 6th re-raises
    (net.py:553-559).  workers > 0 loads the raw samples of a batch with that many threads.

    world > 1 (data parallel, one process per GPU): every rank walks the SAME order over the dataset -- the shuffles come from a private
    RandomState(DP_SHUFFLE_SEED), identical on all ranks, instead of NumPy's global one -- and keeps samples [rank * batch_size,
    (rank + 1) * batch_size) of every global batch of world * batch_size: the ranks' shards are disjoint and together they are the batch a
    single process with GPU_COUNT = world would have drawn.  The global RNG (augmentation draws) is seeded with DP_SHUFFLE_SEED + rank so
    that the ranks do not apply identical warps to their different samples.  world == 1 is the reference's generator, draw for draw."""
    from .net import mold_image
    ids = np.copy(dataset.image_ids)
    cursor, errors = -1, 0
    order_rng = None
    if world > 1:
        order_rng = np.random.RandomState(DP_SHUFFLE_SEED)
        np.random.seed(DP_SHUFFLE_SEED + int(rank))
    pool = None
    if workers > 0:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=workers)
    ft = np.float16 if config.F16 else np.float32

    def safe_load(image_id):
        try:
            return load_sample(dataset, config, image_id)
        except (GeneratorExit, KeyboardInterrupt):
            raise
        except Exception:
            logging.exception("Error processing image {}".format(dataset.image_info[image_id]))
            return None
    def advance():
        nonlocal cursor
        cursor = (cursor + 1) % len(ids)
        if shuffle and cursor == 0:
            (order_rng or np.random).shuffle(ids)
        return ids[cursor]
    while True:
        chosen = []
        for _ in range(rank * batch_size if world > 1 else 0):      # the samples of this global batch that belong to the ranks before this one
            advance()
        while len(chosen) < batch_size:
            want = batch_size - len(chosen)
            todo = [advance() for _ in range(want)]
            loaded = list(pool.map(safe_load, todo)) if pool is not None else [safe_load(i) for i in todo]
            for s in loaded:
                if s is None:
                    errors += 1
                    if errors > 5:
                        raise RuntimeError("more than 5 samples failed to load (net.py:553-559)")
                else:
                    chosen.append(s)
        for _ in range((world - 1 - rank) * batch_size if world > 1 else 0):     # ... and to the ranks behind it
            advance()
        augment_samples(chosen, dataset, config)
        asm = None
        for b, s in enumerate(chosen):
            image, meta = finish_sample(s, config)
            if asm is None:
                asm = BatchAssembler(config, batch_size, image.shape, len(meta), ft if molded else np.uint8)
            asm.put(b, mold_image(image.astype(ft), config) if molded else image, meta, s)
        yield asm


class DeviceFeeder(object):
    """Keeps the engine fed: a producer thread assembles uint8 batches into pinned host buffers (ring of `depth`), the consumer
    side copies batch k+1 to a device staging buffer on a SIDE stream while step k runs, and `next_into(engine)` makes the
    engine's input buffers hold the next batch (device-to-device copy ordered after the upload by an event)."""

    def __init__(self, engine, dataset, config, shuffle=True, workers=4, depth=3, rank=0, world=1):
        import torch
        self.eng, self.torch = engine, torch
        self.q = queue.Queue(maxsize=depth)
        self.stop = False
        self.err = None
        self.side = torch.cuda.Stream(device=engine.device)
        self.stage = [None, None]
        self.events = [torch.cuda.Event(), torch.cuda.Event()]
        self.consumed = [None, None]                           # recorded on the compute stream once a slot's batch has been copied out of it
        self.k = 0
        self.pinned_bytes = 0
        gen = batches(dataset, config, shuffle, engine.B, molded=False, workers=workers, rank=rank, world=world)

        def produce():
            from . import hip
            try:
                while not self.stop:
                    # the generator takes hip.capture_lock itself around its HIP sections (augment_samples: the batched augmentation kernels and
                    # their device-to-host copies); here only hipHostMalloc in pin_memory needs it -- not while the consumer thread captures a hipGraph
                    asm = next(gen)
                    arrays = [np.ascontiguousarray(a) for a in ([asm.images, asm.loc] + ([asm.k1, asm.k2] if asm.ori is None else [asm.ori]))]
                    with hip.capture_lock:
                        host = [torch.from_numpy(a).pin_memory() for a in arrays]
                    self.q.put(host)
            except BaseException as e:                         # surfaced by next_into
                self.err = e
                self.q.put(None)
        self.thread = threading.Thread(target=produce, daemon=True)
        self.thread.start()
        self._upload()

    def _upload(self):
        torch = self.torch
        host = self.q.get()
        if host is None:
            raise self.err
        slot = self.k & 1
        with torch.cuda.stream(self.side):
            if self.consumed[slot] is not None:
                self.side.wait_event(self.consumed[slot])      # the engine's copy out of this slot (two batches ago) must have run first
            if self.stage[slot] is None:
                self.stage[slot] = [torch.empty(h.shape, dtype=h.dtype, device=self.eng.device) for h in host]
            for d, h in zip(self.stage[slot], host):
                d.copy_(h, non_blocking=True)
            self.events[slot].record(self.side)
        self._pending = (slot, host)                           # keep the pinned tensors alive until the copy has been consumed
        self.pinned_bytes = sum(h.numel() * h.element_size() for h in host)

    def next_into(self):
        """Engine inputs <- the uploaded batch; immediately starts uploading the following one."""
        torch, eng = self.torch, self.eng
        slot, _ = self._pending
        torch.cuda.current_stream(eng.device).wait_event(self.events[slot])
        st = self.stage[slot]
        eng.load_batch_u8(st[0], st[1], st[2], st[3] if len(st) > 3 else None)
        if self.consumed[slot] is None:
            self.consumed[slot] = torch.cuda.Event()
        self.consumed[slot].record(torch.cuda.current_stream(eng.device))
        self.k += 1
        self._upload()

    def close(self):
        self.stop = True
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
