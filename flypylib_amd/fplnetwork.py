"""`FplNetwork`: the reference's network wrapper (`flypylib/fplnetwork.py:46-220`)
on the MI355X engine.

Same surface: `FplNetwork(model)`, attributes `rf_size / rf_offset / rf_stride /
infer_sz / n_gpu / compile_args / train_network / train_single / infer_network`,
methods `train`, `make_train_parallel`, `make_infer_parallel`, `infer`,
`save_network`, module function `load_network`.  `model` is a factory from
`fplmodels` returning `(LayerGraph, rf_info, infer_sz, compile_args)`.

Inference runs entirely in libfplhip.so (`fpl_infer_volume`): tile lattice, edge
zero-padding, forward, upsample and stitch happen on the device; there is no CPU
fallback.
"""
import pickle

import numpy as np

from . import _capi, fplutils, multi_gpu, runtime

_PRECISIONS = {'auto': _capi.PREC_AUTO, 'f32': _capi.PREC_F32, 'fp32': _capi.PREC_F32,
               'float32': _capi.PREC_F32, 'bf16': _capi.PREC_BF16,
               'f16': _capi.PREC_F16, 'float16': _capi.PREC_F16,
               # split IEEE halves (hi + lo, three MFMAs per product): fp32-grade
               # probabilities at a third of the 16-bit rate; vgg_like, vgg_like2 and the
               # unet_like / unet_like2 / 3 / 4 skeleton (other graphs are refused)
               'f16s': _capi.PREC_F16S, 'split': _capi.PREC_F16S}


class InferNetwork:
    """what `FplNetwork.infer_network` holds: the fixed-size inference program
    (+ UpSampling3D(rf_stride)), resident on one GPU"""

    def __init__(self, graph, stride, device, lane=0):
        self.graph = graph
        self.stride = stride
        self.device = device
        self.program = _capi.Program(runtime.get_context(device, lane), graph, stride)

    @property
    def input_shape(self):
        return self.graph.input_shape

    def set_weights(self, weights):
        self.graph.set_weights(weights)
        self.program.set_weights_from(self.graph)

    def get_weights(self):
        return self.graph.get_weights()

    def predict(self, data_batch, batch_size=1):
        """(n, I,I,I, 1) -> (n, O,O,O, 1) float32 (Keras `Model.predict` shape
        contract, fplnetwork.py:175-176); batch_size is irrelevant here"""
        return self.program.forward(np.asarray(data_batch, np.float32))


class _ReferenceUnpickler(pickle.Unpickler):
    """a pickle written by the reference names `flypylib.fplnetwork.FplNetwork`, the model
    factory `flypylib.fplmodels.<name>` and, in compile_args, its loss / metric functions:
    all have same-named counterparts here"""

    def find_class(self, module, name):
        if module == 'flypylib' or module.startswith('flypylib.'):
            module = 'flypylib_amd' + module[len('flypylib'):]
        return super().find_class(module, name)


def load_network(filepath, device=None):
    """inverse of `FplNetwork.save_network` (reference fplnetwork.py:32-44).  Reads this
    package's pair (pickle + `<path>.weights.npz`) and the reference's (pickle of a
    `flypylib.fplnetwork.FplNetwork` + Keras `<path>.keras.h5`)."""
    import os
    with open(filepath, 'rb') as fn:
        network = _ReferenceUnpickler(fn).load()
    # attributes a reference-written instance does not carry
    for k, v in (('precision', 'auto'), ('_parallel', None), ('_parallel_devices', None),
                 ('_trainer', None)):
        if not hasattr(network, k):
            setattr(network, k, v)
    network.rf_size, network.rf_offset, network.rf_stride, network.infer_sz = (
        tuple(network.rf_size), tuple(network.rf_offset), tuple(network.rf_stride),
        tuple(network.infer_sz))
    network._device = runtime.default_device() if device is None else device
    network.train_single, _, _, _ = network.model()
    # compile, THEN the saved weights and optimizer state (load_model's order, reference :38-40)
    network.train_single.compile(**network.compile_args)
    if os.path.exists(filepath + '.weights.npz'):
        network.train_single.load(filepath + '.weights.npz')
    elif os.path.exists(filepath + '.keras.h5'):
        network.train_single.load(filepath + '.keras.h5')
    else:
        raise FileNotFoundError('%s: neither %s.weights.npz nor %s.keras.h5 (the reference\'s '
                                'Keras file) is there' % (filepath, filepath, filepath))
    network.train_network = network.train_single
    network._parallel = None
    network._set_infer()
    return network


class FplNetwork:
    """3D-CNN voxel classifier: training by generator, full-volume inference"""

    def __init__(self, model, device=None, precision='auto'):
        self.model = model
        self.train_network, rf_info, infer_sz, compile_args = self.model()
        self.train_network.summary()
        self.train_single = self.train_network

        self.rf_size = tuple(fplutils.to3d(rf_info[0]))
        self.rf_offset = tuple(fplutils.to3d(rf_info[1]))
        self.rf_stride = tuple(fplutils.to3d(rf_info[2]))

        self.infer_network = None
        self.n_gpu = 1
        self.infer_sz = tuple(fplutils.to3d(infer_sz))

        if compile_args is None:
            compile_args = {'loss': 'binary_crossentropy',
                            'optimizer': 'adam',
                            'metrics': ['accuracy']}
        self.train_network.compile(**compile_args)
        self.compile_args = compile_args

        self.precision = precision
        self._device = runtime.default_device() if device is None else device
        self._parallel = None
        self._parallel_devices = None

    # ---- persistence (reference :81-97) ------------------------------------------------
    def save_network(self, filepath, keras_h5=True):
        """pickle + `<path>.weights.npz`, and (keras_h5) the network once more as
        `<path>.keras.h5` - the file Keras' `model.save` writes next to the reference's pickle:
        weights, `model_config`, `training_config` and (after training) `optimizer_weights`
        (keras_io.py; whether Keras' `load_model` accepts the written JSON could not be run
        here - without it, rebuild the model from its factory and `load_weights`)"""
        self.train_single.save(filepath + '.weights.npz')
        if keras_h5:
            self.train_single.save(filepath + '.keras.h5')
        keep = (self.train_single, self.train_network, self.infer_network,
                self._parallel)
        self.train_single = self.train_network = self.infer_network = None
        self._parallel = None
        try:
            with open(filepath, 'wb') as fn:
                pickle.dump(self, fn)
        finally:
            (self.train_single, self.train_network, self.infer_network,
             self._parallel) = keep

    # ---- inference network (reference :99-110) --------------------------------
    def _build_infer(self, device, lane=0):
        graph, _, _, _ = self.model(self.infer_sz)
        graph.set_weights(self.train_single.get_weights())
        return InferNetwork(graph, self.rf_stride, device, lane)

    def _set_infer(self):
        if self._parallel is not None:
            self.make_infer_parallel(self.n_gpu, self._parallel_devices)
        else:
            self.infer_network = self._build_infer(self._device)

    def make_infer_parallel(self, n_gpu, devices=None):
        """reference :130-134 (which rebuilds the inference net, then wraps it); tile
        rows are sharded as Z slabs, one per GPU, each written by its own host thread.
        `devices` (not in the reference): the GPU of each slab, default 0..n_gpu-1; a
        device named twice serves two slabs from two contexts (one-GPU rehearsal)."""
        devices = list(range(n_gpu)) if devices is None else [int(d) for d in devices]
        assert len(devices) == n_gpu, 'make_infer_parallel: one device per slab'
        self.infer_network = self._build_infer(self._device)
        nets, seen = [], {}
        for d in devices:
            lane = seen.get(d, 0)
            seen[d] = lane + 1
            # lanes 32+: contexts of their own, away from the pipeline's lanes 0/1
            nets.append(self.infer_network if (d == self._device and lane == 0)
                        else self._build_infer(d, 32 + lane if lane else 0))
        self._parallel_nets = nets
        self._parallel_devices = devices
        self._parallel = multi_gpu.make_parallel(lambda i: nets[i].program, n_gpu)
        self.n_gpu = n_gpu

    def make_train_parallel(self, n_gpu, batch_size, input_shape, devices=None):
        """reference :124-128: `n_gpu` towers of `batch_size` examples each, the
        generator yields batch_size * n_gpu per step (multi_gpu.py:21-25).  One
        trainer + host thread per GPU in this process, gradients summed by one RCCL
        all-reduce per step; under torchrun (one process per GPU) the ranks are the
        towers.  See flypylib_amd/train.py.  `devices` (not in the reference): the GPU
        of each tower, default 0..n_gpu-1."""
        from . import train
        old = self.train_network
        if isinstance(old, train.ParallelTrainNetwork):
            old.close()
        self.train_network = train.make_parallel(
            self.train_single, n_gpu, batch_size,
            list(fplutils.to3d(input_shape)) + [1], devices)
        self.train_network.compile(**self.compile_args)

    def train(self, generator, steps_per_epoch, epochs, log_file,
              save_filepath):
        """reference :112-122: fit from a batch generator, CSV log, per-epoch
        checkpoint '<save_filepath>_%03d', then rebuild the inference net"""
        from . import train
        train.fit_generator(self, generator, steps_per_epoch, epochs, log_file,
                            save_filepath)
        self._set_infer()

    # ---- full-volume inference (reference :136-189) -----------------------------
    def infer(self, image, normalize=None, precision=None):
        """image: (Z,Y,X) array (already normalised float, as in the reference)
        or uint8 with `normalize=(mean, std)`; or an h5 path with dataset /main.
        Returns float32 predictions of the same shape; the rf_offset border
        shell is zero.

        precision (default: the network's, 'auto'): 'auto' = fp32-grade results on the
        fastest executor that delivers them - split IEEE halves ('f16s': within ~4e-6 of
        fp32, the same detected point set) for all ten factories of fplmodels (the fused
        kernels of vgg_like / vgg_like2 / the U-Net skeletons, the layer-by-layer graph
        executor for the others), rerun on the fp32 MFMA kernels when a weight, an input
        voxel or an activation leaves the IEEE-half range (the kernels check), and the fp32
        MFMA kernels ('f32') for layer programs neither has kernels for; 'f16' / 'bf16' =
        plain 16-bit operands (up to ~1e-3 / ~8e-3 off fp32 on trained weights, 2 - 3x faster
        than 'f16s').

        The returned array lives in memory recycled from earlier results that have died
        (`_capi.host_empty`: a fresh 520^3 result would cost 45 ms of first-touch page faults,
        three times the rest of the call); it is an ordinary writable ndarray, and a caller
        that keeps it - or any slice of it - keeps its memory."""
        if isinstance(image, str):
            from . import keras_io
            image = np.load(image) if image.endswith('.npy') else keras_io.read_main(image)

        assert self.infer_network is not None, \
            'network has not been trained'
        assert self.infer_network.input_shape[1:-1] == self.infer_sz, \
            'network input shape does not match expected infer_sz'

        image = np.asarray(image)
        assert image.ndim == 3, 'image must be (Z,Y,X)'
        mean, std = (0.0, 1.0) if normalize is None else normalize
        if image.dtype != np.uint8:
            image = np.ascontiguousarray(image, dtype=np.float32)
        prec = _PRECISIONS[precision or self.precision]
        kw = dict(mean=mean, std=std, precision=prec)
        if self._parallel is not None:
            return self._parallel.infer_volume(image, self.infer_sz,
                                               self.rf_offset, **kw)
        return self.infer_network.program.infer_volume(
            image, self.infer_sz, self.rf_offset, **kw)

    def voxel_loss(self, image, lm_prefix, l0_thresh=None, l1_thresh=None,
                   normalize=None):
        """per-voxel log loss of the prediction on labelled, unmasked voxels
        (reference :191-220): confident negatives (loss < 0.005) are dropped, losses
        are clamped to the optional [lo, hi] thresholds.  `lm_prefix`: the reference's
        '<prefix>labels.h5' / '<prefix>mask.h5' prefix or a
        (labels, mask) pair of arrays / .npy paths."""
        from .fplobjdetect import _load_main
        pred = self.infer(image, normalize=normalize)
        if isinstance(lm_prefix, str):
            labels = np.array(_load_main('%slabels.h5' % lm_prefix))
            mask = np.array(_load_main('%smask.h5' % lm_prefix))
        else:
            labels, mask = (np.array(_load_main(a)) for a in lm_prefix)
        # voxels closer to a face than half the receptive field cannot be sampled
        for ax, cc in enumerate(self.rf_size):
            edge = int(round(cc / 2))
            sl = [slice(None)] * 3
            sl[ax] = slice(0, edge); mask[tuple(sl)] = 0
            sl[ax] = slice(-edge, None); mask[tuple(sl)] = 0

        def clamp(loss, bounds, where):
            if bounds is None:
                return loss
            return np.minimum(np.maximum(loss, bounds[0] * where), bounds[1] * where)

        neg = (mask == 1) & (labels == 0)
        l0_loss = -1. * neg * np.log(np.maximum(1 - pred, 1e-8))
        confident = neg & (l0_loss < 0.005)          # already learnt: never sampled again
        l0_loss[confident] = 0
        mask[confident] = 0
        l0_loss = clamp(l0_loss, l0_thresh, (labels == 0) * (mask == 1))
        pos = (mask == 1) & (labels == 1)
        l1_loss = clamp(-1. * pos * np.log(np.maximum(pred, 1e-8)), l1_thresh, pos)
        return (l0_loss + l1_loss).astype('float32')

    # pickling: device handles never travel
    def __getstate__(self):
        d = dict(self.__dict__)
        for k in ('train_single', 'train_network', 'infer_network', '_parallel',
                  '_parallel_nets', '_trainer'):
            d[k] = None
        return d
