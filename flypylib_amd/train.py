"""Training loop of `FplNetwork.train` / `make_train_parallel`
(reference `flypylib/fplnetwork.py:112-128`) on the HIP training engine.

`fit_generator(network, generator, steps_per_epoch, epochs, log_file,
save_filepath)` mirrors `train_network.fit_generator` with the reference's two
callbacks: a CSV log (`CSVLogger`: epoch, acc, loss) and a per-epoch checkpoint
'<save_filepath>_%03d' of the single-GPU weights (`multi_gpu_callback`,
fplnetwork.py:9-17).

Data parallelism (reference: in-graph towers slicing the batch,
`flypylib/multi_gpu.py:20-61`): one process per GPU, every rank draws its own
`batch_size` examples, gradients (and BN moving-average deltas) are summed with
ONE all-reduce of the flat gradient arena over RCCL (`torch.distributed`, backend
'nccl') and scaled by 1/world before the identical Adam update on every rank.
BN batch statistics stay per GPU, as each tower normalises its own slice.
"""
import csv

import numpy as np

from . import _capi, runtime

_OPTIMIZERS = {'adam': dict(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8)}


class ParallelTrainNetwork:
    """what `network.train_network` holds after make_train_parallel"""

    def __init__(self, single, n_gpu, batch_size, input_shape):
        self.single, self.n_gpu = single, n_gpu
        self.batch_size, self.input_shape = batch_size, tuple(input_shape)
        self.compile_args = None

    def compile(self, **kw):
        self.compile_args = dict(kw)

    def get_weights(self):
        return self.single.get_weights()


def make_parallel(train_single, n_gpu, batch_size, input_shape):
    return ParallelTrainNetwork(train_single, n_gpu, batch_size, input_shape)


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except ImportError:
        pass
    return None


def allreduce_grads(trainer, force=False):
    """sum the flat gradient arena over all ranks (RCCL over xGMI); returns the
    scale to apply (1/world).  No-op without an initialised process group (or with
    a single rank, unless `force`)."""
    dist = _dist()
    if dist is None or (dist.get_world_size() == 1 and not force):
        return 1.0
    import torch
    ptr, n = trainer.grad_ptr()
    # wrap the library-owned device arena without copying
    arena = _DeviceArena(ptr, n)
    t = torch.as_tensor(arena, device='cuda')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    torch.cuda.current_stream().synchronize()
    return 1.0 / dist.get_world_size()


class _DeviceArena:
    """__cuda_array_interface__ view of `n` float32 at device address `ptr`"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {
            'shape': (int(n),), 'typestr': '<f4', 'data': (int(ptr), False),
            'version': 2}


class _Prefetch:
    """Background consumer of the batch generator (what Keras' `fit_generator` does with
    its enqueuer thread, `max_queue_size` batches ahead): the host-side patch sampling
    and augmentation of batch i+1 overlap the GPU step of batch i - the step is a C call
    that releases the GIL.  Order is preserved (one worker); batches are copied because
    the reference-style generators reuse their output arrays."""

    def __init__(self, generator, depth=2):
        import queue
        import threading
        self._q = queue.Queue(depth)
        self._stop = threading.Event()
        self._full = queue.Full
        self._t = threading.Thread(target=self._work, args=(generator,), daemon=True)
        self._t.start()

    def _put(self, item):
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.1)
                return True
            except self._full:
                pass
        return False

    def _work(self, generator):
        try:
            for item in generator:
                if not self._put(tuple(np.array(a) for a in item)):
                    return
            self._put(StopIteration())
        except BaseException as e:          # surfaces in the training thread
            self._put(e)

    def __next__(self):
        item = self._q.get()
        if isinstance(item, BaseException):
            raise item
        return item

    def close(self):
        self._stop.set()
        self._t.join(5)


def fit_generator(network, generator, steps_per_epoch, epochs, log_file,
                  save_filepath, seed=0):
    graph = network.train_single
    args = network.compile_args or {}
    loss = args.get('loss', 'binary_crossentropy')
    loss = getattr(loss, '__name__', loss)
    # Keras names a metric column after the function ('accuracy' -> 'acc')
    metric_names = ['acc' if m in ('accuracy', 'acc') else getattr(m, '__name__', m)
                    for m in args.get('metrics', ['accuracy'])]
    opt = args.get('optimizer', 'adam')
    if opt not in _OPTIMIZERS:
        raise NotImplementedError('optimizer %r' % (opt,))
    ctx = runtime.get_context(network._device)
    trainer = _capi.Trainer(ctx, graph, loss=loss, **_OPTIMIZERS[opt])
    dist = _dist()
    rank = dist.get_rank() if dist else 0
    cols = sorted(set(metric_names + ['loss']))
    for k in cols:
        if k not in ('loss', 'acc', 'masked_accuracy', 'lb0l1err', 'lb1l1err'):
            raise NotImplementedError('metric %r' % (k,))
    writer = None
    if rank == 0 and log_file:
        f = open(log_file, 'w', newline='')
        writer = csv.writer(f)
        # CSVLogger: 'epoch' then the log keys in sorted order
        writer.writerow(['epoch'] + cols)
    step_no = 0
    history = []
    batches = _Prefetch(generator)
    try:
        for epoch in range(epochs):
            tot = dict.fromkeys(cols, 0.0)
            for _ in range(steps_per_epoch):
                data, labels = next(batches)
                trainer.step(data, labels, seed=seed + step_no)
                trainer.apply(allreduce_grads(trainer))
                m = trainer.metrics()
                for k in cols:
                    tot[k] += m[k]
                step_no += 1
            graph.set_weights(trainer.get_weights())
            row = (epoch,) + tuple(tot[k] / steps_per_epoch for k in cols)
            history.append(row)
            if writer:
                writer.writerow(row)
                f.flush()
            if rank == 0 and save_filepath:
                graph.save('%s_%03d' % (save_filepath, epoch))
    finally:
        batches.close()
    if writer:
        f.close()
    trainer.close()
    return history
