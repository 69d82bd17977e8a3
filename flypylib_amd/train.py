"""Training loop of `FplNetwork.train` / `make_train_parallel`
(reference `flypylib/fplnetwork.py:112-128`) on the HIP training engine.

`fit_generator(network, generator, steps_per_epoch, epochs, log_file,
save_filepath)` mirrors `train_network.fit_generator` with the reference's two
callbacks: a CSV log (`CSVLogger`: epoch, acc, loss) and a per-epoch checkpoint
'<save_filepath>_%03d.h5' (Keras weight layout, keras_io.py; plus the package's .npz)
of the single-GPU weights (`multi_gpu_callback`, fplnetwork.py:9-17).

Data parallelism.  The reference builds in-graph towers: the generator yields
`batch_size * n_gpu` examples, tower i takes rows [i*batch_size, (i+1)*batch_size)
(`flypylib/multi_gpu.py:21-25`), the loss is the mean over the concatenated outputs
and the towers share their variables, so the applied gradient is the mean of the
tower gradients; each tower normalises its own slice (BN batch statistics per tower).
Here a tower is a `fpl_trainer` on its own GPU and the gradient mean is ONE sum
all-reduce of the flat gradient arena over RCCL (`fpl_allreduce_grads`) followed by
the identical Adam update scaled by 1/n on every tower.  BN moving averages: every
tower's pending delta `(1 - momentum) * (batch_stat - moving)` rides in the same
arena, so the applied delta is the mean over the towers.  Two launch styles:

  * one process, `make_train_parallel(n_gpu, batch_size, input_shape)` as the
    reference's scripts call it: one host thread + context + trainer per GPU
    (`TowerGroup`); the threads join one RCCL communicator.
  * one process per GPU (`torchrun`): an initialised `torch.distributed` process
    group carries the RCCL unique id to every rank (`setup_rank_comm`) and the
    library all-reduces over its own communicator.  After `make_train_parallel(world,
    batch_size, ...)` rank r takes rows [r*batch_size, (r+1)*batch_size) of the
    generator's batch (the reference's slicing; the generators of all ranks must then
    yield the same batches, i.e. be seeded alike); without it every rank trains on the
    whole batch its own generator yields.
    Ranks that share a GPU (a rehearsal on a one-GPU box) cannot form an RCCL
    communicator; they reduce through `torch.distributed` instead (host-staged for the
    gloo backend).
"""
import csv
import os
import queue
import threading

import numpy as np

from . import _capi, runtime

_OPTIMIZERS = {'adam': dict(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8)}
_METRICS = ('loss', 'acc', 'masked_accuracy', 'lb0l1err', 'lb1l1err')


class ParallelTrainNetwork:
    """what `network.train_network` holds after `make_train_parallel`: the tower
    layout (reference `multi_gpu.make_parallel(train_single, n_gpu, batch_size,
    input_shape)`); `fit_generator` trains through it.  `devices` defaults to GPUs
    0..n_gpu-1; naming a device twice puts two towers on it (rehearsal on one GPU)."""

    def __init__(self, single, n_gpu, batch_size, input_shape, devices=None):
        self.single, self.n_gpu = single, int(n_gpu)
        self.batch_size, self.input_shape = int(batch_size), tuple(input_shape)
        self.devices = (list(range(self.n_gpu)) if devices is None
                        else [int(d) for d in devices])
        if len(self.devices) != self.n_gpu:
            raise ValueError('make_parallel: %d devices for %d towers'
                             % (len(self.devices), self.n_gpu))
        self.compile_args = None
        self._towers = None

    def compile(self, **kw):
        # (a new Keras model, a new optimizer: `multi_gpu.make_parallel` + compile,
        # flypylib/fplnetwork.py:124-128 - the towers' Adam starts from zero)
        self.compile_args = dict(kw)
        self.single.opt_state = None
        self.single.compile_generation = getattr(self.single, 'compile_generation', 0) + 1
        self.close()

    def get_weights(self):
        return self.single.get_weights()

    def close(self):
        if self._towers is not None:
            self._towers.close()
            self._towers = None

    def __getstate__(self):
        d = dict(self.__dict__)
        d['_towers'] = None
        return d


def make_parallel(train_single, n_gpu, batch_size, input_shape, devices=None):
    return ParallelTrainNetwork(train_single, n_gpu, batch_size, input_shape, devices)


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except ImportError:
        pass
    return None


# ---- gradient reducers ------------------------------------------------------------
class RcclReducer:
    """the library's own communicator (fpl_comm_init): stream-ordered, no host sync"""
    kind = 'rccl'

    def __init__(self, nranks):
        self.nranks = int(nranks)

    def __call__(self, trainer):
        trainer.allreduce_grads()
        return 1.0 / self.nranks


class TorchDistReducer:
    """all-reduce through an initialised torch.distributed group: for ranks that
    cannot form an RCCL communicator of their own (several ranks on one GPU).  With
    the nccl backend the arena is reduced in place through a zero-copy device view;
    any other backend stages it through host memory."""
    kind = 'torch.distributed'

    def __init__(self, dist):
        self.dist = dist
        self.nranks = dist.get_world_size()

    def __call__(self, trainer):
        import torch
        dist = self.dist
        if dist.get_backend() == 'nccl':
            dev = torch.device('cuda', trainer.ctx.device)
            ptr, n = trainer.grad_ptr()
            trainer.ctx.synchronize()        # the step ran on the context's stream
            with torch.cuda.device(dev):
                t = torch.as_tensor(_DeviceArena(ptr, n), device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                torch.cuda.current_stream(dev).synchronize()
        else:
            t = torch.from_numpy(trainer.get_grads_flat())
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            trainer.set_grads_flat(t.numpy())
        return 1.0 / self.nranks


class _DeviceArena:
    """__cuda_array_interface__ view of `n` float32 at device address `ptr`"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {
            'shape': (int(n),), 'typestr': '<f4', 'data': (int(ptr), False),
            'version': 2}


class HostTowerReducer:
    """sum over the towers of one process through host memory: for towers that share
    a device, which RCCL refuses to put in one communicator"""
    kind = 'host'

    def __init__(self, n):
        self.nranks = int(n)
        self._barrier = threading.Barrier(self.nranks)
        self._parts = [None] * self.nranks
        self._sum = None

    def __call__(self, trainer, rank):
        self._parts[rank] = trainer.get_grads_flat()
        if self._barrier.wait() == 0:
            self._sum = np.sum(np.stack(self._parts), axis=0, dtype=np.float32)
        self._barrier.wait()
        trainer.set_grads_flat(self._sum)
        self._barrier.wait()
        return 1.0 / self.nranks

    def abort(self):
        self._barrier.abort()


def allreduce_grads(trainer, force=False):
    """sum the flat gradient arena over all ranks; returns the scale to apply
    (1/world).  The context's own RCCL communicator when it has one, else the
    torch.distributed group; a no-op without either (or with a single rank, unless
    `force`)."""
    info = trainer.ctx.comm_info()
    if info['nranks'] > 1 or (force and info['nranks'] == 1):
        return RcclReducer(info['nranks'])(trainer)
    dist = _dist()
    if dist is None or (dist.get_world_size() == 1 and not force):
        return 1.0
    return TorchDistReducer(dist)(trainer)


def setup_rank_comm(ctx, dist=None):
    """one process per GPU: rank 0 makes the RCCL unique id, the process group
    carries it, every rank joins (fpl_comm_init).  Returns the reducer to use: the
    library's communicator when every rank has a GPU of its own, else the
    torch.distributed one."""
    dist = dist or _dist()
    if dist is None or dist.get_world_size() == 1:
        return None
    world, rank = dist.get_world_size(), dist.get_rank()
    info = ctx.comm_info()
    if info['nranks'] == world and info['rank'] == rank:
        return RcclReducer(world)
    import socket
    where = [None] * world
    dist.all_gather_object(where, (socket.gethostname(), ctx.device_uuid()))
    if len(set(where)) < world or os.environ.get('FPL_TRAIN_REDUCE') == 'torch':
        return TorchDistReducer(dist)
    box = [_capi.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    if info['nranks']:
        ctx.comm_destroy()
    ctx.comm_init(rank, world, box[0])
    return RcclReducer(world)


# ---- towers of one process ----------------------------------------------------------
class TowerGroup:
    """`n` trainers, one host thread each, stepping in lockstep on slices of one
    batch (the single-process form of make_train_parallel)"""

    def __init__(self, graph, devices, loss, opt_args):
        self.n = len(devices)
        self.devices = list(devices)
        shared = len(set(devices)) < self.n
        self.ctxs, self.trainers = [], []
        lanes = {}
        for d in devices:
            lane = lanes.get(d, 0)
            lanes[d] = lane + 1
            # towers that share a device need contexts of their own: lanes 16+
            ctx = runtime.get_context(d, 16 + lane if shared else 0)
            self.ctxs.append(ctx)
            self.trainers.append(_capi.Trainer(ctx, graph, loss=loss, **opt_args))
        self._jobs = [queue.Queue() for _ in range(self.n)]
        self._done = queue.Queue()
        self._broken = None
        # a tower that neither answers nor fails within this many seconds is taken for
        # stuck in a collective whose peer is gone (FPL_TOWER_TIMEOUT overrides)
        self.timeout = float(os.environ.get('FPL_TOWER_TIMEOUT', '600'))
        self._host = HostTowerReducer(self.n) if (shared or os.environ.get(
            'FPL_TRAIN_REDUCE') == 'host') and self.n > 1 else None
        self.reduce_kind = 'host' if self._host else ('rccl' if self.n > 1 else 'none')
        uid = _capi.comm_unique_id() if self.reduce_kind == 'rccl' else None
        self._threads = [threading.Thread(target=self._work, args=(r, uid), daemon=True)
                         for r in range(self.n)]
        for t in self._threads:
            t.start()
        self._collect()                      # communicator formed (or failed)

    def _work(self, rank, uid):
        ctx, tr = self.ctxs[rank], self.trainers[rank]
        try:
            if uid is not None:
                if ctx.comm_info()['nranks']:
                    ctx.comm_destroy()
                ctx.comm_init(rank, self.n, uid)
            self._done.put((rank, None))
        except BaseException as e:
            self._done.put((rank, e))
            return
        while True:
            job = self._jobs[rank].get()
            if job is None:
                return
            kind, args = job
            try:
                if kind == 'step':
                    data, labels, seed = args
                    tr.step(data, labels, seed=seed)
                    if self._host is not None:
                        scale = self._host(tr, rank)
                    elif self.n > 1:
                        tr.allreduce_grads()
                        scale = 1.0 / self.n
                    else:
                        scale = 1.0
                    tr.apply(scale)
                    self._done.put((rank, tr.metrics()))
                elif kind == 'set_weights':
                    tr.set_weights(args)
                    self._done.put((rank, None))
            except BaseException as e:
                if self._host is not None:
                    self._host.abort()
                self._done.put((rank, e))

    def _collect(self):
        """one answer per tower.  A tower that raised leaves its peers inside an RCCL
        collective (ncclCommInitRank / ncclAllReduce) that can never complete: as soon as
        one error - or no answer within `timeout` - is seen, the communicators of ALL
        towers are aborted from this thread, which releases the blocked ones; the group is
        then unusable and says so."""
        if self._broken is not None:
            raise RuntimeError('this tower group failed earlier (%s); build a new one with '
                               'make_train_parallel' % self._broken)
        out = [None] * self.n
        errs = []
        got = 0
        while got < self.n:
            try:
                rank, res = self._done.get(timeout=30.0 if errs else self.timeout)
            except queue.Empty:
                if not errs:
                    errs.append(TimeoutError('a training tower did not answer within %.0f s'
                                             % self.timeout))
                    self._abort_comms()
                    continue
                break                                   # aborted and still silent: give up
            got += 1
            if isinstance(res, BaseException):
                if not errs:
                    self._abort_comms()
                errs.append(res)
            out[rank] = res
        if errs:
            # the cause, not the peers' "barrier broken" echo of it
            err = next((e for e in errs if not isinstance(e, threading.BrokenBarrierError)),
                       errs[0])
            self._broken = '%s: %s' % (type(err).__name__, err)
            raise err
        return out

    def _abort_comms(self):
        if self._host is not None:
            self._host.abort()
        if self.reduce_kind == 'rccl':
            for ctx in self.ctxs:
                try:
                    ctx.comm_abort()
                except Exception:                       # best effort: we are already failing
                    pass

    def set_weights(self, weights):
        for q in self._jobs:
            q.put(('set_weights', weights))
        self._collect()

    def get_weights(self):
        return self.trainers[0].get_weights()

    def get_opt_state(self):
        return self.trainers[0].get_opt_state()       # the towers' Adam states are identical

    def set_opt_state(self, m, v, steps):
        for tr in self.trainers:
            tr.set_opt_state(m, v, steps)

    def step(self, data, labels, batch_size, seed):
        """tower i trains on rows [i*batch_size, (i+1)*batch_size); returns the
        metrics of the whole batch (mean over the equally sized towers)"""
        need = batch_size * self.n
        if data.shape[0] != need:
            raise ValueError(
                'make_train_parallel(%d, %d, ...): the generator must yield batches of '
                '%d examples (batch_size * n_gpu, as the reference towers slice them), '
                'got %d' % (self.n, batch_size, need, data.shape[0]))
        for r, q in enumerate(self._jobs):
            sl = slice(r * batch_size, (r + 1) * batch_size)
            q.put(('step', (data[sl], labels[sl], seed * self.n + r)))
        ms = self._collect()
        return {k: float(np.mean([m[k] for m in ms])) for k in ms[0]}

    def close(self):
        for q in self._jobs:
            q.put(None)
        for t in self._threads:
            t.join(None if self._broken is None else 5.0)    # daemon threads: never hang here
        for t, tr, ctx in zip(self._threads, self.trainers, self.ctxs):
            if t.is_alive():
                # a tower still inside a library call (after a failure): its trainer and context
                # are NOT torn down under it - leaked with the daemon thread
                continue
            tr.close()
            if self.reduce_kind == 'rccl' and ctx.comm_info()['nranks']:
                ctx.comm_destroy()


class _Prefetch:
    """Background consumer of the batch generator (what Keras' `fit_generator` does with
    its enqueuer thread, `max_queue_size` batches ahead): the host-side patch sampling
    and augmentation of batch i+1 overlap the GPU step of batch i - the step is a C call
    that releases the GIL.  Order is preserved (one worker); batches are copied because
    the reference-style generators reuse their output arrays."""

    def __init__(self, generator, depth=2, stage=None):
        self._q = queue.Queue(depth)
        self._stop = threading.Event()
        self._stage = stage
        self._t = threading.Thread(target=self._work, args=(generator,), daemon=True)
        self._t.start()

    def _put(self, item):
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def _work(self, generator):
        try:
            for item in generator:
                if self._stage is not None:
                    item = self._stage(*item)     # its own copies, already on the GPU
                else:
                    item = tuple(np.array(a) for a in item)
                if not self._put(item):
                    return
            self._put(StopIteration())
        except BaseException as e:          # surfaces in the training thread
            self._put(e)

    def __next__(self):
        item = self._q.get()
        if isinstance(item, BaseException):
            raise item
        return item

    def close(self, timeout=60.0):
        """stop the worker and wait for it: a generator still executing in the worker
        would raise 'generator already executing' in the next train() call"""
        self._stop.set()
        self._t.join(timeout)
        if self._t.is_alive():
            raise RuntimeError('the batch generator did not return within %.0f s of '
                               'the end of training' % timeout)


class _DeviceStager:
    """Upload of batch i+1 overlapped with the GPU step of batch i: runs in the prefetch
    worker, copies this rank's rows of the generator's (host) batch to the trainer's GPU on
    a side stream and waits for that copy, so the training thread receives device tensors
    and fpl_trainer_step starts without the 33 MB host-to-device copy in front of it (about
    0.8 ms of a 12.4 ms vgg_like step, configs[3]).  The reference's generators yield host
    arrays (`gen_batches`, flypylib/fplobjdetect.py:27-130); where they live afterwards is plumbing."""

    def __init__(self, device, rows=None, need=None, need_msg=None):
        import torch
        self._torch = torch
        self._dev = torch.device('cuda', device)
        if device >= torch.cuda.device_count():
            raise ValueError('torch sees %d GPUs; the trainer is on device %d (the stager needs the '
                             'same device numbering as the library)' % (torch.cuda.device_count(), device))
        self._stream = torch.cuda.Stream(self._dev)
        self._rows, self._need, self._need_msg = rows, need, need_msg

    def __call__(self, data, labels):
        torch = self._torch
        if self._need is not None and data.shape[0] != self._need:
            raise ValueError(self._need_msg % data.shape[0])
        if self._rows is not None:
            data, labels = data[self._rows], labels[self._rows]
        x = np.ascontiguousarray(data, np.float32)
        y = np.ascontiguousarray(labels, np.uint8)
        with torch.cuda.stream(self._stream):
            xd = torch.from_numpy(x).to(self._dev)
            yd = torch.from_numpy(y).to(self._dev)
        self._stream.synchronize()
        return xd, yd


def _single_trainer(network, graph, loss, opt, opt_args):
    """one trainer per network, kept across train() calls like a compiled Keras
    model keeps its optimizer: Adam moments and the step count persist"""
    key = (network._device, loss, opt, len(graph.weights), getattr(graph, 'compile_generation', 0))
    cached = getattr(network, '_trainer', None)
    if cached is not None and cached[0] == key and cached[1].h is not None \
            and cached[1].ctx.h is not None:
        trainer = cached[1]
        trainer.set_weights(graph.get_weights())
        return trainer
    if cached is not None:
        cached[1].close()
    ctx = runtime.get_context(network._device)
    trainer = _capi.Trainer(ctx, graph, loss=loss, **opt_args)
    if getattr(graph, 'opt_state', None) is not None:
        # a loaded network resumes with the optimizer it was saved with (Keras' load_model
        # restores `optimizer_weights`, flypylib/fplnetwork.py:32-44)
        trainer.set_opt_state(*graph.opt_state)
    network._trainer = (key, trainer)
    return trainer


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
    cols = sorted(set(metric_names + ['loss']))
    for k in cols:
        if k not in _METRICS:
            raise NotImplementedError('metric %r' % (k,))

    dist = _dist()
    world = dist.get_world_size() if dist else 1
    rank = dist.get_rank() if dist else 0
    par = network.train_network if isinstance(network.train_network,
                                              ParallelTrainNetwork) else None
    towers = trainer = reducer = None
    if world > 1:
        # one process per GPU
        if par is not None and par.n_gpu != world:
            raise ValueError('make_train_parallel(n_gpu=%d) under a process group of '
                             '%d ranks: they must agree' % (par.n_gpu, world))
        trainer = _single_trainer(network, graph, loss, opt, _OPTIMIZERS[opt])
        reducer = setup_rank_comm(trainer.ctx, dist)
        if reducer.kind == 'rccl':
            trainer.broadcast_state(0)
        else:
            box = [graph.get_weights() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            trainer.set_weights(box[0])
    elif par is not None and par.n_gpu > 1:
        # one process, one tower per GPU
        key = (tuple(par.devices), loss, opt, len(graph.weights))
        if par._towers is None or par._towers_key != key:
            par.close()
            par._towers = TowerGroup(graph, par.devices, loss, _OPTIMIZERS[opt])
            if getattr(graph, 'opt_state', None) is not None:
                par._towers.set_opt_state(*graph.opt_state)
            par._towers_key = key
        towers = par._towers
        towers.set_weights(graph.get_weights())
    else:
        trainer = _single_trainer(network, graph, loss, opt, _OPTIMIZERS[opt])
    network.train_reduce_kind = (towers.reduce_kind if towers else
                                 reducer.kind if reducer else 'none')

    writer = None
    if rank == 0 and log_file:
        f = open(log_file, 'w', newline='')
        writer = csv.writer(f)
        # CSVLogger: 'epoch' then the log keys in sorted order
        writer.writerow(['epoch'] + cols)
    step_no = getattr(network, '_train_steps_done', 0)
    history = []
    stage = None
    try:
        import torch  # noqa: F401   (the stager moves batches with torch; without it: host batches)
        have_torch = True
    except ImportError:
        have_torch = False
    if towers is None and have_torch and not os.environ.get('FPL_TRAIN_HOST_BATCHES'):
        if world > 1 and par is not None:
            need = par.batch_size * world
            stage = _DeviceStager(
                trainer.ctx.device, slice(rank * par.batch_size, (rank + 1) * par.batch_size),
                need, 'make_train_parallel(%d, %d, ...): the generator must yield batches of '
                      '%d examples, got %%d' % (world, par.batch_size, need))
        else:
            stage = _DeviceStager(trainer.ctx.device)
    batches = _Prefetch(generator, stage=stage)
    try:
        for epoch in range(epochs):
            tot = dict.fromkeys(cols, 0.0)
            for _ in range(steps_per_epoch):
                data, labels = next(batches)
                if towers is not None:
                    m = towers.step(data, labels, par.batch_size, seed + step_no)
                else:
                    if stage is None and world > 1 and par is not None:
                        need = par.batch_size * world
                        if data.shape[0] != need:
                            raise ValueError(
                                'make_train_parallel(%d, %d, ...): the generator must '
                                'yield batches of %d examples, got %d'
                                % (world, par.batch_size, need, data.shape[0]))
                        sl = slice(rank * par.batch_size, (rank + 1) * par.batch_size)
                        data, labels = data[sl], labels[sl]
                    trainer.step(data, labels, seed=(seed + step_no) * world + rank)
                    trainer.apply(reducer(trainer) if reducer else 1.0)
                    m = trainer.metrics()
                for k in cols:
                    tot[k] += m[k]
                step_no += 1
            weights = towers.get_weights() if towers else trainer.get_weights()
            graph.set_weights(weights)
            graph.opt_state = (towers or trainer).get_opt_state()     # travels with every checkpoint
            vals = [tot[k] / steps_per_epoch for k in cols]
            if world > 1:
                # the towers' mean, as Keras reports it for the concatenated batch
                import torch
                t = torch.tensor(vals, dtype=torch.float64)
                if dist.get_backend() == 'nccl':
                    t = t.to(torch.device('cuda', trainer.ctx.device))
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                vals = [float(v) / world for v in t.cpu()]
            row = (epoch,) + tuple(vals)
            history.append(row)
            if writer:
                writer.writerow(row)
                f.flush()
            if rank == 0 and save_filepath:
                # the reference's per-epoch '<prefix>_%03d.h5' (Keras layout) next to the
                # package's own .npz
                graph.save('%s_%03d' % (save_filepath, epoch))
                graph.save('%s_%03d.h5' % (save_filepath, epoch))
    except BaseException:
        # the training error is the one to report: a generator that does not stop in
        # time must not replace it (nor keep the log file open)
        network._train_steps_done = step_no
        if writer:
            f.close()
        try:
            batches.close(timeout=5.0)
        except RuntimeError as e:
            import warnings
            warnings.warn(str(e))
        raise
    network._train_steps_done = step_no
    if writer:
        f.close()
    batches.close()
    return history
