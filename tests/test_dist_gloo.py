"""CPU, world_size 2 over gloo: the N>1 paths.

* inference: every rank computes its Z slab as a standalone sub-volume (the CPU
  oracle stands in for the GPU compute); gathering rows reproduces the
  whole-volume result bit-for-bit, and the bench's max-over-ranks clock works.
* training: summing per-rank gradients with one all-reduce and scaling by
  1/world equals the gradient of the concatenated batch (BN-free graph; with BN
  each rank normalises its own slice, as the reference's towers do).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from flypylib_amd import fplmodels, multi_gpu, synth
from flypylib_amd.program import LayerGraph
from oracle import cnn_oracle, infer_oracle, train_oracle


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)


def _infer_worker(rank, world, port, out_q):
    _init(rank, world, port)
    tile, off = 30, 7
    g = fplmodels.vgg_like(tile)[0]
    synth.synthetic_weights(g, 31)
    Z, Y, X = 100, 38, 40
    u8 = synth.em_volume_u8(17, (Z, Y, X))
    img = (u8.astype(np.float32) - np.float32(128)) / np.float32(33)
    pitch = tile - 2 * off

    def predict(b):
        return cnn_oracle.vgg_like_forward(b.astype(np.float32), g.weights, 4)

    n_rows = multi_gpu.n_tile_rows(Z, tile, off)
    zb, ze = multi_gpu.slab_partition(n_rows, world)[rank]
    z_lo, z_hi = zb * pitch, min(ze * pitch + 2 * off, Z)
    # the rank's slab is a standalone volume whose lattice coincides with the
    # global one (what bench.py does with device-generated rows)
    part = infer_oracle.infer_lattice(img[z_lo:z_hi], (tile,) * 3, (off,) * 3, predict)
    lo, hi = multi_gpu.slab_rows((zb, ze), Z, tile, off)
    rows = torch.zeros((Z, Y, X), dtype=torch.float32)
    rows[lo:hi] = torch.from_numpy(part[lo - z_lo:hi - z_lo])
    dist.all_reduce(rows, op=dist.ReduceOp.SUM)          # disjoint rows -> gather
    t = torch.tensor([0.1 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        whole = infer_oracle.infer_lattice(img, (tile,) * 3, (off,) * 3, predict)
        out_q.put((bool(np.array_equal(rows.numpy(), whole)), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def _train_worker(rank, world, port, out_q):
    _init(rank, world, port)
    g = LayerGraph(None, seed=5)
    x = g.pool(g.conv(g.input(), 6, 3, use_bias=True))
    g.finish(g.conv(x, 1, 1, use_bias=True, activation='sigmoid'))
    rng = np.random.default_rng(100)
    data = rng.standard_normal((4, 8, 8, 8, 1)).astype(np.float32)
    labels = (rng.random((4, 3, 3, 3, 1)) > 0.5).astype(np.uint8)
    mine = slice(rank * 2, rank * 2 + 2)
    _, _, grads = train_oracle.train_step(g, g.weights, data[mine], labels[mine], 0)
    flat = torch.from_numpy(np.concatenate([x.reshape(-1) for x in grads]))
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat *= 1.0 / world
    if rank == 0:
        _, _, full = train_oracle.train_step(g, g.weights, data, labels, 0)
        ref = np.concatenate([x.reshape(-1) for x in full])
        out_q.put(float(np.max(np.abs(flat.numpy() - ref))))
    dist.barrier()
    dist.destroy_process_group()


def _run(worker):
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    return q.get()


@pytest.mark.timeout(600)
def test_slab_sharded_inference_two_ranks_gloo():
    same, tmax = _run(_infer_worker)
    assert same, 'gathered slabs differ from the whole-volume result'
    assert tmax == pytest.approx(1.1)


@pytest.mark.timeout(600)
def test_gradient_allreduce_two_ranks_gloo():
    err = _run(_train_worker)
    assert err < 1e-12


# ---- the product's own reducers (flypylib_amd/train.py) with a stand-in trainer ---------
class _FakeCtx:
    def __init__(self, device, uuid):
        self.device, self._uuid = device, uuid

    def comm_info(self):
        return dict(rank=0, nranks=0, lib='')

    def device_uuid(self):
        return self._uuid


class _FakeTrainer:
    """the three calls the reducers make on a _capi.Trainer"""

    def __init__(self, grads, ctx):
        self.g = np.asarray(grads, np.float32)
        self.ctx = ctx

    def get_grads_flat(self):
        return self.g.copy()

    def set_grads_flat(self, g):
        self.g = np.asarray(g, np.float32).copy()


def _reducer_worker(rank, world, port, out_q):
    _init(rank, world, port)
    from flypylib_amd import train
    # both ranks report the same GPU: they cannot form an RCCL communicator, so
    # setup_rank_comm must hand back the torch.distributed reducer (host-staged on gloo)
    ctx = _FakeCtx(0, '0000:05:00.0')
    red = train.setup_rank_comm(ctx, dist)
    tr = _FakeTrainer(np.arange(5) * (rank + 1), ctx)
    scale = red(tr)
    # and through the module-level entry point fit_generator's protocol uses
    tr2 = _FakeTrainer(np.full(3, 2.0 + rank), ctx)
    scale2 = train.allreduce_grads(tr2)
    out_q.put((rank, red.kind, scale, tr.g.copy(), scale2, tr2.g.copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_product_reducers_two_ranks_sharing_a_gpu():
    """train.setup_rank_comm / TorchDistReducer / allreduce_grads under a real 2-rank
    gloo group (the towers' code path when ranks share a GPU; the RCCL branch needs two
    devices and is covered on the GPU box)"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r = q.get(timeout=120)
        got[r[0]] = r
    for p in procs:
        p.join(30)
    for rank in (0, 1):
        _, kind, scale, g, scale2, g2 = got[rank]
        assert kind == 'torch.distributed' and scale == 0.5 and scale2 == 0.5
        assert np.array_equal(g, np.arange(5) * 3.0)          # sum over the two ranks
        assert np.array_equal(g2, np.full(3, 5.0))


def test_host_tower_reducer_threads():
    """HostTowerReducer: n towers of one process meet at a barrier, every one leaves
    with the same sum"""
    import threading
    from flypylib_amd import train
    n = 3
    red = train.HostTowerReducer(n)
    trainers = [_FakeTrainer(np.arange(4) + 10 * r, _FakeCtx(0, 'x')) for r in range(n)]
    scales = [None] * n

    def work(r):
        for _ in range(2):                                   # two steps: the barrier re-arms
            scales[r] = red(trainers[r], r)

    ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(30)
    first = np.arange(4) * 3 + 30.0                          # sum of the towers, step 1
    for r in range(n):
        assert scales[r] == 1.0 / n
        assert np.array_equal(trainers[r].g, first * 3)      # step 2 summed three equal arenas
