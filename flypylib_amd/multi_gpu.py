"""Multi-GPU fan-out for inference.

The reference replicates the Keras model as in-graph towers and concatenates
tower outputs on the CPU (`flypylib/multi_gpu.py:20-61`).  Inference tiles are
independent, so here the tile lattice is cut into contiguous slabs of tile rows
along Z, one slab per GPU, with no inter-GPU traffic: every GPU reads its slab
(plus rf_offset halo) and writes its own rows of the output.

Two launch styles share `slab_partition`:
  * one process per GPU (`torchrun`, bench.py): rank r computes slab r;
  * `make_parallel(network, n_gpu)`: the single-process drop-in used by
    `FplNetwork.make_infer_parallel`, one host thread + one context per GPU.
"""
import threading

import numpy as np


def n_tile_rows(dim, tile_in, offset):
    """number of tile origins along one axis (fplnetwork.py:151-155)"""
    out = tile_in - 2 * offset
    return len(range(offset, dim - offset, out))


def slab_partition(n_rows, n_parts):
    """contiguous [begin, end) ranges, sizes balanced to +-1, in rank order"""
    base, extra = divmod(int(n_rows), int(n_parts))
    out, b = [], 0
    for r in range(n_parts):
        e = b + base + (1 if r < extra else 0)
        out.append((b, e))
        b = e
    return out


def slab_rows(z_range, dim, tile_in, offset):
    """output rows [lo, hi) of the volume owned by tile rows z_range; the first
    slab also owns the leading border shell and the last one the trailing one"""
    out = tile_in - 2 * offset
    n = n_tile_rows(dim, tile_in, offset)
    b, e = z_range
    if b >= e:
        return (0, 0)
    lo = 0 if b == 0 else offset + b * out
    hi = dim if e == n else offset + e * out
    return (lo, hi)


class ParallelInfer:
    """slab-sharded inference over `n_gpu` devices from one process"""

    def __init__(self, build_program, n_gpu):
        # build_program(device) -> _capi.Program resident on that device
        self.programs = [build_program(d) for d in range(n_gpu)]
        self.n_gpu = n_gpu

    def infer_volume(self, image, tile_in, offset, **kw):
        dims = image.shape
        parts = slab_partition(n_tile_rows(dims[0], tile_in[0], offset[0]),
                               self.n_gpu)
        from . import _capi
        out = _capi.host_empty(dims, np.float32)
        if not all(hi > lo for lo, hi in (slab_rows(p, dims[0], tile_in[0], offset[0])
                                          for p in parts)):
            out[...] = 0                 # more GPUs than tile rows: idle ranks write nothing
        errs = []

        def work(rank):
            try:
                zr = parts[rank]
                lo, hi = slab_rows(zr, dims[0], tile_in[0], offset[0])
                if hi <= lo:
                    return
                # the library writes exactly the rows [lo, hi) of a host `dst`
                # (its slab + the border shell at the volume's ends)
                self.programs[rank].infer_volume(
                    image, tile_in, offset, z_range=zr, dst=out, **kw)
            except Exception as e:       # surfaced after join
                errs.append(e)

        threads = [threading.Thread(target=work, args=(r,))
                   for r in range(self.n_gpu)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errs:
            raise errs[0]
        return out


def make_parallel(build_program, gpu_count):
    return ParallelInfer(build_program, gpu_count)
