"""ctypes binding of libfplhip.so (include/fplhip.h).

The product path has no CPU fallback: if the library is missing or a call fails,
`FplHipError` is raised with the library's own message.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libfplhip.so')

MEM_HOST, MEM_DEVICE = 0, 1
U8, F32, F64 = 0, 1, 2
PREC_AUTO, PREC_F32, PREC_BF16, PREC_F16, PREC_F16S = -1, 0, 1, 2, 3
ABI_VERSION = 8
COMM_ID_BYTES = 128


class FplHipError(RuntimeError):
    pass


class fpl_op(C.Structure):
    _fields_ = [('kind', C.c_int32), ('src0', C.c_int32), ('src1', C.c_int32),
                ('dst', C.c_int32), ('k', C.c_int32), ('cin', C.c_int32),
                ('cout', C.c_int32), ('act', C.c_int32), ('w_off', C.c_int64),
                ('scale_off', C.c_int64), ('shift_off', C.c_int64),
                ('p', C.c_int32 * 6)]


class fpl_layer(C.Structure):
    _fields_ = [('kind', C.c_int32), ('src0', C.c_int32), ('src1', C.c_int32),
                ('dst', C.c_int32), ('k', C.c_int32), ('cin', C.c_int32),
                ('cout', C.c_int32), ('use_bias', C.c_int32), ('act', C.c_int32),
                ('rate', C.c_float), ('p', C.c_int32 * 6),
                ('w_off', C.c_int64 * 4)]


_vp, _i32, _i64, _f32, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
_pi32, _pi64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)

# name -> (restype, argtypes); every symbol include/fplhip.h declares
SIGNATURES = {
    'fpl_abi_version': (C.c_int, []),
    'fpl_ctx_create': (C.c_int, [C.c_int, C.POINTER(_vp)]),
    'fpl_ctx_destroy': (C.c_int, [_vp]),
    'fpl_last_error': (C.c_char_p, [_vp]),
    'fpl_ctx_set_stream': (C.c_int, [_vp, _vp]),
    'fpl_ctx_synchronize': (C.c_int, [_vp]),
    'fpl_device_info': (C.c_int, [_vp, _pi32, _pi64, C.c_char_p, C.c_size_t]),
    'fpl_device_pci_bus_id': (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    'fpl_malloc': (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    'fpl_free': (C.c_int, [_vp, _vp]),
    'fpl_memcpy': (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_size_t]),
    'fpl_program_create': (C.c_int, [_vp, C.POINTER(fpl_op), _i32, _i32, _i32,
                                     _vp, _i64, _pi32, C.POINTER(_vp)]),
    'fpl_program_destroy': (C.c_int, [_vp]),
    'fpl_program_set_arena': (C.c_int, [_vp, _vp, _i64]),
    'fpl_program_forward': (C.c_int, [_vp, _vp, _vp, C.c_int, _i32, _pi32,
                                      C.c_int, _vp, C.c_int, _pi32]),
    'fpl_infer_volume': (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _f32, _f32,
                                   _pi64, _pi32, _pi32, C.c_int, _i32, _i32,
                                   _vp, C.c_int]),
    'fpl_v2o_smooth': (C.c_int, [_vp, _vp, C.c_int, _pi64, _i32, _vp, _i32,
                                 _pi64, _i32, _vp]),
    'fpl_v2o_set_floor': (C.c_int, [_vp, _f32]),
    'fpl_v2o_nms': (C.c_int, [_vp, _f64, _vp, _i64, _pi64, _pi32]),
    'fpl_v2o_set_seg': (C.c_int, [_vp, _vp, _i32, C.c_int, _pi64, _i64]),
    'fpl_v2o_select': (C.c_int, [_vp, _pi64, _i32, _vp]),
    'fpl_v2o_smooth_f64': (C.c_int, [_vp, _vp, C.c_int, _pi64, _i32, _vp, _i32]),
    'fpl_v2o_set_integer': (C.c_int, [_vp, _i32]),
    'fpl_v2o_select_f64': (C.c_int, [_vp, _pi64, _i32, _vp]),
    'fpl_v2o_rank_f64': (C.c_int, [_vp, C.c_double, _pi64]),
    'fpl_v2o_values_f64': (C.c_int, [_vp, _pi64, _i64, _vp]),
    'fpl_v2o_nms_seg': (C.c_int, [_vp, C.c_double, _i32, _i32, _vp, _i64, _pi64, _pi32]),
    'fpl_v2o_copy_smoothed': (C.c_int, [_vp, _vp, C.c_int]),
    'fpl_trainer_create': (C.c_int, [_vp, C.POINTER(fpl_layer), _i32, _i32, _i32,
                                     _vp, _i64, _f32, _f32, _f32, _f32,
                                     C.POINTER(_vp)]),
    'fpl_trainer_destroy': (C.c_int, [_vp]),
    'fpl_trainer_step': (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _i32, _pi32,
                                   C.c_uint64, C.POINTER(_f32), C.POINTER(_f32)]),
    'fpl_trainer_set_loss': (C.c_int, [_vp, C.c_int]),
    'fpl_trainer_metric_sums': (C.c_int, [_vp, C.POINTER(C.c_double)]),
    'fpl_trainer_apply': (C.c_int, [_vp, _f32]),
    'fpl_trainer_grad_ptr': (C.c_int, [_vp, C.POINTER(_vp), _pi64]),
    'fpl_trainer_get_weights': (C.c_int, [_vp, _vp, _i64]),
    'fpl_trainer_set_weights': (C.c_int, [_vp, _vp, _i64]),
    'fpl_trainer_get_opt_state': (C.c_int, [_vp, _vp, _vp, _i64, C.POINTER(C.c_int64)]),
    'fpl_trainer_set_opt_state': (C.c_int, [_vp, _vp, _vp, _i64, _i64]),
    'fpl_trainer_get_grads': (C.c_int, [_vp, _vp, _i64]),
    'fpl_trainer_set_grads': (C.c_int, [_vp, _vp, _i64]),
    'fpl_last_path': (C.c_char_p, [_vp]),
    'fpl_comm_unique_id': (C.c_int, [_vp]),
    'fpl_comm_init': (C.c_int, [_vp, _i32, _i32, _vp]),
    'fpl_comm_destroy': (C.c_int, [_vp]),
    'fpl_comm_abort': (C.c_int, [_vp]),
    'fpl_comm_info': (C.c_int, [_vp, _pi32, _pi32, C.c_char_p, C.c_size_t]),
    'fpl_comm_allreduce_sum_f32': (C.c_int, [_vp, _vp, _i64]),
    'fpl_comm_broadcast_f32': (C.c_int, [_vp, _vp, _i64, _i32]),
    'fpl_allreduce_grads': (C.c_int, [_vp]),
    'fpl_trainer_broadcast_state': (C.c_int, [_vp, _i32]),
    'fpl_synth_volume_u8': (C.c_int, [_vp, C.c_uint64, _pi64, _pi64, _vp,
                                      C.c_int]),
    'fpl_synth_substack_u8': (C.c_int, [_vp, C.c_uint64, _pi64, _pi64, _pi64, _vp, C.c_int]),
    'fpl_crop_substack_u8': (C.c_int, [_vp, _vp, _pi64, _pi64, _pi64, _vp]),
    'fpl_histogram_u8': (C.c_int, [_vp, _vp, C.c_int, _i64, C.POINTER(C.c_uint64)]),
    'fpl_timing_enable': (C.c_int, [_vp, C.c_int]),
    'fpl_timing_reset': (C.c_int, [_vp]),
    'fpl_timing_get': (C.c_int, [_vp, _vp, _vp, _vp, _i32, _pi32]),
}

_lib = None


def load_library(path=None):
    """dlopen libfplhip.so and bind every declared symbol (no GPU needed)"""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise FplHipError(
            'libfplhip.so not found at %s - build it with '
            '`python -m flypylib_amd.csrc.build` (there is no CPU fallback)'
            % path)
    # One HIP runtime per process: PyTorch-ROCm carries its own libamdhip64, and whichever
    # copy is loaded second finds "no HIP GPUs".  With torch imported first, libfplhip.so's
    # NEEDED libamdhip64 resolves to the copy torch already loaded, and device tensors,
    # streams and this library's contexts share one runtime (train._DeviceStager,
    # pipeline buffers, torch.distributed).
    # (FPL_NO_TORCH_PRELOAD=1 skips it - a host without torch-side device buffers does not
    # need it; a torch install that fails to import must not take this package down with it)
    if not os.environ.get('FPL_NO_TORCH_PRELOAD'):
        try:
            import torch  # noqa: F401
        except Exception:       # noqa: BLE001 - ImportError, or OSError / RuntimeError of a broken install
            pass
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.fpl_abi_version() != ABI_VERSION:
        raise FplHipError('libfplhip.so ABI %d, binding expects %d'
                          % (lib.fpl_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def _ptr(a):
    """void* of a numpy array, torch tensor (data_ptr) or raw int address"""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(_vp)
    if hasattr(a, 'data_ptr'):
        return _vp(a.data_ptr())
    return _vp(int(a))


def _mem_of(a):
    if isinstance(a, np.ndarray):
        return MEM_HOST
    if hasattr(a, 'is_cuda'):
        return MEM_DEVICE if a.is_cuda else MEM_HOST
    return MEM_DEVICE


def _arr(vals, ctype):
    return (ctype * len(vals))(*[int(v) for v in vals])


def comm_unique_id():
    """128-byte RCCL unique id (fpl_comm_unique_id): made by rank 0, carried to the
    other ranks by the host, consumed by `Context.comm_init`"""
    lib = load_library()
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    if lib.fpl_comm_unique_id(buf) != 0:
        raise FplHipError(lib.fpl_last_error(None).decode())
    return bytes(buf)


class _HostPool:
    """Recycled host memory for the arrays `infer_volume` returns (FplNetwork.infer: a full-resolution
    float32 volume per call, flypylib/fplnetwork.py:136-189).  Memory fresh from the kernel is zeroed page
    by page at its first touch - 45 of the 68 ms of a 520^3 call when that touch is the device-to-host
    copy - so blocks are kept when their array (and every view of it) has died and handed out again:
    22 ms from the third call of a loop on.  A block is owned by the ctypes buffer the array is built on
    (`ndarray.base` of every view leads there), never by one ndarray object, so a caller that keeps a
    slice keeps the block.  At most FPL_HOST_POOL_MB (default 4096) of dead blocks are kept; 0 disables."""

    def __init__(self):
        import threading
        self.lock = threading.Lock()
        self.free = {}            # nbytes -> [address]
        self.cached = 0
        self.libc = C.CDLL(None)
        self.libc.malloc.restype = C.c_void_p
        self.libc.malloc.argtypes = [C.c_size_t]
        self.libc.free.argtypes = [C.c_void_p]
        self.cap = int(os.environ.get('FPL_HOST_POOL_MB', '4096')) << 20

    def empty(self, shape, dtype):
        import weakref
        dtype = np.dtype(dtype)
        shape = tuple(int(d) for d in shape)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        if self.cap <= 0 or nbytes < (8 << 20):
            return np.empty(shape, dtype)
        with self.lock:
            lst = self.free.get(nbytes)
            addr = lst.pop() if lst else None
            if addr is not None:
                self.cached -= nbytes
        if addr is None:
            addr = self.libc.malloc(nbytes)
            if not addr:
                raise MemoryError('host pool: %d bytes' % nbytes)
        buf = (C.c_char * nbytes).from_address(addr)
        weakref.finalize(buf, self._release, addr, nbytes)
        return np.ndarray(shape, dtype, buffer=buf)

    def _release(self, addr, nbytes):
        with self.lock:
            if self.cached + nbytes <= self.cap:
                self.free.setdefault(nbytes, []).append(addr)
                self.cached += nbytes
                return
        self.libc.free(addr)


_host_pool = None


def host_empty(shape, dtype=np.float32):
    """an uninitialised host array from the recycling pool (see _HostPool)"""
    global _host_pool
    if _host_pool is None:
        _host_pool = _HostPool()
    return _host_pool.empty(shape, dtype)


class Context:
    """one GPU (fpl_ctx).  Create after fork(); use from one thread at a time."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = _vp()
        rc = self.lib.fpl_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise FplHipError(self.lib.fpl_last_error(None).decode())
        self.h = h
        self.device = int(device)

    def check(self, rc):
        if rc != 0:
            raise FplHipError(self.lib.fpl_last_error(self.h).decode())

    def close(self):
        if getattr(self, 'h', None):
            self.lib.fpl_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        self.check(self.lib.fpl_ctx_set_stream(self.h, _vp(hip_stream or 0)))

    def synchronize(self):
        self.check(self.lib.fpl_ctx_synchronize(self.h))

    def device_info(self):
        ncu, hbm = C.c_int32(), C.c_int64()
        name = C.create_string_buffer(256)
        self.check(self.lib.fpl_device_info(self.h, C.byref(ncu), C.byref(hbm),
                                            name, 256))
        return dict(n_cu=ncu.value, hbm_bytes=hbm.value,
                    name=name.value.decode())

    def device_uuid(self):
        """PCI bus id of this context's GPU"""
        buf = C.create_string_buffer(64)
        self.check(self.lib.fpl_device_pci_bus_id(self.h, buf, 64))
        return buf.value.decode()

    def last_path(self):
        """executor the last infer_volume / forward of this context ran on"""
        return self.lib.fpl_last_path(self.h).decode()

    # ---- RCCL communicator of this GPU (data-parallel training)
    def comm_init(self, rank, nranks, unique_id):
        """collective: returns when all `nranks` ranks have joined"""
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError('unique id must be %d bytes' % COMM_ID_BYTES)
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self.check(self.lib.fpl_comm_init(self.h, int(rank), int(nranks), buf))

    def comm_destroy(self):
        self.check(self.lib.fpl_comm_destroy(self.h))

    def comm_abort(self):
        """drop the communicator without waiting for pending collectives (a peer died)"""
        self.check(self.lib.fpl_comm_abort(self.h))

    def comm_info(self):
        rank, n = C.c_int32(), C.c_int32()
        path = C.create_string_buffer(256)
        self.check(self.lib.fpl_comm_info(self.h, C.byref(rank), C.byref(n), path, 256))
        return dict(rank=rank.value, nranks=n.value, lib=path.value.decode())

    def comm_allreduce_sum_f32(self, dev, n):
        self.check(self.lib.fpl_comm_allreduce_sum_f32(self.h, _ptr(dev), int(n)))

    def comm_broadcast_f32(self, dev, n, root=0):
        self.check(self.lib.fpl_comm_broadcast_f32(self.h, _ptr(dev), int(n), int(root)))

    # ---- timing
    def timing(self, on=True):
        self.check(self.lib.fpl_timing_enable(self.h, int(bool(on))))

    def timing_reset(self):
        self.check(self.lib.fpl_timing_reset(self.h))

    def timing_get(self, cap=256):
        names = C.create_string_buffer(64 * cap)
        ms = (C.c_double * cap)()
        cnt = (C.c_int64 * cap)()
        n = C.c_int32()
        self.check(self.lib.fpl_timing_get(self.h, names, ms, cnt, cap,
                                           C.byref(n)))
        out = {}
        for i in range(n.value):
            nm = names.raw[64 * i:64 * (i + 1)].split(b'\0')[0].decode()
            out[nm] = dict(ms=ms[i], launches=cnt[i])
        return out

    # ---- synthetic data
    def synth_volume_u8(self, seed, dims, origin=(0, 0, 0), out=None):
        if out is None:
            out = np.empty(tuple(int(d) for d in dims), np.uint8)
        self.check(self.lib.fpl_synth_volume_u8(
            self.h, C.c_uint64(int(seed)), _arr(dims, C.c_int64),
            _arr(origin, C.c_int64), _ptr(out), _mem_of(out)))
        return out

    # ---- raw device memory (the pipeline keeps substacks / predictions resident)
    def malloc(self, shape, dtype):
        return DeviceBuffer(self, shape, dtype)

    def memcpy(self, dst, src, nbytes):
        self.check(self.lib.fpl_memcpy(self.h, _ptr(dst), _mem_of(dst), _ptr(src),
                                       _mem_of(src), int(nbytes)))

    def synth_substack_u8(self, seed, extent, dims, origin, out):
        self.check(self.lib.fpl_synth_substack_u8(
            self.h, C.c_uint64(int(seed)), _arr(extent, C.c_int64), _arr(dims, C.c_int64),
            _arr(origin, C.c_int64), _ptr(out), _mem_of(out)))
        return out

    def crop_substack_u8(self, src, extent, dims, origin, out):
        """`out` := the (dims) box at `origin` of the device-resident uint8 volume `src`
        (extents `extent`), zeros outside it; stream-ordered"""
        if _mem_of(src) != MEM_DEVICE or _mem_of(out) != MEM_DEVICE:
            raise ValueError('crop_substack_u8: source and destination must be device memory')
        self.check(self.lib.fpl_crop_substack_u8(
            self.h, _ptr(src), _arr(extent, C.c_int64), _arr(dims, C.c_int64),
            _arr(origin, C.c_int64), _ptr(out)))
        return out

    def histogram_u8(self, src, n=None):
        if isinstance(src, np.ndarray):
            src = np.ascontiguousarray(src, np.uint8)
            n = src.size
        elif n is None:
            n = int(np.prod(src.shape))
        out = (C.c_uint64 * 256)()
        self.check(self.lib.fpl_histogram_u8(self.h, _ptr(src), _mem_of(src), int(n), out))
        return np.array(out[:], np.int64)

    # ---- voxel2obj stages
    def v2o_smooth(self, pred, dims, r, weights, ranks):
        weights = np.ascontiguousarray(weights, np.float64)
        wr = (weights.size - 1) // 2
        ranks = np.ascontiguousarray(ranks, np.int64)
        vals = np.zeros(max(ranks.size, 1), np.float32)
        self.check(self.lib.fpl_v2o_smooth(
            self.h, _ptr(pred), _mem_of(pred), _arr(dims, C.c_int64), int(r),
            _ptr(weights), int(wr), ranks.ctypes.data_as(_pi64),
            int(ranks.size), _ptr(vals)))
        return vals[:ranks.size]

    def v2o_set_floor(self, floor):
        """the next NMS threshold will be >= floor (valid for one v2o_smooth)"""
        self.check(self.lib.fpl_v2o_set_floor(self.h, float(floor)))

    def v2o_nms(self, thresh, cap=1 << 20):
        out = np.zeros((cap, 4), np.float64)
        n, rounds = C.c_int64(), C.c_int32()
        self.check(self.lib.fpl_v2o_nms(self.h, float(thresh), _ptr(out),
                                        int(cap), C.byref(n), C.byref(rounds)))
        return out[:n.value].copy(), rounds.value

    def v2o_set_seg(self, seg, dims, sz_thd=None):
        if isinstance(seg, np.ndarray):
            if seg.dtype.itemsize not in (4, 8) or seg.dtype.kind not in 'iu':
                seg = seg.astype(np.uint64)
            seg = np.ascontiguousarray(seg)
            nbytes = seg.dtype.itemsize
        else:
            nbytes = np.dtype(getattr(seg, 'dtype', np.uint64)).itemsize
        self.check(self.lib.fpl_v2o_set_seg(self.h, _ptr(seg), nbytes, _mem_of(seg),
                                            _arr(dims, C.c_int64),
                                            -1 if sz_thd is None else int(sz_thd)))

    def v2o_select(self, ranks):
        ranks = np.ascontiguousarray(ranks, np.int64)
        vals = np.zeros(max(ranks.size, 1), np.float32)
        self.check(self.lib.fpl_v2o_select(self.h, ranks.ctypes.data_as(_pi64),
                                           int(ranks.size), _ptr(vals)))
        return vals[:ranks.size]

    # float64 predictions (include/fplhip.h: fpl_v2o_smooth_f64 ...)
    def v2o_smooth_f64(self, pred, dims, r, weights):
        w = np.ascontiguousarray(weights, np.float64)
        self.check(self.lib.fpl_v2o_smooth_f64(self.h, _ptr(pred), _mem_of(pred),
                                               _arr(dims, C.c_int64), int(r), _ptr(w),
                                               int((w.size - 1) // 2)))

    def v2o_set_integer(self, on=True):
        """the next v2o_smooth_f64 filters an INTEGER volume (truncation after every pass)"""
        self.check(self.lib.fpl_v2o_set_integer(self.h, int(bool(on))))

    def v2o_select_f64(self, ranks):
        ranks = np.ascontiguousarray(ranks, np.int64)
        vals = np.zeros(max(ranks.size, 1), np.float64)
        self.check(self.lib.fpl_v2o_select_f64(self.h, ranks.ctypes.data_as(_pi64),
                                               int(ranks.size), _ptr(vals)))
        return vals[:ranks.size]

    def v2o_rank_f64(self, thresh):
        n = C.c_int64()
        self.check(self.lib.fpl_v2o_rank_f64(self.h, float(thresh), C.byref(n)))
        return n.value

    def v2o_values_f64(self, flat):
        flat = np.ascontiguousarray(flat, np.int64)
        out = np.zeros(max(flat.size, 1), np.float64)
        self.check(self.lib.fpl_v2o_values_f64(self.h, flat.ctypes.data_as(_pi64), int(flat.size),
                                               _ptr(out)))
        return out[:flat.size]

    def v2o_nms_seg(self, thresh, seg_dilate, seg_force, cap=1 << 20):
        out = np.zeros((cap, 4), np.float64)
        n, rounds = C.c_int64(), C.c_int32()
        self.check(self.lib.fpl_v2o_nms_seg(self.h, float(thresh), int(seg_dilate or 0),
                                            int(seg_force or 0), _ptr(out), int(cap),
                                            C.byref(n), C.byref(rounds)))
        return out[:n.value].copy(), rounds.value

    def v2o_smoothed(self, pdims):
        out = np.empty(tuple(int(d) for d in pdims), np.float32)
        self.check(self.lib.fpl_v2o_copy_smoothed(self.h, _ptr(out), MEM_HOST))
        return out


class DeviceBuffer:
    """`shape` x `dtype` in the HBM of one context (fpl_malloc / fpl_free); accepted
    wherever the bindings take a device array"""
    is_cuda = True

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.shape = tuple(int(d) for d in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = _vp()
        ctx.check(ctx.lib.fpl_malloc(ctx.h, max(self.nbytes, 16), C.byref(p)))
        self.ptr = p.value

    def data_ptr(self):
        return self.ptr

    def view(self, shape, dtype=None):
        """the leading bytes of this buffer under another shape (no copy)"""
        v = object.__new__(DeviceBuffer)
        v.ctx, v.ptr = self.ctx, self.ptr
        v.shape = tuple(int(d) for d in shape)
        v.dtype = np.dtype(dtype or self.dtype)
        v.nbytes = int(np.prod(v.shape)) * v.dtype.itemsize
        assert v.nbytes <= self.nbytes
        v._base = self
        return v

    def to_host(self):
        out = np.empty(self.shape, self.dtype)
        self.ctx.memcpy(out, self, self.nbytes)
        return out

    def from_host(self, a):
        a = np.ascontiguousarray(a, self.dtype)
        assert a.nbytes == self.nbytes
        self.ctx.memcpy(self, a, self.nbytes)
        return self

    def free(self):
        if getattr(self, 'ptr', None) and not hasattr(self, '_base') \
                and getattr(self.ctx, 'h', None):
            self.ctx.lib.fpl_free(self.ctx.h, _vp(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Program:
    """lowered layer program resident on one GPU (fpl_program)"""

    def __init__(self, ctx, graph, stride=(1, 1, 1)):
        self.ctx = ctx
        ops, arena, out_tensor, n_tensors = graph.lower_inference()
        self.n_arena = arena.size
        c_ops = (fpl_op * len(ops))()
        for i, o in enumerate(ops):
            c_ops[i] = fpl_op(o['kind'], o['src0'], o['src1'], o['dst'], o['k'],
                              o['cin'], o['cout'], o['act'], o['w_off'],
                              o['scale_off'], o['shift_off'],
                              (C.c_int32 * 6)(*[int(v) for v in o['p']]))
        h = _vp()
        arena = np.ascontiguousarray(arena, np.float32)
        ctx.check(ctx.lib.fpl_program_create(
            ctx.h, c_ops, len(ops), n_tensors, out_tensor, _ptr(arena),
            arena.size, _arr(stride, C.c_int32), C.byref(h)))
        self.h = h
        self.stride = tuple(int(s) for s in stride)

    def set_weights_from(self, graph):
        _, arena, _, _ = graph.lower_inference()
        arena = np.ascontiguousarray(arena, np.float32)
        self.ctx.check(self.ctx.lib.fpl_program_set_arena(
            self.h, _ptr(arena), arena.size))

    def close(self):
        if getattr(self, 'h', None) and getattr(self.ctx, 'h', None):
            self.ctx.lib.fpl_program_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def out_dims(self, in_dims):
        od = (C.c_int32 * 4)()
        dummy = np.zeros(1, np.float32)
        self.ctx.check(self.ctx.lib.fpl_program_forward(
            self.ctx.h, self.h, _ptr(dummy), MEM_HOST, 1,
            _arr(in_dims, C.c_int32), PREC_F32, None, MEM_HOST, od))
        return tuple(od)

    def forward(self, batch, precision=PREC_F32):
        """batch (n, D, H, W[, 1]) float32 host array -> (n, d, h, w, c)"""
        x = np.ascontiguousarray(batch, np.float32)
        if x.ndim == 5:
            x = x[..., 0]
        n, in_dims = x.shape[0], x.shape[1:]
        od = self.out_dims(in_dims)
        out = np.empty((n,) + od, np.float32)
        odc = (C.c_int32 * 4)()
        self.ctx.check(self.ctx.lib.fpl_program_forward(
            self.ctx.h, self.h, _ptr(x), MEM_HOST, n, _arr(in_dims, C.c_int32),
            precision, _ptr(out), MEM_HOST, odc))
        return out

    def infer_volume(self, src, tile_in, offset, mean=0.0, std=1.0,
                     precision=PREC_F32, z_range=(0, -1), dst=None, dims=None):
        """src: numpy (Z,Y,X) uint8/float32, or a device tensor/pointer (then
        `dims` and dtype via `src.dtype`/uint8 are required)"""
        if isinstance(src, np.ndarray):
            if src.dtype != np.uint8:
                src = np.ascontiguousarray(src, np.float32)
            src = np.ascontiguousarray(src)
            dims = src.shape
            dt = U8 if src.dtype == np.uint8 else F32
        else:
            dt = U8 if 'uint8' in str(getattr(src, 'dtype', 'uint8')) else F32
            if dims is None:
                dims = tuple(src.shape)
        if dst is None:
            dst = host_empty(dims, np.float32)
        self.ctx.check(self.ctx.lib.fpl_infer_volume(
            self.ctx.h, self.h, _ptr(src), dt, _mem_of(src), float(mean),
            float(std), _arr(dims, C.c_int64), _arr(tile_in, C.c_int32),
            _arr(offset, C.c_int32), precision, int(z_range[0]),
            int(z_range[1]), _ptr(dst), _mem_of(dst)))
        return dst


# include/fplhip.h fpl_loss; names are the reference's (fplnetwork.py:74-77,
# fplmodels.py:28-50)
LOSS_KINDS = {'binary_crossentropy': 0, 'masked_binary_crossentropy': 1,
              'masked_weighted_binary_crossentropy': 2, 'masked_focal_loss': 3}


def metrics_from_sums(s):
    """the reference's metrics (fplmodels.py:52-65 + Keras 'accuracy') of one batch
    from fpl_trainer_metric_sums"""
    n = max(s[7], 1.0)
    return {'loss': s[0] / n, 'acc': s[1] / n, 'masked_accuracy': s[2] / n,
            'lb0l1err': s[3] / max(s[4], 1.0), 'lb1l1err': s[5] / max(s[6], 1.0)}


class Trainer:
    """training engine for one LayerGraph on one GPU (fpl_trainer)"""

    def __init__(self, ctx, graph, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
                 loss='binary_crossentropy'):
        from .program import lower_training
        self.ctx = ctx
        self.graph = graph
        layers, arena, offsets, out_tensor, n_tensors = lower_training(graph)
        self.offsets = offsets
        self.n_w = int(arena.size)
        c_layers = (fpl_layer * len(layers))()
        for i, d in enumerate(layers):
            c_layers[i] = fpl_layer(d['kind'], d['src0'], d['src1'], d['dst'],
                                    d['k'], d['cin'], d['cout'], d['use_bias'],
                                    d['act'], d['rate'],
                                    (C.c_int32 * 6)(*[int(v) for v in d['p']]),
                                    (C.c_int64 * 4)(*[int(v) for v in d['w_off']]))
        h = _vp()
        arena = np.ascontiguousarray(arena, np.float32)
        ctx.check(ctx.lib.fpl_trainer_create(
            ctx.h, c_layers, len(layers), n_tensors, out_tensor, _ptr(arena),
            arena.size, lr, beta1, beta2, eps, C.byref(h)))
        self.h = h
        loss = getattr(loss, '__name__', loss)
        if loss not in LOSS_KINDS:
            raise NotImplementedError('loss %r (have %s)' % (loss, sorted(LOSS_KINDS)))
        ctx.check(ctx.lib.fpl_trainer_set_loss(h, LOSS_KINDS[loss]))

    def close(self):
        if getattr(self, 'h', None) and getattr(self.ctx, 'h', None):
            self.ctx.lib.fpl_trainer_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, data, labels, seed=0):
        """forward + loss + backward; data (B,D,H,W[,1]) f32, labels
        (B,d,h,w[,1]) u8.  Returns (loss, accuracy); gradients stay on device."""
        if isinstance(data, np.ndarray) or not hasattr(data, 'data_ptr'):
            x = np.ascontiguousarray(data, np.float32)
            y = np.ascontiguousarray(labels, np.uint8)
        else:
            # batch already resident on this trainer's GPU (train._DeviceStager: uploaded on
            # a side stream while the previous step ran); contiguous f32 / u8 tensors
            x, y = data, labels
            if 'float32' not in str(x.dtype) or 'uint8' not in str(y.dtype) or \
                    not x.is_contiguous() or not y.is_contiguous():
                raise TypeError('device batches must be contiguous float32 data / uint8 labels')
            for t in (x, y):        # a raw pointer of another GPU would be a memory fault, not an error
                if not getattr(t, 'is_cuda', False) or t.device.index != self.ctx.device:
                    raise TypeError('device batches must live on this trainer\'s GPU (cuda:%d), got %s'
                                    % (self.ctx.device, getattr(t, 'device', type(t))))
        shape = tuple(int(v) for v in x.shape)
        if len(shape) == 5:
            shape = shape[:4]                     # trailing channel axis of 1
        loss, acc = C.c_float(), C.c_float()
        self.ctx.check(self.ctx.lib.fpl_trainer_step(
            self.h, _ptr(x), _mem_of(x), _ptr(y), _mem_of(y), shape[0],
            _arr(shape[1:], C.c_int32), C.c_uint64(int(seed)),
            C.byref(loss), C.byref(acc)))
        return loss.value, acc.value

    def metrics(self):
        """metrics of the last step by the reference's names"""
        sums = (C.c_double * 8)()
        self.ctx.check(self.ctx.lib.fpl_trainer_metric_sums(self.h, sums))
        return metrics_from_sums(list(sums))

    def apply(self, grad_scale=1.0):
        self.ctx.check(self.ctx.lib.fpl_trainer_apply(self.h, float(grad_scale)))

    def grad_ptr(self):
        p, n = _vp(), C.c_int64()
        self.ctx.check(self.ctx.lib.fpl_trainer_grad_ptr(self.h, C.byref(p),
                                                         C.byref(n)))
        return p.value, n.value

    def _split(self, flat):
        out = []
        for w, o in zip(self.graph.weights, self.offsets):
            out.append(flat[o:o + w.size].reshape(w.shape).copy())
        return out

    def get_weights(self):
        flat = np.empty(self.n_w, np.float32)
        self.ctx.check(self.ctx.lib.fpl_trainer_get_weights(self.h, _ptr(flat),
                                                            self.n_w))
        return self._split(flat)

    def get_grads(self):
        flat = np.empty(self.n_w, np.float32)
        self.ctx.check(self.ctx.lib.fpl_trainer_get_grads(self.h, _ptr(flat),
                                                          self.n_w))
        return self._split(flat)

    def set_weights(self, weights):
        flat = np.concatenate([np.asarray(w, np.float32).reshape(-1)
                               for w in weights])
        self.ctx.check(self.ctx.lib.fpl_trainer_set_weights(self.h, _ptr(flat),
                                                            flat.size))

    def get_opt_state(self):
        """Adam state as (m, v, steps): lists in get_weights() order + the update count"""
        m, v = np.empty(self.n_w, np.float32), np.empty(self.n_w, np.float32)
        steps = C.c_int64()
        self.ctx.check(self.ctx.lib.fpl_trainer_get_opt_state(self.h, _ptr(m), _ptr(v), self.n_w,
                                                              C.byref(steps)))
        return self._split(m), self._split(v), int(steps.value)

    def set_opt_state(self, m, v, steps):
        fm = np.concatenate([np.asarray(w, np.float32).reshape(-1) for w in m])
        fv = np.concatenate([np.asarray(w, np.float32).reshape(-1) for w in v])
        if fm.size != self.n_w or fv.size != self.n_w:
            raise ValueError('optimizer state of %d / %d values for %d weights' % (fm.size, fv.size, self.n_w))
        self.ctx.check(self.ctx.lib.fpl_trainer_set_opt_state(self.h, _ptr(fm), _ptr(fv), self.n_w,
                                                              int(steps)))

    # ---- data parallelism
    def get_grads_flat(self):
        flat = np.empty(self.n_w, np.float32)
        self.ctx.check(self.ctx.lib.fpl_trainer_get_grads(self.h, _ptr(flat), self.n_w))
        return flat

    def set_grads_flat(self, flat):
        flat = np.ascontiguousarray(flat, np.float32).reshape(-1)
        self.ctx.check(self.ctx.lib.fpl_trainer_set_grads(self.h, _ptr(flat), flat.size))

    def allreduce_grads(self):
        """ONE RCCL all-reduce (sum) of the gradient arena over the context's
        communicator, stream-ordered before `apply`"""
        self.ctx.check(self.ctx.lib.fpl_allreduce_grads(self.h))

    def broadcast_state(self, root=0):
        self.ctx.check(self.ctx.lib.fpl_trainer_broadcast_state(self.h, int(root)))
