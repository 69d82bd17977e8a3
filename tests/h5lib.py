"""libhdf5 through ctypes - TEST INFRASTRUCTURE only (h5py is not installed; the C library
is, under /opt/conda/lib).  Two uses:

  * `read_tree(path)`: open a file the package's pure-Python writer (flypylib_amd/h5min.py)
    produced with the real library - what h5py / Keras would do - and return its groups,
    attributes and datasets;
  * `write_tree(path, tree)`: write a file WITH the real library (variable-length string
    attributes as h5py >= 3 stores Python `str`, chunked datasets, ...) for
    tests/golden/make_h5_fixture.py, so that the package's reader is pinned by bytes it
    did not write.

`available()` is False when no libhdf5 can be loaded; callers skip.
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

_CANDIDATES = ['/opt/conda/lib/libhdf5.so.103', '/opt/conda/lib/libhdf5.so',
               ctypes.util.find_library('hdf5') or '']
_lib = None

hid_t = C.c_int64
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT = 0
H5S_SCALAR, H5S_SIMPLE = 0, 1
H5T_VARIABLE = C.c_size_t(-1).value
H5T_CSET_UTF8 = 1
H5T_STRING, H5T_VLEN = 3, 9
H5O_TYPE_GROUP, H5O_TYPE_DATASET = 0, 1
H5D_CHUNKED = 2


def _load():
    global _lib
    if _lib is not None:
        return _lib
    for p in _CANDIDATES:
        if p and (os.path.exists(p) or '/' not in p):
            try:
                lib = C.CDLL(p)
            except OSError:
                continue
            lib.H5open()
            lib.H5Eset_auto2(hid_t(0), None, None)          # no error stack on stderr
            for name, res, args in [
                ('H5Fopen', hid_t, [C.c_char_p, C.c_uint, hid_t]),
                ('H5Fcreate', hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
                ('H5Fclose', C.c_int, [hid_t]),
                ('H5Gopen2', hid_t, [hid_t, C.c_char_p, hid_t]),
                ('H5Gcreate2', hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
                ('H5Gclose', C.c_int, [hid_t]),
                ('H5Oopen', hid_t, [hid_t, C.c_char_p, hid_t]),
                ('H5Oclose', C.c_int, [hid_t]),
                ('H5Iget_type', C.c_int, [hid_t]),
                ('H5Dopen2', hid_t, [hid_t, C.c_char_p, hid_t]),
                ('H5Dcreate2', hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
                ('H5Dget_space', hid_t, [hid_t]),
                ('H5Dget_type', hid_t, [hid_t]),
                ('H5Dread', C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
                ('H5Dwrite', C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
                ('H5Dclose', C.c_int, [hid_t]),
                ('H5Screate', hid_t, [C.c_int]),
                ('H5Screate_simple', hid_t, [C.c_int, C.POINTER(C.c_uint64), C.c_void_p]),
                ('H5Sget_simple_extent_ndims', C.c_int, [hid_t]),
                ('H5Sget_simple_extent_dims', C.c_int, [hid_t, C.POINTER(C.c_uint64), C.c_void_p]),
                ('H5Sget_simple_extent_type', C.c_int, [hid_t]),
                ('H5Sclose', C.c_int, [hid_t]),
                ('H5Tcopy', hid_t, [hid_t]),
                ('H5Tset_size', C.c_int, [hid_t, C.c_size_t]),
                ('H5Tset_cset', C.c_int, [hid_t, C.c_int]),
                ('H5Tget_size', C.c_size_t, [hid_t]),
                ('H5Tget_class', C.c_int, [hid_t]),
                ('H5Tis_variable_str', C.c_int, [hid_t]),
                ('H5Tget_sign', C.c_int, [hid_t]),
                ('H5Tclose', C.c_int, [hid_t]),
                ('H5Acreate2', hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]),
                ('H5Awrite', C.c_int, [hid_t, hid_t, C.c_void_p]),
                ('H5Aread', C.c_int, [hid_t, hid_t, C.c_void_p]),
                ('H5Aget_space', hid_t, [hid_t]),
                ('H5Aget_type', hid_t, [hid_t]),
                ('H5Aget_name', C.c_ssize_t, [hid_t, C.c_size_t, C.c_char_p]),
                ('H5Aopen_by_idx', hid_t, [hid_t, C.c_char_p, C.c_int, C.c_int, C.c_uint64, hid_t, hid_t]),
                ('H5Aclose', C.c_int, [hid_t]),
                ('H5Pcreate', hid_t, [hid_t]),
                ('H5Pset_chunk', C.c_int, [hid_t, C.c_int, C.POINTER(C.c_uint64)]),
                ('H5Pclose', C.c_int, [hid_t]),
                ('H5Lget_name_by_idx', C.c_ssize_t, [hid_t, C.c_char_p, C.c_int, C.c_int, C.c_uint64,
                                                     C.c_char_p, C.c_size_t, hid_t]),
                ('H5Gget_num_objs', C.c_int, [hid_t, C.POINTER(C.c_uint64)]),
            ]:
                f = getattr(lib, name)
                f.restype, f.argtypes = res, args
            _lib = lib
            return lib
    return None


def available():
    return _load() is not None


def _g(name):
    """a library global type id (H5T_NATIVE_FLOAT and friends are macros over these)"""
    return hid_t.in_dll(_lib, name).value


def _np_type(lib, tid):
    cls, size = lib.H5Tget_class(tid), lib.H5Tget_size(tid)
    if cls == 0:
        return np.dtype('%s%d' % ('i' if lib.H5Tget_sign(tid) else 'u', size))
    if cls == 1:
        return np.dtype('f%d' % size)
    if cls == H5T_STRING:
        return 'vstr' if lib.H5Tis_variable_str(tid) > 0 else np.dtype('S%d' % size)
    raise NotImplementedError('HDF5 type class %d' % cls)


def _mem_type(lib, dt):
    """(memory type id, needs H5Tclose)"""
    if dt == 'vstr':
        t = lib.H5Tcopy(_g('H5T_C_S1_g'))
        lib.H5Tset_size(t, H5T_VARIABLE)
        lib.H5Tset_cset(t, H5T_CSET_UTF8)
        return t, True
    if dt.kind == 'S':
        t = lib.H5Tcopy(_g('H5T_C_S1_g'))
        lib.H5Tset_size(t, max(dt.itemsize, 1))
        return t, True
    names = {'f4': 'H5T_NATIVE_FLOAT_g', 'f8': 'H5T_NATIVE_DOUBLE_g', 'u1': 'H5T_NATIVE_UINT8_g',
             'i1': 'H5T_NATIVE_INT8_g', 'i4': 'H5T_NATIVE_INT32_g', 'u4': 'H5T_NATIVE_UINT32_g',
             'i8': 'H5T_NATIVE_INT64_g', 'u8': 'H5T_NATIVE_UINT64_g', 'f2': None,
             'i2': 'H5T_NATIVE_INT16_g', 'u2': 'H5T_NATIVE_UINT16_g'}
    n = names.get(dt.str[1:])
    if n is None:
        raise NotImplementedError('dtype %s' % dt)
    return _g(n), False


def _shape(lib, sid):
    if lib.H5Sget_simple_extent_type(sid) == H5S_SCALAR:
        return ()
    nd = lib.H5Sget_simple_extent_ndims(sid)
    dims = (C.c_uint64 * max(nd, 1))()
    lib.H5Sget_simple_extent_dims(sid, dims, None)
    return tuple(int(d) for d in dims[:nd])


def _read(lib, reader, obj, tid, sid):
    dt, shape = _np_type(lib, tid), _shape(lib, sid)
    n = int(np.prod(shape)) if shape else 1
    if dt != 'vstr' and dt.kind == 'S':
        mt, own = lib.H5Tcopy(tid), True           # the file's own padding convention
    else:
        mt, own = _mem_type(lib, dt)
    try:
        if dt == 'vstr':
            buf = (C.c_char_p * n)()
            assert reader(obj, mt, buf) >= 0
            vals = [buf[i].decode('utf8') for i in range(n)]     # (the C strings leak: a test helper)
            return vals[0] if shape == () else np.array(vals, dtype=object).reshape(shape)
        out = np.empty(shape, dt)
        assert reader(obj, mt, out.ctypes.data_as(C.c_void_p)) >= 0
        return out[()] if shape == () else out
    finally:
        if own:
            lib.H5Tclose(mt)


def _attrs(lib, obj):
    out, i = {}, 0
    while True:
        a = lib.H5Aopen_by_idx(obj, b'.', 0, 0, i, H5P_DEFAULT, H5P_DEFAULT)
        if a < 0:
            return out
        name = C.create_string_buffer(256)
        lib.H5Aget_name(a, 256, name)
        tid, sid = lib.H5Aget_type(a), lib.H5Aget_space(a)
        out[name.value.decode()] = _read(lib, lambda o, mt, b: lib.H5Aread(o, mt, b), a, tid, sid)
        lib.H5Tclose(tid); lib.H5Sclose(sid); lib.H5Aclose(a)
        i += 1


def _read_group(lib, gid):
    tree = {'attrs': _attrs(lib, gid), 'groups': {}, 'datasets': {}, 'dataset_attrs': {}}
    n = C.c_uint64()
    assert lib.H5Gget_num_objs(gid, C.byref(n)) >= 0
    for i in range(n.value):
        name = C.create_string_buffer(1024)
        lib.H5Lget_name_by_idx(gid, b'.', 0, 0, i, name, 1024, H5P_DEFAULT)
        o = lib.H5Oopen(gid, name.value, H5P_DEFAULT)
        assert o >= 0, name.value
        kind = lib.H5Iget_type(o)                 # H5I_GROUP = 2, H5I_DATASET = 5
        key = name.value.decode()
        if kind == 2:
            tree['groups'][key] = _read_group(lib, o)
        elif kind == 5:
            tid, sid = lib.H5Dget_type(o), lib.H5Dget_space(o)
            tree['datasets'][key] = _read(
                lib, lambda d, mt, b: lib.H5Dread(d, mt, hid_t(0), hid_t(0), H5P_DEFAULT, b), o, tid, sid)
            tree['dataset_attrs'][key] = _attrs(lib, o)
            lib.H5Tclose(tid); lib.H5Sclose(sid)
        lib.H5Oclose(o)
    return tree


def read_tree(path):
    lib = _load()
    f = lib.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT)
    if f < 0:
        raise IOError('libhdf5 cannot open %s' % path)
    try:
        g = lib.H5Gopen2(f, b'/', H5P_DEFAULT)
        assert g >= 0, 'libhdf5 cannot open the root group'
        try:
            return _read_group(lib, g)
        finally:
            lib.H5Gclose(g)
    finally:
        lib.H5Fclose(f)


# ---- writing ----------------------------------------------------------------------------
def _space(lib, shape):
    if shape == ():
        return lib.H5Screate(H5S_SCALAR)
    dims = (C.c_uint64 * len(shape))(*shape)
    return lib.H5Screate_simple(len(shape), dims, None)


def _write_attr(lib, obj, name, value):
    if isinstance(value, str):                     # h5py >= 3: a variable-length UTF-8 string
        t, _ = _mem_type(lib, 'vstr')
        s = lib.H5Screate(H5S_SCALAR)
        a = lib.H5Acreate2(obj, name.encode(), t, s, H5P_DEFAULT, H5P_DEFAULT)
        buf = (C.c_char_p * 1)(value.encode('utf8'))
        assert lib.H5Awrite(a, t, buf) >= 0
        lib.H5Aclose(a); lib.H5Sclose(s); lib.H5Tclose(t)
        return
    arr = np.asarray(value)
    if arr.dtype.kind == 'O':                      # array of str: variable-length strings
        t, _ = _mem_type(lib, 'vstr')
        s = _space(lib, arr.shape)
        a = lib.H5Acreate2(obj, name.encode(), t, s, H5P_DEFAULT, H5P_DEFAULT)
        flat = [str(v).encode('utf8') for v in arr.ravel()]
        buf = (C.c_char_p * len(flat))(*flat)
        assert lib.H5Awrite(a, t, buf) >= 0
        lib.H5Aclose(a); lib.H5Sclose(s); lib.H5Tclose(t)
        return
    arr = np.ascontiguousarray(arr) if arr.shape else arr.copy()
    t, own = _mem_type(lib, arr.dtype)
    s = _space(lib, arr.shape)
    a = lib.H5Acreate2(obj, name.encode(), t, s, H5P_DEFAULT, H5P_DEFAULT)
    assert a >= 0, name
    assert lib.H5Awrite(a, t, arr.ctypes.data_as(C.c_void_p)) >= 0
    lib.H5Aclose(a); lib.H5Sclose(s)
    if own:
        lib.H5Tclose(t)


def _write_group(lib, gid, tree):
    for k, v in (tree.get('attrs') or {}).items():
        _write_attr(lib, gid, k, v)
    for k, v in (tree.get('datasets') or {}).items():
        chunks = None
        if isinstance(v, tuple):                   # (array, chunk shape): a chunked dataset
            v, chunks = v
        arr = np.ascontiguousarray(v) if np.ndim(v) else np.asarray(v).copy()
        t, own = _mem_type(lib, arr.dtype)
        s = _space(lib, arr.shape)
        dcpl = H5P_DEFAULT
        if chunks is not None:
            dcpl = lib.H5Pcreate(_g('H5P_CLS_DATASET_CREATE_ID_g'))
            lib.H5Pset_chunk(dcpl, len(chunks), (C.c_uint64 * len(chunks))(*chunks))
        d = lib.H5Dcreate2(gid, k.encode(), t, s, H5P_DEFAULT, dcpl, H5P_DEFAULT)
        assert d >= 0, k
        assert lib.H5Dwrite(d, t, hid_t(0), hid_t(0), H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p)) >= 0
        for ak, av in ((tree.get('dataset_attrs') or {}).get(k) or {}).items():
            _write_attr(lib, d, ak, av)
        lib.H5Dclose(d); lib.H5Sclose(s)
        if own:
            lib.H5Tclose(t)
        if chunks is not None:
            lib.H5Pclose(dcpl)
    for k, sub in (tree.get('groups') or {}).items():
        g = lib.H5Gcreate2(gid, k.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        assert g >= 0, k
        _write_group(lib, g, sub)
        lib.H5Gclose(g)


def write_tree(path, tree):
    """tree: {'attrs': {name: array | str}, 'datasets': {name: array | (array, chunks)},
    'dataset_attrs': {name: {...}}, 'groups': {name: tree}} - `str` values become
    variable-length UTF-8 strings (h5py >= 3), bytes arrays fixed-length ones (h5py 2)"""
    lib = _load()
    f = lib.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
    assert f >= 0, path
    try:
        g = lib.H5Gopen2(f, b'/', H5P_DEFAULT)
        _write_group(lib, g, tree)
        lib.H5Gclose(g)
    finally:
        lib.H5Fclose(f)
