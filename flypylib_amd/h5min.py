"""Minimal pure-Python HDF5 reader / writer for Keras weight files.

The reference keeps its networks as a pickle + a Keras `.h5`
(`flypylib/fplnetwork.py:32-44,81-97`: `model.save(path + '.keras.h5')`, per-epoch
`'%s_%03d.h5'`); h5py is not available to this build, so the subset of HDF5 that
h5py / libhdf5 emit for such files (default `libver='earliest'`) is read here directly:

  superblock version 0 (8-byte offsets and lengths)  .  old-style groups: symbol-table
  message -> version-1 B-tree ('TREE') -> symbol-table nodes ('SNOD') -> local heap
  ('HEAP')  .  version-1 object headers with continuation blocks  .  dataspace (v1, v2),
  datatype (fixed-point, IEEE float, fixed-length string), layout v3 (compact and
  contiguous), attribute messages v1 - v3

plus variable-length strings (how h5py >= 3 stores a Python `str` attribute such as
`keras_version`, `backend`, `model_config`): a (length, global-heap collection, index)
triple resolved through the 'GCOL' heap.  Attributes are parsed LAZILY, one name at a
time: an attribute or dataset the loader never touches cannot make a file unreadable.
What remains outside the subset (chunked or filtered datasets, dense attribute storage,
new-style groups, version-2 object headers, superblock >= 2) raises `H5Unsupported`
naming the feature when - and only when - the object that needs it is accessed; convert
such a file with h5py (`tools/keras_h5_to_npz.py`).  The writer emits the same subset
(one symbol-table node per group, leaf K raised so that it fits) for
`FplNetwork.save_network`.  Pinned both ways by the C library (tests/test_keras_io.py):
libhdf5 opens and reads every file the writer produces, and the reader reads
tests/golden/keras_libhdf5.h5, which libhdf5 wrote (tests/golden/make_h5_fixture.py).

    f = h5min.File(path)            # or File(bytes)
    f.attrs['layer_names']          # numpy arrays (strings as bytes)
    f['model_weights/conv3d_1/conv3d_1/kernel:0'][...]   # numpy array
    f['x'].keys(), 'name' in group

    h5min.write(path, tree)         # tree: {'attrs': {...}, 'groups': {...}, 'datasets': {...}}
"""
import struct

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Unsupported(NotImplementedError):
    pass


VLEN_STR = 'vlen_str'          # what _parse_datatype returns for variable-length strings


# ---- reader ---------------------------------------------------------------------------
class _Buf:
    def __init__(self, data):
        self.d = memoryview(data)

    def u(self, off, n):
        return int.from_bytes(self.d[off:off + n], 'little')

    def bytes(self, off, n):
        return bytes(self.d[off:off + n])


def _pad8(n):
    return (n + 7) & ~7


def _parse_datatype(b, off):
    """-> (numpy dtype, bytes consumed is not needed: sizes come from the enclosing message)"""
    cv = b.u(off, 1)
    cls, ver = cv & 0x0F, cv >> 4
    bits = b.u(off + 1, 3)
    size = b.u(off + 4, 4)
    if ver not in (1, 2, 3):
        raise H5Unsupported('datatype message version %d' % ver)
    order = '>' if bits & 1 else '<'
    if cls == 0:                                   # fixed point
        signed = bool(bits & 0x08)
        return np.dtype('%s%s%d' % (order, 'i' if signed else 'u', size))
    if cls == 1:                                   # IEEE float
        if size not in (2, 4, 8):
            raise H5Unsupported('%d-byte floating-point type' % size)
        return np.dtype('%sf%d' % (order, size))
    if cls == 3:                                   # fixed-length string
        return np.dtype('S%d' % size)
    if cls == 9:
        if bits & 0x0F == 1:                       # variable-length STRING (h5py `str`)
            return VLEN_STR
        raise H5Unsupported('variable-length sequence datatype')
    raise H5Unsupported('datatype class %d' % cls)


def _parse_dataspace(b, off):
    ver = b.u(off, 1)
    rank = b.u(off + 1, 1)
    flags = b.u(off + 2, 1)
    if ver == 1:
        p = off + 8
    elif ver == 2:
        if b.u(off + 3, 1) == 2:                   # null dataspace
            return None
        p = off + 4
    else:
        raise H5Unsupported('dataspace message version %d' % ver)
    return tuple(b.u(p + 8 * i, 8) for i in range(rank))


class _Object:
    """parsed object header: messages by type"""

    def __init__(self, f, addr):
        self.f, self.addr = f, addr
        b = f._b
        self.msgs = []
        ver = b.u(addr, 1)
        if b.bytes(addr, 4) == b'OHDR':
            raise H5Unsupported('version-2 object headers (file written with libver="latest")')
        if ver != 1:
            raise H5Unsupported('object header version %d' % ver)
        nmsg = b.u(addr + 2, 2)
        size = b.u(addr + 8, 4)
        blocks = [(addr + 16, size)]
        while blocks and len(self.msgs) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(self.msgs) < nmsg:
                mtype, msize, mflags = b.u(p, 2), b.u(p + 2, 2), b.u(p + 4, 1)
                body = p + 8
                if mtype == 0x0010:                # continuation
                    blocks.append((b.u(body, 8), b.u(body + 8, 8)))
                if mflags & 0x02:
                    raise H5Unsupported('shared header messages')
                self.msgs.append((mtype, body, msize))
                p = body + msize

    def find(self, mtype):
        return [(o, n) for t, o, n in self.msgs if t == mtype]

    def attrs(self):
        return Attrs(self)


class Attrs:
    """the attributes of one object, parsed one name at a time (`[]`, `in`, `get`, `keys`)"""

    def __init__(self, obj):
        self._o = obj
        self._index = None
        self._cache = {}

    def _scan(self):
        """name -> (datatype offset, dataspace offset, data offset); no value is parsed"""
        if self._index is not None:
            return self._index
        b = self._o.f._b
        idx = {}
        for off, n in self._o.find(0x000C):
            ver = b.u(off, 1)
            name_sz, dt_sz, ds_sz = b.u(off + 2, 2), b.u(off + 4, 2), b.u(off + 6, 2)
            if ver == 1:
                p = off + 8
                name = b.bytes(p, name_sz).split(b'\0')[0].decode('utf8')
                p += _pad8(name_sz)
                dt_off, p = p, p + _pad8(dt_sz)
                ds_off, p = p, p + _pad8(ds_sz)
            elif ver in (2, 3):
                p = off + 8 + (1 if ver == 3 else 0)
                name = b.bytes(p, name_sz).split(b'\0')[0].decode('utf8')
                p += name_sz
                dt_off, p = p, p + dt_sz
                ds_off, p = p, p + ds_sz
            else:
                continue                           # an attribute nobody may ask for
            idx[name] = (dt_off, ds_off, p)
        self._index = idx
        return idx

    def _dense(self):
        """attributes moved out of the object header into a fractal heap (attribute-info
        message with a defined heap address)?"""
        info = self._o.find(0x0015)
        if not info:
            return False
        b, off = self._o.f._b, info[0][0]
        flags = b.u(off + 1, 1)
        return b.u(off + 2 + (2 if flags & 1 else 0), 8) != UNDEF

    def keys(self):
        if self._dense():
            raise H5Unsupported('dense attribute storage (more attributes than fit the object '
                                'header): the attribute list is not readable')
        return sorted(self._scan())

    def __contains__(self, name):
        if name in self._scan():
            return True
        if self._dense():
            raise H5Unsupported('dense attribute storage: cannot tell whether %r exists' % name)
        return False

    def get(self, name, default=None):
        return self[name] if name in self else default

    def __getitem__(self, name):
        if name in self._cache:
            return self._cache[name]
        if name not in self:
            raise KeyError(name)
        f = self._o.f
        b = f._b
        dt_off, ds_off, p = self._scan()[name]
        dtype = _parse_datatype(b, dt_off)
        shape = _parse_dataspace(b, ds_off)
        if shape is None:
            val = None
        elif dtype is VLEN_STR:
            count = int(np.prod(shape)) if shape else 1
            vals = [f._global_heap_object(b.u(p + 16 * i + 4, 8), b.u(p + 16 * i + 12, 4),
                                          b.u(p + 16 * i, 4)) for i in range(count)]
            val = vals[0] if shape == () else np.array(vals, dtype=object).reshape(shape)
        else:
            count = int(np.prod(shape)) if shape else 1
            arr = np.frombuffer(b.bytes(p, count * dtype.itemsize), dtype=dtype).reshape(shape)
            val = arr[()] if shape == () else arr.copy()
        self._cache[name] = val
        return val


class Dataset:
    def __init__(self, f, obj, name):
        self._f, self._o, self.name = f, obj, name
        b = f._b
        (dt_off, _), = obj.find(0x0003)
        (ds_off, _), = obj.find(0x0001)
        self.dtype = _parse_datatype(b, dt_off)
        shape = _parse_dataspace(b, ds_off)
        self.shape = () if shape is None else shape
        self.attrs = obj.attrs()

    def __getitem__(self, key):
        return self.read()[key]

    def read(self):
        b = self._f._b
        (lo, _), = self._o.find(0x0008)
        ver, cls = b.u(lo, 1), b.u(lo + 1, 1)
        if ver != 3:
            raise H5Unsupported('data layout message version %d' % ver)
        count = int(np.prod(self.shape)) if self.shape else 1
        nbytes = count * self.dtype.itemsize
        if cls == 0:                               # compact
            raw = b.bytes(lo + 4, b.u(lo + 2, 2))[:nbytes]
        elif cls == 1:                             # contiguous
            addr = b.u(lo + 2, 8)
            if addr == UNDEF:                      # never written: fill value 0
                return np.zeros(self.shape, self.dtype.newbyteorder('='))
            raw = b.bytes(self._f._base + addr, nbytes)
        else:
            raise H5Unsupported('chunked / filtered dataset %r (Keras writes contiguous '
                                'ones; re-save without compression)' % self.name)
        a = np.frombuffer(raw, dtype=self.dtype).reshape(self.shape)
        return a.astype(self.dtype.newbyteorder('='))


class Group:
    def __init__(self, f, obj, name):
        self._f, self._o, self.name = f, obj, name
        self.attrs = obj.attrs()
        self._links = None

    def _load(self):
        if self._links is not None:
            return
        b, f = self._f._b, self._f
        st = self._o.find(0x0011)
        if not st:
            if self._o.find(0x0002) or self._o.find(0x0006):
                raise H5Unsupported('new-style groups (link messages)')
            self._links = {}
            return
        btree, heap = b.u(st[0][0], 8), b.u(st[0][0] + 8, 8)
        if b.bytes(f._base + heap, 4) != b'HEAP':
            raise H5Unsupported('local heap signature')
        heap_data = f._base + b.u(f._base + heap + 24, 8)
        links = {}

        def walk(node):
            p = f._base + node
            if b.bytes(p, 4) != b'TREE':
                raise H5Unsupported('B-tree node signature')
            level, used = b.u(p + 5, 1), b.u(p + 6, 2)
            q = p + 24                               # key 0
            for i in range(used):
                child = b.u(q + 8, 8)
                if level > 0:
                    walk(child)
                else:
                    s = f._base + child
                    if b.bytes(s, 4) != b'SNOD':
                        raise H5Unsupported('symbol table node signature')
                    for k in range(b.u(s + 6, 2)):
                        e = s + 8 + 40 * k
                        noff, ohdr = b.u(e, 8), b.u(e + 8, 8)
                        nm = heap_data + noff
                        end = nm
                        while b.d[end] != 0:
                            end += 1
                        links[b.bytes(nm, end - nm).decode('utf8')] = ohdr
                q += 16

        walk(btree)
        self._links = links

    def keys(self):
        self._load()
        return sorted(self._links)

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split('/') if p]:
            if not isinstance(node, Group):
                raise KeyError(path)
            node._load()
            if part not in node._links:
                raise KeyError('%s (no %r in %s)' % (path, part, node.name))
            node = node._f._open(node._links[part], node.name.rstrip('/') + '/' + part)
        return node


class File(Group):
    def __init__(self, src):
        if isinstance(src, (bytes, bytearray, memoryview)):
            data = bytes(src)
        else:
            with open(src, 'rb') as fh:
                data = fh.read()
        self._b = _Buf(data)
        b = self._b
        if b.bytes(0, 8) != SIGNATURE:
            raise H5Unsupported('not an HDF5 file (or a user block precedes the superblock)')
        ver = b.u(8, 1)
        if ver not in (0, 1):
            raise H5Unsupported('superblock version %d (file written with libver="latest")' % ver)
        if b.u(13, 1) != 8 or b.u(14, 1) != 8:
            raise H5Unsupported('offset / length size other than 8 bytes')
        p = 24 + (4 if ver == 1 else 0)
        self._base = b.u(p, 8)
        root = p + 32                                  # root symbol table entry
        self._cache = {}
        Group.__init__(self, self, _Object(self, self._base + b.u(root + 8, 8)), '/')

    def _global_heap_object(self, collection, index, length):
        """object `index` of the global heap collection at `collection` (the payload of a
        variable-length value), as bytes of `length`"""
        b = self._b
        p = self._base + collection
        if collection == UNDEF or collection == 0:
            return b''                               # an empty / never-written string
        if b.bytes(p, 4) != b'GCOL':
            raise H5Unsupported('global heap collection signature')
        size = b.u(p + 8, 8)
        q, end = p + 16, p + size
        while q + 16 <= end:
            idx, osz = b.u(q, 2), b.u(q + 8, 8)
            if idx == index:
                return b.bytes(q + 16, min(length, osz))
            if idx == 0:                             # the free-space object ends the list
                break
            q += 16 + _pad8(osz)
        raise H5Unsupported('global heap object %d not found' % index)

    def _open(self, ohdr, name):
        if ohdr not in self._cache:
            obj = _Object(self, self._base + ohdr)
            self._cache[ohdr] = Dataset(self, obj, name) if obj.find(0x0008) else Group(self, obj, name)
        return self._cache[ohdr]


# ---- writer -----------------------------------------------------------------------------
def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind == 'f':
        exp, mant, bias = {2: (5, 10, 15), 4: (8, 23, 127), 8: (11, 52, 1023)}[dt.itemsize]
        bits = 8 * dt.itemsize
        return (struct.pack('<B3BI', 0x11, 0x20, bits - 1, 0, dt.itemsize) +
                struct.pack('<HHBBBBI', 0, bits, mant, exp, 0, mant, bias))
    if dt.kind in 'iu':
        return (struct.pack('<B3BI', 0x10, 0x08 if dt.kind == 'i' else 0, 0, 0, dt.itemsize) +
                struct.pack('<HH', 0, 8 * dt.itemsize))
    if dt.kind == 'S':
        return struct.pack('<B3BI', 0x13, 0x01, 0, 0, dt.itemsize)      # null-padded ASCII
    raise H5Unsupported('cannot write dtype %s' % dt)


def _space_msg(shape):
    return struct.pack('<BBB5x', 1, len(shape), 0) + b''.join(struct.pack('<Q', d) for d in shape)


def _msg(mtype, body):
    body = body + b'\0' * (_pad8(len(body)) - len(body))
    return struct.pack('<HHB3x', mtype, len(body), 0) + body


def _attr_msg(name, value):
    a = np.asarray(value)
    if a.dtype.kind == 'U':
        a = np.char.encode(a, 'utf8')
    if a.dtype.kind == 'O':
        raise H5Unsupported('object arrays as attributes')
    a = (a.astype(a.dtype.newbyteorder('<')) if a.dtype.kind in 'fiu' else a).copy(order='C')
    nm = name.encode('utf8') + b'\0'
    dt, ds = _dtype_msg(a.dtype), _space_msg(a.shape)
    body = struct.pack('<BxHHH', 1, len(nm), len(dt), len(ds))
    for part in (nm, dt, ds):
        body += part + b'\0' * (_pad8(len(part)) - len(part))
    return _msg(0x000C, body + a.tobytes())


class _Writer:
    def __init__(self, leaf_k):
        self.buf = bytearray(96)                   # superblock, patched at the end
        self.leaf_k = leaf_k

    def alloc(self, data, align=8):
        while len(self.buf) % align:
            self.buf.append(0)
        off = len(self.buf)
        self.buf += data
        return off

    def header(self, msgs):
        body = b''.join(msgs)
        if len(body) > 0xFFFF0:
            raise H5Unsupported('object header too large')
        return self.alloc(struct.pack('<BxHII4x', 1, len(msgs), 1, len(body)) + body)

    def dataset(self, arr):
        a = np.asarray(arr)
        if a.dtype.kind in 'fiu':
            a = a.astype(a.dtype.newbyteorder('<'))
        a = a.copy(order='C')                      # (ascontiguousarray would make 0-d 1-d)
        data = self.alloc(a.tobytes() or b'\0')
        layout = struct.pack('<BBQQ', 3, 1, data, a.nbytes)
        return self.header([_msg(0x0001, _space_msg(a.shape)), _msg(0x0003, _dtype_msg(a.dtype)),
                            _msg(0x0008, layout)])

    def group(self, tree):
        """-> (object header address, B-tree address, heap address)"""
        children = {}
        for name, sub in (tree.get('groups') or {}).items():
            children[name] = self.group(sub)[0]
        for name, arr in (tree.get('datasets') or {}).items():
            children[name] = self.dataset(arr)
        names = sorted(children, key=lambda s: s.encode('utf8'))
        # local heap: offset 0 = "" (key 0 of the B-tree), then the names, 8-byte aligned
        heap = bytearray(b'\0' * 8)
        offs = {}
        for n in names:
            offs[n] = len(heap)
            raw = n.encode('utf8') + b'\0'
            heap += raw + b'\0' * (_pad8(len(raw)) - len(raw))
        heap_data = self.alloc(bytes(heap))
        # 'offset to head of free list' = 1 is libhdf5's H5HL_FREE_NULL ("no free block");
        # the library rejects anything else that is not a valid offset ("bad heap free list")
        heap_addr = self.alloc(b'HEAP' + struct.pack('<B3xQQQ', 0, len(heap), 1, heap_data))
        snod = b'SNOD' + struct.pack('<BxH', 1, len(names))
        for n in names:
            snod += struct.pack('<QQII16x', offs[n], children[n], 0, 0)
        snod += b'\0' * (40 * (2 * self.leaf_k - len(names)))
        snod_addr = self.alloc(snod)
        last = offs[names[-1]] if names else 0
        tree_node = (b'TREE' + struct.pack('<BBHQQ', 0, 0, 1 if names else 0, UNDEF, UNDEF) +
                     struct.pack('<QQQ', 0, snod_addr, last))
        tree_node += b'\0' * (16 * (2 * 16) + 8 - 24)         # room for 2 * internal K entries
        btree = self.alloc(tree_node)
        msgs = [_msg(0x0011, struct.pack('<QQ', btree, heap_addr))]
        for k, v in (tree.get('attrs') or {}).items():
            msgs.append(_attr_msg(k, v))
        return self.header(msgs), btree, heap_addr


def to_bytes(tree):
    """serialise {'attrs': {name: array}, 'groups': {name: tree}, 'datasets': {name: array}}.
    Every group is ONE symbol-table node: the file's 'group leaf node K' (a superblock
    field; libhdf5's default is 4) is raised to half the largest group."""
    most = max(len((t.get('groups') or {})) + len((t.get('datasets') or {})) for t in _walk(tree))
    leaf_k = max(4, (most + 1) // 2)
    if leaf_k > 0x7FFF:
        raise H5Unsupported('%d links in one group' % most)
    w = _Writer(leaf_k)
    ohdr, btree, heap = w.group(tree)
    eof = len(w.buf)
    sb = SIGNATURE + struct.pack('<BBBBBBBxHHI', 0, 0, 0, 0, 0, 8, 8, leaf_k, 16, 0)
    sb += struct.pack('<QQQQ', 0, UNDEF, eof, UNDEF)
    sb += struct.pack('<QQII', 0, ohdr, 1, 0) + struct.pack('<QQ', btree, heap)
    assert len(sb) == 96
    w.buf[:96] = sb
    return bytes(w.buf)


def _walk(tree):
    yield tree
    for sub in (tree.get('groups') or {}).values():
        yield from _walk(sub)


def write(path, tree):
    with open(path, 'wb') as fh:
        fh.write(to_bytes(tree))
