"""Synthetic inputs for benchmarks and tests.

`em_volume_u8` is a pure function of (seed, global z, y, x) built on splitmix64,
implemented identically in csrc/synth.hip so the device can generate any
subvolume of a 4096^3 volume in place.  `synthetic_weights` fills a layer
program with seeded glorot-uniform kernels and non-trivial BN statistics.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def em_volume_u8(seed, dims, origin=(0, 0, 0)):
    """EM-like uint8 volume: 4-byte-sum noise around 128 (sigma ~33) plus one dark
    blob of radius 7 per 64^3 lattice cell."""
    with np.errstate(over='ignore'):
        seed = np.uint64(seed)
        z = (np.arange(dims[0], dtype=np.int64) + origin[0])[:, None, None]
        y = (np.arange(dims[1], dtype=np.int64) + origin[1])[None, :, None]
        x = (np.arange(dims[2], dtype=np.int64) + origin[2])[None, None, :]
        zu, yu, xu = (a.astype(np.uint64) for a in (z, y, x))
        idx = (zu << np.uint64(42)) | (yu << np.uint64(21)) | xu
        h = _splitmix64(seed ^ _splitmix64(idx))
        b = np.uint64(255)
        s4 = ((h & b) + ((h >> np.uint64(8)) & b) + ((h >> np.uint64(16)) & b)
              + ((h >> np.uint64(24)) & b)).astype(np.int64)
        v = 128 + (((s4 - 510) * 57) >> 8)
        cz, cy, cx = z >> 6, y >> 6, x >> 6
        cidx = ((cz.astype(np.uint64) << np.uint64(42))
                | (cy.astype(np.uint64) << np.uint64(21)) | cx.astype(np.uint64))
        hc = _splitmix64((seed + np.uint64(0x5851F42D4C957F2D))
                         ^ _splitmix64(cidx))
        bz = (cz << 6) + 16 + (hc & np.uint64(31)).astype(np.int64)
        by = (cy << 6) + 16 + ((hc >> np.uint64(8)) & np.uint64(31)).astype(np.int64)
        bx = (cx << 6) + 16 + ((hc >> np.uint64(16)) & np.uint64(31)).astype(np.int64)
        d2 = (z - bz) ** 2 + (y - by) ** 2 + (x - bx) ** 2
        v = v - np.where(d2 < 49, (60 * (49 - d2)) // 49, 0)
        return np.clip(v, 0, 255).astype(np.uint8)


def synthetic_weights(graph, seed=1234):
    """seeded glorot-uniform kernels (already drawn at graph construction from
    `seed`), BN gamma~U(.5,1.5), beta~N(0,.1), mean~N(0,.1), var~U(.5,1.5)"""
    rng = np.random.default_rng(seed)
    new = []
    for n in graph.nodes:
        for s in n.weight_slots:
            w = graph.weights[s]
            name = graph.weight_names[s].split('/')[-1]
            if name == 'kernel':
                k = w.shape[0]
                lim = np.sqrt(6.0 / (k ** 3 * (w.shape[3] + w.shape[4])))
                w = rng.uniform(-lim, lim, w.shape)
            elif name == 'bias':
                w = np.zeros(w.shape)
            elif name in ('gamma', 'moving_variance'):
                w = rng.uniform(0.5, 1.5, w.shape)
            else:
                w = 0.1 * rng.standard_normal(w.shape)
            new.append(w.astype(np.float32))
    graph.set_weights(new)
    return graph


def hash_uniform_f32(seed, shape):
    """float32 uniform [0,1) volume, k / 2^24 from splitmix64 of the flat index
    (integer arithmetic only -> identical on every machine)"""
    with np.errstate(over='ignore'):
        n = int(np.prod(shape))
        idx = np.arange(n, dtype=np.uint64)
        h = _splitmix64(np.uint64(seed) ^ _splitmix64(idx))
        return ((h >> np.uint64(40)).astype(np.float32)
                / np.float32(1 << 24)).reshape(shape)


def blob_prob_volume(seed, shape, period=24, radius=6.0, noise=0.05):
    """T-bar-like probability volume: one compact blob per `period`^3 lattice
    cell (jittered centre, hashed peak height) over low uniform noise.  Only
    + - * / on float64, then one cast: bit-reproducible everywhere."""
    with np.errstate(over='ignore'):
        z = np.arange(shape[0], dtype=np.int64)[:, None, None]
        y = np.arange(shape[1], dtype=np.int64)[None, :, None]
        x = np.arange(shape[2], dtype=np.int64)[None, None, :]
        cz, cy, cx = z // period, y // period, x // period
        cidx = ((cz.astype(np.uint64) << np.uint64(42))
                | (cy.astype(np.uint64) << np.uint64(21)) | cx.astype(np.uint64))
        hc = _splitmix64(np.uint64(seed) ^ _splitmix64(cidx))
        span = np.uint64(max(period // 2, 1))
        q = period // 4
        bz = cz * period + q + (hc % span).astype(np.int64)
        by = cy * period + q + ((hc >> np.uint64(8)) % span).astype(np.int64)
        bx = cx * period + q + ((hc >> np.uint64(16)) % span).astype(np.int64)
        height = 0.5 + ((hc >> np.uint64(32)) & np.uint64(1023)).astype(
            np.float64) / 2048.0
        d2 = ((z - bz) ** 2 + (y - by) ** 2 + (x - bx) ** 2).astype(np.float64)
        fall = np.maximum(0.0, 1.0 - d2 / (radius * radius))
        v = height * fall * fall
        v = v + noise * hash_uniform_f32(seed + 1, shape).astype(np.float64)
        return np.minimum(v, 1.0).astype(np.float32)


def dropout_keep_mask(seed, layer, n, rate):
    """keep-mask of the training engine's Dropout (csrc/train.hip::drop_keep):
    element i of lowered layer `layer` is kept iff u_i >= rate with
    u_i = (splitmix64(seed ^ splitmix64((layer << 48) ^ i)) >> 40) / 2^24"""
    with np.errstate(over='ignore'):
        i = np.arange(n, dtype=np.uint64)
        key = (np.uint64(layer) << np.uint64(48)) ^ i
        h = _splitmix64(np.uint64(seed) ^ _splitmix64(key))
        u = (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        return u >= np.float32(rate)


def voronoi_segmentation(seed, shape, n_sites, tiny=0):
    """synthetic segmentation: Voronoi cells of `n_sites` hashed sites (labels are
    large sparse uint64 ids, as DVID body ids are), plus `tiny` isolated 2^3 specks
    with their own labels.  Deterministic in (seed, shape, n_sites, tiny)."""
    rs = np.random.RandomState(seed)
    sites = rs.rand(n_sites, 3) * np.asarray(shape)
    ids = (rs.randint(1, 2 ** 31, n_sites).astype(np.uint64) << np.uint64(20)) | \
        np.arange(n_sites, dtype=np.uint64)
    z, y, x = np.meshgrid(*(np.arange(s) for s in shape), indexing='ij')
    best = np.full(shape, np.inf)
    lab = np.zeros(shape, np.uint64)
    for k in range(n_sites):
        d = (z - sites[k, 0]) ** 2 + (y - sites[k, 1]) ** 2 + (x - sites[k, 2]) ** 2
        m = d < best
        best[m] = d[m]
        lab[m] = ids[k]
    for k in range(tiny):
        c = [rs.randint(1, s - 3) for s in shape]
        lab[c[0]:c[0] + 2, c[1]:c[1] + 2, c[2]:c[2] + 2] = np.uint64(7000000 + k)
    return lab
