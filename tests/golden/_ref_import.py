"""Dev-time helper (THIS container only): import the read-only reference package
with inert stubs for the third-party modules that are absent here and that the
hot path never touches.  Never imported by the test-suite."""
import sys
from unittest import mock

_STUBS = [
    'skimage', 'skimage.exposure', 'h5py', 'keras', 'keras.models', 'keras.layers',
    'keras.layers.core', 'keras.callbacks', 'keras.backend', 'tensorflow',
    'tensorflow.python', 'tensorflow.python.framework',
    'tensorflow.python.framework.ops', 'tensorflow.python.ops',
    'tensorflow.python.ops.nn', 'tensorflow.python.ops.clip_ops',
    'tensorflow.python.ops.math_ops', 'diced', 'libdvid', 'libdvid._dvid_python',
    'z5py', 'pulp', 'flyem_syn_eval', 'matplotlib', 'matplotlib.pyplot',
]


def import_reference(path='/root/reference'):
    for name in _STUBS:
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock()
    sys.modules['keras.callbacks'].Callback = object
    # fri_get_image does isinstance(node, z5py.dataset.Dataset): needs a type
    sys.modules['z5py'].dataset.Dataset = type('Dataset', (), {})
    sys.dont_write_bytecode = True
    if path not in sys.path:
        sys.path.insert(0, path)
    from flypylib import fplutils, fplobjdetect, fplnetwork  # noqa
    return fplutils, fplobjdetect, fplnetwork
