"""flypylib_amd - MI355X-native T-bar detection path with flypylib's API.

    from flypylib_amd import FplNetwork, fplmodels, fplobjdetect

mirrors `from flypylib import FplNetwork` (reference flypylib/__init__.py:21).
Importing the package needs no GPU; the first call that computes anything opens
libfplhip.so and fails loudly if it (or a gfx950 device) is missing.
"""
from . import fplutils, fplmodels, fplobjdetect, fplnetwork, fplsynapses, multi_gpu  # noqa
from .fplnetwork import FplNetwork  # noqa

__all__ = ['FplNetwork', 'fplutils', 'fplmodels', 'fplobjdetect', 'fplnetwork',
           'fplsynapses', 'multi_gpu']
