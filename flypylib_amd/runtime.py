"""Process-wide GPU contexts (one `fpl_ctx` per device per process).

HIP state does not survive fork(): contexts are keyed by pid, so a forked child
that touches the GPU gets an explicit error from the driver rather than a stale
handle (the reference forks its post-processing workers,
`flypylib/fplobjdetect.py:497-503`; here post-processing runs in the inferring
process or in spawned workers).
"""
import os

from . import _capi

_contexts = {}


def get_context(device=0, lane=0):
    """`lane` > 0 names additional contexts on the same device - each has its own HIP
    stream and voxel2obj state, so e.g. the substack pipeline post-processes substack
    i on lane 1 while lane 0 infers substack i+1 (a context serves one thread at a
    time; device buffers are shared across the lanes of a device)."""
    key = (os.getpid(), int(device), int(lane))
    ctx = _contexts.get(key)
    if ctx is None or ctx.h is None:
        ctx = _capi.Context(device)
        _contexts[key] = ctx
    return ctx


def default_device():
    return int(os.environ.get('LOCAL_RANK', '0'))
