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


def get_context(device=0):
    key = (os.getpid(), int(device))
    ctx = _contexts.get(key)
    if ctx is None or ctx.h is None:
        ctx = _capi.Context(device)
        _contexts[key] = ctx
    return ctx


def default_device():
    return int(os.environ.get('LOCAL_RANK', '0'))
