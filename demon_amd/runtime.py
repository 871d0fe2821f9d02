"""Process-wide context / weight registry shared by the three network classes.

The reference shares one tf.Session and one set of variables between BootstrapNet, IterativeNet and
RefinementNet (examples/example.py:70-83); here they share one DemonContext per (device, batch, size).
"""
import os

from .engine import DemonContext, DemonError

_contexts = {}
_default_weights = {1: None, 2: None}   # per model version (original / v2)


def set_default_weights(weights):
    """weights: dict tf variable name -> array (TF layout).  Replaces Saver.restore (example.py:82-83,
    example_v2.py:88-89); whether it is the original or the v2 model is read off the variable names."""
    from .weights import weights_version
    version = weights_version(weights)
    _default_weights[version] = weights
    for ctx in _contexts.values():
        if ctx.version == version:
            ctx.set_weights(weights)


def default_weights(version=1):
    return _default_weights[version]


def get_context(batch_size=1, height=192, width=256, device=None, version=1):
    if device is None:
        device = int(os.environ.get("DEMON_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    key = (device, batch_size, height, width, version)
    ctx = _contexts.get(key)
    if ctx is None:
        ctx = DemonContext(device, batch_size, height, width, version)
        if os.environ.get("DEMON_HIPGRAPH", "1") == "0":
            ctx.set_option("hipgraph", 0)
        ctx.load_tuned_plan(batch_size)   # measured launch plan for this shape, when one is shipped (demon_amd/tuned)
        if _default_weights[version] is not None:
            ctx.set_weights(_default_weights[version])
        _contexts[key] = ctx
    return ctx


def release_all():
    for ctx in _contexts.values():
        ctx.close()
    _contexts.clear()
