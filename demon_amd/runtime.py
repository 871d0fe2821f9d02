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
        if ctx.version == version and getattr(ctx, "_owner", None) is None:   # session-owned contexts keep their own set
            ctx.set_weights(weights)


def default_weights(version=1):
    return _default_weights[version]


def _owner(session):
    """sessions that carry their own variables (python/tf_stub: an object with a `demon_weights` attribute) own their contexts,
    like a tf.Session owns its variables; anything else (None, a real tf.Session) shares the process default"""
    return session if session is not None and hasattr(session, "demon_weights") else None


def get_context(batch_size=1, height=192, width=256, device=None, version=1, session=None):
    """One context per (device, batch, size, model version, owning session).  Nets built on different weight-carrying
    sessions get different contexts, so restoring a checkpoint into one session does not replace the weights of nets built
    on another (the reference's variables live in the tf.Session, examples/example.py:70-83)."""
    if device is None:
        device = int(os.environ.get("DEMON_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    owner = _owner(session)
    key = (device, batch_size, height, width, version, id(owner) if owner is not None else 0)
    ctx = _contexts.get(key)
    if ctx is None:
        ctx = DemonContext(device, batch_size, height, width, version)
        if os.environ.get("DEMON_HIPGRAPH", "1") == "0":
            ctx.set_option("hipgraph", 0)
        ctx.load_tuned_plan(batch_size)   # measured launch plan for this shape, when one is shipped (demon_amd/tuned)
        w = owner.demon_weights if owner is not None and owner.demon_weights is not None else _default_weights[version]
        if w is not None:
            ctx.set_weights(w)
        ctx._owner = owner   # keeps the session alive, so its id stays unique while the context exists
        _contexts[key] = ctx
    return ctx


def set_session_weights(session, weights):
    """Saver.restore(session, ...) of python/tf_stub: the weights become the session's variables and go to every context the
    session owns (nets are constructed before restore(), examples/example.py:75-83).  They also become the process default
    when there is none yet, for nets built without a session."""
    from .weights import weights_version
    version = weights_version(weights)
    session.demon_weights = weights
    for ctx in _contexts.values():
        if ctx.version == version and getattr(ctx, "_owner", None) is session:
            ctx.set_weights(weights)
    if _default_weights[version] is None:
        set_default_weights(weights)


_ops_contexts = {}


def get_ops_context(device=None):
    """Stream + workspace only (demon_create_ops): what the lmbspecialops-level entry points need.  No network, no
    activation arena, no packed weights."""
    if device is None:
        device = int(os.environ.get("DEMON_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    ctx = _ops_contexts.get(device)
    if ctx is None:
        ctx = _ops_contexts[device] = DemonContext.ops_only(device)
    return ctx


def release_all():
    for ctx in list(_contexts.values()) + list(_ops_contexts.values()):
        ctx.close()
    _contexts.clear()
    _ops_contexts.clear()


# ---- optional record of what the network classes evaluated (test hook for unmodified reference drivers) ----------------------
_eval_log = {}


def note_eval(net, inputs, outputs):
    """With DEMON_EVAL_DUMP=<path.npz> in the environment every eval() of the network classes leaves its inputs and outputs
    here ("<ClassName>.<key>", the last call wins) and the collection is written to <path.npz> at interpreter exit: a test can
    then run an UNMODIFIED driver script (examples/example.py of the reference) and still see what it computed."""
    path = os.environ.get("DEMON_EVAL_DUMP")
    if not path:
        return
    import atexit
    import numpy as np
    if not _eval_log:
        atexit.register(lambda: np.savez(path, **_eval_log))
    name = type(net).__name__
    for k, v in list(inputs.items()) + list(outputs.items()):
        _eval_log["%s.%s" % (name, k)] = np.array(v, copy=True)
    _eval_log["%s.calls" % name] = np.array(int(_eval_log.get("%s.calls" % name, 0)) + 1)
    _eval_log["data_format"] = np.array(net.data_format)
