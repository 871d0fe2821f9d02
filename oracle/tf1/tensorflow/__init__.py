"""A small TensorFlow-1.4 GRAPH EMULATOR -- test infrastructure like the rest of oracle/, never imported by the product path.

Why it exists: the reference (lmb-freiburg/demon) builds its networks with TensorFlow 1.4 calls in
python/depthmotionnet/{helpers,blocks_original,networks_original}.py and v2/{helpers,blocks,networks}.py.  TensorFlow 1.4 cannot be
installed here (no network, no cp310 wheel), so until round 4 the WIRING of the nets (which layer feeds which, concatenation
orders, channel counts, variable names and shapes, the crop of the transposed convs, the flatten order of the motion head) was
pinned only by reading.  With this package first on sys.path as `tensorflow` (and oracle/tf1/lmbspecialops as `lmbspecialops`)
the reference's OWN graph-building code runs unmodified in this container: tests/golden/make_golden_wiring.py imports
/root/reference/python/depthmotionnet, builds BootstrapNet / IterativeNet / RefinementNet, loads seeded weights by the variable
names that code created, evaluates the nets and commits the outputs as tests/golden/wiring_*.npz.  tools/dump_reference_goldens.py
(the script for the real TensorFlow environment) is also executed end to end on it (tests/test_tf1_emulator.py), so its TensorFlow
branch is no longer unexecuted code.

What it pins and what it does not: the graph topology and the variable table come from executing reference code; the arithmetic of
each primitive below is a restatement of TensorFlow's documented semantics, written in numpy independently
of oracle/net_ref.py (which uses PyTorch's convolutions): two independent restatements that must agree.  It is NOT TensorFlow:
"parity unpinned" still holds for the primitive semantics themselves.

Only the TensorFlow-1.4 API names the reference and the dump tool touch exist here, with TensorFlow 1.4's argument names
(`keep_dims`, `num_or_size_splits`, ...); anything else raises AttributeError / TypeError, which is the point of the dry run.
API (TensorFlow 1.4 names): float32, int32, bool, placeholder, constant, zeros_like, concat, split, slice, pad, transpose,
reshape, norm, where, maximum, minimum, clip_by_value, stop_gradient, identity, reduce_sum, reduce_mean, sqrt, exp, abs, add_n,
name_scope, variable_scope, get_variable_scope,
global_variables, trainable_variables, global_variables_initializer, Graph, get_default_graph, reset_default_graph, Session,
InteractiveSession, ConfigProto, GPUOptions, layers.conv2d, layers.conv2d_transpose, layers.dense, contrib.layers.flatten,
contrib.layers.variance_scaling_initializer, image.resize_nearest_neighbor, test.is_gpu_available, train.Saver, __version__.
"""
import contextlib
import os
import threading

import numpy as np

__version__ = "1.4.0-emulated"
EMULATED = True

# every top-level / sub-module attribute a caller touched: tests assert it is a subset of the TF-1.4 names listed above
TOUCHED = set()


class DType(object):
    def __init__(self, name, np_dtype):
        self.name, self.as_numpy_dtype = name, np_dtype

    def __repr__(self):
        return "tf." + self.name


float32 = DType("float32", np.float32)
int32 = DType("int32", np.int32)
bool = DType("bool", np.bool_)   # noqa: A001  (tf.bool)


def _np_dtype(dtype):
    if dtype is None:
        return None
    if isinstance(dtype, DType):
        return dtype.as_numpy_dtype
    return np.dtype(dtype).type


# =====================================================================================================================================
# graph, tensors, variables
# =====================================================================================================================================
class TensorShape(object):
    def __init__(self, dims):
        self._dims = None if dims is None else [None if d is None else int(d) for d in dims]

    def as_list(self):
        if self._dims is None:
            raise ValueError("as_list() is not defined on an unknown TensorShape.")
        return list(self._dims)

    @property
    def ndims(self):
        return None if self._dims is None else len(self._dims)

    def __len__(self):
        return len(self._dims)

    def __iter__(self):
        return iter(self._dims)

    def __getitem__(self, i):
        return self._dims[i]

    def __repr__(self):
        return "TensorShape(%r)" % (self._dims,)


class Graph(object):
    def __init__(self):
        self.variables = []
        self.var_by_name = {}
        self.scope = []               # variable_scope stack (names)
        self.layer_names = {}         # (scope, base) -> count, for tf.layers' default names
        self.nodes = 0

    @contextlib.contextmanager
    def as_default(self):
        _STATE.stack.append(self)
        try:
            yield self
        finally:
            _STATE.stack.pop()

    def scope_name(self):
        return "/".join(self.scope)


class _State(threading.local):
    def __init__(self):
        self.stack = [Graph()]
        self.default_session = None


_STATE = _State()


def get_default_graph():
    return _STATE.stack[-1]


def reset_default_graph():
    if len(_STATE.stack) != 1:
        raise AssertionError("Do not use tf.reset_default_graph() to clear nested graphs.")
    _STATE.stack[0] = Graph()


class Tensor(object):
    """A node: fn(*input arrays) -> array, with a static shape known at construction (TensorFlow's shape inference is done by
    evaluating cheap ops on zeros and by the documented formulas for the convolutions)."""
    __array_priority__ = 100

    def __init__(self, fn, inputs, shape, dtype=float32, name=None):
        self.graph = get_default_graph()
        self.fn, self.inputs, self.dtype = fn, list(inputs), dtype
        self._shape = TensorShape(shape)
        self.graph.nodes += 1
        self.name = (name or "node_%d" % self.graph.nodes) + ":0"
        for t in self.inputs:
            if t.graph is not self.graph:
                raise ValueError("Tensor %s is not an element of this graph." % t.name)

    def get_shape(self):
        return self._shape

    @property
    def shape(self):
        return self._shape

    def eval(self, feed_dict=None, session=None):
        session = session or _STATE.default_session
        if session is None:
            raise ValueError("Cannot evaluate tensor using `eval()`: No default session is registered.")
        return session.run(self, feed_dict=feed_dict)

    # the operators the reference uses on tensors: s * depth (blocks_original.py:283), norm < 1.0 (:166)
    def __mul__(self, other): return _elementwise(np.multiply, self, other)
    def __rmul__(self, other): return _elementwise(np.multiply, other, self)
    def __add__(self, other): return _elementwise(np.add, self, other)
    def __radd__(self, other): return _elementwise(np.add, other, self)
    def __sub__(self, other): return _elementwise(np.subtract, self, other)
    def __rsub__(self, other): return _elementwise(np.subtract, other, self)
    def __truediv__(self, other): return _elementwise(np.divide, self, other)
    def __neg__(self): return _elementwise(np.negative, self)
    def __pow__(self, other): return _elementwise(np.power, self, other)       # x**2 (v2/losses.py:29, :50)
    def __lt__(self, other): return _elementwise(np.less, self, other, dtype=bool)
    def __le__(self, other): return _elementwise(np.less_equal, self, other, dtype=bool)
    def __gt__(self, other): return _elementwise(np.greater, self, other, dtype=bool)
    def __ge__(self, other): return _elementwise(np.greater_equal, self, other, dtype=bool)

    def __getitem__(self, key):
        return _cheap(lambda a: a[key], [self], dtype=self.dtype)

    def __bool__(self):
        raise TypeError("Using a `tf.Tensor` as a Python `bool` is not allowed.")

    __nonzero__ = __bool__

    def __hash__(self):
        return id(self)

    def __eq__(self, other):
        return self is other

    def __repr__(self):
        return "<tf.Tensor '%s' shape=%s dtype=%s>" % (self.name, tuple(self._shape.as_list()), self.dtype.name)


class Variable(Tensor):
    """tf.Variable as tf.layers creates it through tf.get_variable: named '<scope>/<layer>/kernel:0'; its value lives in the graph
    (one session per graph in every caller here); reading it before it was loaded / initialised raises like TensorFlow does."""

    def __init__(self, name, shape, initializer=None):
        g = get_default_graph()
        if name in g.var_by_name:
            raise ValueError("Variable %s already exists, disallowed. Did you mean to set reuse=True or reuse=tf.AUTO_REUSE in VarScope?" % name)
        Tensor.__init__(self, self._read, [], shape, float32, name)
        self.value, self.initializer = None, initializer
        g.variables.append(self)
        g.var_by_name[name] = self

    def _read(self):
        if self.value is None:
            raise RuntimeError("FailedPreconditionError: Attempting to use uninitialized value %s" % self.name.split(":")[0])
        return self.value

    def load(self, value, session=None):
        session = session or _STATE.default_session
        if session is None:
            raise ValueError("Either session argument should be provided or default session should be established")
        value = np.asarray(value, np.float32)
        if list(value.shape) != self._shape.as_list():
            raise ValueError("Cannot feed value of shape %r for Tensor %r, which has shape %r" % (value.shape, self.name, tuple(self._shape.as_list())))
        self.value = value.copy()


def _as_tensor(x, dtype=None):
    if isinstance(x, Tensor):
        return x
    return constant(x, dtype=dtype or (float32 if not isinstance(x, (np.ndarray,)) or x.dtype.kind == "f" else None))


def _cheap(fn, inputs, dtype=float32, name=None):
    """an op whose static shape is inferred by running it on zeros of the inputs' static shapes (elementwise ops, data movement)"""
    zeros = [np.zeros(t.get_shape().as_list(), t.dtype.as_numpy_dtype) for t in inputs]
    return Tensor(fn, inputs, np.asarray(fn(*zeros)).shape, dtype, name)


def _elementwise(op, *args, **kw):
    dtype = kw.pop("dtype", float32)
    ts = [_as_tensor(a) for a in args]
    return _cheap(lambda *arrays: op(*arrays), ts, dtype=dtype)


# =====================================================================================================================================
# sessions
# =====================================================================================================================================
class GPUOptions(object):
    def __init__(self, per_process_gpu_memory_fraction=None, allow_growth=None, visible_device_list=None):
        self.per_process_gpu_memory_fraction, self.allow_growth = per_process_gpu_memory_fraction, allow_growth


class ConfigProto(object):
    def __init__(self, allow_soft_placement=None, log_device_placement=None, gpu_options=None, device_count=None,
                 intra_op_parallelism_threads=None, inter_op_parallelism_threads=None):
        self.allow_soft_placement, self.gpu_options = allow_soft_placement, gpu_options


class Session(object):
    def __init__(self, target="", graph=None, config=None):
        self.graph = graph or get_default_graph()
        self._closed = False

    def close(self):
        self._closed = True

    def __enter__(self):
        self._prev = _STATE.default_session
        _STATE.default_session = self
        return self

    def __exit__(self, *exc):
        _STATE.default_session = self._prev
        self.close()
        return False

    def as_default(self):
        return self

    def run(self, fetches, feed_dict=None, options=None, run_metadata=None):
        if self._closed:
            raise RuntimeError("Attempted to use a closed Session.")
        cache = {}
        for k, v in (feed_dict or {}).items():
            if not isinstance(k, Tensor):
                raise TypeError("Cannot interpret feed_dict key as Tensor: %r" % (k,))
            v = np.asarray(v, k.dtype.as_numpy_dtype)
            want = k.get_shape().as_list()
            if len(want) != v.ndim or any(w is not None and w != s for w, s in zip(want, v.shape)):
                raise ValueError("Cannot feed value of shape %r for Tensor %r, which has shape %r" % (v.shape, k.name, tuple(want)))
            cache[id(k)] = v

        def ev(t):
            if id(t) in cache:
                return cache[id(t)]
            if t.graph is not self.graph:
                raise ValueError("Tensor %s is not an element of this graph." % t.name)
            # iterative post-order walk: the nets are a few hundred nodes deep
            stack = [t]
            while stack:
                node = stack[-1]
                if id(node) in cache:
                    stack.pop()
                    continue
                missing = [i for i in node.inputs if id(i) not in cache]
                if missing:
                    stack.extend(missing)
                    continue
                out = node.fn(*[cache[id(i)] for i in node.inputs])
                out = np.asarray(out, node.dtype.as_numpy_dtype)
                static = node.get_shape().as_list()
                if list(out.shape) != static:
                    raise AssertionError("emulator bug: %s evaluated to shape %r, static shape %r" % (node.name, out.shape, static))
                cache[id(node)] = out
                stack.pop()
            return cache[id(t)]

        def walk(f):
            if isinstance(f, Tensor):
                return ev(f)
            if isinstance(f, dict):
                return dict((k, walk(v)) for k, v in f.items())
            if isinstance(f, (list, tuple)):
                return type(f)(walk(v) for v in f) if not hasattr(f, "_fields") else type(f)(*[walk(v) for v in f])
            if isinstance(f, _InitOp):
                f.run()
                return None
            raise TypeError("Fetch argument %r has invalid type %r, must be a string or Tensor." % (f, type(f)))

        return walk(fetches)


class InteractiveSession(Session):
    def __init__(self, target="", graph=None, config=None):
        Session.__init__(self, target, graph, config)
        _STATE.default_session = self


class _InitOp(object):
    def __init__(self, variables):
        self.variables = variables

    def run(self, feed_dict=None, session=None):
        for v in self.variables:
            if v.value is None:
                shape = v.get_shape().as_list()
                v.value = np.zeros(shape, np.float32) if v.initializer is None else np.asarray(v.initializer(shape), np.float32)


def global_variables_initializer():
    return _InitOp(list(get_default_graph().variables))


def global_variables():
    return list(get_default_graph().variables)


def trainable_variables():
    return list(get_default_graph().variables)


# =====================================================================================================================================
# variable scopes
# =====================================================================================================================================
class VariableScope(object):
    def __init__(self, name):
        self.name = name

    @property
    def original_name_scope(self):
        return self.name + "/"


@contextlib.contextmanager
def variable_scope(name_or_scope, default_name=None, values=None, initializer=None, regularizer=None, reuse=None, dtype=None):
    g = get_default_graph()
    if isinstance(name_or_scope, VariableScope):
        saved, g.scope = g.scope, name_or_scope.name.split("/") if name_or_scope.name else []
        try:
            yield name_or_scope
        finally:
            g.scope = saved
        return
    if name_or_scope is None:
        name_or_scope = default_name
    if reuse:
        raise NotImplementedError("emulator: reuse=True is not used by the reference's inference graphs")
    parts = [p for p in str(name_or_scope).split("/") if p]
    g.scope.extend(parts)
    try:
        yield VariableScope(g.scope_name())
    finally:
        del g.scope[len(g.scope) - len(parts):]


def get_variable_scope():
    return VariableScope(get_default_graph().scope_name())


# =====================================================================================================================================
# basic ops
# =====================================================================================================================================
def placeholder(dtype, shape=None, name=None):
    if shape is None or any(s is None for s in shape):
        raise NotImplementedError("emulator: placeholders need a fully defined shape (the reference's all have one)")

    def unfed():
        raise RuntimeError("InvalidArgumentError: You must feed a value for placeholder tensor '%s'" % t.name)

    t = Tensor(unfed, [], shape, dtype, name)
    return t


def constant(value, dtype=None, shape=None, name="Const", verify_shape=False):
    nd = _np_dtype(dtype)
    arr = np.array(value, dtype=nd) if nd is not None else np.array(value)
    if nd is None:
        if arr.dtype.kind == "f":
            arr = arr.astype(np.float32)        # tf.constant of python floats / float64 arrays without dtype: float32 / float64 --
        elif arr.dtype.kind in "iu":             # the callers here always pass dtype for float64 data
            arr = arr.astype(np.int32)
    if shape is not None:
        arr = np.broadcast_to(arr, shape).copy() if arr.size == 1 else arr.reshape(shape)
    dt = float32 if arr.dtype == np.float32 else (int32 if arr.dtype.kind in "iu" else (bool if arr.dtype == np.bool_ else None))
    if dt is None:
        raise TypeError("emulator: constants of dtype %s are not supported" % arr.dtype)
    arr = arr.astype(dt.as_numpy_dtype)
    return Tensor(lambda: arr, [], arr.shape, dt, name)


def zeros_like(tensor, dtype=None, name=None, optimize=True):
    tensor = _as_tensor(tensor)
    dt = dtype or tensor.dtype
    return _cheap(lambda a: np.zeros(a.shape, dt.as_numpy_dtype), [tensor], dtype=dt, name=name)


def identity(input, name=None):   # noqa: A002
    return _cheap(lambda a: a, [_as_tensor(input)], name=name)


def stop_gradient(input, name=None):   # noqa: A002
    t = _as_tensor(input)
    return _cheap(lambda a: a, [t], dtype=t.dtype, name=name)


def concat(values, axis, name="concat"):
    ts = [_as_tensor(v) for v in values]
    return _cheap(lambda *arrays: np.concatenate(arrays, axis=axis), ts, dtype=ts[0].dtype, name=name)


def split(value, num_or_size_splits, axis=0, num=None, name="split"):
    value = _as_tensor(value)
    dim = value.get_shape().as_list()[axis]
    if isinstance(num_or_size_splits, int):
        if dim % num_or_size_splits:
            raise ValueError("Dimension size must be evenly divisible by %d but is %d" % (num_or_size_splits, dim))
        sizes = [dim // num_or_size_splits] * num_or_size_splits
    else:
        sizes = [int(s) for s in num_or_size_splits]
        if sum(sizes) != dim:
            raise ValueError("Sum of split sizes %r must match the dimension size %d" % (sizes, dim))
    outs, at = [], 0
    for s in sizes:
        index = [np.s_[:]] * len(value.get_shape())
        index[axis] = np.s_[at:at + s]
        outs.append(_cheap(lambda a, index=tuple(index): a[index], [value], dtype=value.dtype))
        at += s
    return outs


def slice(input_, begin, size, name=None):   # noqa: A001
    input_ = _as_tensor(input_)
    shape = input_.get_shape().as_list()
    index = []
    for b, s, d in zip(begin, size, shape):
        e = d if s == -1 else b + s
        if b < 0 or e > d:
            raise ValueError("slice [%d, %d) out of range for dimension of size %d" % (b, e, d))
        index.append(np.s_[b:e])
    return _cheap(lambda a: a[tuple(index)], [input_], dtype=input_.dtype, name=name)


def pad(tensor, paddings, mode="CONSTANT", name=None, constant_values=0):
    if mode != "CONSTANT":
        raise NotImplementedError("emulator: tf.pad mode %s" % mode)
    tensor = _as_tensor(tensor)
    paddings = [tuple(int(q) for q in p) for p in paddings]
    return _cheap(lambda a: np.pad(a, paddings, mode="constant", constant_values=constant_values), [tensor], name=name)


def transpose(a, perm=None, name="transpose"):
    a = _as_tensor(a)
    return _cheap(lambda x: np.transpose(x, perm), [a], dtype=a.dtype, name=name)


def reshape(tensor, shape, name=None):
    tensor = _as_tensor(tensor)
    shape = [int(s) for s in shape]
    return _cheap(lambda x: x.reshape(shape), [tensor], dtype=tensor.dtype, name=name)


def norm(tensor, ord="euclidean", axis=None, keep_dims=False, name=None):   # noqa: A002  (TensorFlow 1.4: keep_dims)
    if ord not in ("euclidean", 2):
        raise NotImplementedError("emulator: tf.norm ord=%r" % (ord,))
    tensor = _as_tensor(tensor)
    return _cheap(lambda x: np.sqrt(np.sum(x * x, axis=axis, keepdims=keep_dims)), [tensor], name=name)


def where(condition, x=None, y=None, name=None):
    if x is None or y is None:
        raise NotImplementedError("emulator: tf.where(condition) without x, y")
    c, x, y = _as_tensor(condition), _as_tensor(x), _as_tensor(y)
    if c.dtype is not bool:
        raise TypeError("Expected bool for argument 'condition'")
    if c.get_shape().as_list() != x.get_shape().as_list():      # TensorFlow 1.4's tf.where does not broadcast (rank-1 condition aside)
        raise ValueError("Shapes %r and %r are incompatible" % (c.get_shape().as_list(), x.get_shape().as_list()))
    return _cheap(lambda cc, a, b: np.where(cc, a, b), [c, x, y], name=name)


def maximum(x, y, name=None):
    return _elementwise(np.maximum, x, y)


def minimum(x, y, name=None):
    return _elementwise(np.minimum, x, y)


def _reduce(fn, input_tensor, axis, keep_dims):
    t = _as_tensor(input_tensor)
    return _cheap(lambda a: np.asarray(fn(a, axis=None if axis is None else (tuple(axis) if isinstance(axis, (list, tuple)) else axis), keepdims=keep_dims), np.float32), [t])


def reduce_sum(input_tensor, axis=None, keep_dims=False, name=None, reduction_indices=None):   # (TensorFlow 1.4: keep_dims)
    return _reduce(np.sum, input_tensor, axis if axis is not None else reduction_indices, keep_dims)


def reduce_mean(input_tensor, axis=None, keep_dims=False, name=None, reduction_indices=None):
    return _reduce(np.mean, input_tensor, axis if axis is not None else reduction_indices, keep_dims)


def sqrt(x, name=None):
    return _elementwise(np.sqrt, x)


def exp(x, name=None):
    return _elementwise(np.exp, x)


def abs(x, name=None):   # noqa: A001
    return _elementwise(np.abs, x)


def add_n(inputs, name=None):
    ts = [_as_tensor(t) for t in inputs]
    return _cheap(lambda *arrays: sum(arrays[1:], arrays[0]), ts)


@contextlib.contextmanager
def name_scope(name, default_name=None, values=None):
    yield name          # op names carry no meaning here; variable names come from variable_scope only (as in TensorFlow)


def clip_by_value(t, clip_value_min, clip_value_max, name=None):
    # TensorFlow: minimum(maximum(t, min), max); NaN propagates (np.clip agrees)
    return _cheap(lambda a: np.minimum(np.maximum(a, np.float32(clip_value_min)), np.float32(clip_value_max)), [_as_tensor(t)], name=name)


# =====================================================================================================================================
# tf.layers
# =====================================================================================================================================
def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def _layer_scope(base, name):
    """tf.layers' naming: the given name, or '<base>', '<base>_1', ... unique within the current variable scope"""
    g = get_default_graph()
    if name is None:
        key = (g.scope_name(), base)
        n = g.layer_names.get(key, 0)
        g.layer_names[key] = n + 1
        name = base if n == 0 else "%s_%d" % (base, n)
    return "/".join([p for p in (g.scope_name(), name) if p])


def _same_pads(size, k, s):
    """TensorFlow's 'SAME': out = ceil(in / s); total padding split with the extra element at the END"""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2, total - total // 2


def _conv_nchw(x, kernel, strides, padding):
    """tf.nn.conv2d on NCHW data, HWIO kernel: cross-correlation (no flip), out[o,y,x] = sum_{a,b,i} in[i, s y + a, s x + b] K[a,b,i,o]"""
    kh, kw, cin, cout = kernel.shape
    sh, sw = strides
    n, c, h, w = x.shape
    if c != cin:
        raise ValueError("input depth %d does not match the kernel's %d" % (c, cin))
    if padding == "SAME":
        ho, pt, pb = _same_pads(h, kh, sh)
        wo, pl, pr = _same_pads(w, kw, sw)
        x = np.pad(x, ((0, 0), (0, 0), (pt, pb), (pl, pr)), mode="constant")
    else:
        ho, wo = (h - kh) // sh + 1, (w - kw) // sw + 1
    out = np.zeros((n, cout, ho, wo), np.float32)
    for a in range(kh):
        for b in range(kw):
            win = x[:, :, a:a + (ho - 1) * sh + 1:sh, b:b + (wo - 1) * sw + 1:sw]          # [n, ci, ho, wo]
            out += np.tensordot(kernel[a, b], win, axes=([0], [1])).transpose(1, 0, 2, 3)   # [co, n, ho, wo] -> [n, co, ho, wo]
    return out


def _conv_out_shape(size, k, s, padding):
    return -(-size // s) if padding == "SAME" else (size - k) // s + 1


def _check_padding(padding):
    p = str(padding).upper()
    if p not in ("VALID", "SAME"):
        raise ValueError("The `padding` argument must be one of \"valid\", \"same\". Received: " + str(padding))
    return p


def _check_format(data_format):
    if data_format not in ("channels_first", "channels_last"):
        raise ValueError("The `data_format` argument must be one of \"channels_first\", \"channels_last\". Received: " + str(data_format))
    return data_format == "channels_first"


class _Layers(object):
    def conv2d(self, inputs, filters, kernel_size, strides=(1, 1), padding="valid", data_format="channels_last", dilation_rate=(1, 1),
               activation=None, use_bias=True, kernel_initializer=None, bias_initializer=None, kernel_regularizer=None,
               bias_regularizer=None, activity_regularizer=None, kernel_constraint=None, bias_constraint=None, trainable=True,
               name=None, reuse=None):
        TOUCHED.add("layers.conv2d")
        if _pair(dilation_rate) != (1, 1) or reuse:
            raise NotImplementedError("emulator: dilation / reuse")
        nchw, padding = _check_format(data_format), _check_padding(padding)
        (kh, kw), (sh, sw) = _pair(kernel_size), _pair(strides)
        inputs = _as_tensor(inputs)
        shp = inputs.get_shape().as_list()
        if len(shp) != 4:
            raise ValueError("Input 0 of layer conv2d is incompatible with the layer: expected ndim=4, found ndim=%d" % len(shp))
        n, cin, h, w = shp if nchw else (shp[0], shp[3], shp[1], shp[2])
        scope = _layer_scope("conv2d", name)
        kernel = Variable(scope + "/kernel", (kh, kw, cin, int(filters)), kernel_initializer)
        ins = [inputs, kernel]
        if use_bias:
            ins.append(Variable(scope + "/bias", (int(filters),), None))
        ho, wo = _conv_out_shape(h, kh, sh, padding), _conv_out_shape(w, kw, sw, padding)
        if ho < 1 or wo < 1:
            raise ValueError("Negative dimension size caused by subtracting %d from %d" % (kh, h))

        def fn(x, k, b=None):
            y = _conv_nchw(x if nchw else x.transpose(0, 3, 1, 2), k, (sh, sw), padding)
            if b is not None:
                y = y + b.reshape(1, -1, 1, 1)
            return y if nchw else y.transpose(0, 2, 3, 1)

        out = Tensor(fn, ins, (n, int(filters), ho, wo) if nchw else (n, ho, wo, int(filters)), float32, scope + "/BiasAdd")
        return activation(out) if activation is not None else out

    def conv2d_transpose(self, inputs, filters, kernel_size, strides=(1, 1), padding="valid", data_format="channels_last",
                         activation=None, use_bias=True, kernel_initializer=None, bias_initializer=None, kernel_regularizer=None,
                         bias_regularizer=None, activity_regularizer=None, kernel_constraint=None, bias_constraint=None,
                         trainable=True, name=None, reuse=None):
        """The gradient of conv2d with respect to its input (TensorFlow's definition).  For the conv whose input is this op's OUTPUT:
        VALID reads out[s y + a] for in[y]  =>  scatter  out[o, s y + a, s x + b] += in[i, y, x] K[a, b, o, i],  size (in - 1) s + k;
        SAME  (out = in s) pads the conv's input by pad_before = (k - s) // 2  =>  the same scatter shifted by -pad_before and cut to in s.
        Kernel variable: [kh, kw, filters, Cin]."""
        TOUCHED.add("layers.conv2d_transpose")
        if reuse:
            raise NotImplementedError("emulator: reuse")
        nchw, padding = _check_format(data_format), _check_padding(padding)
        (kh, kw), (sh, sw) = _pair(kernel_size), _pair(strides)
        inputs = _as_tensor(inputs)
        shp = inputs.get_shape().as_list()
        n, cin, h, w = shp if nchw else (shp[0], shp[3], shp[1], shp[2])
        scope = _layer_scope("conv2d_transpose", name)
        kernel = Variable(scope + "/kernel", (kh, kw, int(filters), cin), kernel_initializer)
        ins = [inputs, kernel]
        if use_bias:
            ins.append(Variable(scope + "/bias", (int(filters),), None))
        fh, fw = (h - 1) * sh + kh, (w - 1) * sw + kw          # full scatter extent
        if padding == "VALID":
            # TensorFlow (deconv_output_length): in * s + max(k - s, 0)
            ho, wo, oy, ox = h * sh + max(kh - sh, 0), w * sw + max(kw - sw, 0), 0, 0
        else:
            ho, wo = h * sh, w * sw
            # the forward SAME conv on an (in s)-sized map with stride s produces `in` outputs: total padding max((in-1) s + k - in s, 0)
            oy, ox = max((h - 1) * sh + kh - ho, 0) // 2, max((w - 1) * sw + kw - wo, 0) // 2

        def fn(x, k, b=None):
            x = x if nchw else x.transpose(0, 3, 1, 2)
            full = np.zeros((n, int(filters), max(fh, ho + oy), max(fw, wo + ox)), np.float32)
            for a in range(kh):
                for c in range(kw):
                    contrib = np.tensordot(k[a, c], x, axes=([1], [1])).transpose(1, 0, 2, 3)     # [o, n, h, w] -> [n, o, h, w]
                    full[:, :, a:a + (h - 1) * sh + 1:sh, c:c + (w - 1) * sw + 1:sw] += contrib
            y = full[:, :, oy:oy + ho, ox:ox + wo]
            if b is not None:
                y = y + b.reshape(1, -1, 1, 1)
            return y if nchw else y.transpose(0, 2, 3, 1)

        out = Tensor(fn, ins, (n, int(filters), ho, wo) if nchw else (n, ho, wo, int(filters)), float32, scope + "/BiasAdd")
        return activation(out) if activation is not None else out

    def dense(self, inputs, units, activation=None, use_bias=True, kernel_initializer=None, bias_initializer=None,
              kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None,
              bias_constraint=None, trainable=True, name=None, reuse=None):
        TOUCHED.add("layers.dense")
        if reuse:
            raise NotImplementedError("emulator: reuse")
        inputs = _as_tensor(inputs)
        shp = inputs.get_shape().as_list()
        scope = _layer_scope("dense", name)
        kernel = Variable(scope + "/kernel", (shp[-1], int(units)), kernel_initializer)
        ins = [inputs, kernel]
        if use_bias:
            ins.append(Variable(scope + "/bias", (int(units),), None))

        def fn(x, k, b=None):
            y = np.tensordot(x, k, axes=([x.ndim - 1], [0]))
            return y + b if b is not None else y

        out = Tensor(fn, ins, tuple(shp[:-1]) + (int(units),), float32, scope + "/BiasAdd")
        return activation(out) if activation is not None else out


layers = _Layers()


# =====================================================================================================================================
# tf.contrib.layers, tf.image, tf.test, tf.train
# =====================================================================================================================================
class _ContribLayers(object):
    def flatten(self, inputs, outputs_collections=None, scope=None):
        TOUCHED.add("contrib.layers.flatten")
        inputs = _as_tensor(inputs)
        return _cheap(lambda x: x.reshape(x.shape[0], -1), [inputs])

    def variance_scaling_initializer(self, factor=2.0, mode="FAN_IN", uniform=False, seed=None, dtype=float32):
        """He-style truncated normal on FAN_IN (tf.contrib.layers, TensorFlow 1.4): stddev = sqrt(1.3 * factor / fan_in), truncated at 2 sigma"""
        TOUCHED.add("contrib.layers.variance_scaling_initializer")
        rs = np.random.RandomState(0 if seed is None else seed)

        def init(shape):
            fan_in = float(np.prod(shape[:-1])) if len(shape) > 1 else float(shape[0])
            std = np.sqrt(1.3 * factor / max(1.0, fan_in))
            v = rs.standard_normal(shape)
            bad = np.abs(v) > 2
            while bad.any():
                v[bad] = rs.standard_normal(int(bad.sum()))
                bad = np.abs(v) > 2
            return (v * std).astype(np.float32)
        return init


class _Contrib(object):
    layers = _ContribLayers()


contrib = _Contrib()


class _Image(object):
    def resize_nearest_neighbor(self, images, size, align_corners=False, name=None):
        """NHWC; without align_corners: src = min(floor(dst * in / out), in - 1)"""
        TOUCHED.add("image.resize_nearest_neighbor")
        if align_corners:
            raise NotImplementedError("emulator: align_corners")
        images = _as_tensor(images)
        ho, wo = int(size[0]), int(size[1])

        def fn(x):
            h, w = x.shape[1], x.shape[2]
            ys = np.minimum(np.floor(np.arange(ho) * (float(h) / ho)).astype(np.int64), h - 1)
            xs = np.minimum(np.floor(np.arange(wo) * (float(w) / wo)).astype(np.int64), w - 1)
            return x[:, ys][:, :, xs]
        return _cheap(fn, [images], name=name)


image = _Image()


class _Test(object):
    def is_gpu_available(self, cuda_only=False):
        """the emulator has no GPU unless TF1_EMU_GPU=1 pretends one (it can evaluate channels_first graphs either way)"""
        TOUCHED.add("test.is_gpu_available")
        return os.environ.get("TF1_EMU_GPU", "0") == "1"


test = _Test()


class _Saver(object):
    """tf.train.Saver.save / restore in the TensorBundle format, through this repo's writer / reader (demon_amd/tf_checkpoint.py): good
    enough for a dry run of the dump tool; a bundle written here says nothing about TensorFlow's own writer."""

    def __init__(self, var_list=None, max_to_keep=5):
        self.var_list = list(var_list) if var_list is not None else global_variables()

    def save(self, sess, save_path, global_step=None, latest_filename=None, meta_graph_suffix="meta", write_meta_graph=True, write_state=True):
        from demon_amd import tf_checkpoint as ck
        ck.save_tf_checkpoint(save_path, dict((v.name.split(":")[0], v._read()) for v in self.var_list))
        return save_path

    def restore(self, sess, save_path):
        from demon_amd import tf_checkpoint as ck
        have = ck.load_tf_checkpoint(save_path)
        for v in self.var_list:
            name = v.name.split(":")[0]
            if name not in have:
                raise KeyError("NotFoundError: Key %s not found in checkpoint" % name)
            v.load(have[name], sess)


class _Train(object):
    def Saver(self, var_list=None, max_to_keep=5, **kwargs):   # noqa: N802
        TOUCHED.add("train.Saver")
        if kwargs:
            raise TypeError("emulator: tf.train.Saver arguments %r" % sorted(kwargs))
        return _Saver(var_list, max_to_keep)


train = _Train()
