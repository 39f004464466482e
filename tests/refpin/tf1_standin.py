"""A numpy-backed stand-in for the ~30 TensorFlow-1.x symbols the reference's hot-path files use
(TEST INFRASTRUCTURE, build container only -- see tests/golden/make_ref_fixtures.py).

Why it exists: /root/reference/src/model/MVIN/{model.py,aggregators.py,util.py,train.py} import
`tensorflow` (1.x graph mode), which cannot be installed in this image.  Registering this module as
``sys.modules['tensorflow']`` lets the reference's OWN, UNMODIFIED Python run: its graph-construction
control flow (loops over hops / mix blocks / levels, concat orders, reshapes, which weights are
created and used where, feed assembly, the evaluation loops of util.py) executes as written, and the
fixtures it produces pin this repo's two restatements against that WIRING.  What it does not pin is
TensorFlow's arithmetic: every op below is numpy's (np.matmul, np.exp, ...), evaluated in fp32 or,
with ``set_float(np.float64)``, in fp64.  DESIGN.md section 2 says so wherever the pin is cited.

Model: a lazy graph like TF1's.  An op call creates a Node holding (numpy function, input nodes) and
a PROTOTYPE value -- the op evaluated on zero-filled placeholders with every unknown (None) dimension
set to ``UNKNOWN_DIM`` -- from which the static ``.shape`` the reference reads at graph-build time is
taken.  ``Session.run(fetches, feed_dict)`` evaluates the graph on the fed values with memoisation.
Variables take their initial values from a provider callback ``VARIABLE_PROVIDER(full_name, shape)``
so that the reference graph and the repo's oracles see identical weights.
"""
import contextlib

import numpy as np

UNKNOWN_DIM = 1          # value of a None placeholder dimension in the prototypes (set to batch_size)
VARIABLE_PROVIDER = None  # callable(full_name, shape) -> ndarray
FLOAT = np.float32        # evaluation precision of every float tensor
_SCOPE = []
_VARIABLES = {}

float32, float64, int32, int64 = "float32", "float64", "int32", "int64"


def reset(unknown_dim, provider, float_dtype=np.float32):
    global UNKNOWN_DIM, VARIABLE_PROVIDER, FLOAT
    UNKNOWN_DIM, VARIABLE_PROVIDER, FLOAT = int(unknown_dim), provider, float_dtype
    _SCOPE.clear()
    _VARIABLES.clear()


def global_variables():
    return list(_VARIABLES.values())


def _np_dtype(dtype):
    if dtype in (None, float32, float64, np.float32, np.float64):
        return FLOAT
    return {"int32": np.int32, "int64": np.int64}.get(dtype, dtype)


class Node(object):
    """A graph tensor: ``fn(*inputs)`` evaluated lazily; ``proto`` carries the static shape."""

    def __init__(self, fn, inputs, proto, name=None, kind="op"):
        self.fn, self.inputs, self.proto, self.name, self.kind = fn, inputs, proto, name, kind

    # static shape, as the reference reads it (`.shape[1]`, `.get_shape()[2]`)
    @property
    def shape(self):
        return tuple(self.proto.shape)

    def get_shape(self):
        return self.shape

    def __add__(self, o):
        return _op(np.add, self, o)

    def __radd__(self, o):
        return _op(np.add, o, self)

    def __sub__(self, o):
        return _op(np.subtract, self, o)

    def __rsub__(self, o):
        return _op(np.subtract, o, self)

    def __mul__(self, o):
        return _op(np.multiply, self, o)

    def __rmul__(self, o):
        return _op(np.multiply, o, self)

    def __neg__(self):
        return _op(np.negative, self)


def _proto_of(x):
    return x.proto if isinstance(x, Node) else x


def _op(fn, *inputs):
    proto = fn(*[_proto_of(i) for i in inputs])
    return Node(fn, inputs, np.asarray(proto))


def _evaluate(node, feeds, memo):
    if not isinstance(node, Node):
        return node
    key = id(node)
    if key in memo:
        return memo[key]
    if node.kind == "placeholder":
        if node not in feeds:
            raise KeyError(f"placeholder {node.name} was not fed")
        val = np.asarray(feeds[node], dtype=node.proto.dtype)
        want = node.static_shape
        assert val.ndim == len(want) and all(w is None or w == s for w, s in zip(want, val.shape)), \
            (node.name, val.shape, want)
    elif node.kind == "variable":
        val = node.value
    else:
        val = node.fn(*[_evaluate(i, feeds, memo) for i in node.inputs])
    memo[key] = val
    return val


# ----------------------------------------------------------------------------- graph inputs
def placeholder(dtype=None, shape=None, name=None):
    proto = np.zeros([UNKNOWN_DIM if s is None else s for s in shape], dtype=_np_dtype(dtype))
    n = Node(None, (), proto, name=name, kind="placeholder")
    n.static_shape = tuple(shape)
    return n


@contextlib.contextmanager
def variable_scope(name):
    _SCOPE.append(str(name))
    try:
        yield
    finally:
        _SCOPE.pop()


def get_variable(name=None, shape=None, dtype=None, initializer=None):
    full = "/".join(_SCOPE + [str(name)])
    if full in _VARIABLES:
        raise ValueError(f"variable {full} already exists (reuse is not set in the reference)")
    value = np.asarray(VARIABLE_PROVIDER(full, tuple(shape)), dtype=FLOAT)
    assert value.shape == tuple(shape), (full, value.shape, shape)
    n = Node(None, (), value, name=full + ":0", kind="variable")
    n.value = value
    _VARIABLES[full] = n
    return n


def zeros_initializer():
    return "zeros"


def set_random_seed(seed):
    pass


def reset_default_graph():
    _VARIABLES.clear()


# ----------------------------------------------------------------------------- ops
def _axis(axis, dim):
    return dim if axis is None and dim is not None else axis


def gather(params, indices):
    return _op(lambda p, i: np.asarray(p)[np.asarray(i)], params, indices)


def expand_dims(x, axis):
    return _op(lambda v: np.expand_dims(v, axis), x)


def squeeze(x, axis=None):
    return _op(lambda v: np.squeeze(v, axis=axis), x)


def tile(x, multiples):
    return _op(lambda v: np.tile(v, [int(m) for m in multiples]), x)


def concat(values, axis):
    return _op(lambda *v: np.concatenate(v, axis=axis), *values)


def reshape(x, shape):
    return _op(lambda v: np.reshape(v, [int(s) for s in shape]), x)


def matmul(a, b):
    return _op(np.matmul, a, b)


def reduce_sum(x, axis=None):
    return _op(lambda v: np.sum(v, axis=axis), x)


def reduce_mean(x, axis=None):
    return _op(lambda v: np.mean(v, axis=axis), x)


def sigmoid(x):
    return _op(lambda v: 1.0 / (1.0 + np.exp(-v)), x)


def _softmax(v, axis):
    e = np.exp(v - np.max(v, axis=axis, keepdims=True))
    return e / np.sum(e, axis=axis, keepdims=True)


class _NN(object):
    @staticmethod
    def embedding_lookup(params, ids):
        return gather(params, ids)

    @staticmethod
    def softmax(logits, axis=None, dim=None):
        ax = _axis(axis, dim)
        return _op(lambda v: _softmax(v, -1 if ax is None else ax), logits)

    @staticmethod
    def relu(x):
        return _op(lambda v: np.maximum(v, 0), x)

    @staticmethod
    def dropout(x, keep_prob):
        if keep_prob != 1:
            raise NotImplementedError("the reference only ever builds dropout with keep_prob = 1")
        return x

    @staticmethod
    def l2_loss(x):
        return _op(lambda v: np.sum(v * v) / 2, x)

    @staticmethod
    def sigmoid_cross_entropy_with_logits(labels=None, logits=None):
        # TF's stable form: max(x, 0) - x*z + log(1 + exp(-|x|))
        return _op(lambda z, x: np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x))), labels, logits)


nn = _NN()


class _Layers(object):
    @staticmethod
    def xavier_initializer(seed=None):
        return ("xavier", seed)


class _Contrib(object):
    layers = _Layers()


contrib = _Contrib()


# ----------------------------------------------------------------------------- training / session
class _MinimizeOp(object):
    """`AdamOptimizer(lr).minimize(loss)`: recorded, never executed (no autodiff here)."""

    def __init__(self, lr, loss):
        self.lr, self.loss = lr, loss


class _Adam(object):
    def __init__(self, lr):
        self.lr = lr

    def minimize(self, loss):
        return _MinimizeOp(self.lr, loss)


class _Saver(object):
    def __init__(self, var_list=None):
        self.var_list = list(var_list or [])
        self.saved = []

    def save(self, sess, path):
        self.saved.append(path)

    def restore(self, sess, path):
        raise NotImplementedError


class _Train(object):
    AdamOptimizer = _Adam
    Saver = _Saver


train = _Train()


def global_variables_initializer():
    return None


class _GPUOptions(object):
    allow_growth = False


class ConfigProto(object):
    def __init__(self):
        self.gpu_options = _GPUOptions()


class Session(object):
    def __init__(self, config=None):
        self.runs = 0

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def run(self, fetches, feed_dict=None):
        if fetches is None:
            return None
        self.runs += 1
        memo = {}
        feeds = feed_dict or {}

        def one(f):
            if isinstance(f, (list, tuple)):
                return [one(g) for g in f]
            if isinstance(f, _MinimizeOp) or f is None:
                return None
            if isinstance(f, Node):
                return np.array(_evaluate(f, feeds, memo))
            return f  # python constants, e.g. importance_list_1 = 0 at depth 1 (model.py:323)
        return one(fetches)
