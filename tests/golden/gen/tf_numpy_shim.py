"""numpy-backed stand-in for the `tensorflow` / `keras` modules, used ONLY in the
development container to execute the reference's own Python source
(/root/reference, read-only) eagerly and record golden vectors.

TEST INFRASTRUCTURE.  Nothing here is imported by the product (`synthsr_amd`),
by the `-m gpu` tests or by bench.py; the GPU box never sees /root/reference.
Only `make_goldens.py` (same directory) uses it.

Design:
  * tensors are a float32/int32-by-default ndarray subclass with
    `.get_shape().as_list()` and NON-mutating `+= *= -= /=` (TF tensors are
    immutable; the reference writes `vec += ...` and `prod *= p`);
  * `tf.random.normal/uniform` are served from a recorded *tape* of raw N(0,1)
    / U[0,1) draws; TF's own affine maps (`rnd*stddev+mean`,
    `rnd*(maxval-minval)+minval`) are applied in float32 on top;
  * `tf.matmul` is a k-ordered float32 multiply/add chain without FMA.  TF's
    real accumulation order is unpinned (Eigen, not vendored); this order is
    OUR convention and the oracle / HIP kernels follow it bit for bit;
  * `keras.layers.Input` pops real arrays from a feed queue so a Keras "graph"
    runs eagerly; `Layer.__call__` = build(input_shape) once + call().
"""
import sys
import types
import itertools
import numpy as np
import scipy.ndimage
import scipy.stats

F32 = np.float32
I32 = np.int32


# ----------------------------------------------------------------------------- tensor type
class _Shape(list):
    def as_list(self):
        return list(self)

    def __getitem__(self, item):
        r = list.__getitem__(self, item)
        return _Shape(r) if isinstance(item, slice) else r


class TensorShape:  # isinstance(x.shape, tf.TensorShape) must be False for our tensors
    pass


class T(np.ndarray):
    """ndarray subclass standing in for tf.Tensor"""
    _none_batch = False

    def get_shape(self):
        s = _Shape(self.shape)
        if self._none_batch and len(s):
            s[0] = None
        return s

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        """dtype fidelity: whenever a tensor takes part in an op, numpy float64/int64 operands
        (np.float64 scalars are NOT weak under NEP 50) are converted to float32/int32 first, as
        TF does when it converts a numpy operand to the tensor's dtype."""
        args = []
        for x in inputs:
            if isinstance(x, T):
                x = x.view(np.ndarray)
            if isinstance(x, (np.ndarray, np.generic)):
                if x.dtype == np.float64:
                    x = x.astype(np.float32)
                elif x.dtype == np.int64:
                    x = x.astype(np.int32)
            args.append(x)
        res = getattr(ufunc, method)(*args, **kwargs)
        if isinstance(res, tuple):
            return tuple(r.view(T) if isinstance(r, np.ndarray) else r for r in res)
        if isinstance(res, np.ndarray):
            return res.view(T)
        if isinstance(res, np.generic):
            return np.asarray(res).view(T)
        return res

    # TF tensors are immutable: in-place operators must rebind, not mutate
    def __iadd__(self, o):
        return self + o

    def __isub__(self, o):
        return self - o

    def __imul__(self, o):
        return self * o

    def __itruediv__(self, o):
        return self / o

    def __len__(self):
        return self.shape[0]


def _dt(dtype):
    if dtype is None:
        return None
    if isinstance(dtype, str):
        return np.dtype({'float32': 'float32', 'int32': 'int32', 'bool': 'bool', 'int64': 'int64',
                         'float64': 'float64', 'float': 'float32', 'int': 'int32'}[dtype])
    return np.dtype(dtype)


def t(x, dtype=None):
    """to tensor with TF defaults: python float -> float32, python int -> int32"""
    if isinstance(x, T) and dtype is None:
        return x
    a = np.asarray(x)
    if dtype is not None:
        a = a.astype(_dt(dtype))
    elif a.dtype == np.float64:
        a = a.astype(F32)
    elif a.dtype == np.int64:
        a = a.astype(I32)
    return a.view(T)


def _ax(axis):
    if isinstance(axis, np.ndarray):
        axis = axis.tolist()
    if isinstance(axis, list):
        axis = tuple(int(a) for a in axis)
    return axis


# ----------------------------------------------------------------------------- random tape
class Tape:
    """records raw draws in call order; can also replay a given list"""

    def __init__(self, seed=0, replay=None):
        self.rng = np.random.default_rng(seed)
        self.entries = []  # list of (kind, array)
        self.replay = replay
        self.pos = 0

    def _draw(self, kind, shape):
        shape = tuple(int(s) for s in np.asarray(shape).reshape(-1))
        if self.replay is not None:
            k, a = self.replay[self.pos]
            assert k == kind and tuple(a.shape) == shape, (k, kind, a.shape, shape)
            self.pos += 1
        else:
            if kind == 'n':
                a = self.rng.standard_normal(shape, dtype=np.float32)
            else:
                a = self.rng.random(shape, dtype=np.float32)
        self.entries.append((kind, a.copy()))
        return a.view(T)


TAPE = Tape()


def set_tape(tape):
    global TAPE
    TAPE = tape


def _rand_normal(shape, mean=0.0, stddev=1.0, dtype=None, **kw):
    rnd = TAPE._draw('n', shape)
    return t(rnd * t(stddev) + t(mean))


def _rand_uniform(shape, minval=0, maxval=None, dtype='float32', **kw):
    if maxval is None:
        maxval = 1
    rnd = TAPE._draw('u', shape)
    if _dt(dtype) == np.int32:
        return t(np.floor(rnd * F32(maxval - minval)).astype(I32) + I32(minval))
    minval = t(np.asarray(minval, dtype=F32))
    maxval = t(np.asarray(maxval, dtype=F32))
    return t(rnd * (maxval - minval) + minval)


# ----------------------------------------------------------------------------- tf functions
def _matmul(a, b):
    a = np.asarray(a, dtype=F32)
    b = np.asarray(b, dtype=F32)
    K_ = a.shape[-1]
    acc = a[..., :, 0:1] * b[..., 0:1, :]
    for k in range(1, K_):
        acc = acc + a[..., :, k:k + 1] * b[..., k:k + 1, :]
    return t(acc)


def _cast(x, dtype=None, **kw):
    x = np.asarray(x)
    return t(x.astype(_dt(dtype)))  # float->int truncates toward zero, as TF


def _stack(vals, axis=0, **kw):
    if isinstance(vals, np.ndarray) and not isinstance(vals, (list, tuple)):
        return t(vals)
    return t(np.stack([np.asarray(v) for v in vals], axis=axis))


def _concat(vals, axis=0, **kw):
    return t(np.concatenate([np.atleast_1d(np.asarray(t(v))) for v in vals], axis=axis))


def _split(x, num_or_size_splits, axis=0, **kw):
    x = np.asarray(x)
    if isinstance(num_or_size_splits, (int, np.integer)):
        return [t(p) for p in np.split(x, int(num_or_size_splits), axis=axis)]
    sizes = [int(s) for s in np.asarray(num_or_size_splits).reshape(-1)]
    if -1 in sizes:
        i = sizes.index(-1)
        sizes[i] = x.shape[axis] - (sum(sizes) + 1)
    idx = np.cumsum(sizes)[:-1]
    return [t(p) for p in np.split(x, idx, axis=axis)]


def _reshape(x, shape=None, **kw):
    shape = [int(s) for s in np.asarray(shape).reshape(-1)]
    return t(np.reshape(np.asarray(x), shape))


def _tile(x, multiples, **kw):
    return t(np.tile(np.asarray(x), [int(m) for m in np.asarray(multiples).reshape(-1)]))


def _gather(params, indices, axis=0, **kw):
    return t(np.take(np.asarray(params), np.asarray(indices), axis=axis))


def _range(start, limit=None, delta=1, dtype=None, **kw):
    if limit is None:
        start, limit = 0, start
    return t(np.arange(start, limit, delta), dtype=dtype)


def _unstack(x, axis=0, **kw):
    x = np.asarray(x)
    return [t(np.take(x, i, axis=axis)) for i in range(x.shape[axis])]


def _where(cond, x=None, y=None, **kw):
    if x is None:
        return t(np.argwhere(np.asarray(cond)).astype(np.int64))
    return t(np.where(np.asarray(cond), np.asarray(t(x)), np.asarray(t(y))))


def _scatter_nd(indices, updates, shape, **kw):
    shape = [int(s) for s in np.asarray(shape).reshape(-1)]
    updates = np.asarray(updates)
    out = np.zeros(shape, dtype=updates.dtype)
    idx = np.asarray(indices)
    np.add.at(out, tuple(idx[..., i] for i in range(idx.shape[-1])), updates)  # duplicates accumulate
    return t(out)


def _tensor_scatter_nd_update(tensor, indices, updates, **kw):
    out = np.array(tensor)
    idx = np.asarray(indices)
    out[tuple(idx[..., i] for i in range(idx.shape[-1]))] = np.asarray(updates)
    return t(out)


def _reverse(x, axis, **kw):
    axis = [int(a) for a in np.asarray(axis).reshape(-1)]
    return t(np.flip(np.asarray(x), axis=axis))


def _slice(x, begin, size, **kw):
    x = np.asarray(x)
    begin = [int(b) for b in np.asarray(begin).reshape(-1)]
    size = [int(s) for s in np.asarray(size).reshape(-1)]
    sl = tuple(slice(b, None if s == -1 else b + s) for b, s in zip(begin, size))
    return t(x[sl])


def _pad(x, paddings, mode='CONSTANT', constant_values=0, **kw):
    return t(np.pad(np.asarray(x), np.asarray(paddings).astype(int), constant_values=constant_values))


def _map_fn(fn, elems, dtype=None, **kw):
    if isinstance(elems, (list, tuple)):
        n = np.asarray(elems[0]).shape[0]
        outs = [fn([t(np.asarray(e)[i]) for e in elems]) for i in range(n)]
    else:
        n = np.asarray(elems).shape[0]
        outs = [fn(t(np.asarray(elems)[i])) for i in range(n)]
    if isinstance(outs[0], (list, tuple)):
        return [t(np.stack([np.asarray(o[j]) for o in outs], 0)) for j in range(len(outs[0]))]
    return t(np.stack([np.asarray(o) for o in outs], 0))


def _conv3d(x, filters, strides, padding, **kw):
    """cross-correlation with zero padding ('SAME'), float32 accumulation over taps in
    (z, y, x) raster order of the filter."""
    assert padding == 'SAME'
    x = np.asarray(x, dtype=F32)
    w = np.asarray(filters, dtype=F32)
    assert w.shape[3] == 1 and w.shape[4] == 1 and x.shape[-1] == 1
    kz, ky, kx = w.shape[:3]
    pz, py, px = kz // 2, ky // 2, kx // 2
    xp = np.pad(x[..., 0], ((0, 0), (pz, pz), (py, py), (px, px)))
    B_, Z, Y, X = x.shape[:4]
    acc = np.zeros((B_, Z, Y, X), dtype=F32)
    for a in range(kz):
        for b in range(ky):
            for c in range(kx):
                acc = acc + xp[:, a:a + Z, b:b + Y, c:c + X] * w[a, b, c, 0, 0]
    return t(acc[..., None])


def _u(fn):
    def f(x, *a, **kw):
        kw.pop('name', None)
        return t(fn(np.asarray(t(x)), *a, **kw))
    return f


def _b(fn):
    def f(x, y, *a, **kw):
        kw.pop('name', None)
        return t(fn(np.asarray(t(x)), np.asarray(t(y))))
    return f


def _reduce(fn):
    def f(x, axis=None, keepdims=False, **kw):
        r = fn(np.asarray(x), axis=_ax(axis), keepdims=keepdims)
        return t(r)
    return f


def _round(x, **kw):
    return t(np.round(np.asarray(x)))  # half-to-even, as tf.round


def _clip(x, lo, hi, **kw):
    x = np.asarray(t(x))
    lo_ = None if lo is None else np.asarray(t(lo)).astype(x.dtype)
    hi_ = None if hi is None else np.asarray(t(hi)).astype(x.dtype)
    return t(np.clip(x, lo_, hi_))


def _expand_dims(x, axis=-1, **kw):
    return t(np.expand_dims(np.asarray(t(x)), int(axis)))


def _shape(x, **kw):
    return t(np.array(np.asarray(x).shape, dtype=I32))


def _size(x, **kw):
    return int(np.asarray(x).size)


def _ones(shape, dtype='float32', **kw):
    return t(np.ones([int(s) for s in np.asarray(shape).reshape(-1)], dtype=_dt(dtype)))


def _zeros(shape, dtype='float32', **kw):
    return t(np.zeros([int(s) for s in np.asarray(shape).reshape(-1)], dtype=_dt(dtype)))


def _convert(x, dtype=None, **kw):
    return t(x, dtype=dtype)


def _eye(n, dtype='float32', **kw):
    return t(np.eye(int(n), dtype=_dt(dtype)))


def _diag(x, **kw):
    x = np.asarray(x)
    out = np.zeros(x.shape + (x.shape[-1],), dtype=x.dtype)
    i = np.arange(x.shape[-1])
    out[..., i, i] = x
    return t(out)


def _inv(x, **kw):
    # float32 inverse (TF: LU in float32).  Computed in float64 and rounded: tolerance-checked only.
    return t(np.linalg.inv(np.asarray(x, dtype=np.float64)).astype(F32))


def _transpose(x, perm=None, **kw):
    return t(np.transpose(np.asarray(x), perm))


def _squeeze(x, axis=None, **kw):
    return t(np.squeeze(np.asarray(x), axis=_ax(axis)))


def _pow(x, y, **kw):
    return t(np.power(np.asarray(t(x)), np.asarray(t(y))))


def build_modules(feed_queue):
    """returns dict name -> module to be placed in sys.modules"""
    tf = types.ModuleType('tensorflow')
    tf.Tensor = T
    tf.TensorShape = TensorShape
    tf.float32 = 'float32'
    tf.int32 = 'int32'
    tf.stack = _stack
    tf.cast = _cast
    tf.floor = _u(np.floor)
    tf.round = _round
    tf.clip_by_value = _clip
    tf.gather = _gather
    tf.reshape = _reshape
    tf.range = _range
    tf.tile = _tile
    tf.size = _size
    tf.transpose = _transpose
    tf.matmul = _matmul
    tf.unstack = _unstack
    tf.ones = _ones
    tf.zeros = _zeros
    tf.ones_like = _u(np.ones_like)
    tf.zeros_like = _u(np.zeros_like)
    tf.convert_to_tensor = _convert
    tf.is_tensor = lambda x: isinstance(x, T)
    tf.split = _split
    tf.exp = _u(np.exp)
    tf.cos = _u(np.cos)
    tf.sin = _u(np.sin)
    tf.sqrt = _u(np.sqrt)
    tf.square = _u(np.square)
    tf.abs = _u(np.abs)
    tf.reduce_sum = _reduce(np.sum)
    tf.reduce_mean = _reduce(np.mean)
    tf.reduce_max = _reduce(np.max)
    tf.reduce_min = _reduce(np.min)
    tf.expand_dims = _expand_dims
    tf.equal = _b(np.equal)
    tf.less = _b(np.less)
    tf.less_equal = _b(np.less_equal)
    tf.where = _where
    tf.concat = _concat
    tf.shape = _shape
    tf.map_fn = _map_fn
    tf.scatter_nd = _scatter_nd
    tf.tensor_scatter_nd_update = _tensor_scatter_nd_update
    tf.reverse = _reverse
    tf.slice = _slice
    tf.pad = _pad
    tf.squeeze = _squeeze
    tf.eye = _eye
    tf.sort = lambda x, axis=-1, **kw: t(np.sort(np.asarray(x), axis=axis))

    tf.nn = types.ModuleType('tensorflow.nn')
    tf.nn.conv3d = _conv3d

    tf.linalg = types.ModuleType('tensorflow.linalg')
    tf.linalg.inv = _inv
    tf.linalg.diag = _diag

    m = types.ModuleType('tensorflow.math')
    m.floormod = _b(np.mod)
    m.minimum = _b(np.minimum)
    m.maximum = _b(np.maximum)
    m.sqrt = _u(np.sqrt)
    m.square = _u(np.square)
    m.exp = _u(np.exp)
    m.log = _u(np.log)
    m.pow = _pow
    m.multiply = _b(np.multiply)
    m.ceil = _u(np.ceil)
    m.floor = _u(np.floor)
    m.reduce_sum = _reduce(np.sum)
    m.equal = _b(np.equal)
    m.abs = _u(np.abs)
    tf.math = m

    r = types.ModuleType('tensorflow.random')
    r.normal = _rand_normal
    r.uniform = _rand_uniform
    tf.random = r

    dbg = types.ModuleType('tensorflow.debugging')
    dbg.check_numerics = lambda x, msg=None: x
    tf.debugging = dbg

    # ---- keras.backend
    K = types.ModuleType('keras.backend')
    K.expand_dims = _expand_dims
    K.square = _u(np.square)
    K.abs = _u(np.abs)
    K.sum = _reduce(np.sum)
    K.mean = _reduce(np.mean)
    K.min = _reduce(np.min)
    K.max = _reduce(np.max)
    K.less = _b(np.less)
    K.clip = _clip
    K.epsilon = lambda: 1e-7
    K.reshape = _reshape
    K.reverse = lambda x, axes: _reverse(x, axes)
    K.permute_dimensions = lambda x, p: _transpose(x, p)
    K.switch = lambda c, a, b: (a if bool(np.asarray(c).reshape(-1)[0]) else b)
    K.int_shape = lambda x: tuple(np.asarray(x).shape)

    # ---- keras.layers
    class Layer:
        def __init__(self, **kwargs):
            self.built = False
            self.name = kwargs.get('name')

        def get_config(self):
            return {}

        def build(self, input_shape):
            self.built = True

        def compute_output_shape(self, s):
            return s

        def __call__(self, inputs, **kw):
            if not self.built:
                def shp(x):
                    s = list(np.asarray(x).shape)
                    s[0] = None
                    return tuple(s)
                if isinstance(inputs, (list, tuple)):
                    ishape = [shp(x) for x in inputs]
                else:
                    ishape = shp(inputs)
                self.build(ishape)
                self.built = True
            if isinstance(inputs, (list, tuple)):
                inputs = [t(x) for x in inputs]
            return self.call(inputs, **kw)

    NAMED = {}

    def Input(shape=None, name=None, dtype=None, **kw):
        nm, arr = feed_queue.pop(0)
        assert nm == name, (nm, name)
        arr = t(arr, dtype=dtype)
        assert list(arr.shape[1:]) == [int(s) for s in shape], (name, arr.shape, shape)
        return arr

    class Lambda:
        def __init__(self, fn, name=None, **kw):
            self.fn = fn
            self.name = name

        def __call__(self, x):
            out = self.fn(x)
            if self.name is not None:
                NAMED[self.name] = out
            return out

    class _Merge:   # keras.layers.Add / Subtract: elementwise on a list of two tensors
        op = None

        def __init__(self, name=None, **kw):
            self.name = name

        def __call__(self, xs):
            out = type(self).op(t(xs[0]), t(xs[1]))
            if self.name is not None:
                NAMED[self.name] = out
            return out

    class Add(_Merge):
        op = staticmethod(lambda a, b: a + b)

    class Subtract(_Merge):
        op = staticmethod(lambda a, b: a - b)

    KL = types.ModuleType('keras.layers')
    KL.Layer = Layer
    KL.Input = Input
    KL.Lambda = Lambda
    KL.Add = Add
    KL.Subtract = Subtract

    class Model:
        def __init__(self, inputs=None, outputs=None, **kw):
            self.inputs = inputs
            self.outputs = outputs
            self.output = outputs
            self.named = NAMED

    KM = types.ModuleType('keras.models')
    KM.Model = Model

    class _Auto(types.ModuleType):
        def __getattr__(self, item):
            if item.startswith('__'):
                raise AttributeError(item)
            return _Auto(self.__name__ + '.' + item)

        def __call__(self, *a, **k):
            return None

    keras = types.ModuleType('keras')
    keras.backend = K
    keras.layers = KL
    keras.models = KM
    keras.callbacks = _Auto('keras.callbacks')
    keras.optimizers = _Auto('keras.optimizers')
    keras.activations = _Auto('keras.activations')
    keras.initializers = _Auto('keras.initializers')
    keras.constraints = _Auto('keras.constraints')
    keras.regularizers = _Auto('keras.regularizers')

    mods = {'tensorflow': tf, 'tensorflow.nn': tf.nn, 'tensorflow.math': tf.math,
            'tensorflow.random': tf.random, 'tensorflow.linalg': tf.linalg,
            'keras': keras, 'keras.backend': K, 'keras.layers': KL, 'keras.models': KM,
            'keras.callbacks': keras.callbacks, 'keras.optimizers': keras.optimizers,
            'keras.activations': keras.activations, 'keras.initializers': keras.initializers,
            'keras.constraints': keras.constraints, 'keras.regularizers': keras.regularizers,
            'nibabel': _Auto('nibabel'), 'h5py': _Auto('h5py')}
    return mods, NAMED


def install(feed_queue, reference_root='/root/reference'):
    """patch numpy/scipy removed aliases, install fake modules, put the reference on sys.path"""
    np.int = int
    np.float = float
    np.bool = bool
    if not hasattr(scipy.stats, 'median_absolute_deviation'):
        # scipy 1.4.1 (the version the reference pins, requirements.txt:53): the result is `scale * MAD` with the
        # default scale 1.4826; today's median_abs_deviation defaults to scale 1.0, so the old default is restated
        def median_absolute_deviation(x, axis=0, center=np.median, scale=1.4826, nan_policy='propagate'):
            return scale * scipy.stats.median_abs_deviation(x, axis=axis, center=center, scale=1.0,
                                                            nan_policy=nan_policy)
        scipy.stats.median_absolute_deviation = median_absolute_deviation
    mods, named = build_modules(feed_queue)
    sys.modules.update(mods)
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    return named
