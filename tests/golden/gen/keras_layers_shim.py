"""numpy stand-ins for the *built-in* Keras layers the reference's network builders instantiate
(`Conv3D`, `BatchNormalization`, `MaxPooling3D`, `UpSampling3D`, `concatenate`, `Activation`, `LeakyReLU`, `Flatten`,
`Dense`, `Dropout`, `add`), plus a `Model` that supports `get_layer(name).output`, `.layers` and being *called* on new
tensors (graph replay) and a reverse-mode `K.gradients` over the recorded layer calls.

TEST INFRASTRUCTURE, development container only (see tf_numpy_shim.py).  With these, the reference's own

    ext/neuron/models.py            unet / conv_enc / conv_dec
    SynthSR/metrics_model.py        metrics_model / add_seg_loss_to_model      (+ ext/lab2im/layers.py DiceLoss)
    SynthSR/fine_tuning_with_adversary.py   make_discriminator / build_generator_loss / build_discriminator_loss

execute verbatim and eagerly; `make_unet_goldens.py` records what they compute.

What this pins and what it does not.  The WIRING is the reference's: which layers exist, their names, kernel shapes,
order, skip taps, concatenation order, BatchNorm placement, heads, loss composition.  The ARITHMETIC of each built-in
layer below is ours, written from the documented Keras 2.3.1 / TensorFlow 2.0 semantics (both third party, neither
vendored nor installable here):
  * Conv3D: cross-correlation, kernel [k,k,k,Cin,Cout], 'same' zero padding, TensorFlow's asymmetric padding for
    strides (total = max((ceil(n/s)-1)*s + k - n, 0), before = total // 2), bias, then the activation;
  * BatchNormalization(axis=-1): eps 1e-3, momentum .99; training phase = biased batch statistics over every axis but
    the last; inference = moving statistics;
  * MaxPooling3D(2, padding='same') on even sizes, UpSampling3D = nearest repeat, Flatten = C-order reshape,
    Dense = x @ kernel + bias, ELU(alpha=1), LeakyReLU(alpha), softmax.
Everything is evaluated in float64 and rounded to float32 at the layer boundary, so a golden is closer to the exact
result than either float32 implementation checked against it.
"""
import types
import numpy as np
import tf_numpy_shim as shim

T, t = shim.T, shim.t

GRAPH = []        # (layer, inputs, output) in call order; holds references so id() stays unique
LAYERS = {}       # name -> layer object (first call wins, like keras' get_layer)
PARAMS = {}       # '<layer>/<weight>' -> float32 array (what the weight provider handed out)
STATE = {'learning_phase': 1, 'bn_batch': {}, 'uid': {}, 'rng': np.random.default_rng(0)}


def reset(seed=0, learning_phase=1):
    GRAPH.clear()
    LAYERS.clear()
    PARAMS.clear()
    STATE.update(learning_phase=learning_phase, bn_batch={}, uid={}, rng=np.random.default_rng(seed),
                 frozen_bn_inference=False, dropout={})


def _uid(prefix):
    STATE['uid'][prefix] = STATE['uid'].get(prefix, 0) + 1          # keras.backend.get_uid: first is 1
    return '%s_%d' % (prefix, STATE['uid'][prefix])


def _weight(layer_name, wname, shape, kind):
    """seeded stand-in for the initialisers (the values are arbitrary test data, not Keras' defaults: non-zero biases
    and non-trivial BatchNorm parameters exercise more of the arithmetic)"""
    rng = STATE['rng']
    shape = tuple(int(s) for s in shape)
    if kind == 'kernel':
        rf = int(np.prod(shape[:-2]))
        lim = np.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))
        a = rng.uniform(-lim, lim, shape)
    elif kind == 'bias':
        a = rng.normal(0, .05, shape)
    elif kind == 'gamma':
        a = rng.uniform(.5, 1.5, shape)
    elif kind == 'beta':
        a = rng.normal(0, .1, shape)
    elif kind == 'moving_mean':
        a = rng.normal(0, .2, shape)
    elif kind == 'moving_variance':
        a = rng.uniform(.5, 2., shape)
    a = a.astype(np.float32)
    PARAMS['%s/%s' % (layer_name, wname)] = a
    return a


def _act(name):
    if name is None or name == 'linear':
        return lambda z: z, lambda z, y: np.ones_like(z)
    if name == 'elu':
        return (lambda z: np.where(z > 0, z, np.expm1(np.minimum(z, 0))),
                lambda z, y: np.where(z > 0, 1.0, np.exp(np.minimum(z, 0))))
    if name == 'relu':
        return lambda z: np.maximum(z, 0), lambda z, y: (z > 0).astype(z.dtype)
    raise NotImplementedError(name)


class _ShapeList(list):
    def as_list(self):
        return list(self)


def _patch_tensor_shape():
    """`tensor.shape.as_list()` (ext/neuron/models.py:394): numpy's tuple has no as_list; hand out a tuple subclass"""
    class _ShapeTuple(tuple):
        def as_list(self):
            return list(self)

    base = np.ndarray.shape

    def _get(self):
        return _ShapeTuple(base.__get__(self))

    def _set(self, v):
        base.__set__(self, v)
    T.shape = property(_get, _set)


class KerasLayer:
    prefix = 'layer'

    def __init__(self, name=None, **kw):
        self.name = name if name is not None else _uid(self.prefix)
        self.trainable = True
        self.built = False

    def __call__(self, x):
        xin = [t(v) for v in x] if isinstance(x, (list, tuple)) else t(x)
        if not self.built:
            self.build(xin)
            self.built = True
        out = self.forward(xin)
        out = [t(np.asarray(o, dtype=np.float32)) for o in out] if isinstance(out, list) else \
            t(np.asarray(out, dtype=np.float32))
        GRAPH.append((self, xin, out))
        if self.name not in LAYERS:
            LAYERS[self.name] = self
            self.output = out
            self.input = xin
        return out

    def build(self, x):
        pass

    def get_weights(self):
        return []


def _same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2, out


class Conv3D(KerasLayer):
    prefix = 'conv3d'

    def __init__(self, filters, kernel_size, strides=1, padding='valid', activation=None, data_format=None,
                 dilation_rate=1, name=None, **kw):
        super().__init__(name=name)
        self.k = int(kernel_size) if np.isscalar(kernel_size) else int(kernel_size[0])
        assert (padding == 'same' or self.k == 1) and data_format in (None, 'channels_last')
        assert dilation_rate in (1, (1, 1, 1))
        self.filters = int(filters)
        self.s = int(strides) if np.isscalar(strides) else int(strides[0])
        self.activation = activation

    def build(self, x):
        cin = x.shape[-1]
        self.kernel = _weight(self.name, 'kernel', (self.k,) * 3 + (cin, self.filters), 'kernel')
        self.bias = _weight(self.name, 'bias', (self.filters,), 'bias')

    def _pads(self, shape):
        return [_same_pad(n, self.k, self.s) for n in shape[1:4]]

    def forward(self, x):
        x = np.asarray(x, dtype=np.float64)
        w = self.kernel.astype(np.float64)
        pads = self._pads(x.shape)
        xp = np.pad(x, [(0, 0)] + [(p[0], p[1]) for p in pads] + [(0, 0)])
        o = [p[2] for p in pads]
        s = self.s
        z = np.zeros((x.shape[0], o[0], o[1], o[2], self.filters))
        for a in range(self.k):
            for b in range(self.k):
                for c in range(self.k):
                    sl = xp[:, a:a + (o[0] - 1) * s + 1:s, b:b + (o[1] - 1) * s + 1:s, c:c + (o[2] - 1) * s + 1:s]
                    z += sl @ w[a, b, c]
        z += self.bias.astype(np.float64)
        self._z = z
        return _act(self.activation)[0](z)

    def backward_input(self, x, dy):
        """d/dx of sum(dy * forward(x)) (used by K.gradients)"""
        x = np.asarray(x, dtype=np.float64)
        z = self._recompute_z(x)
        dz = np.asarray(dy, dtype=np.float64) * _act(self.activation)[1](z, None)
        w = self.kernel.astype(np.float64)
        pads = self._pads(x.shape)
        o = [p[2] for p in pads]
        s = self.s
        dxp = np.zeros([x.shape[0]] + [n + p[0] + p[1] for n, p in zip(x.shape[1:4], pads)] + [x.shape[-1]])
        for a in range(self.k):
            for b in range(self.k):
                for c in range(self.k):
                    dxp[:, a:a + (o[0] - 1) * s + 1:s, b:b + (o[1] - 1) * s + 1:s, c:c + (o[2] - 1) * s + 1:s] += \
                        dz @ w[a, b, c].T
        sl = tuple(slice(p[0], p[0] + n) for n, p in zip(x.shape[1:4], pads))
        return dxp[(slice(None),) + sl]

    def _recompute_z(self, x):
        self.forward(x)
        return self._z


class BatchNormalization(KerasLayer):
    prefix = 'batch_normalization'

    def __init__(self, axis=-1, momentum=.99, epsilon=1e-3, name=None, **kw):
        super().__init__(name=name)
        assert axis == -1
        self.eps = epsilon
        self.momentum = momentum

    def build(self, x):
        c = x.shape[-1]
        self.gamma = _weight(self.name, 'gamma', (c,), 'gamma')
        self.beta = _weight(self.name, 'beta', (c,), 'beta')
        self.moving_mean = _weight(self.name, 'moving_mean', (c,), 'moving_mean')
        self.moving_variance = _weight(self.name, 'moving_variance', (c,), 'moving_variance')

    def forward(self, x):
        x = np.asarray(x, dtype=np.float64)
        frozen_inference = STATE.get('frozen_bn_inference', False) and not self.trainable
        if STATE['learning_phase'] and not frozen_inference:
            ax = tuple(range(x.ndim - 1))
            mean, var = x.mean(ax), x.var(ax)        # biased variance
            STATE['bn_batch'].setdefault(self.name, (mean.astype(np.float32), var.astype(np.float32)))
        else:
            mean, var = self.moving_mean.astype(np.float64), self.moving_variance.astype(np.float64)
        return (x - mean) / np.sqrt(var + self.eps) * self.gamma + self.beta


class MaxPooling3D(KerasLayer):
    prefix = 'max_pooling3d'

    def __init__(self, pool_size=2, strides=None, padding='valid', name=None, **kw):
        super().__init__(name=name)
        self.p = tuple(int(v) for v in (pool_size if not np.isscalar(pool_size) else (pool_size,) * 3))

    def forward(self, x):
        x = np.asarray(x)
        b, d0, d1, d2, c = x.shape
        p = self.p
        assert d0 % p[0] == 0 and d1 % p[1] == 0 and d2 % p[2] == 0      # 'same' == 'valid' on divisible sizes
        return x.reshape(b, d0 // p[0], p[0], d1 // p[1], p[1], d2 // p[2], p[2], c).max((2, 4, 6))


class UpSampling3D(KerasLayer):
    prefix = 'up_sampling3d'

    def __init__(self, size=2, name=None, **kw):
        super().__init__(name=name)
        self.size = tuple(int(v) for v in (size if not np.isscalar(size) else (size,) * 3))

    def forward(self, x):
        x = np.asarray(x)
        for ax, r in enumerate(self.size):
            x = np.repeat(x, r, axis=ax + 1)
        return x


class Concatenate(KerasLayer):
    prefix = 'concatenate'

    def __init__(self, axis=-1, name=None, **kw):
        super().__init__(name=name)
        self.axis = axis

    def forward(self, xs):
        return np.concatenate([np.asarray(v) for v in xs], axis=self.axis)


class Add(KerasLayer):
    prefix = 'add'

    def forward(self, xs):
        return sum(np.asarray(v, dtype=np.float64) for v in xs)


class Subtract(KerasLayer):
    prefix = 'subtract'

    def forward(self, xs):
        return np.asarray(xs[0], dtype=np.float64) - np.asarray(xs[1], dtype=np.float64)


class Activation(KerasLayer):
    prefix = 'activation'

    def __init__(self, activation, name=None, **kw):
        super().__init__(name=name)
        self.activation = activation

    def forward(self, x):
        return _act(self.activation)[0](np.asarray(x, dtype=np.float64))

    def backward_input(self, x, dy):
        return dy * _act(self.activation)[1](np.asarray(x, dtype=np.float64), None)


class LeakyReLU(KerasLayer):
    prefix = 'leaky_re_lu'

    def __init__(self, alpha=.3, name=None, **kw):
        super().__init__(name=name)
        self.alpha = float(np.float32(alpha))

    def forward(self, x):
        x = np.asarray(x, dtype=np.float64)
        return np.where(x > 0, x, self.alpha * x)

    def backward_input(self, x, dy):
        return np.asarray(dy) * np.where(np.asarray(x) > 0, 1.0, self.alpha)


class Flatten(KerasLayer):
    prefix = 'flatten'

    def __init__(self, data_format=None, name=None, **kw):
        super().__init__(name=name)
        assert data_format in (None, 'channels_last')

    def forward(self, x):
        x = np.asarray(x)
        return x.reshape(x.shape[0], -1)

    def backward_input(self, x, dy):
        return np.asarray(dy).reshape(np.asarray(x).shape)


class Dense(KerasLayer):
    prefix = 'dense'

    def __init__(self, units, activation=None, name=None, **kw):
        super().__init__(name=name)
        self.units = int(units)
        self.activation = activation

    def build(self, x):
        self.kernel = _weight(self.name, 'kernel', (x.shape[-1], self.units), 'kernel')
        self.bias = _weight(self.name, 'bias', (self.units,), 'bias')

    def forward(self, x):
        z = np.asarray(x, dtype=np.float64) @ self.kernel.astype(np.float64) + self.bias
        return _act(self.activation)[0](z)

    def backward_input(self, x, dy):
        z = np.asarray(x, dtype=np.float64) @ self.kernel.astype(np.float64) + self.bias
        return (np.asarray(dy) * _act(self.activation)[1](z, None)) @ self.kernel.astype(np.float64).T


class Dropout(KerasLayer):
    """keras.layers.Dropout(rate, noise_shape): in the learning phase x * keep / (1 - rate) with a keep mask of shape
    noise_shape (None -> the tensor's own extent), identity otherwise (documented Keras semantics, backend.dropout ->
    tf.nn.dropout).  The mask is drawn from a generator seeded with the layer name; the per-feature factor is recorded in
    STATE['dropout'][name] so that a golden can hand the same factors to the code under test."""
    prefix = 'dropout'

    def __init__(self, rate, noise_shape=None, name=None, **kw):
        super().__init__(name=name)
        self.rate = float(rate)
        self.noise_shape = noise_shape

    def forward(self, x):
        x = np.asarray(x)
        if self.rate == 0 or not STATE['learning_phase']:
            return x
        import zlib
        shape = list(x.shape) if self.noise_shape is None else [x.shape[i] if d is None else int(d)
                                                                 for i, d in enumerate(self.noise_shape)]
        rng = np.random.default_rng(zlib.crc32(self.name.encode()))
        keep = rng.random(shape) >= self.rate
        scale = (keep / (1.0 - self.rate)).astype(np.float32)
        STATE.setdefault('dropout', {})[self.name] = scale
        return x * scale


class Lambda:
    """records the call so that Model replay / K.gradients can walk through it"""

    def __init__(self, fn, name=None, **kw):
        self.fn = fn
        self.name = name
        self.trainable = True

    def __call__(self, x):
        out = self.fn(x)
        GRAPH.append((self, x, out))
        if self.name is not None and self.name not in LAYERS:
            LAYERS[self.name] = self
            self.output = out
        return out

    def forward(self, x):
        return self.fn(x)

    def get_weights(self):
        return []


def _aslist(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


class Model:
    """holder of inputs / outputs over the eagerly evaluated graph; calling it on new tensors replays the recorded
    layer calls that lie between its inputs and its outputs (shared weights, like keras)"""

    def __init__(self, inputs=None, outputs=None, name=None, **kw):
        self.inputs = _aslist(inputs)
        self.outputs = _aslist(outputs)
        self.input = self.inputs[0] if len(self.inputs) == 1 else self.inputs
        self.output = self.outputs[0] if len(self.outputs) == 1 else self.outputs
        self.name = name
        self.trainable = True
        self._records = self._subgraph()

    def _subgraph(self):
        """the recorded calls that lie between the inputs and the outputs (what keras collects into model.layers)"""
        stop = {id(v) for v in self.inputs}
        need = {id(v) for v in self.outputs} - stop
        recs = []
        for rec in reversed(GRAPH):
            layer, ins, out = rec
            if not any(id(o) in need for o in _aslist(out)):
                continue
            recs.append(rec)
            for v in _aslist(ins):
                if id(v) not in stop:
                    need.add(id(v))
        return recs[::-1]

    @property
    def layers(self):
        seen, out = set(), []
        for layer, _, _ in self._records:
            if id(layer) not in seen:
                seen.add(id(layer))
                out.append(layer)
        return out

    def get_layer(self, name):
        for layer, _, out in self._records:
            if getattr(layer, 'name', None) == name:
                return types.SimpleNamespace(output=out, name=name, layer=layer)
        raise ValueError('No such layer: ' + name)

    def __call__(self, new_inputs):
        env = {id(o): n for o, n in zip(self.inputs, _aslist(new_inputs))}
        want = [id(o) for o in self.outputs]
        for layer, ins, out in self._records:
            if all(w in env for w in want):
                break
            lst = _aslist(ins)
            if not all(id(v) in env for v in lst):
                continue
            new_in = [env[id(v)] for v in lst]
            new_in = new_in if isinstance(ins, (list, tuple)) else new_in[0]
            new_out = layer.forward(new_in)
            if isinstance(layer, KerasLayer):
                new_out = t(np.asarray(new_out, dtype=np.float32))
            GRAPH.append((layer, new_in, new_out))
            for o, n in zip(_aslist(out), _aslist(new_out)):
                env[id(o)] = n
        res = [env[w] for w in want]
        return res[0] if len(res) == 1 else res


def gradients(y, xs):
    """K.gradients(y, x): reverse sweep over the recorded calls from y back to x.  Supports the layers of
    make_discriminator and element-wise Lambdas of the form x[0] * cast(x[1]) (the mask product)."""
    xs_l = _aslist(xs)
    grads = {id(y): np.ones(np.asarray(y).shape)}
    stop = {id(x) for x in xs_l}
    for layer, ins, out in reversed(GRAPH):
        if isinstance(out, (list, tuple)) or id(out) not in grads or id(out) in stop:
            continue
        dy = grads[id(out)]
        if isinstance(ins, (list, tuple)):
            if isinstance(layer, Lambda) and len(ins) == 2:         # x[0] * cast(x[1]): gradient w.r.t. x[0]
                probe = np.asarray(layer.fn([t(np.ones_like(np.asarray(ins[0]))), ins[1]]), dtype=np.float64)
                dx = dy * probe
                grads[id(ins[0])] = grads.get(id(ins[0]), 0) + dx
                continue
            raise NotImplementedError(type(layer))
        dx = layer.backward_input(ins, dy)
        grads[id(ins)] = grads.get(id(ins), 0) + dx
    return [t(np.asarray(grads[id(x)], dtype=np.float32)) for x in xs_l]


def install(feed_queue, reference_root='/root/reference'):
    """tf_numpy_shim.install + the built-in Keras layers"""
    import sys
    named = shim.install(feed_queue, reference_root)
    _patch_tensor_shape()
    KL = sys.modules['keras.layers']
    KM = sys.modules['keras.models']
    K = sys.modules['keras.backend']
    keras = sys.modules['keras']
    tf = sys.modules['tensorflow']

    base_layer = KL.Layer
    base_call = base_layer.__call__

    def layer_call(self, inputs, **kw):                 # reference Layer subclasses: record by name, too
        out = base_call(self, inputs, **kw)
        GRAPH.append((self, inputs, out))
        nm = getattr(self, 'name', None)
        if nm is not None and nm not in LAYERS:
            LAYERS[nm] = self
            self.output = out
        return out
    base_layer.__call__ = layer_call
    base_layer.forward = lambda self, x: self.call([t(v) for v in x] if isinstance(x, (list, tuple)) else x)
    base_layer.trainable = True

    def Input(shape=None, name=None, dtype=None, **kw):
        nm, arr = feed_queue.pop(0)
        assert nm == name, (nm, name)
        arr = t(arr, dtype=dtype)
        assert list(arr.shape[1:]) == [int(s) for s in shape], (name, arr.shape, shape)
        return arr
    KL.Input = Input
    for cls in (Conv3D, BatchNormalization, MaxPooling3D, UpSampling3D, Activation, LeakyReLU, Flatten, Dense, Dropout,
                Add, Subtract):
        setattr(KL, cls.__name__, cls)
    KL.Lambda = Lambda
    KL.AvgPool3D = None          # only looked up by DiceLoss.build (boundary weighting, unused here)
    KL.Concatenate = Concatenate
    KL.concatenate = lambda xs, axis=-1, name=None, **kw: Concatenate(axis=axis, name=name)(xs)
    KL.add = lambda xs, name=None, **kw: Add(name=name)(xs)
    KM.Model = Model
    keras.models = KM
    K.gradients = gradients
    K.sqrt = shim._u(np.sqrt)

    act = types.ModuleType('keras.activations')

    def softmax(x, axis=-1):
        x = np.asarray(x, dtype=np.float64)
        e = np.exp(x - x.max(axis=axis, keepdims=True))
        return t((e / e.sum(axis=axis, keepdims=True)).astype(np.float32))
    act.softmax = softmax
    keras.activations = act
    sys.modules['keras.activations'] = act
    keras.layers = KL

    tf.math.reduce_mean = tf.reduce_mean
    tf.math.reduce_max = tf.reduce_max
    tfk = types.ModuleType('tensorflow.keras')
    tfk.backend = K
    tf.keras = tfk
    return named
