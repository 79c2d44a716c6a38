"""WGAN-GP critic of the adversarial fine-tuning (SynthSR/fine_tuning_with_adversary.py:482-508 `make_discriminator`,
:579-595 `build_discriminator_loss`, :604-642 `RandomWeightedAverage` / `Gradients`) on the HIP kernels.

Network: n_levels x [Conv3D(f, 3, stride 1) + LeakyReLU(.2), Conv3D(f, 3, stride 2) + LeakyReLU(.2)], f = n_filters 2^level,
Flatten (channels last), Dense(n_filters 2^n_levels) + LeakyReLU(.2), Dense(1).  Strided layers run on the parity kernels of
the folded decoder conv (ops.conv3d_stride2*: a stride-2 conv is a sum of eight 2x2x2-window convs on the parity
sub-lattices of its input).

Critic loss on (real, fake):  -D(real) + D(fake) + lambda (1 - ||grad_x D(x_hat)||_2)^2,  x_hat = w real + (1 - w) fake.
The network is piecewise linear, so the gradient of the penalty w.r.t. the weights needs no second derivatives of the
activations: with delta_l the back-propagated signals of grad_x D and u_0 = d penalty / d grad_x D, a "masked forward" pass
u_l = mask_l * conv_l(u_{l-1}) gives  d penalty / d W_l = weight-gradient(u_{l-1}, delta_l)  (the backward pass is linear in
every W_l); biases get no penalty gradient.

float32: the critic's channel counts (32 ... 256) are not multiples of 24, so every convolution takes the generic (CK = 8)
kernels.  dtype='bf16' ("mixed bf16", BASELINE.json configs[4]): the conv stack runs on csrc/conv_bf16.hip, whose
32-channel K-chunks fit these layers exactly -- bf16 activations / gradient signals / packed weights, fp32 accumulation,
fp32 master weights, gradients, Dense layers, loss and penalty norm; LeakyReLU and its backward are conv epilogues; a
stride-2 layer is the stride-1 conv sampled at the odd output positions (ops.subsample_odd_bf16), its backward goes through
the zero-inserted full-resolution signal."""
import numpy as np
import torch

from . import ops

ALPHA = 0.2


class Critic3D:
    def __init__(self, input_shape, n_filters=32, n_levels=4, device=None, seed=0, name='discriminator', dtype='f32'):
        if dtype not in ('f32', 'bf16'):
            raise ValueError("dtype should be 'f32' or 'bf16'")
        self.bf16 = dtype == 'bf16'
        if len(input_shape) != 4:
            raise NotImplementedError('3-D volumes only')
        shape = [int(s) for s in input_shape[:3]]
        if any(s % (2 ** n_levels) for s in shape):
            raise ValueError('spatial shape %s must be divisible by 2**n_levels = %d' % (shape, 2 ** n_levels))
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.input_shape, self.name = shape + [int(input_shape[3])], name
        self.convs, self.specs = [], []
        cin = self.input_shape[3]
        for level in range(n_levels):
            f = n_filters * 2 ** level
            for stride in (1, 2):
                i = len(self.convs)
                self.convs.append(dict(cin=cin, cout=f, stride=stride, shape=list(shape),
                                       w=self._add('%s_conv_%d/kernel' % (name, i), (3, 3, 3, cin, f)),
                                       b=self._add('%s_conv_%d/bias' % (name, i), (f,))))
                cin = f
                if stride == 2:
                    shape = [s // 2 for s in shape]
        self.flat_shape = shape + [cin]
        n_flat, n_dense = int(np.prod(self.flat_shape)), n_filters * 2 ** n_levels
        self.dense = [dict(n_in=n_flat, n_out=n_dense, w=self._add('%s_dense_0/kernel' % name, (n_flat, n_dense)),
                           b=self._add('%s_dense_0/bias' % name, (n_dense,))),
                      dict(n_in=n_dense, n_out=1, w=self._add('%s_dense_1/kernel' % name, (n_dense, 1)),
                           b=self._add('%s_dense_1/bias' % name, (1,)))]
        self.n_params = sum(int(np.prod(s)) for _, s in self.specs)
        self.offsets, off = {}, 0
        for nm, shp in self.specs:
            self.offsets[nm] = (off, shp)
            off += int(np.prod(shp))
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros_like(self.params)
        self.adam_m, self.adam_v = torch.zeros_like(self.params), torch.zeros_like(self.params)
        self.iterations = 0
        self._bufs = {}
        g = torch.Generator().manual_seed(int(seed))
        for nm, shp in self.specs:   # Keras defaults: glorot_uniform kernels, zero biases
            if nm.endswith('/kernel'):
                rf = int(np.prod(shp[:-2])) if len(shp) > 2 else 1
                limit = float(np.sqrt(6.0 / (rf * shp[-2] + rf * shp[-1])))
                self.view(nm).copy_(((torch.rand(shp, generator=g) * 2 - 1) * limit).to(self.device))
        self.repack()

    # ------------------------------------------------------------------ bookkeeping
    def _add(self, name, shape):
        self.specs.append((name, tuple(shape)))
        return name

    def view(self, name, buf=None):
        off, shp = self.offsets[name]
        return (self.params if buf is None else buf)[off:off + int(np.prod(shp))].view(*shp)

    def buf(self, key, shape, dtype=torch.float32):
        n = int(np.prod(shape))
        key = (key, dtype)
        t = self._bufs.get(key)
        if t is None or t.numel() < n:
            t = torch.empty(n, dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t[:n].view(*shape)

    def repack(self):
        epoch = ops.conv_layout_epoch()
        if getattr(self, '_pack_epoch', epoch) != epoch:   # arithmetic / plan options changed: the old buffers belong to
            for c in self.convs:                           # another layout (size and fragment order), pack into fresh ones
                c.pop('wp', None)
                c.pop('wpd', None)
        self._pack_epoch = epoch
        if self.bf16:
            for c in self.convs:
                c['wp'] = ops.pack_conv_weights_bf16(self.view(c['w']), 0, out=c.get('wp'))
                c['wpd'] = ops.pack_conv_weights_bf16(self.view(c['w']), 1, out=c.get('wpd'))
            return
        for c in self.convs:
            if c['stride'] == 2:   # parity weight sets on the OUTPUT grid (ops.conv3d_stride2*)
                lo = [v // 2 for v in c['shape']]
                c['wp'] = ops.pack_stride2_weights(self.view(c['w']), lo, 0, out=c.get('wp'))
                c['wpd'] = ops.pack_stride2_weights(self.view(c['w']), lo, 1, out=c.get('wpd'))
            else:
                c['wp'] = ops.pack_conv_weights(self.view(c['w']), c['shape'], 0, out=c.get('wp'))
                c['wpd'] = ops.pack_conv_weights(self.view(c['w']), c['shape'], 1, out=c.get('wpd'))

    def _conv(self, c, x, out, bias=True):
        """conv layer c on x; the stride-1 kernel adds the bias itself, the parity kernels of a stride-2 layer do not"""
        if c['stride'] == 2:
            return ops.conv3d_stride2(x, c['wp'], c['cout'], out=out)
        return ops.conv3d(x, c['wp'], self.view(c['b']) if bias else None, c['cout'], 0, out=out)

    def _out_shape(self, c):
        return [v // c['stride'] for v in c['shape']] + [c['cout']]

    def _wgrad(self, c, x, delta, bias):
        """self.grads += weight (and bias) gradient of conv layer c for input x and output gradient delta"""
        G = self.grads
        dbias = self.view(c['b'], G) if bias else None
        if c['stride'] == 2:
            ops.conv3d_stride2_wgrad(x, delta, self.view(c['w'], G), self.buf('dwc', [8, 27, c['cout'], c['cin']]),
                                     dbias=dbias)
        else:
            ops.conv3d_wgrad(x, delta, self.view(c['w'], G), dbias=dbias)

    def state_dict(self):
        return {nm: self.view(nm).detach().cpu().clone() for nm, _ in self.specs}

    def load_state_dict(self, sd):
        for nm, _ in self.specs:
            self.view(nm).copy_(torch.as_tensor(sd[nm]).to(self.device).reshape(self.view(nm).shape))
        self.repack()

    # ------------------------------------------------------------------ forward / backward
    def forward(self, x, tag='a'):
        """x [d0,d1,d2,C] -> D(x) as a 1-element device tensor; the LeakyReLU outputs are kept under `tag`"""
        if self.bf16:
            return self._forward_bf16(x, tag)
        ops.check_layout_epoch(getattr(self, '_pack_epoch', None), 'Critic3D.forward')
        hs = [x]
        cur = x
        for i, c in enumerate(self.convs):
            cur = self._conv(c, cur, self.buf('h%s%d' % (tag, i), self._out_shape(c)))
            if c['stride'] == 2:
                ops.bias_leaky_relu(cur, self.view(c['b']), ALPHA)
            else:
                ops.leaky_relu(cur, ALPHA)
            hs.append(cur)
        d0, d1 = self.dense
        h9 = ops.dense_fwd(cur.reshape(-1), self.view(d0['w']), self.view(d0['b']), out=self.buf('h%s_d' % tag, [d0['n_out']]))
        ops.leaky_relu(h9, ALPHA)
        out = ops.dense_fwd(h9, self.view(d1['w']), self.view(d1['b']), out=self.buf('out' + tag, [1]))
        self._saved = dict(hs=hs, h9=h9, tag=tag)
        self.last_output = out
        return out

    def backward(self, dout=1.0, weight_grads=True, input_grad=False, keep_deltas=False):
        """back-propagates dD = dout through the pass stored by the last forward(): accumulates the weight gradients into
        self.grads (weight_grads), returns grad_x D * dout (input_grad), keeps the per-layer signals for the penalty"""
        if self.bf16:
            return self._backward_bf16(dout, weight_grads, input_grad, keep_deltas)
        ops.check_layout_epoch(getattr(self, '_pack_epoch', None), 'Critic3D.backward')
        hs, h9 = self._saved['hs'], self._saved['h9']
        d0, d1 = self.dense
        G = self.grads
        dD = self.buf('dD', [1])
        dD.fill_(float(dout))
        dh9 = self.buf('dh9', [d1['n_in']])
        ops.dense_bwd(h9, self.view(d1['w']), dD, dx=dh9, dW=self.view(d1['w'], G) if weight_grads else None)
        delta9 = ops.leaky_relu_bwd(dh9, h9, ALPHA, out=self.buf('delta9', [d1['n_in']]))
        flat = hs[-1].reshape(-1)
        dflat = self.buf('dflat', [d0['n_in']])
        ops.dense_bwd(flat, self.view(d0['w']), delta9, dx=dflat, dW=self.view(d0['w'], G) if weight_grads else None)
        if weight_grads:
            self.view(d1['b'], G).add_(dD)
            self.view(d0['b'], G).add_(delta9)
        deltas = [None] * len(self.convs)
        delta = ops.leaky_relu_bwd(dflat.view(*hs[-1].shape), hs[-1], ALPHA,
                                   out=self.buf('delta%d' % (len(self.convs) - 1), list(hs[-1].shape)))
        g = None
        for i in range(len(self.convs) - 1, -1, -1):
            c = self.convs[i]
            deltas[i] = delta
            if weight_grads:
                self._wgrad(c, hs[i], delta, bias=True)
            if i > 0 or input_grad:
                gbuf = self.buf('g%d' % (i & 1), c['shape'] + [c['cin']])
                g = ops.conv3d_stride2_dgrad(delta, c['wpd'], c['cin'], out=gbuf) if c['stride'] == 2 else \
                    ops.conv3d(delta, c['wpd'], None, c['cin'], 0, out=gbuf)
                if i > 0:
                    delta = ops.leaky_relu_bwd(g, hs[i], ALPHA, out=self.buf('delta%d' % (i - 1), list(hs[i].shape)))
        if keep_deltas:
            self._deltas, self._delta9 = deltas, delta9
        return g if input_grad else None

    def input_gradient(self, x, dout=1.0, mask=None):
        """dout * grad_x D(x [* mask]) (critic frozen): the adversarial term of the generator loss"""
        self.forward(x if mask is None else ops.mul(x, mask, out=self.buf('xm', list(x.shape))), tag='g')
        g = self.backward(dout, weight_grads=False, input_grad=True)
        return g if mask is None else ops.mul(g, mask, out=g)

    # ------------------------------------------------------------------ WGAN-GP
    def critic_loss_and_grads(self, real, fake, u_mix, gp_weight=10.0, mask=None, accumulate=False):
        """loss = -D(real) + D(fake) + gp_weight (1 - ||grad D(x_hat)||)^2 with x_hat = u real + (1 - u) fake
        (build_discriminator_loss, ONE sample); self.grads = its gradient (accumulate=True: += , the further samples of a
        batch -- the caller divides by the batch size).  mask (optional, `labels_to_mask`): the critic
        sees x * mask (make_discriminator(mask_input=True)); the penalty's gradient is the one w.r.t. x_hat itself.
        Returns (loss, D(real), D(fake), ||grad||)"""
        if not accumulate:
            self.grads.zero_()
        if mask is not None:
            real = ops.mul(real, mask, out=self.buf('real_m', list(real.shape)))
            fake = ops.mul(fake, mask, out=self.buf('fake_m', list(fake.shape)))
        d_real = self.forward(real, 'a').clone()
        self.backward(-1.0)
        d_fake = self.forward(fake, 'a').clone()
        self.backward(+1.0)
        x_hat = ops.axpby(real, fake, float(u_mix), 1.0 - float(u_mix), out=self.buf('x_hat', list(real.shape)))
        self.forward(x_hat, 'p')
        g0 = self.backward(1.0, weight_grads=False, input_grad=True, keep_deltas=True)
        if mask is not None:
            g0 = ops.mul(g0, mask, out=g0)
        nsq = self.buf('nsq', [1])
        nsq.zero_()
        ops.sumsq(g0, nsq)
        norm = float(torch.sqrt(nsq).item())          # one host sync per critic step, like Keras' train_on_batch return
        penalty = gp_weight * (1.0 - norm) ** 2
        if norm > 0:
            self._penalty_backward(g0, gp_weight * 2.0 * (norm - 1.0) / norm, mask)
        loss = -float(d_real.item()) + float(d_fake.item()) + penalty
        return loss, float(d_real.item()), float(d_fake.item()), norm

    def _penalty_backward(self, g0, scale, mask=None):
        """adds d penalty / d W to self.grads: masked forward pass of u_0 = scale * grad_x D(x_hat)"""
        if self.bf16:
            return self._penalty_backward_bf16(g0, scale, mask)
        hs, h9 = self._saved['hs'], self._saved['h9']
        G = self.grads
        u = ops.axpby(g0, None, scale, 0.0, out=self.buf('u0', list(g0.shape)))
        if mask is not None:   # d(x_hat * mask)/d x_hat
            ops.mul(u, mask, out=u)
        for i, c in enumerate(self.convs):
            self._wgrad(c, u, self._deltas[i], bias=False)
            v = self._conv(c, u, self.buf('v%d' % (i & 1), self._out_shape(c)), bias=False)
            u = ops.leaky_relu_bwd(v, hs[i + 1], ALPHA, out=self.buf('u%d' % ((i + 1) & 1), list(hs[i + 1].shape)))
        d0, d1 = self.dense
        uflat = u.reshape(-1)
        ops.dense_bwd(uflat, self.view(d0['w']), self._delta9, dx=None, dW=self.view(d0['w'], G))
        v9 = ops.dense_fwd(uflat, self.view(d0['w']), None, out=self.buf('v9', [d0['n_out']]))
        self.view(d1['w'], G).view(-1).add_(ops.leaky_relu_bwd(v9, h9, ALPHA, out=self.buf('m9v9', [d0['n_out']])))

    # ------------------------------------------------------------------ bf16 conv stack (dtype='bf16')
    BF = torch.bfloat16

    def _to_bf16_input(self, x, key):
        """fp32 [d0,d1,d2,C] -> bf16 with the channels zero-padded to a multiple of 8 (16-byte K-groups of the MFMA)"""
        C = int(x.shape[-1])
        Cp = (C + 7) // 8 * 8
        return ops.to_bf16_pad(x.contiguous(), Cp, out=self.buf(key, list(x.shape[:3]) + [Cp], self.BF))

    def _conv_bf16(self, c, x, key, act, below=None, bias=True):
        """layer c on bf16 x: act 3 = + bias, LeakyReLU; act 4 = * LeakyReLU'(below); act 0 linear.  A stride-2 layer is
        evaluated at full resolution and sampled at the odd positions (TensorFlow's (0, 1) 'same' padding); with act 4 the
        mask `below` lives on the low-resolution grid and is applied by the sampling kernel"""
        b = self.view(c['b']) if bias else None
        if c['stride'] == 1:
            return ops.conv3d_bf16(x, c['wp'], b, c['cout'], act, below=below, alpha=ALPHA,
                                   out=self.buf(key, c['shape'] + [c['cout']], self.BF))
        full = ops.conv3d_bf16(x, c['wp'], b, c['cout'], 3 if act == 3 else 0, alpha=ALPHA,
                               out=self.buf('full', c['shape'] + [c['cout']], self.BF))
        return ops.subsample_odd_bf16(full, below=below if act == 4 else None, alpha=ALPHA,
                                      out=self.buf(key, self._out_shape(c), self.BF))

    def _full_delta(self, c, delta):
        """the gradient signal of layer c on the grid its stride-1 conv runs on"""
        if c['stride'] == 1:
            return delta
        return ops.zero_insert_odd_bf16(delta, out=self.buf('dfull', c['shape'] + [c['cout']], self.BF))

    def _forward_bf16(self, x, tag):
        hs = [self._to_bf16_input(x, 'x8' + tag)]
        cur = hs[0]
        for i, c in enumerate(self.convs):
            cur = self._conv_bf16(c, cur, 'h%s%d' % (tag, i), 3)
            hs.append(cur)
        d0, d1 = self.dense
        flat = self.buf('flat32' + tag, [d0['n_in']])
        flat.copy_(cur.reshape(-1))                                        # bf16 -> fp32 for the Dense layers
        h9 = ops.dense_fwd(flat, self.view(d0['w']), self.view(d0['b']), out=self.buf('h%s_d' % tag, [d0['n_out']]))
        ops.leaky_relu(h9, ALPHA)
        out = ops.dense_fwd(h9, self.view(d1['w']), self.view(d1['b']), out=self.buf('out' + tag, [1]))
        self._saved = dict(hs=hs, h9=h9, tag=tag, flat=flat)
        self.last_output = out
        return out

    def _backward_bf16(self, dout, weight_grads, input_grad, keep_deltas):
        hs, h9, flat = self._saved['hs'], self._saved['h9'], self._saved['flat']
        d0, d1 = self.dense
        G = self.grads
        dD = self.buf('dD', [1])
        dD.fill_(float(dout))
        dh9 = self.buf('dh9', [d1['n_in']])
        ops.dense_bwd(h9, self.view(d1['w']), dD, dx=dh9, dW=self.view(d1['w'], G) if weight_grads else None)
        delta9 = ops.leaky_relu_bwd(dh9, h9, ALPHA, out=self.buf('delta9', [d1['n_in']]))
        dflat = self.buf('dflat', [d0['n_in']])
        ops.dense_bwd(flat, self.view(d0['w']), delta9, dx=dflat, dW=self.view(d0['w'], G) if weight_grads else None)
        if weight_grads:
            self.view(d1['b'], G).add_(dD)
            self.view(d0['b'], G).add_(delta9)
        last = hs[-1]
        delta = self.buf('delta%d' % (len(self.convs) - 1), list(last.shape), self.BF)
        torch.mul(dflat.view(*last.shape), torch.where(last > 0, 1.0, ALPHA), out=self.buf('dlast32', list(last.shape)))
        delta.copy_(self.buf('dlast32', list(last.shape)))
        deltas = [None] * len(self.convs)
        g = None
        for i in range(len(self.convs) - 1, -1, -1):
            c = self.convs[i]
            deltas[i] = delta
            dfull = self._full_delta(c, delta) if (weight_grads or i > 0 or input_grad) else None
            if weight_grads:
                ops.conv3d_wgrad_bf16(hs[i], dfull, self.view(c['w'], G), self.view(c['b'], G))
            if i > 0:      # data gradient fused with the LeakyReLU backward of the layer below -> its delta
                delta = ops.conv3d_bf16(dfull, c['wpd'], None, hs[i].shape[3], 4, below=hs[i], alpha=ALPHA,
                                        out=self.buf('delta%d' % (i - 1), list(hs[i].shape), self.BF))
            elif input_grad:
                g8 = ops.conv3d_bf16(dfull, c['wpd'], None, hs[0].shape[3], 0, out=self.buf('g8', list(hs[0].shape), self.BF))
                g = self.buf('g32', c['shape'] + [c['cin']])
                g.copy_(g8[..., :c['cin']])
        if keep_deltas:
            self._deltas, self._delta9 = deltas, delta9
        return g if input_grad else None

    def _penalty_backward_bf16(self, g0, scale, mask):
        hs, h9 = self._saved['hs'], self._saved['h9']
        G = self.grads
        u32 = ops.axpby(g0, None, scale, 0.0, out=self.buf('u0', list(g0.shape)))
        if mask is not None:
            ops.mul(u32, mask, out=u32)
        u = self._to_bf16_input(u32, 'u8')
        for i, c in enumerate(self.convs):
            ops.conv3d_wgrad_bf16(u, self._full_delta(c, self._deltas[i]), self.view(c['w'], G), None)
            u = self._conv_bf16(c, u, 'u%d' % (i & 1), 4, below=hs[i + 1], bias=False)
        d0, d1 = self.dense
        uflat = self.buf('uflat32', [d0['n_in']])
        uflat.copy_(u.reshape(-1))
        ops.dense_bwd(uflat, self.view(d0['w']), self._delta9, dx=None, dW=self.view(d0['w'], G))
        v9 = ops.dense_fwd(uflat, self.view(d0['w']), None, out=self.buf('v9', [d0['n_out']]))
        self.view(d1['w'], G).view(-1).add_(ops.leaky_relu_bwd(v9, h9, ALPHA, out=self.buf('m9v9', [d0['n_out']])))

    def adam_step(self, lr=1e-4, decay=0.0, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        """keras.optimizers.Adam (2.3.1) update of the critic, then re-packs the conv weights; grad_scale = 1 / world
        size when self.grads holds the all-reduced sum of a data-parallel step"""
        if decay > 0:
            lr = lr * (1.0 / (1.0 + decay * self.iterations))
        self.iterations += 1
        t = self.iterations
        lr_t = lr * (np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t))
        ops.adam_step(self.params, self.grads, self.adam_m, self.adam_v, lr_t, beta1, beta2, eps, grad_scale)
        self.repack()
