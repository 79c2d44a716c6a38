"""3-D U-Net of ext/neuron/models.py:26-145 (`unet` = `conv_enc` :256-360 + `conv_dec` :363-498) as SynthSR
builds it (SynthSR/training.py:330-341): per level Conv3D(3^3,'same')+ELU x nb_conv_per_level ->
BatchNormalization(axis=-1) -> MaxPooling3D(2); decoder UpSampling3D(2) -> concatenate([skip, up]) with
skip = PRE-BN output of conv_downarm_{l}_{last} -> convs -> BN; head Conv3D(nb_labels, 1x1x1), linear.

MI355X-native: explicit forward/backward over hand-written HIP kernels (ops.py), no autograd.
All trainable parameters live in ONE flat float32 buffer (same for gradients and the two Adam
moments): one Adam launch, one RCCL all-reduce over a contiguous buffer.  Parameter names are the Keras
layer names (`unet_conv_downarm_0_0/kernel` ...), so a Keras .h5 maps 1:1 later.

Batch size 1 per replica (the configuration of every BASELINE config); BN uses batch statistics in
training and moving statistics in inference, moving averages updated with Keras 2.3.1 semantics.
"""
import math
import numpy as np
import torch

from . import ops


class UNet3D:
    def __init__(self, nb_features, input_shape, nb_levels, conv_size, nb_labels, name='unet', prefix=None,
                 feat_mult=1, nb_conv_per_level=1, batch_norm=None, activation='elu', device=None, seed=0,
                 final_pred_activation='linear', fold_upsample='auto', table_only=False, dtype='f32', conv_dropout=0):
        self.overlap_wgrad = False  # weight gradients on a second HIP stream (see _fork): measured 0.5 ms SLOWER per step
                                    # on one MI355X (cross-stream event waits cost more than the tails they fill) ...
        self.overlap_max_voxels = 40 ** 3  # ... on the small levels only: big persistent kernels just disturb each other
        # frozen use (segmentation network of the segmentation-regularised loss) with BatchNorm on BATCH statistics: what Keras
        # 2.3.1 does with a non-trainable BatchNormalization while the learning phase is 1 (`trainable` only stops the
        # weight / moving-average updates, keras/layers/normalization.py) -- predict_probs(batch_stats=True)
        self.frozen_batch_stats = False
        self.fuse_pool_bwd = True  # encoder levels: max-pool + BatchNorm + ELU backward in one pass (False: separate kernels)
        self._side_stream = None
        self._side_busy = False
        # dtype 'bf16' (BASELINE.json configs[3] / [4]): activations, activation gradients and the packed conv weights are
        # bfloat16 (csrc/conv_bf16.hip + the _bf16 pointwise kernels); master weights, gradients, Adam moments, BatchNorm
        # statistics / reductions, the head's prediction and the loss stay float32.  The reference has no such mode.
        if dtype not in ('f32', 'bf16'):
            raise ValueError("dtype should be 'f32' or 'bf16'")
        self.bf16 = dtype == 'bf16'
        self.act_dtype = torch.bfloat16 if self.bf16 else torch.float32
        # conv_dropout (ext/neuron/models.py:320-324, 448-451): KL.Dropout(rate, noise_shape=[None, 1, 1, 1, C]) after every
        # conv + ELU, i.e. ONE factor per feature map and step (0, or 1/(1-rate)).  A per-channel factor never has to touch
        # the activations: the factor of a conv that feeds another conv rides on that conv's input-channel weights
        # (forward and data gradient use the scaled copy; its weight gradient is scaled back the same way), and the factor
        # of a level's last conv is absorbed by the BatchNorm that follows it (see _dropout_bn).  The skip connection reads
        # the conv layer's own output, before the dropout, in the reference too (models.py:431-432).
        self.conv_dropout = float(conv_dropout)
        if not 0.0 <= self.conv_dropout < 1.0:
            raise ValueError('conv_dropout should be in [0, 1)')
        # the masks are a function of (dropout seed, optimizer iteration t, number of training forwards since that step): a run
        # resumed from a checkpoint (which restores `iterations`) continues the exact mask sequence of an uninterrupted one,
        # on every rank -- and forwards that share an iteration (the 10-100 critic updates per generator step of the
        # adversarial schedule, repeated loss() calls) still draw a fresh mask each, as Keras does on every forward
        self._drop_forwards = 0
        self._drop_gen = torch.Generator(device='cpu')
        self._drop_seed = int(seed) + 0x5eed
        self._drop_next = None      # scales for the next training forward (tests); None: drawn
        self._drop = None           # conv name -> per-channel scale of the step in flight
        self._drop_ps = None        # batchsize > 1: conv name -> [B, C] scales, one mask per sample (see _start_dropout_batch)
        self._mult = None
        self._packed_scaled = False
        if conv_size != 3:
            raise NotImplementedError('only conv_size=3 is supported')
        if activation != 'elu':
            raise NotImplementedError("only activation='elu' is supported")
        if batch_norm not in (-1, len(input_shape) - 1 + 1, 4):
            raise NotImplementedError('batch_norm=-1 (channels-last BatchNormalization after each level) is required')
        # nb_labels > 1 / 'softmax': the (frozen) segmentation network of the segmentation-regularised loss
        # (SynthSR/training.py:373-389); it is only run through predict_probs() / backward_input()
        if final_pred_activation not in ('linear', 'softmax'):
            raise NotImplementedError("final_pred_activation should be 'linear' or 'softmax'")
        # linear heads: one channel per regression target ('l1' / 'l2'), or intensity + spread channels per target for the
        # 'laplace' loss (SynthSR/training.py:246-249, 325-326)
        if final_pred_activation == 'softmax' and nb_labels < 2:
            raise NotImplementedError('a softmax head needs nb_labels > 1')
        if final_pred_activation == 'linear' and not 1 <= nb_labels <= 4:
            raise NotImplementedError('linear heads have 1 to 4 output channels')
        self.final_pred_activation = final_pred_activation
        self.nb_labels = int(nb_labels)
        self.need_input_grad = False  # True: also keep the data-gradient weights of the first conv (backward_input)
        if len(input_shape) != 4:
            raise NotImplementedError('3-D volumes only')
        self.prefix = name if prefix is None else prefix
        self.device = torch.device(device) if device is not None else (
            torch.device('cpu') if table_only else torch.device('cuda', torch.cuda.current_device()))
        self.input_shape = [int(s) for s in input_shape]
        self.nb_levels = L = int(nb_levels)
        self.nconv = int(nb_conv_per_level)
        self.feats = [int(np.round(nb_features * feat_mult ** l)) for l in range(L)]
        for s in self.input_shape[:3]:
            if s % (2 ** (L - 1)) != 0:
                raise ValueError('spatial shape must be divisible by 2**(nb_levels-1)')
        self.shapes = [[s // (2 ** l) for s in self.input_shape[:3]] for l in range(L)]
        cin = self.input_shape[3]

        # ---- parameter table (forward order); BN stored as [beta | gamma] so that the backward reduction
        # [sum dy | sum dy*xhat] lands directly on [dbeta | dgamma]
        self.specs = []  # (name, shape, kind)
        self.enc, self.dec = [], []
        c = cin
        for l in range(L):
            convs = []
            for k in range(self.nconv):
                nm = '%s_conv_downarm_%d_%d' % (self.prefix, l, k)
                convs.append(self._add_conv(nm, c, self.feats[l], self.shapes[l]))
                c = self.feats[l]
            bn = self._add_bn('%s_bn_down_%d' % (self.prefix, l), c)
            self.enc.append(dict(convs=convs, bn=bn))
        for k in range(L - 1):
            l = L - 2 - k
            c_in = self.feats[l] + c
            convs = []
            for j in range(self.nconv):
                nm = '%s_conv_uparm_%d_%d' % (self.prefix, L + k, j)
                convs.append(self._add_conv(nm, c_in, self.feats[l], self.shapes[l]))
                c_in = self.feats[l]
            c = self.feats[l]
            bn = self._add_bn('%s_bn_up_%d' % (self.prefix, k), c)
            # nearest-upsample folding of the first conv of the stage (ops.conv3d_up): 3.4x fewer FLOPs on the up-sampled
            # channels; measured faster on every level of the 160^3 network (parity = grid.z keeps the GPU filled)
            # bf16: the folded kernels of the small deep levels are launch / latency-bound (8 parities x a handful of
            # tiles): measured at 160^3, folding pays from a 40^3 low-res grid up (level 1: -0.2 ms, level 0: -1.3 ms per
            # step), is neutral at 20^3 and costs 0.1 ms at 10^3 (profiles/r03_bf16_c1_bench.json.log)
            lo_vox = int(np.prod(self.shapes[l + 1]))
            fold = (lo_vox >= (32768 if self.bf16 else 512)) if fold_upsample == 'auto' else bool(fold_upsample)
            convs[0]['fold'] = fold
            convs[0]['cs'] = self.feats[l]
            self.dec.append(dict(convs=convs, bn=bn, level=l, fold=fold))
        self.head = dict(name='%s_likelihood' % self.prefix, cin=c,
                         w=self._add('%s_likelihood/kernel' % self.prefix, (c, self.nb_labels), 'head_w'),
                         b=self._add('%s_likelihood/bias' % self.prefix, (self.nb_labels,), 'bias'))
        self.n_params = sum(int(np.prod(s[1])) for s in self.specs)
        self.bn_layers = [e['bn'] for e in self.enc] + [d['bn'] for d in self.dec]
        if table_only:      # host-side inspection of the layer table (names / shapes / count): nothing is allocated
            return
        dev = self.device
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.iterations = 0
        self.offsets = {}
        off = 0
        for nm, shp, kind in self.specs:
            self.offsets[nm] = (off, tuple(shp), kind)
            off += int(np.prod(shp))
        # BN statistics (batch + moving), flat [per layer: mean(C) | var(C)]
        self.bn_layers = [e['bn'] for e in self.enc] + [d['bn'] for d in self.dec]
        nst = sum(2 * b['C'] for b in self.bn_layers)
        self.bn_batch = torch.zeros(nst, dtype=torch.float32, device=dev)
        self.bn_moving = torch.zeros(nst, dtype=torch.float32, device=dev)
        self.bn_corr = torch.ones(nst, dtype=torch.float32, device=dev)
        self.bn_ws = torch.zeros(2 * max(b['C'] for b in self.bn_layers), dtype=torch.float64, device=dev)
        o = 0
        lvl_of = [l for l in range(L)] + [d['level'] for d in self.dec]
        for b, l in zip(self.bn_layers, lvl_of):
            b['soff'] = o
            C = b['C']
            self.bn_moving[o + C:o + 2 * C] = 1.0  # moving_variance initialised to ones
            n = float(np.prod(self.shapes[l]))
            self.bn_corr[o + C:o + 2 * C] = n / (n - (1.0 + ops.BN_EPS))  # Keras 2.3.1 sample-variance correction
            o += 2 * C
        self.bn_momentum = 0.99
        self.training = True
        self.batch = 1
        self._bufs = {}
        self.init_weights(seed)

    # ------------------------------------------------------------------ parameter bookkeeping
    def _add(self, name, shape, kind):
        self.specs.append((name, tuple(shape), kind))
        return name

    def _add_conv(self, name, cin, cout, shape):
        return dict(name=name, cin=cin, cout=cout, shape=list(shape), w=self._add(name + '/kernel', (3, 3, 3, cin, cout), 'kernel'),
                    b=self._add(name + '/bias', (cout,), 'bias'), wp=None, wpd=None)

    def _add_bn(self, name, C):
        return dict(name=name, C=C, beta=self._add(name + '/beta', (C,), 'beta'),
                    gamma=self._add(name + '/gamma', (C,), 'gamma'))

    def view(self, name, buf=None):
        off, shp, _ = self.offsets[name]
        buf = self.params if buf is None else buf
        return buf[off:off + int(np.prod(shp))].view(*shp)

    def named_parameters(self):
        return [(nm, self.view(nm)) for nm, _, _ in self.specs]

    def init_weights(self, seed=0):
        """Keras defaults: glorot_uniform kernels, zero biases, gamma=1, beta=0"""
        g = torch.Generator(device='cpu')
        g.manual_seed(int(seed))
        for nm, shp, kind in self.specs:
            v = self.view(nm)
            if kind == 'kernel':
                fan_in, fan_out = 27 * shp[3], 27 * shp[4]
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                v.copy_((torch.rand(shp, generator=g) * 2 - 1) * lim)
            elif kind == 'head_w':
                lim = math.sqrt(6.0 / (shp[0] + shp[1]))
                v.copy_((torch.rand(shp, generator=g) * 2 - 1) * lim)
            elif kind == 'gamma':
                v.fill_(1.0)
            else:
                v.zero_()
        self.repack()

    def all_convs(self):
        for e in self.enc:
            for c in e['convs']:
                yield c
        for d in self.dec:
            for c in d['convs']:
                yield c

    def _pack_jobs(self):
        """job table for synthsr_conv3d_pack_all: every packed weight set of the network, one flat destination"""
        import ctypes
        from . import _lib
        lib = _lib.load()
        jobs, off = [], 0

        def plan(shape, cin_e, cout_e, plain):
            out = (ctypes.c_int64 * 8)()
            _lib.check(lib.synthsr_conv3d_plan(ops.conv_ctx_host(), _lib.i3(shape), cin_e, cout_e, int(plain), out), 'conv3d_plan')
            return [int(v) for v in out]

        def add(c, key, shape, ci_off, cin, mode, up):
            nonlocal off
            cin_total, cout = c['cin'], c['cout']
            cin_e, cout_e = (cout, cin) if mode else (cin, cout)
            # kind: 1 plain conv, 2 forward parity convs of a folded decoder conv, 0 their data gradient
            ck, ncc, nt, nchunks, _, _, nv, per = plan(shape, cin_e, cout_e, ((2 if mode == 0 else 0) if up else 1))
            # last job field: size of the MFMA section of a mixed layout; the split layout (nt <= -100) wants its co-chunk count
            mfma_count = nchunks if nt <= -100 else nchunks * ncc * 27 * (ck // 8) * nt * 128
            w_off = self.offsets[c['w']][0]
            c[key + '_off'] = (off, per * (8 if up else 1))
            for p in range(8 if up else 1):
                jobs.append([w_off, off, per, cin_total, ci_off, cin, cout, mode, ck, ncc, nt, p if up else -1, nv,
                             mfma_count])
                off += per

        first = True
        for c in self.all_convs():
            if c.get('fold'):
                cs, cl = c['cs'], c['cin'] - c['cs']
                lo_shape = [s // 2 for s in c['shape']]
                add(c, 'wp_s', c['shape'], 0, cs, 0, False)
                add(c, 'wpd_s', c['shape'], 0, cs, 1, False)
                add(c, 'wp_u', lo_shape, cs, cl, 0, True)
                add(c, 'wpd_u', lo_shape, cs, cl, 1, True)
            else:
                add(c, 'wp', c['shape'], 0, c['cin'], 0, False)
                if not first or self.need_input_grad:  # the first layer's input normally has no gradient
                    add(c, 'wpd', c['shape'], 0, c['cin'], 1, False)
            first = False
        # zeros: synthsr_conv3d_pack_all never writes the 19 structurally empty slots of a 27-slot parity set
        self._packed = torch.zeros(off, dtype=torch.float32, device=self.device)
        self._jobs = torch.tensor(jobs, dtype=torch.int64, device=self.device)
        for c in self.all_convs():
            for key in ('wp', 'wpd', 'wp_s', 'wpd_s', 'wp_u', 'wpd_u'):
                if key + '_off' in c:
                    o, n = c[key + '_off']
                    c[key] = self._packed[o:o + n]

    def _bf16_pack_jobs(self):
        """job table of synthsr_conv3d_bf16_pack_all: bf16 fragment-ordered copies of every conv kernel (forward, and
        data-gradient where a gradient flows on) in one flat buffer"""
        import ctypes
        from . import _lib
        lib = _lib.load()
        jobs, off, first = [], 0, True

        def add(c, key, ci_off, cin, mode, up):
            nonlocal off
            start = off
            for par in (range(8) if up else (-1,)):
                job = (ctypes.c_int64 * 13)()
                _lib.check(lib.synthsr_conv3d_bf16_pack_job(c['cin'], ci_off, cin, c['cout'], mode, par, job), 'bf16_pack_job')
                job[0], job[1] = self.offsets[c['w']][0], off
                off += int(job[2])
                jobs.append([int(v) for v in job])
            c[key + '_off'] = (start, off - start)

        for c in self.all_convs():
            if c.get('fold'):   # skip-channel half at full resolution + the 8 parity sets of the up-sampled half (conv_bf16.hip)
                cs, cl = c['cs'], c['cin'] - c['cs']
                add(c, 'wp_s', 0, cs, 0, False)
                add(c, 'wpd_s', 0, cs, 1, False)
                add(c, 'wp_u', cs, cl, 0, True)
                add(c, 'wpd_u', cs, cl, 1, True)
            else:
                add(c, 'wp', 0, c['cin'], 0, False)
                if not first or self.need_input_grad:  # the first layer's input normally has no gradient
                    add(c, 'wpd', 0, c['cin'], 1, False)
            first = False
        self._packed_bf16 = torch.empty(off, dtype=torch.bfloat16, device=self.device)
        self._jobs_bf16 = torch.tensor(jobs, dtype=torch.int64, device=self.device)
        for c in self.all_convs():
            for key in ('wp', 'wpd', 'wp_s', 'wpd_s', 'wp_u', 'wpd_u'):
                if key + '_off' in c:
                    o, n = c[key + '_off']
                    c[key] = self._packed_bf16[o:o + n]

    def _repack_bf16(self, src=None):
        from . import _lib
        if getattr(self, '_jobs_bf16', None) is None:
            self._bf16_pack_jobs()
        _lib.check(_lib.load().synthsr_conv3d_bf16_pack_all(_lib.ptr(self.params if src is None else src),
                                                            _lib.ptr(self._packed_bf16), _lib.ptr(self._jobs_bf16),
                                                            int(self._jobs_bf16.shape[0]), _lib.stream()), 'bf16_pack_all')

    def repack(self, src=None):
        """refresh the MFMA-fragment-ordered copies of ALL conv kernels in one launch (after init / optimizer step /
        load); src: a flat buffer laid out like self.params to pack instead (the dropout-scaled copy)"""
        self._packed_scaled = src is not None
        if self.bf16:
            return self._repack_bf16(src)
        from . import _lib
        if getattr(self, '_jobs', None) is None or getattr(self, '_jobs_epoch', None) != ops.conv_layout_epoch():
            self._jobs_epoch = ops.conv_layout_epoch()   # the packed layouts depend on the arithmetic / plan options
            self._pack_jobs()
        _lib.check(_lib.load().synthsr_conv3d_pack_all(_lib.ptr(self.params if src is None else src), _lib.ptr(self._packed),
                                                       _lib.ptr(self._jobs), int(self._jobs.shape[0]), _lib.stream()),
                   'conv3d_pack_all')

    def state_dict(self):
        sd = {nm: self.view(nm).detach().cpu().clone() for nm, _, _ in self.specs}
        for b in self.bn_layers:
            o, C = b['soff'], b['C']
            sd[b['name'] + '/moving_mean'] = self.bn_moving[o:o + C].cpu().clone()
            sd[b['name'] + '/moving_variance'] = self.bn_moving[o + C:o + 2 * C].cpu().clone()
        return sd

    def load_state_dict(self, sd, strict=True):
        for nm, _, _ in self.specs:
            if nm in sd:
                self.view(nm).copy_(torch.as_tensor(sd[nm]).to(self.device).reshape(self.view(nm).shape))
            elif strict:
                raise KeyError(nm)
        for b in self.bn_layers:
            o, C = b['soff'], b['C']
            if b['name'] + '/moving_mean' in sd:
                self.bn_moving[o:o + C].copy_(torch.as_tensor(sd[b['name'] + '/moving_mean']).to(self.device))
                self.bn_moving[o + C:o + 2 * C].copy_(torch.as_tensor(sd[b['name'] + '/moving_variance']).to(self.device))
        self.repack()

    # ------------------------------------------------------------------ buffers
    _F32_BUFS = ('loss', 'dpred', 'pred', 'loss_unused', 'zero_t', 'probs', 'dwc', 'head_ab')

    def buf(self, key, shape):
        """persistent scratch tensor: activations / activation gradients in the network's dtype, the head's outputs,
        losses and weight-gradient scratch always float32"""
        t = self._bufs.get(key)
        n = int(np.prod(shape))
        if t is None or t.numel() < n:
            dt = torch.float32 if key in self._F32_BUFS or key.startswith('ssim') else self.act_dtype
            t = torch.empty(n, dtype=dt, device=self.device)
            self._bufs[key] = t
        return t[:n].view(*shape)

    def _stats(self, bn):
        o, C = bn['soff'], bn['C']
        src = self.bn_batch if (self.training or self.frozen_batch_stats) else self.bn_moving
        return src[o:o + 2 * C]

    # ------------------------------------------------------------------ batches of volumes (SynthSR/training.py: batchsize)
    def set_batch(self, batch):
        """B volumes per step: every activation tensor holds the B volumes stacked along the first spatial axis
        ([B*d0, d1, d2, C]).  BatchNorm statistics / reductions, pooling, up-sampling + concatenation, the head and the
        loss are per-voxel or per-channel-sum operations and run on the stack as one volume (d0 is even on every level that
        is pooled, so no 2x2x2 window straddles two volumes); the 3x3x3 convolutions (zero padding at every volume's own
        border) run volume by volume on slices of the stack, their weight gradients accumulate."""
        batch = int(batch)
        if batch < 1:
            raise ValueError('batch should be >= 1')
        if batch != self.batch:
            self.batch = batch
            lvl_of = [l for l in range(self.nb_levels)] + [d['level'] for d in self.dec]
            for b, l in zip(self.bn_layers, lvl_of):
                n = float(batch * np.prod(self.shapes[l]))
                o, C = b['soff'], b['C']
                self.bn_corr[o + C:o + 2 * C] = n / (n - (1.0 + ops.BN_EPS))

    def _bshape(self, l):
        s = self.shapes[l]
        return [self.batch * s[0], s[1], s[2]]

    def _pb(self, fn, *stacked):
        """fn(*tensors) on each volume of the stack (tensors: stacked along axis 0, sliced without copies)"""
        if self.batch == 1:
            return fn(*stacked)
        for parts in zip(*[t.chunk(self.batch, 0) for t in stacked]):
            fn(*parts)

    # ------------------------------------------------------------------ feature-wise dropout
    def set_dropout_seed(self, seed):
        self._drop_seed = int(seed)

    def set_dropout_scales(self, scales):
        """explicit per-channel factors {conv layer name: [Cout] (or [B, Cout]: one row per sample of the batch) of
        0 | 1/(1-rate)} for the NEXT training forward (parity tests against the oracle); None: drawn from the network's own
        generator"""
        self._drop_next = scales

    def _draw_dropout(self, rows, total):
        # tf.nn.dropout: keep where uniform >= rate, scale the kept features by 1 / (1 - rate); ONE draw for all layers
        p = self.conv_dropout
        self._drop_gen.manual_seed((self._drop_seed * 1000003 + self.iterations + 7046029254386353131 * self._drop_forwards)
                                   & 0x7fffffffffffffff)
        self._drop_forwards += 1
        return (torch.rand(rows * total, generator=self._drop_gen) >= p).float() / (1.0 - p)

    def _upload_dropout(self, flat):
        n = flat.numel()
        if getattr(self, '_drop_host', None) is None or self._drop_host.numel() != n:
            self._drop_host = torch.empty(n, dtype=torch.float32).pin_memory()  # one pinned staging + one device buffer
            self._drop_dev = torch.empty(n, dtype=torch.float32, device=self.device)
            self._drop_event = None
        if self._drop_event is not None:
            self._drop_event.synchronize()  # the previous step's upload has left the staging buffer
        self._drop_host.copy_(flat)
        self._drop_dev.copy_(self._drop_host, non_blocking=True)
        self._drop_event = torch.cuda.Event()
        self._drop_event.record()
        return self._drop_dev

    def _start_dropout_batch(self):
        """batchsize > 1: KL.Dropout(noise_shape=[None, 1, 1, 1, C]) draws one feature mask PER SAMPLE
        (ext/neuron/models.py:320-324).  A per-sample factor cannot ride on the kernels shared by the batch nor on the
        BatchNorm of the whole stack, so the dropped-out tensors d = s_b * y are materialised (ops.scale_channels) next to
        the conv outputs y: convs, BatchNorm statistics, pooling and up-sampling read d; the skip connections and ELU' read
        y (models.py:431-432); the backward multiplies by s_b in the ELU-backward kernel (ops.elu_bwd_drop)."""
        convs = list(self.all_convs())
        B, total = self.batch, sum(c['cout'] for c in convs)
        if self._drop_next is not None:
            rows = []
            for c in convs:
                a = np.asarray(self._drop_next[c['name']], dtype=np.float32)
                rows.append(torch.as_tensor(np.array(np.broadcast_to(a, (B, c['cout'])))).reshape(-1))
            flat = torch.cat(rows)
        else:  # sample-major draw: sample 0 sees the masks a batch of one would
            r = self._draw_dropout(B, total).view(B, total)
            o, rows = 0, []
            for c in convs:
                rows.append(r[:, o:o + c['cout']].reshape(-1))
                o += c['cout']
            flat = torch.cat(rows)
        dev = self._upload_dropout(flat)
        drop, o = {}, 0
        for c in convs:
            drop[c['name']] = dev[o:o + B * c['cout']].view(B, c['cout'])
            o += B * c['cout']
        self._drop_next = None
        self._drop = None
        self._drop_ps = drop
        if self._packed_scaled:
            self.repack()

    def _dropped(self, y, conv, key):
        return ops.scale_channels(y, self._drop_ps[conv['name']], out=self.buf(key, list(y.shape)))

    def _start_dropout(self):
        p = self.conv_dropout
        convs = list(self.all_convs())
        total = sum(c['cout'] for c in convs)
        if self._drop_next is not None:
            flat = torch.cat([torch.as_tensor(np.asarray(self._drop_next[c['name']], dtype=np.float32).reshape(-1))
                              for c in convs])
        else:
            flat = self._draw_dropout(1, total)
        dev = self._upload_dropout(flat)
        drop, o = {}, 0
        for c in convs:
            drop[c['name']] = dev[o:o + c['cout']]
            o += c['cout']
        self._drop_next = None
        self._drop = drop
        self._drop_ps = None
        if self._mult is None:
            self._mult = torch.ones_like(self.params)
            self._params_eff = torch.empty_like(self.params)
            self.bn_true = torch.zeros_like(self.bn_batch)
        self._mult.fill_(1.0)
        for grp in self.enc + self.dec:
            for k in range(1, len(grp['convs'])):
                self.view(grp['convs'][k]['w'], self._mult).mul_(drop[grp['convs'][k - 1]['name']][None, None, None, :, None])
        torch.mul(self.params, self._mult, out=self._params_eff)
        self.repack(self._params_eff)

    def _dropout_bn(self, bn, s):
        """BatchNorm of x' = s * x (s per channel) written in terms of the stored x: xhat' = (x - mean_x) * r with
        r = s / sqrt(s^2 var_x + eps) = 1 / sqrt(var_x + eps / s^2); the kernels form rsqrt(var + eps), so the batch-variance
        slot gets var_x + eps (1/s^2 - 1) (inf for a dropped feature: xhat' = 0, the output is beta).  dr/dvar_x = -r^3/2
        as for a plain BatchNorm, so the backward kernels hold unchanged.  The moving averages see the statistics of x'."""
        o, C = bn['soff'], bn['C']
        mean, var = self.bn_batch[o:o + C], self.bn_batch[o + C:o + 2 * C]
        self.bn_true[o:o + C] = mean * s
        self.bn_true[o + C:o + 2 * C] = var * s * s
        inv2 = torch.where(s > 0, 1.0 / (s * s).clamp_min(1e-30), torch.full_like(s, float('inf')))
        var.add_(ops.BN_EPS * (inv2 - 1.0))

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        """x [d0,d1,d2,Cin] -> saves activations; returns the last decoder activation (pre-BN) and its BN"""
        L = self.nb_levels
        if not self.bf16:   # packed weights of another arithmetic would be read under this one's plans
            ops.check_layout_epoch(getattr(self, '_jobs_epoch', None), 'UNet3D.forward')
        self.saved = dict(x=[], enc=[], cat=[], dec=[])
        dropping = self.training and self.conv_dropout > 0
        per_sample = dropping and self.batch > 1
        if per_sample:
            self._start_dropout_batch()
            dropping = False    # no factors folded into the kernels / BatchNorm slots: the dropped-out tensors exist
            self.saved['encd'], self.saved['decd'] = [], []
        elif dropping:
            self._start_dropout()
        else:
            self._drop = self._drop_ps = None
            if self._packed_scaled:  # a training forward without its optimizer step left scaled kernels behind
                self.repack()
        if self.bf16 and x.dtype != torch.bfloat16:
            # generator output (float32, Cin channels) -> bf16 with the channel count padded to a multiple of 8 (zeros):
            # 16-byte K-groups for the MFMA; the first conv's kernel is packed / its gradient taken on the real Cin only
            x = ops.to_bf16_pad(x, (int(x.shape[-1]) + 7) // 8 * 8, out=self.buf('x_bf16', list(x.shape[:3]) + [
                (int(x.shape[-1]) + 7) // 8 * 8]))
        cur = x
        for l in range(L):
            e = self.enc[l]
            self.saved['x'].append(cur)
            acts = []
            nconv = len(e['convs'])
            for k, c in enumerate(e['convs']):
                out = self.buf('enc%d_%d' % (l, k), self._bshape(l) + [c['cout']])
                if self.training and k == nconv - 1 and self.batch == 1:  # the BatchNorm statistics ride in the conv epilogue
                    cur = ops.conv3d_stats(cur, c['wp'], self.view(c['b']), c['cout'], self._stats(e['bn']), self.bn_ws,
                                           1, out=out)
                else:
                    self._pb(lambda x_, o_, c=c: ops.conv3d(x_, c['wp'], self.view(c['b']), c['cout'], 1, out=o_), cur, out)
                    cur = out
                    if self.training and k == nconv - 1 and not per_sample:
                        ops.bn_stats(cur, self._stats(e['bn']), self.bn_ws)
                acts.append(cur)
                if per_sample:
                    cur = self._dropped(cur, c, 'encd%d_%d' % (l, k))
                    if k == 0:
                        self.saved['encd'].append([])
                    self.saved['encd'][l].append(cur)
                    if k == nconv - 1:
                        ops.bn_stats(cur, self._stats(e['bn']), self.bn_ws)
            self.saved['enc'].append(acts)
            if dropping:
                self._dropout_bn(e['bn'], self._drop[e['convs'][-1]['name']])
            if l < L - 1:
                cur = ops.bn_maxpool(cur, self._stats(e['bn']), self.view(e['bn']['gamma']), self.view(e['bn']['beta']),
                                     out=self.buf('pool%d' % l, self._bshape(l + 1) + [e['bn']['C']]))
        low, low_bn = cur, self.enc[L - 1]['bn']
        for k, d in enumerate(self.dec):
            l = d['level']
            skip = self.saved['enc'][l][-1]
            acts, dacts = [], []
            if d['fold']:
                c0 = d['convs'][0]
                lo_bn = ops.bn_apply(low, self._stats(low_bn), self.view(low_bn['gamma']), self.view(low_bn['beta']),
                                     out=self.buf('lobn%d' % k, list(low.shape)))
                self.saved['cat'].append((skip, lo_bn))
                # the parity convs write the raw up-sampled part (strided stores), the skip-channel conv then adds it
                # in place with coalesced reads and applies bias + ELU
                cur = self.buf('dec%d_0' % k, self._bshape(l) + [c0['cout']])

                def folded(lo_, skip_, o_, c0=c0):   # (bf16: the partial sums are rounded to bf16 once before the addition)
                    ops.conv3d_up(lo_, c0['wp_u'], None, None, c0['cout'], 0, out=o_)
                    ops.conv3d_add(skip_, c0['wp_s'], self.view(c0['b']), o_, c0['cout'], 1, out=o_)
                self._pb(folded, lo_bn, skip, cur)
                acts.append(cur)
                if per_sample:
                    cur = self._dropped(cur, c0, 'decd%d_0' % k)
                    dacts.append(cur)
            else:
                cat = ops.upsample_concat(skip, low, self._stats(low_bn), self.view(low_bn['gamma']),
                                          self.view(low_bn['beta']),
                                          out=self.buf('cat%d' % k, self._bshape(l) + [skip.shape[3] + low.shape[3]]))
                self.saved['cat'].append(cat)
                cur = cat
            nconv = len(d['convs'])
            stats_done = False
            for j, c in enumerate(d['convs']):
                if d['fold'] and j == 0:
                    continue
                out = self.buf('dec%d_%d' % (k, j), self._bshape(l) + [c['cout']])
                if self.training and j == nconv - 1 and self.batch == 1:
                    cur = ops.conv3d_stats(cur, c['wp'], self.view(c['b']), c['cout'], self._stats(d['bn']), self.bn_ws,
                                           1, out=out)
                    stats_done = True
                else:
                    self._pb(lambda x_, o_, c=c: ops.conv3d(x_, c['wp'], self.view(c['b']), c['cout'], 1, out=o_), cur, out)
                    cur = out
                acts.append(cur)
                if per_sample:
                    cur = self._dropped(cur, c, 'decd%d_%d' % (k, j))
                    dacts.append(cur)
            self.saved['dec'].append(acts)
            if per_sample:
                self.saved['decd'].append(dacts)
            if self.training and not stats_done:  # single-conv level whose only conv was the folded one (or per-sample dropout)
                ops.bn_stats(cur, self._stats(d['bn']), self.bn_ws)
            if dropping:
                self._dropout_bn(d['bn'], self._drop[d['convs'][-1]['name']])
            low, low_bn = cur, d['bn']
        self.saved['last'] = (low, low_bn)
        return low, low_bn

    def loss(self, x, target, kind='l1', loss_cropping=None, residual=None, res_stride=1, res_off=0, want_pred=False,
             fuse_head_bwd=False):
        """forward + unet_likelihood + regression loss (SynthSR/metrics_model.py:30-132): kind 'l1' | 'l2' | 'laplace' | 'ssim'
        with n = target.numel() / nvox regression targets (head channels: n, or 2n for laplace = intensities, spreads);
        loss_cropping = sizes of the centred box the loss is averaged over (metrics_model.py:70-90); residual
        [nvox, res_stride]: channel(s) res_off added to the intensities.  fuse_head_bwd (one l1 / l2 target): the head kernel
        also accumulates the two sums backward() needs, saving its pass over the last feature map -- only valid if
        self.dpred is NOT modified between loss() and backward() (the segmentation-regularised loss adds to it).
        Returns (loss tensor[1], pred [nvox*K] | None)"""
        K = self.nb_labels
        nvox_in = self.batch * int(np.prod(self.input_shape[:3]))
        n = K // 2 if kind == 'laplace' else K
        if kind == 'ssim' and (K != 1 or target.numel() != nvox_in):
            raise Exception('SSIM metric does not currently support multiple channels')  # metrics_model.py:108-109
        if self.final_pred_activation != 'linear' or (kind == 'laplace' and K % 2) or target.numel() != nvox_in * n:
            raise ValueError('the %s loss on %d regression target(s) needs a linear head with %d output channels, this '
                             'network has %d (%s)' % (kind, target.numel() // nvox_in, (2 if kind == 'laplace' else 1) *
                                                      (target.numel() // nvox_in), K, self.final_pred_activation))
        low, bn = self.forward(x)
        nvox = low.numel() // low.shape[3]
        C_last = int(low.shape[3])
        self._head_ab = None
        crop = None
        if loss_cropping is not None:
            size = [int(loss_cropping)] * 3 if np.ndim(loss_cropping) == 0 else [int(v) for v in loss_cropping]
            shape = [int(low.shape[0]) // self.batch, int(low.shape[1]), int(low.shape[2])]   # of ONE volume of the stack
            if len(size) != 3 or any(c < 1 or c > s for c, s in zip(size, shape)):
                raise ValueError('loss_cropping %s does not fit the output shape %s' % (size, shape))
            crop = ([int((s - c) / 2) for s, c in zip(shape, size)], size)
        self.loss_buf = self.buf('loss', [1])
        self.loss_buf.zero_()
        self.dpred = self.buf('dpred', [nvox * K])
        pred = self.buf('pred', [nvox * K]) if want_pred else None
        if self.batch > 1 and (kind == 'ssim' or crop is not None):
            # a loss defined per volume (the centred loss_cropping box, the slice-wise SSIM windows) on a batch: volume by
            # volume on slices of the stack -- BatchNorm statistics stay those of the whole batch -- then the mean over the
            # batch (the reference's K.mean over [B, ...], metrics_model.py:100-125): loss and gradient times 1 / B
            B, nvb = self.batch, nvox // self.batch
            pred = self.buf('pred', [nvox * K])
            stats, gam, bet = self._stats(bn), self.view(bn['gamma']), self.view(bn['beta'])
            hw, hb = self.view(self.head['w']), self.view(self.head['b'])
            for b_, lo_b in enumerate(low.chunk(B, 0)):
                tgt_b = target.reshape(B, -1)[b_]
                res_b = None if residual is None else residual.reshape(B, nvb, -1)[b_]
                pred_b, dpred_b = pred.view(B, nvb * K)[b_], self.dpred.view(B, nvb * K)[b_]
                if kind == 'ssim':
                    ops.head_loss_fwd(lo_b, stats, gam, bet, hw, hb, tgt_b, self.buf('loss_unused', [1]), kind='l1', pred=pred_b,
                                      residual=res_b, res_stride=res_stride, res_off=res_off)
                    ops.ssim_loss(pred_b, tgt_b, lo_b.shape[:3], self.loss_buf, dpred_b, crop=crop,
                                  scratch=lambda key, numel: self.buf(key, [numel]))
                else:
                    ops.head_loss_fwd(lo_b, stats, gam, bet, hw, hb, tgt_b, self.loss_buf, kind=kind, crop=crop, pred=pred_b,
                                      dpred=dpred_b, residual=res_b, res_stride=res_stride, res_off=res_off)
            self.loss_buf.mul_(1.0 / B)
            self.dpred.mul_(1.0 / B)
            return self.loss_buf, pred
        if kind == 'ssim':  # the head kernel only produces the prediction; loss and gradient come from the SSIM kernels
            pred = self.buf('pred', [nvox])
            ops.head_loss_fwd(low, self._stats(bn), self.view(bn['gamma']), self.view(bn['beta']),
                              self.view(self.head['w']), self.view(self.head['b']), target, self.buf('loss_unused', [1]),
                              kind='l1', pred=pred, residual=residual, res_stride=res_stride, res_off=res_off)
            ops.ssim_loss(pred, target, low.shape[:3], self.loss_buf, self.dpred, crop=crop,
                          scratch=lambda key, numel: self.buf(key, [numel]))
            return self.loss_buf, pred
        self._head_ab = None
        if fuse_head_bwd and self.training and K == 1 and kind in ('l1', 'l2') and C_last <= 120:
            self._head_ab = self.buf('head_ab', [C_last + 1])
            self._head_ab.zero_()
        ops.head_loss_fwd(low, self._stats(bn), self.view(bn['gamma']), self.view(bn['beta']), self.view(self.head['w']),
                          self.view(self.head['b']), target, self.loss_buf, kind=kind, crop=crop, pred=pred,
                          dpred=self.dpred, residual=residual, res_stride=res_stride, res_off=res_off, ab=self._head_ab)
        return self.loss_buf, pred

    def loss_l1(self, x, target, residual=None, res_stride=1, res_off=0, want_pred=False):
        """forward + unet_likelihood + L1 (SynthSR/metrics_model.py:102-104). Returns (loss tensor[1], pred|None)"""
        return self.loss(x, target, 'l1', None, residual, res_stride, res_off, want_pred)

    def predict(self, x):
        """inference forward (moving statistics): x [d0,d1,d2,Cin] -> [d0,d1,d2,K] (K = head channels: one per
        regression target; intensities then spreads for a laplace head)"""
        if self.final_pred_activation != 'linear':
            raise ValueError('predict() is for linear heads; use predict_probs() for a softmax head')
        K = self.nb_labels
        was = self.training
        self.training = False
        try:
            low, bn = self.forward(x)
            nvox = low.numel() // low.shape[3]
            zero_t = self.buf('zero_t', [nvox * K])
            zero_t.zero_()
            loss = self.buf('loss', [1])
            loss.zero_()
            pred = self.buf('pred', [nvox * K])
            ops.head_loss_fwd(low, self._stats(bn), self.view(bn['gamma']), self.view(bn['beta']),
                              self.view(self.head['w']), self.view(self.head['b']), zero_t, loss, kind='l1', pred=pred)
        finally:
            self.training = was
        return pred.view(self.batch * self.input_shape[0], self.input_shape[1], self.input_shape[2], K)

    # ------------------------------------------------------------------ frozen use (segmentation network)
    def enable_input_grad(self):
        """keep data-gradient weights for the first conv as well, so that backward_input() can reach the network input"""
        if not self.need_input_grad:
            self.need_input_grad = True
            self._jobs = None
            self._jobs_bf16 = None
            self.repack()

    def predict_probs(self, x, batch_stats=False):
        """forward of a (frozen) softmax-headed network: x [d0,d1,d2,Cin] -> probs [nvox, nb_labels] (activations are kept
        for backward_input).  batch_stats: BatchNorm normalises with the statistics of x's own activations (a frozen Keras
        network inside a model that is being fitted) instead of the moving averages; nothing is updated either way"""
        assert self.nb_labels > 1
        self.frozen_batch_stats = bool(batch_stats)
        if batch_stats:
            # Keras' learning phase: the forward pass gathers the batch statistics (conv epilogues / bn_stats) and the
            # Dropout layers of a network built with conv_dropout > 0 are active, frozen or not (the reference builds its
            # segmentation unet with conv_dropout=dropout, SynthSR/training.py:381); backward_input sees the same masks
            self.training = True
            try:
                low, bn = self.forward(x)
            finally:
                self.training = False
        else:
            self.training = False
            low, bn = self.forward(x)
        nvox = low.numel() // low.shape[3]
        probs = self.buf('probs', [nvox, self.nb_labels])
        ops.seg_head_fwd(low, self._stats(bn), self.view(bn['gamma']), self.view(bn['beta']), self.view(self.head['w']),
                         self.view(self.head['b']), probs)
        return probs

    def backward_input(self, dbn):
        """gradient w.r.t. the network input of a loss whose gradient w.r.t. the LAST BatchNorm output is `dbn`
        [d0,d1,d2,C]; the network is frozen: inference-mode BatchNorm (moving statistics), no weight gradients"""
        assert self.need_input_grad and not self.training
        return self.backward(g_last=dbn, frozen=True)   # BatchNorm as in the forward pass (predict_probs: batch_stats)

    # ------------------------------------------------------------------ backward
    def backward(self, on_grad_ready=None, g_last=None, frozen=False):
        """gradients of the L1 loss w.r.t. every parameter into self.grads (zeroed here).
        frozen=True (with g_last = gradient w.r.t. the last BatchNorm output): data gradients only, inference-mode
        BatchNorm; returns the gradient w.r.t. the network input.
        on_grad_ready(offset_lo): optional hook called when every gradient at flat offset >= offset_lo is final
        (used to overlap the RCCL all-reduce with the rest of the backward)."""
        L = self.nb_levels
        G = self.grads
        if not self.bf16:
            ops.check_layout_epoch(getattr(self, '_jobs_epoch', None), 'UNet3D.backward')
        self._frozen = frozen
        self._pending_bn = None
        if ops._deterministic:  # ordered in-workgroup sums need a fixed channel group per thread (unet_pointwise.hip)
            bad = [c for c in set(self.feats) if c % 4 or 384 % (c // 4)]
            if bad:
                raise ValueError('deterministic mode: feature counts %s need 384 %% (C / 4) == 0' % bad)
        if self._drop is not None and not frozen:
            # the convs ran on W * diag(s_in): dL/dW = dL/d(W diag(s_in)) * diag(s_in), applied to each finished range
            # of the flat gradient before it is handed on (bucketed all-reduce)
            user, mult, done = on_grad_ready, self._mult, [G.numel()]

            def on_grad_ready(lo):
                G[lo:done[0]].mul_(mult[lo:done[0]])
                done[0] = lo
                if user is not None:
                    user(lo)
            out = self._backward_inner(on_grad_ready, g_last, frozen)
            if done[0] > 0:
                on_grad_ready(0)
            return out
        return self._backward_inner(on_grad_ready, g_last, frozen)

    def _backward_inner(self, on_grad_ready, g_last, frozen):
        G = self.grads
        low, bn = self.saved['last']
        C = low.shape[3]
        if frozen:
            if getattr(self, '_zero_sums', None) is None:
                self._zero_sums = torch.zeros(2 * max(b['C'] for b in self.bn_layers), device=self.device)
                self._frozen_sums = torch.zeros_like(self._zero_sums)
            return self._backward_body(g_last, on_grad_ready)
        G.zero_()
        # the gradient w.r.t. the last BatchNorm output is rank-1 (dpred[v] * w_head[c]): it is neither stored nor
        # reduced; head_bwd emits that BN's backward sums and the first ELU backward forms it on the fly
        if self.nb_labels > 1:  # K-channel linear head (laplace): the gradient w.r.t. the BN output is written out
            dbn = self.buf('dbn_head', list(low.shape))
            ops.head_bwd_multi(self.dpred, low, self._stats(bn), self.view(bn['gamma']), self.view(bn['beta']),
                               self.view(self.head['w']), dbn, self.view(self.head['w'], G), self.view(self.head['b'], G))
            return self._backward_body(dbn, on_grad_ready)
        off = self.offsets[bn['beta']][0]
        sums = self.grads[off:off + 2 * bn['C']]
        if getattr(self, '_head_ab', None) is not None:  # the head kernel of loss() already holds the sums (fuse_head_bwd)
            ops.head_bwd_from_sums(self._head_ab, self.view(bn['gamma']), self.view(bn['beta']), self.view(self.head['w']),
                                   self.view(self.head['w'], G), self.view(self.head['b'], G), bn_sums=sums)
        else:
            ops.head_bwd(self.dpred, low, self._stats(bn), self.view(bn['gamma']), self.view(bn['beta']),
                         self.view(self.head['w']), None, self.view(self.head['w'], G), self.view(self.head['b'], G),
                         bn_sums=sums)
        self._pending_bn = (bn, sums)
        self._rank1 = (self.dpred, self.view(self.head['w']))
        return self._backward_body(None, on_grad_ready)

    def _backward_body(self, g, on_grad_ready):
        L = self.nb_levels
        G = self.grads
        frozen = self._frozen
        dskips = [None] * L
        for k in range(len(self.dec) - 1, -1, -1):
            d = self.dec[k]
            l = d['level']
            acts = self.saved['dec'][k]
            ps = self._drop_ps is not None
            dacts = self.saved['decd'][k] if ps else acts  # what the consumers of the conv outputs read (per-sample dropout)
            g = self._bn_backward(g, dacts[-1], d['bn'])
            Cs = self.feats[l]
            if d['fold']:
                skip, lo_bn = self.saved['cat'][k]
                Cl = lo_bn.shape[3]
                # all convs but the first: regular; the first one through the folded kernels
                if len(d['convs']) > 1:
                    dz = self._convs_backward(g, None, d['convs'][1:], acts[1:], dacts[0], need_dx=True, tag='d%d' % k,
                                              elu_below=acts[0], below_conv=d['convs'][0], dacts=dacts[1:])
                else:  # nb_conv_per_level = 1: the folded conv feeds the BatchNorm itself
                    self._join()
                    dz = self._elu_backward(g, acts[0], None, None, conv=d['convs'][0])
                c0 = d['convs'][0]
                if not frozen:
                    dW = self.view(c0['w'], self.grads)
                    self._join()
                    dwc = self.buf('dwc', [8, 27, Cl, c0['cout']])
                    # the unpack kernel leaves the partials it has folded at zero: the buffer is zeroed when it is (re)allocated
                    # or outgrows what has been zeroed so far, not once per call (85 MB of memsets per step at configs[1])
                    base = self._bufs['dwc']
                    known = getattr(self, '_dwc_zeroed', (None, 0))
                    dwc_is_zero = known[0] is base and known[1] >= dwc.numel()
                    self._dwc_zeroed = (base, max(dwc.numel(), known[1] if known[0] is base else 0))

                    def c0_wgrads(skip=skip, dz=dz, dW=dW, lo_bn=lo_bn, dwc=dwc, c0=c0, Cs=Cs, dwc_is_zero=dwc_is_zero):
                        def one(skip_, dz_, lo_):
                            ops.conv3d_wgrad_part(skip_, dz_, dW, 0, dbias=self.view(c0['b'], self.grads))
                            ops.conv3d_up_wgrad(lo_, dz_, dwc, dW, Cs, dwc_is_zero=dwc_is_zero)
                        self._pb(one, skip, dz, lo_bn)
                    self._fork(c0_wgrads, skip[..., 0].numel())
                dskips[l] = self.buf('dskip%d' % l, self._bshape(l) + [Cs])
                g = self.buf('dlo%d' % k, self._bshape(l + 1) + [Cl])

                def c0_dgrads(dz_, ds_, dl_, c0=c0, Cs=Cs, Cl=Cl):
                    ops.conv3d(dz_, c0['wpd_s'], None, Cs, 0, out=ds_)
                    ops.conv3d_up_dgrad(dz_, c0['wpd_u'], Cl, out=dl_)
                self._pb(c0_dgrads, dz, dskips[l], g)
            else:
                g = self._convs_backward(g, None, d['convs'], acts, self.saved['cat'][k], need_dx=True, tag='d%d' % k,
                                         dacts=dacts)
                # g = d(concat)
                Cl = g.shape[3] - Cs
                dskips[l], g = ops.upsample_concat_bwd(g, Cs, Cl, dskip=self.buf('dskip%d' % l, self._bshape(l) + [Cs]),
                                                       dlo=self.buf('dlo%d' % k, self._bshape(l + 1) + [Cl]))
            if on_grad_ready is not None:
                self._join()
                on_grad_ready(self.offsets[d['convs'][0]['w']][0])
        for l in range(L - 1, -1, -1):
            e = self.enc[l]
            acts = self.saved['enc'][l]
            ps = self._drop_ps is not None
            dacts = self.saved['encd'][l] if ps else acts
            if l < L - 1:
                # the pool backward also emits the sums of this level's BatchNorm backward (its output is the BN-output
                # gradient): no separate reduction pass
                off = self.offsets[e['bn']['beta']][0]
                sums = None if frozen else self.grads[off:off + 2 * e['bn']['C']]
                if frozen and self.frozen_batch_stats:  # batch-statistics BatchNorm: its backward needs the two sums
                    sums = self._frozen_sums[:2 * e['bn']['C']]
                    sums.zero_()
                if self.fuse_pool_bwd and not frozen and not ps:
                    # only the BatchNorm-backward sums now; the routed gradient (7/8 zeros) is never written: the fused
                    # pool + BatchNorm + ELU backward re-derives it from the pooled gradient (ops.bn_pool_elu_bwd)
                    ops.bn_maxpool_bwd(g, acts[-1], self._stats(e['bn']), self.view(e['bn']['gamma']),
                                       self.view(e['bn']['beta']), out=False, sums=sums)
                    self._pending_bn = (e['bn'], sums, 'pooled')
                else:
                    g = ops.bn_maxpool_bwd(g, dacts[-1], self._stats(e['bn']), self.view(e['bn']['gamma']),
                                           self.view(e['bn']['beta']), out=self.buf('gpool%d' % l, list(acts[-1].shape)),
                                           sums=sums)
                    self._pending_bn = (e['bn'], self._zero_sums[:2 * e['bn']['C']] if (frozen and sums is None) else sums)
            else:
                g = self._bn_backward(g, dacts[-1], e['bn'])
            g = self._convs_backward(g, dskips[l], e['convs'], acts, self.saved['x'][l], need_dx=(l > 0) or frozen,
                                     tag='e%d' % l, dacts=dacts)
            if on_grad_ready is not None:
                self._join()
                on_grad_ready(self.offsets[e['convs'][0]['w']][0])
        self._join()
        return g if frozen else G

    # ---- weight gradients on a second stream: wgrad(layer) and dgrad(layer) both only read dz, so they can share
    # the GPU; on the small deep levels neither fills 256 CUs alone.  _join() before anything overwrites dz / reads grads.
    def _fork(self, fn, nvox=0):
        # deterministic mode: the ordered weight-gradient planes are ONE buffer for the process (csrc/conv3d.hip:
        # syn_det_prepare, "one stream" rule) -- a second stream's launch would memset / accumulate into them concurrently
        if not self.overlap_wgrad or nvox > self.overlap_max_voxels or ops._deterministic:
            fn()
            return
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self._side_stream):
            self._side_stream.wait_event(ev)
            fn()
        self._side_busy = True

    def _join(self):
        if self._side_busy:
            torch.cuda.current_stream().wait_stream(self._side_stream)
            self._side_busy = False

    def _bn_backward(self, g, x, bn):
        """pass 1 (channel sums = dbeta | dgamma); pass 2 is fused into the ELU backward of the conv that produced x"""
        if g is None:  # rank-1 head gradient: head_bwd already produced the sums (backward())
            return None
        if getattr(self, '_frozen', False):
            if self.frozen_batch_stats:     # batch-statistics BatchNorm of a frozen network: the full backward, no parameter gradients
                sums = self._frozen_sums[:2 * bn['C']]
                sums.zero_()
                ops.bn_reduce_bwd(g, x, self._stats(bn), sums)
                self._pending_bn = (bn, sums)
                return g
            # inference-mode BatchNorm: dx = gamma * invstd * dy, i.e. zero batch sums
            self._pending_bn = (bn, self._zero_sums[:2 * bn['C']])
            return g
        off = self.offsets[bn['beta']][0]
        sums = self.grads[off:off + 2 * bn['C']]  # [dbeta | dgamma]
        ops.bn_reduce_bwd(g, x, self._stats(bn), sums)
        self._pending_bn = (bn, sums)
        return g

    def _elu_backward(self, g, y, dy2, dbias, conv=None):
        """ELU backward of a conv output y; consumes a pending BN backward (y was the BN input)"""
        out = self.buf('dz', list(y.shape))
        pend, self._pending_bn = self._pending_bn, None
        if self._drop_ps is not None:  # per-sample dropout sits between y and its consumer
            bn_args = head = None
            if pend is not None:
                assert len(pend) == 2
                bn_args = (self._stats(pend[0]), self.view(pend[0]['gamma']), pend[1])
                if g is None:
                    head = self._rank1
            return ops.elu_bwd_drop(g, y, self._drop_ps[conv['name']], dy2=dy2, dbias=dbias, out=out, bn=bn_args, head=head)
        if pend is not None:
            bn, sums = pend[:2]
            if len(pend) > 2:  # g is the gradient w.r.t. the POOLED tensor: pool + BatchNorm + ELU backward in one pass
                return ops.bn_pool_elu_bwd(g, y, self._stats(bn), self.view(bn['gamma']), self.view(bn['beta']), sums,
                                           dy2=dy2, dbias=dbias, out=out)
            if g is None:  # rank-1 gradient of the head
                dpred, whead = self._rank1
                assert dy2 is None
                return ops.bn_elu_bwd_head(dpred, whead, y, self._stats(bn), self.view(bn['gamma']), sums, dbias=dbias,
                                           out=out)
            return ops.bn_elu_bwd(g, y, self._stats(bn), self.view(bn['gamma']), sums, dy2=dy2, dbias=dbias, out=out)
        return ops.elu_bwd(g, y, dy2=dy2, dbias=dbias, out=out)

    def _convs_backward(self, g, g2, convs, acts, x_in, need_dx, tag, elu_below=None, below_conv=None, dacts=None):
        """g (+g2) = gradient w.r.t. the output of the last conv's ELU. Returns gradient w.r.t. x_in (or None);
        with elu_below = x_in being itself an ELU output (of below_conv), the returned gradient is w.r.t. the pre-activation
        of x_in.  dacts: what the next layer read of each conv output (per-sample dropout: the dropped-out copies)."""
        fused = False  # g already is dz of conv j (ELU backward applied in the data-gradient epilogue of conv j+1)
        dacts = acts if dacts is None else dacts
        ps = self._drop_ps is not None
        for j in range(len(convs) - 1, -1, -1):
            c = convs[j]
            y = acts[j]
            xin = dacts[j - 1] if j > 0 else x_in
            self._join()  # the previous layer's wgrad still reads the buffer dz is about to reuse
            frozen = getattr(self, '_frozen', False)
            if fused:
                dz = g
                if not frozen:
                    self._fork(lambda: self._pb(lambda x_, dz_: ops.conv3d_wgrad(
                        x_, dz_, self.view(c['w'], self.grads), dbias=self.view(c['b'], self.grads)), xin, dz),
                        xin[..., 0].numel())
            else:
                dz = self._elu_backward(g, y, g2, None if frozen else self.view(c['b'], self.grads), conv=c)
                if not frozen:
                    self._fork(lambda: self._pb(lambda x_, dz_: ops.conv3d_wgrad(x_, dz_, self.view(c['w'], self.grads)),
                                                xin, dz), xin[..., 0].numel())
            g2 = None
            fused = False
            if j > 0 or need_dx:
                out = self.buf('dx_%s_%d' % (tag, j & 1), list(xin.shape))
                # the ELU backward of the layer below (no BatchNorm in between) rides in the epilogue: one pass over
                # the activation instead of three; its dbias comes out of the weight-gradient GEMM
                below = acts[j - 1] if j > 0 else elu_below
                if below is not None:
                    self._pb(lambda dz_, b_, o_: ops.conv3d_add(dz_, c['wpd'], None, b_, c['cin'], 2, out=o_), dz, below, out)
                    if ps:  # the conv read s_b * ELU(below): the factor of the layer below's dropout on its gradient
                        ops.scale_channels(out, self._drop_ps[(convs[j - 1] if j > 0 else below_conv)['name']], out=out)
                    fused = True
                else:
                    self._pb(lambda dz_, o_: ops.conv3d(dz_, c['wpd'], None, c['cin'], 0, out=o_), dz, out)
                g = out
            else:
                g = None
        return g

    # ------------------------------------------------------------------ optimizer (keras.optimizers.Adam, 2.3.1)
    def adam_step(self, lr=1e-4, decay=0.0, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        if decay > 0:
            lr = lr * (1.0 / (1.0 + decay * self.iterations))
        t = self.iterations + 1
        lr_t = lr * (math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t))
        ops.adam_step(self.params, self.grads, self.adam_m, self.adam_v, lr_t, beta1, beta2, eps, grad_scale)
        self.iterations = t
        self._drop_forwards = 0   # the first forward of an iteration: the mask of (seed, t) alone (resume-exact)
        self.repack()

    def update_moving_stats(self):
        """K.moving_average_update with momentum .99; variance gets Keras' n/(n-(1+eps)) correction"""
        m = self.bn_momentum
        batch = self.bn_true if self._drop is not None else self.bn_batch  # dropout: statistics of the dropped-out tensor
        self.bn_moving.mul_(m).add_(batch * self.bn_corr, alpha=1.0 - m)


def unet(nb_features, input_shape, nb_levels, conv_size, nb_labels, name='unet', prefix=None, feat_mult=1,
         pool_size=2, use_logp=True, padding='same', dilation_rate_mult=1, activation='elu', skip_n_concatenations=0,
         use_residuals=False, final_pred_activation='softmax', nb_conv_per_level=1, add_prior_layer=False,
         layer_nb_feats=None, conv_dropout=0, batch_norm=None, input_model=None, device=None, seed=0,
         fold_upsample='auto', dtype='f32'):
    """ext/neuron/models.py:26-47 signature.  Unsupported knobs of the over-parametrised reference raise."""
    if pool_size != 2 or padding != 'same' or dilation_rate_mult != 1 or skip_n_concatenations != 0 or \
            use_residuals or add_prior_layer or layer_nb_feats is not None:
        raise NotImplementedError('only the configuration SynthSR.training uses is supported '
                                  '(pool 2, same padding, no dilation/residuals/prior)')
    net = UNet3D(nb_features, input_shape, nb_levels, conv_size, nb_labels, name=name, prefix=prefix,
                 feat_mult=feat_mult, nb_conv_per_level=nb_conv_per_level, batch_norm=batch_norm,
                 activation=activation, device=device, seed=seed, final_pred_activation=final_pred_activation,
                 fold_upsample=fold_upsample, dtype=dtype, conv_dropout=conv_dropout)
    net.input_model = input_model
    return net
