"""ORACLE (test infrastructure, never shipped): plain PyTorch-CPU float32 restatement of the U-Net +
L1 loss + Keras Adam the reference builds through Keras (ext/neuron/models.py:256-498 conv_enc/conv_dec,
SynthSR/metrics_model.py:102-104, SynthSR/training.py:444).

PARITY UNPINNED against TensorFlow/Keras itself: Conv3D / BatchNormalization / MaxPooling3D /
UpSampling3D / Adam are third-party code that is neither vendored in /root/reference nor installable
here.  This file restates their documented Keras 2.3.1 semantics with torch ops and
autograd and serves as the independent float32 implementation the HIP kernels are compared against,
and as the U-Net half of bench.py's cpu_baseline.  Only tests/, smoke() and bench.py import it.
"""
import contextlib
import math
import torch
import torch.nn.functional as F

BN_EPS = 1e-3

_COMPUTE = None   # None: compute in the dtype of the tensors handed in (float32 in every parity test)


@contextlib.contextmanager
def compute_dtype(dtype):
    """Inside this context unet_forward / seg_regularisation evaluate in `dtype` (torch.float64) whatever they are handed:
    inputs and parameters are cast on entry, so autograd still delivers `.grad` on the caller's float32 leaves (the float64
    gradient rounded once).  The parity tests use it as their yardstick: the device's distance from the float64 result is
    bounded by the distance of this same oracle evaluated in float32 (tests/conftest.py: single_shot_parity)."""
    global _COMPUTE
    prev, _COMPUTE = _COMPUTE, dtype
    try:
        yield
    finally:
        _COMPUTE = prev


_PRED_TAP = None   # None or a callable applied to the linear output of unet_forward (not to softmax heads)


@contextlib.contextmanager
def prediction_tap(fn):
    """Inside this context every unet_forward with a LINEAR head passes its output through fn (out -> out) before returning
    it.  The parity tests use it to read d(loss)/d(prediction) of the oracle (retain_grad) and to move the prediction of single
    voxels by a few ulp: the losses have kinks at the prediction level (|pred - target| of L1 / Laplace, the clip of the
    segmentation loss), and on which side of a kink a voxel sits whose argument is within float32 rounding of it is not a
    property of the algorithm (tests/conftest.py: single_shot_parity, exactly as for max-pooling ties)."""
    global _PRED_TAP
    prev, _PRED_TAP = _PRED_TAP, fn
    try:
        yield
    finally:
        _PRED_TAP = prev


def _cd(t):
    if _COMPUTE is None or t is None or not torch.is_tensor(t) or not t.is_floating_point():
        return t
    return t.to(_COMPUTE)


def to_ncdhw(x):  # [d0,d1,d2,C] -> [1,C,d0,d1,d2];  a batch [B,d0,d1,d2,C] -> [B,C,d0,d1,d2]
    return x.permute(0, 4, 1, 2, 3) if x.dim() == 5 else x.permute(3, 0, 1, 2).unsqueeze(0)


def from_ncdhw(x, batched=False):
    return x.permute(0, 2, 3, 4, 1).contiguous() if batched else x[0].permute(1, 2, 3, 0).contiguous()


def conv3d_same(x, w, b=None):
    """x [d0,d1,d2,Cin] (or a batch [B,d0,d1,d2,Cin]); w Keras layout [3,3,3,Cin,Cout] (cross-correlation, zero 'same'
    padding)"""
    wt = w.permute(4, 3, 0, 1, 2)
    return from_ncdhw(F.conv3d(to_ncdhw(x), wt, b, padding=1), x.dim() == 5)


def batchnorm_train(x, gamma, beta, eps=BN_EPS):
    """training-mode BatchNormalization(axis=-1): biased batch variance; returns (y, mean, var)"""
    C = x.shape[-1]
    flat = x.reshape(-1, C)
    mean = flat.mean(0)
    var = flat.var(0, unbiased=False)
    y = (x - mean) * torch.rsqrt(var + eps) * gamma + beta
    return y, mean, var


def maxpool2(x):
    return from_ncdhw(F.max_pool3d(to_ncdhw(x), 2), x.dim() == 5)


def upsample2(x):
    o = x.dim() - 4  # 1 for a batch
    return x.repeat_interleave(2, o).repeat_interleave(2, o + 1).repeat_interleave(2, o + 2)


class _RoundBF16(torch.autograd.Function):
    """y = bf16(x) (round to nearest even) in the forward pass, the gradient is rounded the same way on its way back:
    the storage roundings of the bf16 network (synthsr_amd UNet3D(dtype='bf16')) restated for the fp32 oracle"""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def round_bf16(x):
    return _RoundBF16.apply(x)


def _drop_factor(s, t):
    """per-feature dropout factor [C], or one per sample and feature [B, C] for a batch t [B, d0, d1, d2, C]
    (KL.Dropout(noise_shape=[None, 1, 1, 1, C]), ext/neuron/models.py:320-324)"""
    s = torch.as_tensor(s, dtype=t.dtype)
    return s.reshape(s.shape[0], 1, 1, 1, s.shape[1]) if s.dim() == 2 else s


def _dropped(t, s, q):
    """the dropped-out tensor s * t.  One mask per SAMPLE (s [B, C], batchsize > 1): the bf16 network stores that tensor (the
    factor cannot ride on weights shared by the batch), so its storage rounding `q` applies; a per-feature s [C] is folded into
    the next layer's weights there and never stored"""
    f = _drop_factor(s, t)
    return q(t * f) if f.dim() > 1 else t * f


def unet_forward(x, P, prefix, nb_levels, nconv, training=True, moving=None, collect=None, softmax=False, quant=None,
                 dropout=None, pool_inputs=None, pool_nudge=None):
    """P: dict name -> tensor with the Keras layer names (`<prefix>_conv_downarm_l_k/kernel`, ...).
    Returns prediction [d0,d1,d2,1].  `collect` (dict) receives batch statistics per BN layer.
    quant (optional, e.g. round_bf16): applied wherever the bf16 network stores a tensor -- the input, every conv kernel,
    every conv + ELU output, every pooled tensor and every concatenated tensor (BatchNorm, head and loss stay fp32).
    dropout (optional): conv layer name -> per-channel factor (0 or 1/(1-rate)) of the feature-wise KL.Dropout that
    follows that conv (noise_shape [None,1,1,1,C], ext/neuron/models.py:320-324, 448-451); the skip connection reads the
    conv layer's own output, i.e. the tensor BEFORE the dropout (models.py:431-432).
    pool_inputs (optional list): receives the tensor each MaxPooling3D reads (detached).  pool_nudge (optional list, one
    entry per pooled level, None or a constant tensor of that shape) is ADDED to that tensor before the pooling: the parity
    tests use it to make this oracle break an exact-rounding TIE between two candidates of a 2x2x2 window the way the device
    did (a few ulp on one element of an identified window; tests/conftest.py: align_pool_ties) -- max-pooling is
    discontinuous, and which of two values within float32 rounding of each other wins is not a property of the algorithm."""
    L = nb_levels
    if _COMPUTE is not None:
        x, P = _cd(x), {k: _cd(v) for k, v in P.items()}
        moving = None if moving is None else {k: _cd(v) for k, v in moving.items()}
        pool_nudge = None if pool_nudge is None else [_cd(n) for n in pool_nudge]
    if quant is not None:
        x = quant(x)
        P = {k: (quant(v) if k.endswith('/kernel') and 'likelihood' not in k else v) for k, v in P.items()}
    q = (lambda t: t) if quant is None else quant

    def bn(t, name):
        if training:
            y, m, v = batchnorm_train(t, P[name + '/gamma'], P[name + '/beta'])
            if collect is not None:
                collect[name] = (m.detach(), v.detach())
            return y
        m, v = moving[name + '/moving_mean'], moving[name + '/moving_variance']
        return (t - m) * torch.rsqrt(v + BN_EPS) * P[name + '/gamma'] + P[name + '/beta']

    skips = []
    cur = x
    for l in range(L):
        for k in range(nconv):
            nm = '%s_conv_downarm_%d_%d' % (prefix, l, k)
            pre = q(F.elu(conv3d_same(cur, P[nm + '/kernel'], P[nm + '/bias'])))
            cur = _dropped(pre, dropout[nm], q) if dropout is not None else pre
        skips.append(pre)  # pre-BN (and pre-dropout) skip: the conv layer's output (models.py:431-432)
        cur = bn(cur, '%s_bn_down_%d' % (prefix, l))
        if l < L - 1:
            if pool_inputs is not None:
                pool_inputs.append(cur.detach())
            if pool_nudge is not None and pool_nudge[l] is not None:
                cur = cur + pool_nudge[l]
            cur = q(maxpool2(cur))
    for k in range(L - 1):
        l = L - 2 - k
        cur = q(torch.cat([skips[l], upsample2(cur)], -1))  # concatenate([skip, up]) models.py:434
        for j in range(nconv):
            nm = '%s_conv_uparm_%d_%d' % (prefix, L + k, j)
            cur = q(F.elu(conv3d_same(cur, P[nm + '/kernel'], P[nm + '/bias'])))
            if dropout is not None:
                cur = _dropped(cur, dropout[nm], q)
        cur = bn(cur, '%s_bn_up_%d' % (prefix, k))
    w, b = P['%s_likelihood/kernel' % prefix], P['%s_likelihood/bias' % prefix]
    out = cur @ w.reshape(w.shape[-2], w.shape[-1]) + b
    if _PRED_TAP is not None and not softmax:
        out = _PRED_TAP(out)
    return torch.softmax(out, -1) if softmax else out  # final_pred_activation (models.py:487-494)


def dice_loss(gt, pred, eps=1e-7):
    """ext/lab2im/layers.py DiceLoss.call (:1334-1378) with enable_checks=False, no class / boundary weights:
    gt, pred [..., K] -> mean_k (1 - (2 sum gt pred + eps) / (sum gt^2 + pred^2 + eps))"""
    ax = tuple(range(gt.dim() - 1))
    top = (2 * gt * pred).sum(ax)
    bottom = (gt * gt + pred * pred).sum(ax)
    return (1 - (top + eps) / (bottom + eps)).mean()


def seg_regularisation(pred_image, seg_target, Pseg, prefix, nb_levels, nconv, generation_labels, label_equivalency,
                       m=None, M=None, fs_header=False, loss_cropping=None, pool_inputs=None, pool_nudge=None,
                       bn_batch_stats=False, dropout=None):
    """SynthSR/metrics_model.py:136-215 (add_seg_loss_to_model) for one volume: the predicted image [d0,d1,d2] is
    normalised (:152-155), optionally permuted / flipped to the FreeSurfer orientation (:158-163), pushed through the
    FROZEN segmentation U-Net (softmax head; BatchNorm on its moving averages, or with bn_batch_stats on batch statistics) and
    compared with the generator's label map by the soft Dice over the generation labels that have an equivalent
    (:187-207).  NB the reference builds the ground-truth one-hot as `segmentation_target == i` with i the INDEX of the
    generation label (:191), not its value; mirrored here.  Returns the Dice loss (scalar tensor)."""
    x = _cd(pred_image)
    if m is not None:
        x = (torch.clamp(x, m, M) - m) / (M - m)
    x = x[..., None]
    if fs_header:
        x = torch.flip(x.permute(0, 2, 1, 3), dims=[1])
    # bn_batch_stats: the frozen network's BatchNormalization layers use the statistics of the current activations (Keras
    # 2.3.1 with the learning phase set, `trainable = False` notwithstanding) instead of their moving averages
    # dropout: factors of the frozen network's own Dropout layers (it is built with conv_dropout=dropout, training.py:381,
    # and the learning phase switches them on whatever `trainable` says)
    probs = unet_forward(x, Pseg, prefix, nb_levels, nconv, training=bool(bn_batch_stats), moving=Pseg, softmax=True,
                         pool_inputs=pool_inputs, pool_nudge=pool_nudge, dropout=dropout)
    if fs_header:
        probs = torch.flip(probs, dims=[1]).permute(0, 2, 1, 3)
    if loss_cropping is not None:  # :166-183: posteriors and label map cropped to the centred box
        size = [int(loss_cropping)] * 3 if not hasattr(loss_cropping, '__len__') else [int(v) for v in loss_cropping]
        b = [int((s - c) / 2) for s, c in zip(seg_target.shape, size)]
        sl = tuple(slice(b[i], b[i] + size[i]) for i in range(3))
        probs, seg_target = probs[sl], seg_target[sl]
    gts, preds = [], []
    eq = torch.as_tensor(label_equivalency)
    for i, lab in enumerate(generation_labels):
        idx = torch.nonzero(eq == int(lab)).reshape(-1)
        if len(idx) > 0:
            if len(idx) > 3:
                raise Exception("uuummm weird that you're merging so many labels...")
            gts.append((seg_target == i).to(probs.dtype))
            preds.append(sum(probs[..., int(j)] for j in idx))
    return dice_loss(torch.stack(gts, -1), torch.stack(preds, -1))


def l1_loss(pred, target):
    return (pred - target).abs().mean()


def tf_image_ssim(img1, img2, max_val=1.0, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
    """tf.image.ssim of TensorFlow 2.0 (tensorflow/python/ops/image_ops_impl.py: `ssim` -> `_ssim_per_channel` ->
    `_ssim_helper`, window `_fspecial_gauss`), restated from the published algorithm -- TensorFlow is a third-party
    dependency of the reference (requirements.txt: tensorflow-gpu==2.0.0), not vendored, so this restatement is UNPINNED.
    img1, img2 [..., H, W, C]; returns [...] = mean over channels of the per-channel SSIM."""
    coords = torch.arange(filter_size, dtype=img1.dtype) - (filter_size - 1) / 2.0
    g = coords ** 2 * (-0.5 / (filter_sigma * filter_sigma))
    g = (g.reshape(1, -1) + g.reshape(-1, 1)).reshape(-1)
    kernel = torch.softmax(g, 0).reshape(1, 1, filter_size, filter_size)      # 2-D window, sums to 1
    lead, (H, W, C) = img1.shape[:-3], img1.shape[-3:]

    def reducer(x):                                                             # depthwise_conv2d(..., 'VALID')
        x = x.reshape(-1, H, W, C).permute(0, 3, 1, 2).reshape(-1, 1, H, W)
        y = F.conv2d(x, kernel)
        return y.reshape(-1, C, y.shape[-2], y.shape[-1]).permute(0, 2, 3, 1)
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    mean0, mean1 = reducer(img1), reducer(img2)
    num0 = mean0 * mean1 * 2.0
    den0 = mean0 ** 2 + mean1 ** 2
    luminance = (num0 + c1) / (den0 + c1)
    num1 = reducer(img1 * img2) * 2.0
    den1 = reducer(img1 ** 2 + img2 ** 2)
    cs = (num1 - num0 + c2) / (den1 - den0 + c2)
    ssim_val = (luminance * cs).mean(dim=(1, 2))                               # [-1, C]
    return ssim_val.mean(-1).reshape(lead)


def ssim_loss(pred, target):
    """metrics_model.py:105-125, pred / target [d0,d1,d2,1] (batch of one): the three tf.image.ssim terms exactly as the
    reference composes them -- including `ssim_xz`, whose perm [0,1,3,2,4] only swaps the window axes"""
    if target.shape[-1] > 1:
        raise Exception('SSIM metric does not currently support multiple channels')
    x, y = pred[None], target[None]
    ssim_xy = tf_image_ssim(x, y, 1.0)
    ssim_xz = tf_image_ssim(x.permute(0, 1, 3, 2, 4), y.permute(0, 1, 3, 2, 4), 1.0)
    ssim_yz = tf_image_ssim(x.permute(0, 2, 3, 1, 4), y.permute(0, 2, 3, 1, 4), 1.0)
    return -(1 / 3) * ssim_xy.mean() - (1 / 3) * ssim_xz.mean() - (1 / 3) * ssim_yz.mean()


def regression_loss(pred, target, kind='l1', loss_cropping=None, residual=None):
    """SynthSR/metrics_model.py:27-132.  pred [d0,d1,d2,K] (K = n, or 2n for 'laplace': intensities, spreads), target
    [d0,d1,d2,n], residual [d0,d1,d2,n] or None (added to the intensity channel, :53-64); loss_cropping: sizes of the
    centred box (begin = int((shape - size) / 2), :76-90)."""
    if kind == 'laplace':  # n intensity channels, then their n spread channels (metrics_model.py:31-46)
        n = pred.shape[-1] // 2
        intens, spread = pred[..., :n], pred[..., n:2 * n]
    else:
        intens, spread = pred, None
    if residual is not None:
        intens = intens + residual
    target = target.to(intens.dtype)   # (float64 inside compute_dtype)
    if loss_cropping is not None:
        size = [int(loss_cropping)] * 3 if not hasattr(loss_cropping, '__len__') else [int(v) for v in loss_cropping]
        b = [int((s - c) / 2) for s, c in zip(target.shape[:3], size)]
        sl = tuple(slice(b[i], b[i] + size[i]) for i in range(3))
        intens, target = intens[sl], target[sl]
        spread = None if spread is None else spread[sl]
    if kind == 'ssim':
        return ssim_loss(intens, target)
    err = intens - target
    if kind == 'l1':
        return err.abs().mean()
    if kind == 'l2':
        return (err * err).mean()
    if kind == 'laplace':
        bb = 1e-5 + 0.02 * torch.exp(spread)
        return (torch.log(2 * bb) + err.abs() / bb).mean()
    raise Exception('metrics should either be "l1" or "l2" or "ssim" oro "laplace", got {}'.format(kind))


def adam_keras(p, g, m, v, t, lr=1e-4, b1=0.9, b2=0.999, eps=1e-7, decay=0.0):
    """keras.optimizers.Adam (2.3.1) update; t = iterations + 1. Returns (p, m, v)"""
    if decay > 0:
        lr = lr * (1.0 / (1.0 + decay * (t - 1)))
    lr_t = lr * (math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t))
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    return p - lr_t * m / (v.sqrt() + eps), m, v


# ---------------------------------------------------------------------- WGAN-GP critic (fine_tuning_with_adversary.py)
def critic_forward(x, P, name='discriminator', n_levels=4, alpha=0.2):
    """make_discriminator (SynthSR/fine_tuning_with_adversary.py:482-508) for one volume x [d0,d1,d2,C]: per level a
    stride-1 and a stride-2 Conv3D(3, 'same') each followed by LeakyReLU(.2), Flatten (channels last), Dense + LeakyReLU,
    Dense(1).  TensorFlow's 'same' padding of a stride-2 conv on an even size is (0, 1).  Third-party Keras layers:
    UNPINNED restatement.  Returns the scalar D(x)."""
    h = to_ncdhw(x)
    i = 0
    for _ in range(n_levels):
        for stride in (1, 2):
            w = P['%s_conv_%d/kernel' % (name, i)].permute(4, 3, 0, 1, 2)
            b = P['%s_conv_%d/bias' % (name, i)]
            if stride == 1:
                h = F.conv3d(h, w, b, padding=1)
            else:
                h = F.conv3d(F.pad(h, (0, 1, 0, 1, 0, 1)), w, b, stride=2)
            h = F.leaky_relu(h, alpha)
            i += 1
    f = from_ncdhw(h).reshape(-1)
    h9 = F.leaky_relu(f @ P['%s_dense_0/kernel' % name] + P['%s_dense_0/bias' % name], alpha)
    return (h9 @ P['%s_dense_1/kernel' % name] + P['%s_dense_1/bias' % name]).reshape(())


def critic_loss(real, fake, u_mix, P, name='discriminator', n_levels=4, gp_weight=10.0, mask=None):
    """build_discriminator_loss (:579-595) with RandomWeightedAverage (:604-624) and Gradients (:627-642), batch of one:
    -D(real) + D(fake) + gp_weight (1 - ||grad_x D(x_hat)||_2)^2.  Returns (loss, grad norm); differentiable w.r.t. P."""
    m = 1.0 if mask is None else mask      # make_discriminator(mask_input=True): the network sees x * mask (:485-487)
    d_real = critic_forward(real * m, P, name, n_levels)
    d_fake = critic_forward(fake * m, P, name, n_levels)
    x_hat = (u_mix * real + (1 - u_mix) * fake).detach().requires_grad_(True)
    d_hat = critic_forward(x_hat * m, P, name, n_levels)
    g, = torch.autograd.grad(d_hat, x_hat, create_graph=True)
    norm = torch.sqrt((g * g).sum())
    return -d_real + d_fake + gp_weight * (1 - norm) ** 2, norm
