"""ORACLE (test infrastructure, never shipped): plain PyTorch-CPU float32 restatement of the U-Net +
L1 loss + Keras Adam the reference builds through Keras (ext/neuron/models.py:256-498 conv_enc/conv_dec,
SynthSR/metrics_model.py:102-104, SynthSR/training.py:444).

PARITY UNPINNED against TensorFlow/Keras itself: Conv3D / BatchNormalization / MaxPooling3D /
UpSampling3D / Adam are third-party code that is neither vendored in /root/reference nor installable
here.  This file restates their documented Keras 2.3.1 semantics with torch ops and
autograd and serves as the independent float32 implementation the HIP kernels are compared against,
and as the U-Net half of bench.py's cpu_baseline.  Only tests/, smoke() and bench.py import it.
"""
import math
import torch
import torch.nn.functional as F

BN_EPS = 1e-3


def to_ncdhw(x):  # [d0,d1,d2,C] -> [1,C,d0,d1,d2]
    return x.permute(3, 0, 1, 2).unsqueeze(0)


def from_ncdhw(x):
    return x[0].permute(1, 2, 3, 0).contiguous()


def conv3d_same(x, w, b=None):
    """x [d0,d1,d2,Cin]; w Keras layout [3,3,3,Cin,Cout] (cross-correlation, zero 'same' padding)"""
    wt = w.permute(4, 3, 0, 1, 2)
    return from_ncdhw(F.conv3d(to_ncdhw(x), wt, b, padding=1))


def batchnorm_train(x, gamma, beta, eps=BN_EPS):
    """training-mode BatchNormalization(axis=-1): biased batch variance; returns (y, mean, var)"""
    C = x.shape[-1]
    flat = x.reshape(-1, C)
    mean = flat.mean(0)
    var = flat.var(0, unbiased=False)
    y = (x - mean) * torch.rsqrt(var + eps) * gamma + beta
    return y, mean, var


def maxpool2(x):
    return from_ncdhw(F.max_pool3d(to_ncdhw(x), 2))


def upsample2(x):
    return x.repeat_interleave(2, 0).repeat_interleave(2, 1).repeat_interleave(2, 2)


def unet_forward(x, P, prefix, nb_levels, nconv, training=True, moving=None, collect=None):
    """P: dict name -> tensor with the Keras layer names (`<prefix>_conv_downarm_l_k/kernel`, ...).
    Returns prediction [d0,d1,d2,1].  `collect` (dict) receives batch statistics per BN layer."""
    L = nb_levels

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
            cur = F.elu(conv3d_same(cur, P[nm + '/kernel'], P[nm + '/bias']))
        skips.append(cur)  # pre-BN skip (models.py:431-432)
        cur = bn(cur, '%s_bn_down_%d' % (prefix, l))
        if l < L - 1:
            cur = maxpool2(cur)
    for k in range(L - 1):
        l = L - 2 - k
        cur = torch.cat([skips[l], upsample2(cur)], -1)  # concatenate([skip, up]) models.py:434
        for j in range(nconv):
            nm = '%s_conv_uparm_%d_%d' % (prefix, L + k, j)
            cur = F.elu(conv3d_same(cur, P[nm + '/kernel'], P[nm + '/bias']))
        cur = bn(cur, '%s_bn_up_%d' % (prefix, k))
    w, b = P['%s_likelihood/kernel' % prefix], P['%s_likelihood/bias' % prefix]
    return cur @ w + b


def l1_loss(pred, target):
    return (pred - target).abs().mean()


def adam_keras(p, g, m, v, t, lr=1e-4, b1=0.9, b2=0.999, eps=1e-7, decay=0.0):
    """keras.optimizers.Adam (2.3.1) update; t = iterations + 1. Returns (p, m, v)"""
    if decay > 0:
        lr = lr * (1.0 / (1.0 + decay * (t - 1)))
    lr_t = lr * (math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t))
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    return p - lr_t * m / (v.sqrt() + eps), m, v
