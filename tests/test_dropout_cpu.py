"""CPU: the two identities the device path's feature-wise dropout rests on (synthsr_amd/unet.py _start_dropout /
_dropout_bn), checked on the oracle with autograd -- values AND gradients."""
import numpy as np
import torch

from oracle import unet_ref as U


def test_dropout_before_batchnorm_equals_adjusted_variance():
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(6, 5, 7, 8, generator=g) * 3 + 1).double().requires_grad_(True)
    gamma = (torch.rand(8, generator=g) + .5).double()
    beta = torch.randn(8, generator=g).double()
    s = torch.tensor([0, 1 / .7, 1 / .7, 0, 1 / .7, 1 / .7, 1 / .7, 1 / .7]).double()
    up = torch.randn(6, 5, 7, 8, generator=g).double()
    y_ref, m_ref, v_ref = U.batchnorm_train(x * s, gamma, beta)
    (g_ref,) = torch.autograd.grad((y_ref * up).sum(), x)
    # stored x, statistics of x, variance slot var_x + eps (1 / s^2 - 1)  (inf where the feature is dropped)
    x2 = x.detach().clone().requires_grad_(True)
    flat = x2.reshape(-1, 8)
    mean, var = flat.mean(0), flat.var(0, unbiased=False)
    inv2 = torch.where(s > 0, 1 / (s * s).clamp_min(1e-30), torch.full_like(s, float('inf')))
    var_eff = var + U.BN_EPS * (inv2 - 1)
    y = (x2 - mean) * torch.rsqrt(var_eff + U.BN_EPS) * gamma + beta
    (g2,) = torch.autograd.grad((y * up).sum(), x2)
    assert torch.allclose(y, y_ref, atol=1e-12) and torch.allclose(g2, g_ref, atol=1e-12)
    assert torch.allclose(mean * s, m_ref, atol=1e-12) and torch.allclose(var * s * s, v_ref, atol=1e-12)  # moving-average inputs
    assert (y[..., 0] == beta[0]).all() and (g2[..., 0] == 0).all()


def test_dropout_between_convs_equals_scaled_input_channel_weights():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(5, 6, 7, 4, generator=g).double().requires_grad_(True)
    w = torch.randn(3, 3, 3, 4, 3, generator=g).double().requires_grad_(True)
    s = torch.tensor([0., 2., 2., 0.]).double()
    up = torch.randn(5, 6, 7, 3, generator=g).double()
    gx_ref, gw_ref = torch.autograd.grad((U.conv3d_same(x * s, w) * up).sum(), (x, w))
    w_eff = (w.detach() * s[None, None, None, :, None]).requires_grad_(True)
    x2 = x.detach().clone().requires_grad_(True)
    gx, gw_eff = torch.autograd.grad((U.conv3d_same(x2, w_eff) * up).sum(), (x2, w_eff))
    assert torch.allclose(gx, gx_ref, atol=1e-12)
    assert torch.allclose(gw_eff * s[None, None, None, :, None], gw_ref, atol=1e-12)


def test_oracle_dropout_matches_keras_definition():
    """KL.Dropout(rate, noise_shape=[None,1,1,1,C]) in the learning phase: x * keep / (1 - rate), one keep flag per feature;
    the skip connection (the conv layer's own output) is not dropped"""
    g = torch.Generator().manual_seed(2)
    from synthsr_amd.unet import UNet3D
    net = UNet3D(4, [8, 8, 8, 1], 2, 3, 1, feat_mult=2, nb_conv_per_level=2, batch_norm=-1, table_only=True)
    P = {nm: torch.randn(shp, generator=g) * .2 for nm, shp, _ in net.specs}
    x = torch.randn(8, 8, 8, 1, generator=g)
    ones = {c['name']: torch.ones(c['cout']) for grp in net.enc + net.dec for c in grp['convs']}
    a = U.unet_forward(x, P, net.prefix, 2, 2, training=True)
    b = U.unet_forward(x, P, net.prefix, 2, 2, training=True, dropout=ones)
    assert torch.equal(a, b)
    # dropping EVERY feature of the last encoder conv of level 0 leaves the skip path alive: the output still depends on x
    drop = dict(ones)
    drop['unet_conv_downarm_0_1'] = torch.zeros(4)
    c = U.unet_forward(x, P, net.prefix, 2, 2, training=True, dropout=drop)
    d = U.unet_forward(x * 2, P, net.prefix, 2, 2, training=True, dropout=drop)
    assert not torch.allclose(c, d) and not torch.allclose(a, c)


def test_oracle_batch_dimension():
    """the oracle on a batch [B, d0, d1, d2, C]: BatchNorm statistics run over batch and voxels (Keras axis=-1), the rest is
    per volume -- a batch of two copies equals the single volume, and a batch of two DIFFERENT volumes does not"""
    from synthsr_amd.unet import UNet3D
    g = torch.Generator().manual_seed(3)
    net = UNet3D(4, [8, 8, 8, 1], 2, 3, 1, feat_mult=2, nb_conv_per_level=2, batch_norm=-1, table_only=True)
    P = {nm: torch.randn(shp, generator=g) * .2 for nm, shp, _ in net.specs}
    x = torch.randn(8, 8, 8, 1, generator=g)
    one = U.unet_forward(x, P, net.prefix, 2, 2, training=True)
    two = U.unet_forward(torch.stack([x, x]), P, net.prefix, 2, 2, training=True)
    assert list(two.shape) == [2, 8, 8, 8, 1]
    assert torch.allclose(two[0], one, atol=1e-5) and torch.allclose(two[1], one, atol=1e-5)
    mixed = U.unet_forward(torch.stack([x, 3 * x + 1]), P, net.prefix, 2, 2, training=True)
    assert not torch.allclose(mixed[0], one, atol=1e-3)
