"""batchsize > 1 per GPU (SynthSR/training.py:52, 344): B volumes stacked along the first spatial axis; BatchNorm statistics
and every reduction run over the whole stack, the convolutions volume by volume (synthsr_amd/unet.py: set_batch).  Checked
against the oracle run on a real batch dimension ([B, d0, d1, d2, C] through torch's conv3d / autograd)."""
import numpy as np
import pytest

from conftest import single_shot_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('fold', [False, True])
@pytest.mark.parametrize('B,feats,levels,shape,cin', [(2, 24, 3, (16, 16, 32), 2), (3, 8, 2, (8, 12, 16), 1), (2, 24, 4, (32, 16, 16), 2)])
def test_batched_unet_vs_oracle(B, feats, levels, shape, cin, fold):
    import torch
    from synthsr_amd.unet import unet
    from oracle import unet_ref as U
    g = torch.Generator()
    tensors = {}

    def run():
        net = unet(nb_features=feats, input_shape=list(shape) + [cin], nb_levels=levels, conv_size=3, nb_labels=1, feat_mult=2,
                   nb_conv_per_level=2, final_pred_activation='linear', batch_norm=-1, activation='elu', seed=3,
                   fold_upsample=fold)
        g.manual_seed(11)
        for nm, v in net.named_parameters():
            if nm.endswith('/gamma'):
                v.copy_(torch.rand(v.shape, generator=g) + .5)
            elif nm.endswith('/beta') or nm.endswith('/bias'):
                v.copy_(torch.randn(v.shape, generator=g) * .1)
        net.repack()
        net.set_batch(B)
        x = torch.rand(B, *shape, cin, generator=g)
        x[1] *= 1.7                                       # the volumes of a batch differ in scale: per-volume stats would show
        target = torch.rand(B, *shape, 1, generator=g)
        xs = x.reshape(B * shape[0], shape[1], shape[2], cin).cuda()
        loss, pred = net.loss_l1(xs, target.reshape(-1).cuda(), want_pred=True)
        net.test_loss, net.test_pred = loss.clone(), pred.clone()
        net.backward()
        tensors.update(x=x, target=target, xs=xs)
        return net

    def oracle(net, nudge):
        P = {nm: v.detach().cpu().clone().requires_grad_(True) for nm, v in net.named_parameters()}
        stats, pin = {}, []
        pr = U.unet_forward(tensors['x'], P, net.prefix, levels, 2, training=True, collect=stats, pool_inputs=pin,
                            pool_nudge=nudge)
        assert list(pr.shape) == [B] + list(shape) + [1]
        lr = U.l1_loss(pr, tensors['target'])
        lr.backward()
        return (P, stats, pr.detach(), lr.detach()), pin

    def compare(net, ref):
        P, stats, pr, lr = ref
        err = (net.test_pred.view(B, *shape, 1).cpu() - pr).abs().max().item() / pr.abs().max().item()
        assert err < 5e-4, err
        assert abs(net.test_loss.item() - lr.item()) < 2e-5 * max(1.0, abs(lr.item()))
        for nm, _, kind in net.specs:
            got = net.view(nm, net.grads).cpu().double()
            ref_g = P[nm].grad.double()
            e = (got - ref_g).abs().max().item() / max(ref_g.abs().max().item(), 1e-12)
            # per-tensor max error relative to the tensor's max-abs; the sums run over B volumes here (more cancelling terms
            # per weight than in the single-volume tests, whose bounds are 2e-3 / 5e-3)
            assert e < (4e-3 if kind in ('kernel', 'head_w') else 8e-3), (nm, e)
        for bn in net.bn_layers:
            o, C = bn['soff'], bn['C']
            m, v = stats[bn['name']]
            assert (net.bn_batch[o:o + C].cpu() - m).abs().max().item() < 1e-4 * max(1.0, m.abs().max().item())
            assert (net.bn_batch[o + C:o + 2 * C].cpu() - v).abs().max().item() < 1e-4 * max(1.0, v.abs().max().item())

    net, _ = single_shot_parity(run, oracle, compare)
    xs = tensors['xs']
    # Keras' sample-variance correction uses the number of values behind the statistics: B * voxels
    l0 = float(B * np.prod(shape))
    o, C = net.bn_layers[0]['soff'], net.bn_layers[0]['C']
    assert abs(net.bn_corr[o + C].item() - l0 / (l0 - (1 + 1e-3))) < 1e-6
    # back to single volumes: same network object, same weights
    net.adam_step(lr=1e-3)
    net.update_moving_stats()
    net.set_batch(1)
    net.training = False
    assert torch.isfinite(net.predict(xs[:shape[0]].contiguous())).all()


def test_batched_bf16_agrees_with_fp32():
    import torch
    from synthsr_amd.unet import unet
    kw = dict(nb_features=24, input_shape=[16, 16, 32, 2], nb_levels=3, conv_size=3, nb_labels=1, feat_mult=2,
              nb_conv_per_level=2, final_pred_activation='linear', batch_norm=-1, activation='elu', seed=4)
    f, h = unet(**kw), unet(dtype='bf16', **kw)
    f.set_batch(2)
    h.set_batch(2)
    x = torch.rand(32, 16, 32, 2).cuda()
    t = torch.rand(32 * 16 * 32).cuda()
    lf, lh = f.loss_l1(x, t)[0], h.loss_l1(x, t)[0]
    assert abs(lf.item() - lh.item()) < 3e-2 * max(1.0, abs(lf.item()))
    f.backward()
    h.backward()
    gf, gh = f.grads.double(), h.grads.double()
    assert ((gf * gh).sum() / (gf.norm() * gh.norm())).item() > 0.97


def test_trainer_and_training_entry_point_with_batchsize_2(tmp_path):
    import os
    import torch
    from synthsr_amd.brain_generator import BrainGenerator
    from synthsr_amd.nifti import write_nifti
    from synthsr_amd.training import Trainer, training
    from synthsr_amd.unet import unet
    from synthsr_amd.synthetic import (synthetic_label_pool, synthetic_label_map, GENERATION_LABELS, GENERATION_CLASSES,
                                       PRIOR_MEANS_T1_HR, PRIOR_STDS_T1_HR)
    pool = synthetic_label_pool(3, (32, 32, 32), 5)
    bg = BrainGenerator(None, PRIOR_MEANS_T1_HR, PRIOR_STDS_T1_HR, 'normal', GENERATION_LABELS,
                        generation_classes=GENERATION_CLASSES, output_shape=32, output_div_by_n=8, nonlin_std=4.,
                        nonlin_shape_factor=.125, bias_shape_factor=.125, build_reliability_maps=True, downsample=True,
                        shearing_bounds=.02, label_maps=pool, batchsize=2, rng=np.random.default_rng(0))
    net = unet(24, bg.model_output_shape, 3, 3, 1, feat_mult=2, nb_conv_per_level=2, final_pred_activation='linear',
               batch_norm=-1, seed=1)
    tr = Trainer(bg, net, lr=1e-3)
    inputs = next(bg.model_inputs_generator)
    assert np.asarray(inputs[0]).shape[0] == 2
    draws = [bg.labels_to_image_model.sample_draws() for _ in range(2)]
    losses = [tr.step(inputs, draws).item() for _ in range(8)]
    assert net.batch == 2 and all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert list(tr._img_b.shape) == [64, 32, 32, 2]
    assert not torch.equal(tr._img_b[:32], tr._img_b[32:])          # two different volumes
    # entry point
    d = tmp_path / 'labels'
    d.mkdir()
    for i in range(2):
        write_nifti(str(d / ('brain%d_labels.nii.gz' % i)), synthetic_label_map((40, 36, 48), 10 + i).astype(np.float32))
    for nm, v in (('gl', GENERATION_LABELS), ('gc', GENERATION_CLASSES), ('pm', PRIOR_MEANS_T1_HR), ('ps', PRIOR_STDS_T1_HR)):
        np.save(tmp_path / (nm + '.npy'), v)
    model_dir = str(tmp_path / 'models')
    net2 = training(str(d), model_dir, str(tmp_path / 'pm.npy'), str(tmp_path / 'ps.npy'), str(tmp_path / 'gl.npy'),
                    path_generation_classes=str(tmp_path / 'gc.npy'), output_shape=32, n_levels=3, unet_feat_count=24,
                    nonlin_shape_factor=.125, bias_shape_factor=.125, steps_per_epoch=3, epochs=1, batchsize=2, verbose=False)
    assert net2.batch == 2 and net2.iterations == 3 and os.path.exists(os.path.join(model_dir, '001.npz'))


@pytest.mark.parametrize('kind,crop', [('l1', (12, 10, 24)), ('l2', (16, 8, 20)), ('ssim', None), ('ssim', (14, 16, 24)),
                                       ('laplace', (12, 10, 24))])
def test_batched_per_volume_losses_vs_oracle(kind, crop):
    """batchsize > 1 (SynthSR/training.py:52) with the losses that are defined per volume: loss_cropping (the centred box of
    EVERY volume, metrics_model.py:70-90) and the slice-wise SSIM (:105-125).  Loss = mean over the batch of the per-volume
    oracle losses; every gradient against autograd through the batched oracle network (deterministic mode, single shot)."""
    import torch
    from synthsr_amd import ops
    from synthsr_amd.unet import unet
    from oracle import unet_ref as U
    B, shape, levels, cin = 2, (16, 16, 32), 2, 2
    K = 2 if kind == 'laplace' else 1
    prev = ops.set_deterministic(True)
    try:
        net = unet(nb_features=8, input_shape=list(shape) + [cin], nb_levels=levels, conv_size=3, nb_labels=K, feat_mult=2,
                   nb_conv_per_level=2, final_pred_activation='linear', batch_norm=-1, activation='elu', seed=3)
        g = torch.Generator().manual_seed(13)
        for nm, v in net.named_parameters():
            if nm.endswith('/gamma'):
                v.copy_(torch.rand(v.shape, generator=g) + .5)
            elif nm.endswith('/beta') or nm.endswith('/bias'):
                v.copy_(torch.randn(v.shape, generator=g) * .1)
        net.repack()
        net.set_batch(B)
        x = torch.rand(B, *shape, cin, generator=g)
        target = torch.rand(B, *shape, 1, generator=g)
        xs = x.reshape(B * shape[0], shape[1], shape[2], cin).cuda()
        loss, pred = net.loss(xs, target.reshape(-1).cuda(), kind, crop, want_pred=True)
        loss, pred = loss.clone(), pred.clone()
        net.backward()
        P = {nm: v.detach().cpu().clone().requires_grad_(True) for nm, v in net.named_parameters()}
        pr = U.unet_forward(x, P, net.prefix, levels, 2, training=True)
        lr = sum(U.regression_loss(pr[b], target[b], kind, crop) for b in range(B)) / B
        lr.backward()
        assert (pred.view(B, *shape, K).cpu() - pr.detach()).abs().max().item() < 5e-4 * pr.abs().max().item()
        assert abs(loss.item() - lr.item()) < 3e-5 * max(1.0, abs(lr.item())), (loss.item(), lr.item())
        for nm, _, kind_ in net.specs:
            got = net.view(nm, net.grads).cpu().double()
            ref = P[nm].grad.double()
            e = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)
            assert e < (4e-3 if kind_ in ('kernel', 'head_w') else 8e-3), (nm, e)
    finally:
        ops.set_deterministic(prev)


@pytest.mark.gpu
@pytest.mark.parametrize('fs_header,crop', [(False, None), (True, (12, 16, 20))])
def test_batched_segmentation_loss_vs_per_volume_oracle(fs_header, crop):
    """segmentation-regularised loss on a stack of two volumes (batchsize 2): the frozen network runs on the stack, the Dice
    is the mean of the per-volume Dice losses.  frozen_bn='inference' (moving averages: the volumes do not interact): Dice
    and d(Dice)/d(prediction) against the oracle evaluated volume by volume under autograd.  frozen_bn='batch': two copies
    of the same volume give the batch the statistics of the single volume -> the single-volume result."""
    import torch
    from synthsr_amd.unet import unet
    from synthsr_amd.seg_loss import SegmentationRegulariser
    from oracle import unet_ref as U
    shape, levels, B = (16, 24, 32), 3, 2
    gen_labels = np.array([0, 14, 2, 3, 41, 42, 17])
    seg_labels = np.array([0, 2, 3, 4, 41, 42, 43, 17, 53])
    equivalency = np.array([0, 2, 3, 3, 41, 42, 42, 17, 17])
    segshape = (shape[0], shape[2], shape[1]) if fs_header else shape
    g = torch.Generator().manual_seed(1)
    segnet = unet(24, list(segshape) + [1], levels, 3, len(seg_labels), feat_mult=2, nb_conv_per_level=2, batch_norm=-1,
                  activation='elu', final_pred_activation='softmax', seed=4)
    segnet.bn_moving.copy_((torch.rand(segnet.bn_moving.shape, generator=g) * 0.5 + 0.25).to(segnet.device))
    preds = [torch.rand(*shape, generator=g) for _ in range(B)]
    segs = [torch.randint(0, len(gen_labels), shape, generator=g, dtype=torch.int32) for _ in range(B)]
    Pseg = {k: v.clone().float() for k, v in segnet.state_dict().items()}
    w = 0.5

    # --- inference-mode BatchNorm: per-volume oracle
    reg = SegmentationRegulariser(segnet, gen_labels, equivalency, w, fs_header=fs_header, frozen_bn='inference')
    pred_b, seg_b = torch.cat(preds, 0).cuda(), torch.cat(segs, 0).cuda()
    dpred = torch.zeros(pred_b.numel(), device='cuda')
    dice = float(reg(pred_b.reshape(-1), seg_b, dpred, crop).item())
    want, grads = 0.0, []
    for p, s in zip(preds, segs):
        pr = p.clone().requires_grad_(True)
        d = U.seg_regularisation(pr, s, Pseg, segnet.prefix, levels, 2, gen_labels, equivalency, fs_header=fs_header,
                                 loss_cropping=crop)
        (w * d / B).backward()
        want += float(d) / B
        grads.append(pr.grad)
    assert abs(dice - want) < 2e-5, (dice, want)
    gref = torch.cat(grads, 0).reshape(-1)
    got = dpred.cpu()
    assert float((got - gref).abs().max()) < 3e-3 * float(gref.abs().max())
    assert float((got * gref).sum() / (got.norm() * gref.norm())) > 0.9999

    # --- batch-statistics BatchNorm: two copies of one volume = that volume alone
    reg = SegmentationRegulariser(segnet, gen_labels, equivalency, w, fs_header=fs_header, frozen_bn='batch')
    one_d = torch.zeros(preds[0].numel(), device='cuda')
    dice1 = float(reg(preds[0].cuda().reshape(-1), segs[0].cuda(), one_d, crop).item())
    two = torch.cat([preds[0], preds[0]], 0).cuda()
    two_d = torch.zeros(two.numel(), device='cuda')
    dice2 = float(reg(two.reshape(-1), torch.cat([segs[0], segs[0]], 0).cuda(), two_d, crop).item())
    assert abs(dice1 - dice2) < 1e-5, (dice1, dice2)
    a, b = two_d.chunk(2), one_d
    # each copy carries half of the single-volume gradient of its own Dice term ... plus the coupling through the shared
    # statistics, which for identical copies adds up to the single-volume gradient split in two
    assert float(((a[0] + a[1]) - b).abs().max()) < 3e-3 * float(b.abs().max())
