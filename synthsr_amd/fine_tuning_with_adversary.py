"""Adversarial fine-tuning of the synthesis U-Net: `training()` with the parameters and schedule of
SynthSR/fine_tuning_with_adversary.py:37-479 on the HIP kernels.

Per outer step: `first_training_ratio` (very first step) or `training_ratio` critic updates, each on a freshly generated
sample - critic loss -D(target) + D(G(x)) + gp (1 - ||grad D(x_hat)||)^2 (critic.py) - then one generator update on another
sample with loss (1 - w_d [- w_dice]) L1 + w_d mean(-D(G(x))) [+ w_dice Dice] (`build_generator_loss`, :511-576).  Both
networks use Keras-semantics Adam.  Per epoch: `logs/{discriminator,generator}_loss.npy` and the two networks as
`generator_<epoch>.h5|.npz`, `discriminator_<epoch>.h5` (Keras weight layout).

One GPU; the critic's convolutions run on the generic kernels (critic.py).  Keras details that are third-party and unpinned here: BatchNormalization of the frozen generator inside the
critic update uses batch statistics (training phase, Keras 2.3.1) and does not move the moving averages."""
import os
import time

import numpy as np

from . import host_math as hm
from . import volumes
from .brain_generator import BrainGenerator
from .critic import Critic3D
from .training import load_checkpoint, save_checkpoint, settle_host_gc
from .unet import unet as build_unet


def make_discriminator(input_shape, n_filters=32, n_levels=4, mask_input=False, device=None, seed=0, dtype='f32'):
    """the critic network of fine_tuning_with_adversary.py:482-502 as a `Critic3D`; with mask_input the volumes are
    multiplied by a mask before the first layer - pass it as `mask=` to the critic's methods"""
    critic = Critic3D(input_shape, n_filters=n_filters, n_levels=n_levels, device=device, seed=seed, dtype=dtype)
    critic.mask_input = bool(mask_input)
    return critic


class AdversarialTrainer:
    """generator (U-Net) + critic for one rank: `critic_step()` / `generator_step()` each draw a new training sample.
    distributed=True (one process per GPU, batch 1 each): every rank draws its own sample; the critic's gradient buffer
    (134.6 M parameters at 160^3 = 538 MB) is all-reduced in 64 MB buckets after its backward, the U-Net's
    through the tail-first bucketed reducer hooked into its backward (training.GradBucketReducer); both optimizers apply
    1 / world.  The gradient penalty is per sample (each rank's own x_hat), as in a Keras batch."""

    def __init__(self, brain_generator, net, critic, lr_generator=1e-4, lr_discriminator=1e-4, lr_decay=0.0,
                 relative_weight_discriminator=0.01, gradient_penalty_weight=10.0, work_with_residual_channel=None,
                 loss_cropping=None, seg_regulariser=None, rng=None, mask_lut=None, distributed=False,
                 force_allreduce=False):
        from .training import GradBucketReducer
        self.gen_reducer = GradBucketReducer(net.grads, force=force_allreduce) if distributed else None
        self.critic_reducer = GradBucketReducer(critic.grads, bucket_elems=16 * 1024 * 1024,
                                                force=force_allreduce) if distributed else None
        self.bg, self.gen, self.net, self.critic = brain_generator, brain_generator.labels_to_image_model, net, critic
        self.lr_g, self.lr_d, self.lr_decay = lr_generator, lr_discriminator, lr_decay
        self.w_d, self.gp = float(relative_weight_discriminator), float(gradient_penalty_weight)
        self.residual, self.loss_cropping, self.seg = work_with_residual_channel, loss_cropping, seg_regulariser
        self.rng = np.random.default_rng(0) if rng is None else rng
        # labels_to_mask: float LUT label value -> mask value (ConvertLabels(generation_labels, labels_to_mask), :363-365)
        self.mask_lut = None if mask_lut is None else __import__('torch').as_tensor(
            np.asarray(mask_lut, dtype=np.float32)).to(net.device)

    def _mask(self, seg, like):
        from . import ops
        if self.mask_lut is None:
            return None
        if list(seg.shape) != list(like.shape[:3]):
            raise ValueError('labels_to_mask with a target resolution different from the label maps: the mask of the label maps\' grid '
                             'cannot multiply a prediction on another grid (the reference fails with incompatible shapes too)')
        m = ops.lut_gather(seg.contiguous(), self.mask_lut).view(*like.shape[:3], 1)
        # several output channels: the same mask on every channel (Keras broadcasts the Multiply, :365)
        return m if like.shape[3] == 1 else m.expand(*like.shape).contiguous()

    def _generate(self):
        """one batch: (image, target, [label map per item]); batchsize > 1: the items are generated one after the other and
        stacked along the first spatial axis (the layout UNet3D.set_batch works on, as training.Trainer does)"""
        import torch
        inputs = next(self.bg.model_inputs_generator)
        labels, means, stds = inputs[:3]
        B = int(np.asarray(means).shape[0])
        self.net.set_batch(B)
        imgs, tgts, segs = None, None, []
        means_b, stds_b = hm.batch_gmm_parameters(means, stds, getattr(self.gen, 'sum_gmm_over_batch', False))
        for b in range(B):
            real = np.asarray(inputs[3])[b, ..., 0] if getattr(self.gen, 'use_real_image', False) else None
            image, target, seg = self.gen.generate(np.asarray(labels)[b, ..., 0], means_b[b], stds_b[b],
                                                   None, real_image=real)
            if B == 1:
                return image, target, [seg]
            if imgs is None:   # generate() re-uses its output buffers: copy every item out
                imgs = torch.empty((B * image.shape[0],) + tuple(image.shape[1:]), dtype=image.dtype, device=image.device)
                tgts = torch.empty((B * target.shape[0],) + tuple(target.shape[1:]), dtype=target.dtype, device=target.device)
            imgs.chunk(B, 0)[b].copy_(image)
            tgts.chunk(B, 0)[b].copy_(target)
            segs.append(seg.clone())
        return imgs, tgts, segs

    def _forward_generator(self, image, target, want_dpred):
        residual, rs, ro = None, 1, 0
        if self.residual is not None:
            residual, rs, ro = image, image.shape[-1], [int(c) for c in self.residual]
        return self.net.loss(image, target.reshape(-1), 'l1', self.loss_cropping, residual=residual, res_stride=rs,
                             res_off=ro, want_pred=True)

    def critic_step(self):
        """one update of the critic (discriminator_model.train_on_batch, :452-453); returns the critic loss"""
        image, target, segs = self._generate()
        _, pred = self._forward_generator(image, target, False)          # generator frozen: forward only
        B = len(segs)
        fakes, reals = pred.view(*target.shape).chunk(B, 0), target.chunk(B, 0)
        loss = 0.0
        for b in range(B):   # the critic sees one sample at a time (Dense on the flattened volume); loss = batch mean
            u = float(self.rng.uniform())                                  # RandomWeightedAverage, one weight per sample
            l, _, _, _ = self.critic.critic_loss_and_grads(reals[b].contiguous(), fakes[b].contiguous(), u, self.gp,
                                                           mask=self._mask(segs[b], reals[b]), accumulate=b > 0)
            loss += l / B
        scale = self.critic_reducer.reduce_all() if self.critic_reducer is not None else 1.0
        self.critic.adam_step(self.lr_d, self.lr_decay, grad_scale=scale / B)
        return loss

    def generator_step(self):
        """one update of the U-Net (generator_model.train_on_batch, :457-458); returns the generator loss"""
        from . import ops
        net = self.net
        image, target, segs = self._generate()
        B = len(segs)
        l1, pred = self._forward_generator(image, target, True)
        w_dice = self.seg.rel_weight if self.seg is not None else 0.0
        w_l1 = 1.0 - self.w_d - w_dice
        ops.axpby(net.dpred, None, w_l1, 0.0, out=net.dpred)              # d(w_l1 L1)/d pred
        fakes, reals, dpreds = pred.view(*target.shape).chunk(B, 0), target.chunk(B, 0), net.dpred.chunk(B, 0)
        d_fake = 0.0
        for b in range(B):   # d(w_d * mean_b -D(G_b))/d pred, sample by sample
            g_adv = self.critic.input_gradient(fakes[b].contiguous(), -self.w_d / B, mask=self._mask(segs[b], reals[b]))
            d_fake += float(self.critic.last_output.item()) / B
            ops.axpby(dpreds[b], g_adv.reshape(-1), 1.0, 1.0, out=dpreds[b])
        loss = w_l1 * float(l1.item()) - self.w_d * d_fake
        if self.seg is not None:
            seg = segs[0] if B == 1 else __import__('torch').cat(segs, 0)   # stacked like the volumes
            if list(seg.shape) != list(image.shape[:3]):
                raise ValueError('segmentation loss with a target resolution different from the label maps: `segmentation_target` '
                                 'has the label maps\' grid, the posteriors the prediction\'s, and DiceLoss multiplies them voxel by voxel '
                                 '(SynthSR/metrics_model.py:191-207) -- the reference\'s graph fails with incompatible shapes too')
            loss += w_dice * float(self.seg(pred, seg, net.dpred, self.loss_cropping).item())
        if self.gen_reducer is not None:
            self.gen_reducer.start()
            net.backward(on_grad_ready=self.gen_reducer.ready)
            scale = self.gen_reducer.finish()
        else:
            net.backward()
            scale = 1.0
        net.adam_step(self.lr_g, self.lr_decay, grad_scale=scale)
        net.update_moving_stats()
        return loss


def training(labels_dir, images_dir, model_dir, prior_means, prior_stds, path_generation_labels,
             path_segmentation_equivalency=None, segmentation_model_file=None, prior_distributions='normal',
             path_generation_classes=None, FS_sort=True, batchsize=1, input_channels=True, output_channel=None,
             target_res=None, output_shape=None, flipping=True, padding_margin=None, scaling_bounds=0.2, rotation_bounds=20,
             shearing_bounds=0.03, translation_bounds=5, nonlin_std=5., nonlin_shape_factor=0.04,
             simulate_registration_error=False, data_res=None, thickness=None, randomise_res=True, downsample=True,
             blur_range=1.03, build_reliability_maps=False, bias_field_std=.4, bias_shape_factor=0.04, n_levels=5,
             nb_conv_per_level=2, conv_size=3, unet_feat_count=24, feat_multiplier=2, dropout=0, activation='elu',
             lr_decay=0, epochs=100, steps_per_epoch=1000, work_with_residual_channel=None, loss_cropping=None,
             lr_generator=1e-4, lr_discriminator=1e-4, relative_weight_segmentation=0.25,
             relative_weight_discriminator=0.01, checkpoint_generator=None, gradient_penalty_weight=10,
             first_training_ratio=100, training_ratio=10, labels_to_mask=None, seed=0, verbose=True, dtype='f32',
             segnet_frozen_bn='batch'):
    """Parameters as documented in SynthSR/fine_tuning_with_adversary.py:92-283 (+ `seed`, `verbose`, `dtype`: 'bf16' runs
    the conv stacks of the generator U-Net AND the critic in bf16 (fp32 accumulation / statistics / master weights / Dense layers /
    losses): the "mixed bf16" of BASELINE.json configs[4]; `segnet_frozen_bn` as in synthsr_amd.training.training).
    Deviation: `work_with_residual_channel` is APPLIED here (prediction = network output + that input channel, as in
    SynthSR/training.py:260-271); the reference's adversarial script validates the argument (fine_tuning_with_adversary.py:
    256-264) and then never uses it.  Pass None for the reference's behaviour.
    Returns (generator U-Net, critic)."""
    import torch
    n_channels = len(hm.reformat_to_list(input_channels))
    if output_channel is not None:
        output_channel = list(hm.reformat_to_list(output_channel))
    n_output_channels = 1 if output_channel is None else len(output_channel)
    # checks, fine_tuning_with_adversary.py:139-157
    if (images_dir is None) & (output_channel is None):
        raise Exception('please provide a value for output_channel or image_dir')
    elif (images_dir is not None) & (output_channel is not None):
        raise Exception('please provide a value either for output_channel or image_dir, but not both at the same time')
    if output_channel is not None and any(x >= n_channels for x in output_channel):
        raise Exception('indices in output_channel cannot be greater than the total number of channels')
    if work_with_residual_channel is not None:
        work_with_residual_channel = hm.reformat_to_list(work_with_residual_channel)
        if output_channel is not None and len(work_with_residual_channel) != len(output_channel):
            raise Exception('The number or residual channels and output channels must be the same')
        if any(x >= n_channels for x in work_with_residual_channel):
            raise Exception('indices in work_with_residual_channel cannot be greater than the total number of channels')
    if n_output_channels != 1 and segmentation_model_file is not None:
        raise ValueError('the segmentation loss needs ONE regression target (the segmentation network takes a single-channel '
                         'image, SynthSR/training.py:375)')
    # data parallel like training(): one process per GPU (torchrun), batch 1 per rank, per-rank random streams
    dist_on = 'RANK' in os.environ and int(os.environ.get('WORLD_SIZE', '1')) > 1
    rank, world = 0, 1
    if dist_on:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
            dist.init_process_group('nccl')
        rank, world = dist.get_rank(), dist.get_world_size()

    generation_labels, n_neutral_labels = volumes.get_list_labels(label_list=path_generation_labels, labels_dir=labels_dir,
                                                                  FS_sort=FS_sort)
    if rank == 0:
        os.makedirs(os.path.join(model_dir, 'logs'), exist_ok=True)
    if loss_cropping == 0:
        padding_margin, loss_cropping = None, None
    elif padding_margin is None:
        padding_margin = hm.get_padding_margin(output_shape, loss_cropping)
    rng = np.random.Generator(np.random.Philox(key=(int(seed) << 20) + rank))
    brain_generator = BrainGenerator(
        labels_dir=labels_dir, images_dir=images_dir, generation_labels=generation_labels,
        n_neutral_labels=n_neutral_labels, padding_margin=padding_margin, batchsize=batchsize,
        input_channels=input_channels, output_channel=output_channel, target_res=target_res, output_shape=output_shape,
        output_div_by_n=2 ** n_levels, generation_classes=path_generation_classes, prior_means=prior_means,
        prior_stds=prior_stds, prior_distributions=prior_distributions, flipping=flipping, scaling_bounds=scaling_bounds,
        rotation_bounds=rotation_bounds, shearing_bounds=shearing_bounds, translation_bounds=translation_bounds,
        nonlin_std=nonlin_std, nonlin_shape_factor=nonlin_shape_factor,
        simulate_registration_error=simulate_registration_error, randomise_res=randomise_res, data_res=data_res,
        thickness=thickness, downsample=downsample, blur_range=blur_range, build_reliability_maps=build_reliability_maps,
        bias_field_std=bias_field_std, bias_shape_factor=bias_shape_factor, rng=rng)
    brain_generator.labels_to_image_model.seed(int(seed), rank)
    unet_input_shape = brain_generator.model_output_shape
    generator = build_unet(nb_features=unet_feat_count, input_shape=unet_input_shape, nb_levels=n_levels,
                           conv_size=conv_size, nb_labels=n_output_channels, feat_mult=feat_multiplier,
                           nb_conv_per_level=nb_conv_per_level, conv_dropout=dropout, final_pred_activation='linear',
                           batch_norm=-1, activation=activation, input_model=brain_generator.labels_to_image_model,
                           seed=seed, dtype=dtype)
    if checkpoint_generator is not None:
        if verbose:
            print('loading', checkpoint_generator)
        load_checkpoint(checkpoint_generator, generator)
    mask_lut = None
    if labels_to_mask is not None:
        mask_lut = hm.get_mapping_lut(generation_labels, hm.load_array_if_path(labels_to_mask))
    critic = make_discriminator(list(unet_input_shape[:-1]) + [n_output_channels], mask_input=mask_lut is not None,
                                seed=seed + 2, dtype=dtype)
    seg_reg = None
    if segmentation_model_file is not None:
        from .seg_loss import SegmentationRegulariser
        equivalency = np.asarray(hm.load_array_if_path(path_segmentation_equivalency))
        seg_net = build_unet(nb_features=unet_feat_count, input_shape=list(unet_input_shape[:-1]) + [1], nb_levels=n_levels,
                             conv_size=conv_size, nb_labels=len(equivalency), feat_mult=feat_multiplier,
                             nb_conv_per_level=nb_conv_per_level, conv_dropout=dropout, final_pred_activation='softmax',
                             batch_norm=-1, activation=activation, input_model=None, seed=seed + 1)
        load_checkpoint(segmentation_model_file, seg_net)
        im = volumes.load_volume(volumes.list_images_in_folder(images_dir)[0], im_only=True)   # :378-380
        seg_reg = SegmentationRegulariser(seg_net, brain_generator.generation_labels, equivalency,
                                          relative_weight_segmentation, m=np.percentile(im, 2), M=np.percentile(im, 98),
                                          frozen_bn=segnet_frozen_bn)
    if dist_on:                       # identical replicas: rank 0's initial / loaded weights everywhere
        dist.broadcast(generator.params, 0)
        dist.broadcast(generator.bn_moving, 0)
        dist.broadcast(critic.params, 0)
        generator.set_dropout_seed(int(seed) + 0x5eed + 7919 * rank)  # per-rank dropout masks
        generator.repack()
        critic.repack()
    trainer = AdversarialTrainer(brain_generator, generator, critic, lr_generator, lr_discriminator, lr_decay,
                                 relative_weight_discriminator, gradient_penalty_weight, work_with_residual_channel,
                                 loss_cropping, seg_reg, rng=rng, mask_lut=mask_lut, distributed=dist_on)
    width = len(str(epochs))
    log_d, log_g = np.array([]), np.array([])
    gc_settled = False
    for epoch in range(epochs):
        t0 = time.time()
        avg_d = avg_g = 0.0
        for step in range(int(steps_per_epoch)):
            ratio = first_training_ratio if (epoch == 0) & (step == 0) else training_ratio
            for _ in range(ratio):
                avg_d += trainer.critic_step() / (steps_per_epoch * ratio)
            avg_g += trainer.generator_step() / steps_per_epoch
            if not gc_settled:
                settle_host_gc()
                gc_settled = True
        if dist_on:                   # epoch averages over the ranks' samples (a Keras batch of `world` samples)
            import torch
            acc = torch.tensor([avg_d, avg_g], dtype=torch.float64, device=generator.device)
            dist.all_reduce(acc)
            avg_d, avg_g = float(acc[0].item()) / world, float(acc[1].item()) / world
        if not (np.isfinite(avg_d) and np.isfinite(avg_g)):
            raise FloatingPointError('Loss not finite')
        if rank != 0:
            continue
        if verbose:
            print('Epoch {0:0{1}d}/{2}   discriminator loss {3:.5f}   generator loss {4:.5f}   {5:.1f}s'.format(
                epoch + 1, width, epochs, avg_d, avg_g, time.time() - t0))
        log_d, log_g = np.append(log_d, avg_d), np.append(log_g, avg_g)
        np.save(os.path.join(model_dir, 'logs', 'discriminator_loss.npy'), log_d)
        np.save(os.path.join(model_dir, 'logs', 'generator_loss.npy'), log_g)
        tag = '{0:0{1}d}'.format(epoch + 1, width)
        save_checkpoint(os.path.join(model_dir, 'generator_%s.h5' % tag), generator)
        save_checkpoint(os.path.join(model_dir, 'generator_%s.npz' % tag), generator)
        from .keras_h5 import save_keras_weights
        save_keras_weights(os.path.join(model_dir, 'discriminator_%s.h5' % tag),
                           {k: v.numpy() for k, v in critic.state_dict().items()})
    return generator, critic
