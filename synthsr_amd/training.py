"""`training()` — same positional order, keyword names and defaults as SynthSR/training.py:38-89; the
Keras graph {generator -> U-Net -> loss} + `fit_generator` becomes an explicit loop over HIP kernels:

    per step:  host input sampler (model_inputs.py)  ->  generator kernels (labels_to_image_model.py)
               ->  U-Net forward + L1 (unet.py)  ->  backward  ->  [RCCL all-reduce of the flat gradient
               buffer, bucketed and overlapped with the rest of the backward]  ->  Keras-semantics Adam.

Data parallel: one process per GPU (`torchrun`), batch 1 per GPU, per-rank random streams, gradient
average over ranks; BN statistics stay per replica (reference semantics at batch 1).
Checkpoints (rank 0 only): `{epoch:03d}.npz` (weights under the Keras layer names + Adam state, exact resume) and
`{epoch:03d}.h5` (Keras save_weights layout, the reference's file name; keras_h5.py).  `checkpoint=` takes either.
"""
import os
import time
import numpy as np

from . import host_math as hm
from . import volumes
from .brain_generator import BrainGenerator
from .unet import unet as build_unet


def settle_host_gc():
    """Called once by the training loops after their first step.  Everything alive at that point (networks, kernels' argument
    tables, the label pool, torch itself: a few hundred thousand tracked objects) is moved to Python's permanent generation, so
    that the cyclic collector's full passes -- triggered every few thousand allocations of the ctypes launch wrappers -- no
    longer walk them.  Measured on the adversarial schedule, whose critic updates read their loss on the host and therefore run
    in lock step with it: one update in ~12 took 45 instead of 13 ms (profiles/r06_adversarial_gc_outlier.txt); with the
    collector disabled or the heap frozen none does.  Nothing is leaked: objects created later are collected as before."""
    import gc
    gc.collect()
    gc.freeze()


class GradBucketReducer:
    """All-reduces a flat gradient buffer in buckets, tail first (the backward produces the gradients of
    the last layers first), overlapping communication with the remaining backward.  Works with any
    torch.distributed backend (RCCL on GPU, gloo in the CPU tests)."""

    def __init__(self, flat_grads, bucket_elems=2 * 1024 * 1024, group=None, force=False):
        import torch.distributed as dist
        self.dist = dist
        self.g = flat_grads
        self.bucket = int(bucket_elems)
        self.group = group
        self.hi = flat_grads.numel()
        self.works = []
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.force = bool(force) and dist.is_available() and dist.is_initialized()  # run the collectives at world 1 too

    def start(self):
        self.hi = self.g.numel()
        self.works = []
        self.n_launched = 0  # collectives issued for this step (diagnostics / tests)
        self.bytes_launched = 0  # payload handed to all_reduce this step (bench.py: allreduce_bytes_per_step)
        self.ranges = []  # [lo, hi) of every collective of this step, in issue order (tail first)

    def ready(self, offset_lo, force=False):
        """every gradient at flat offset >= offset_lo is final"""
        if self.world == 1 and not self.force:
            return
        if offset_lo >= self.hi:
            return
        if force or (self.hi - offset_lo) >= self.bucket:
            self.works.append(self.dist.all_reduce(self.g[offset_lo:self.hi], op=self.dist.ReduceOp.SUM,
                                                   group=self.group, async_op=True))
            self.n_launched += 1
            self.bytes_launched += (self.hi - offset_lo) * self.g.element_size()
            self.ranges.append((int(offset_lo), int(self.hi)))
            self.hi = offset_lo

    def reduce_all(self):
        """all-reduce the WHOLE buffer now, tail first, in `bucket`-sized collectives queued back to back (the critic's
        538 MB gradient: RCCL pipelines the buckets over the xGMI ring); returns the gradient scale 1/world"""
        self.start()
        if self.world > 1 or self.force:
            off = self.g.numel()
            while off > 0:
                off = max(0, off - self.bucket)
                self.ready(off, force=True)
        return self.finish()

    def finish(self):
        self.ready(0, force=True)
        for w in self.works:
            w.wait()
        self.works = []
        return 1.0 / self.world  # gradient scale for the optimizer


class Trainer:
    """generator + U-Net + L1 + Adam for one rank"""

    def __init__(self, brain_generator, net, lr=1e-4, lr_decay=0.0, work_with_residual_channel=None,
                 distributed=False, bucket_elems=2 * 1024 * 1024, force_allreduce=False, seg_regulariser=None,
                 regression_metric='l1', loss_cropping=None):
        if regression_metric not in ('l1', 'l2', 'laplace', 'ssim'):
            raise Exception('metrics should either be "l1" or "l2" or "ssim" oro "laplace", got {}'.format(
                regression_metric))  # the reference's message (metrics_model.py:127), typo included
        n_targets = net.nb_labels // 2 if regression_metric == 'laplace' else net.nb_labels
        if n_targets != 1 and seg_regulariser is not None:
            # the reference builds the segmentation network on [..., 1] (SynthSR/training.py:375) and feeds it the whole
            # `predicted_image` (metrics_model.py:148,165): with several regression targets Keras refuses the graph
            raise ValueError('the segmentation loss needs ONE regression target (the segmentation network takes a '
                             'single-channel image, SynthSR/training.py:375), this network predicts %d' % n_targets)
        self.metric, self.loss_cropping = regression_metric, loss_cropping
        self.bg = brain_generator
        self.seg = seg_regulariser  # synthsr_amd.seg_loss.SegmentationRegulariser or None
        self.gen = brain_generator.labels_to_image_model
        self.net = net
        self.lr, self.lr_decay = lr, lr_decay
        self.residual = work_with_residual_channel
        self.reducer = GradBucketReducer(net.grads, bucket_elems, force=force_allreduce) if distributed else None
        self.resident_labels = None
        self.auto_pool = True  # step() without explicit inputs: label maps drawn by the host sampler stay on the device
        self.fuse_head_bwd = True  # the head kernel also accumulates the sums of its backward pass (UNet3D.loss)
        self.comm_events = None  # a list: (start, end) HIP events around reducer.finish() of every step (bench.py --gpus N)

    def _generate_batch(self, model_inputs, draws, B):
        """batchsize > 1 (SynthSR/training.py:52): the B items of the batch are generated one after the other (each with
        its own label map, GMM parameters and draws -- the reference's batch-wise GMM LUT sum, F9, only when the generator
        says so: gen.sum_gmm_over_batch) and stacked along the first spatial axis, the layout UNet3D.set_batch works on"""
        import torch
        gen = self.gen
        if draws is not None and len(draws) != B:
            raise ValueError('one set of draws per batch item')
        labels, means, stds = model_inputs[:3]
        imgs, tgts = [], []
        means_b, stds_b = hm.batch_gmm_parameters(means, stds, getattr(gen, 'sum_gmm_over_batch', False))
        for b in range(B):
            real = np.asarray(model_inputs[3])[b, ..., 0] if getattr(gen, 'use_real_image', False) else None
            image, target, seg = gen.generate(np.asarray(labels)[b, ..., 0], means_b[b], stds_b[b],
                                              None if draws is None else draws[b], real_image=real)
            if b == 0:
                key = (B,) + tuple(image.shape) + tuple(target.shape)
                if getattr(self, '_batch_key', None) != key:
                    self._batch_key = key
                    self._img_b = torch.empty((B * image.shape[0],) + tuple(image.shape[1:]), dtype=image.dtype,
                                              device=image.device)
                    self._tgt_b = torch.empty((B * target.shape[0],) + tuple(target.shape[1:]), dtype=target.dtype,
                                              device=target.device)
            self._img_b.chunk(B, 0)[b].copy_(image)   # generate() re-uses its output buffers
            self._tgt_b.chunk(B, 0)[b].copy_(target)
            if self.seg is not None:   # the label maps of the batch, stacked like the volumes (segmentation loss)
                if b == 0:
                    self._seg_b = torch.empty((B * seg.shape[0],) + tuple(seg.shape[1:]), dtype=seg.dtype, device=seg.device)
                self._seg_b.chunk(B, 0)[b].copy_(seg)
        return self._img_b, self._tgt_b, (self._seg_b if self.seg is not None else None)

    @staticmethod
    def _narrowest_labels(m):
        """a label map in the narrowest integer type that holds its values (uint8 < 256, int16 < 32768, else int32): the
        generator gathers 1-2 instead of 4 bytes per voxel (csrc/generator.hip: synthsr_deform_params.label_bytes)"""
        m = np.asarray(m)
        lo, hi = (int(m.min()), int(m.max())) if m.size else (0, 0)
        dt = np.uint8 if (lo >= 0 and hi < 256) else (np.int16 if (lo >= -32768 and hi < 32768) else np.int32)
        return np.ascontiguousarray(m, dtype=dt)

    def _to_device_labels(self, m):
        import torch
        gen = self.gen
        m = np.asarray(m).reshape(gen.input_labels_shape)
        if gen.padding_margin is not None:  # PadAroundCentre (ext/lab2im/layers.py:1754) once, on the host copy
            m = np.pad(m, [(p, p) for p in gen.padding_margin])
        return torch.from_numpy(self._narrowest_labels(m)).to(gen.device)

    def make_labels_resident(self, label_maps):
        """upload a pool of label maps once (narrowest integer type); steps then pick from the pool on the device"""
        self.resident_labels = [self._to_device_labels(m) for m in label_maps]

    # training() keeps every label map it has used on the device (SynthSR/model_inputs.py:86-107 re-reads the file each step;
    # a 160^3 map is 4 MB as uint8): up to POOL_FRACTION of the device memory that is free when the pool is first used (the
    # network and its activations are allocated by then), and never more than POOL_BYTES; beyond that, and after any
    # allocation failure, the remaining maps are copied per step as before (one upload per step, never two)
    POOL_BYTES = 64 << 30
    POOL_FRACTION = 0.25

    def _pool_budget(self):
        import torch
        cap = self.__dict__.get('_auto_pool_cap')
        if cap is None:
            free, _ = torch.cuda.mem_get_info(self.gen.device)
            cap = self._auto_pool_cap = min(int(self.POOL_BYTES), int(self.POOL_FRACTION * free))
        return cap

    def _pooled_labels(self, index, host_map):
        """the device copy of label map `index` (uploaded once, narrowest integer type), or None when the pool is full or
        switched off: the caller then hands the host map to the generator"""
        import torch
        pool = self.__dict__.setdefault('_auto_pool', {})
        t = pool.get(index)
        if t is not None or not self.auto_pool:
            return t
        if self.__dict__.get('_auto_pool_full'):
            return None
        gen = self.gen
        m = np.asarray(host_map).reshape(gen.input_labels_shape)
        if gen.padding_margin is not None:  # PadAroundCentre (ext/lab2im/layers.py:1754) once, on the host copy
            m = np.pad(m, [(p, p) for p in gen.padding_margin])
        host = self._narrowest_labels(m)
        used = self.__dict__.get('_auto_pool_used', 0)
        if used + host.nbytes > self._pool_budget():   # full: checked on the host array, BEFORE anything is uploaded
            self._auto_pool_full = True
            return None
        try:
            t = torch.from_numpy(host).to(gen.device)
        except torch.cuda.OutOfMemoryError:            # somebody else took the memory: stop pooling, keep what is there
            self._auto_pool_full = True
            return None
        pool[index] = t
        self._auto_pool_used = used + host.nbytes
        return t

    def step(self, model_inputs=None, draws=None, label_index=None):
        """one training step; returns the loss as a 1-element device tensor (no host sync)"""
        gen, net = self.gen, self.net
        picks = None
        if model_inputs is None:
            mig = self.bg.model_inputs_generator
            model_inputs = next(mig)
            picks = getattr(mig, 'last_picks', None)   # which label maps: they stay on the device once used
        labels, means, stds = model_inputs[:3]
        from . import ops
        B = int(np.asarray(means).shape[0])
        if B > 1 and label_index is not None:
            raise ValueError('label_index (device-resident label pool) picks ONE map: not available with batchsize > 1')
        if B > 1:
            image, target, seg = self._generate_batch(model_inputs, draws, B)
        else:
            real = np.asarray(model_inputs[3])[0, ..., 0] if getattr(gen, 'use_real_image', False) else None
            with ops.timed('generator', gen.output_shape, gen.n_image_channels, gen.n_target_channels):
                dev_labels = None
                if label_index is not None and self.resident_labels is not None and real is None:
                    dev_labels = self.resident_labels[label_index]
                elif picks is not None and real is None and self.auto_pool:
                    dev_labels = self._pooled_labels(picks[0], np.asarray(labels)[0, ..., 0])
                if dev_labels is not None:
                    image, target, seg = gen.generate(dev_labels, np.asarray(means)[0], np.asarray(stds)[0], draws,
                                                      labels_on_device=True)
                else:
                    image, target, seg = gen.generate(np.asarray(labels)[0, ..., 0], np.asarray(means)[0],
                                                      np.asarray(stds)[0], draws, real_image=real)
        net.set_batch(B)
        residual, rs, ro = None, 1, 0
        if self.residual is not None:
            residual, rs, ro = image, image.shape[-1], [int(c) for c in self.residual]
        # fuse_head_bwd: nothing touches net.dpred between the loss and the backward pass unless the segmentation loss adds to it
        with ops.trace_range('forward + loss'):   # (roctx ranges: SYNTHSR_ROCTX=1, rocprofv3 --marker-trace)
            loss, pred = net.loss(image, target.reshape(-1), self.metric, self.loss_cropping, residual=residual,
                                  res_stride=rs, res_off=ro, want_pred=self.seg is not None,
                                  fuse_head_bwd=self.fuse_head_bwd and self.seg is None)
        if self.seg is not None:  # total = image loss + w * Dice(frozen segmentation net(prediction), labels)
            if list(seg.shape) != list(image.shape[:3]):
                raise ValueError('segmentation loss with a target resolution different from the label maps: `segmentation_target` '
                                 'has the label maps\' grid, the posteriors the prediction\'s, and DiceLoss multiplies them voxel by voxel '
                                 '(SynthSR/metrics_model.py:191-207) -- the reference\'s graph fails with incompatible shapes too')
            loss = loss + self.seg.rel_weight * self.seg(pred, seg, net.dpred, self.loss_cropping,
                                                         head_channels=2 if self.metric == 'laplace' else 1)
        if self.reducer is not None:
            self.reducer.start()
            net.backward(on_grad_ready=self.reducer.ready)
            if self.comm_events is not None:  # how long the compute stream waits for the collectives still in flight
                import torch
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                scale = self.reducer.finish()
                e1.record()
                self.comm_events.append((e0, e1))
            else:
                scale = self.reducer.finish()
        else:
            with ops.trace_range('backward'):
                net.backward()
            scale = 1.0
        with ops.trace_range('optimizer'):
            net.adam_step(self.lr, self.lr_decay, grad_scale=scale)
            net.update_moving_stats()
        return loss


def save_checkpoint(path, net):
    """`.npz`: weights + BN moving statistics + Adam state (exact resume).  `.h5`: the weights in Keras' save_weights
    layout under the reference's layer names (keras_h5.save_keras_weights), loadable by the reference's
    `load_weights(path, by_name=True)`; the 1x1x1 head kernel is stored 5-D like every Conv3D kernel."""
    sd = {k: v.numpy() for k, v in net.state_dict().items()}
    if str(path).lower().endswith(('.h5', '.hdf5')):
        from .keras_h5 import save_keras_weights
        save_keras_weights(path, {k: (v.reshape((1, 1, 1) + v.shape) if k.endswith('/kernel') and v.ndim == 2 else v)
                                  for k, v in sd.items()})
        return
    sd['optimizer/iterations'] = np.array(net.iterations)
    sd['optimizer/m'] = net.adam_m.cpu().numpy()
    sd['optimizer/v'] = net.adam_v.cpu().numpy()
    np.savez(path, **sd)


def read_weights(path):
    """{name: ndarray} of a checkpoint: the .npz of save_checkpoint, or a Keras .h5 as written by the reference
    (ModelCheckpoint / save_weights, SynthSR/training.py:430) through the library-free reader of keras_h5.py"""
    if str(path).lower().endswith(('.h5', '.hdf5')):
        from .keras_h5 import load_keras_weights
        return load_keras_weights(path)
    z = np.load(path)
    return {k: z[k] for k in z.files}


def keras_trainable_order(net):
    """names of the network's trainable weights in the order of Keras' `model.trainable_weights` (layer order; Conv3D: kernel,
    bias; BatchNormalization: gamma, beta -- this build stores beta before gamma): the order of the Adam slots in a Keras file"""
    out = []
    specs = [nm for nm, _, _ in net.specs]
    i = 0
    while i < len(specs):
        if specs[i].endswith('/beta') and i + 1 < len(specs) and specs[i + 1].endswith('/gamma'):
            out += [specs[i + 1], specs[i]]
            i += 2
        else:
            out.append(specs[i])
            i += 1
    return out


def restore_keras_optimizer(path, net):
    """Adam moments + iteration count from a full-model Keras .h5 (SynthSR/training.py:434-439 resumes with
    `models.load_model`, "momentum is comprised in checkpoints").  Slots are matched by position and verified by shape.
    Returns True if restored; False if the file carries none, or carries the optimizer of a DIFFERENT model (weights are
    loaded by layer name, so a file of another architecture is a legitimate source of weights: its optimizer state is then
    left alone, with a warning, and Adam restarts)."""
    import warnings
    import torch
    from .keras_h5 import load_keras_optimizer
    slots = load_keras_optimizer(path)
    if slots is None:
        return False
    it, ms, vs = slots
    order = keras_trainable_order(net)

    def fits(nm, m, v):
        shp = tuple(net.offsets[nm][1])
        want = (1, 1, 1) + shp if nm.endswith('likelihood/kernel') and m.ndim == 5 else shp   # Keras keeps the 1x1x1 head 5-D
        return tuple(m.shape) == want and tuple(v.shape) == want

    if len(ms) != len(order) or not all(fits(nm, m, v) for nm, m, v in zip(order, ms, vs)):
        warnings.warn('%s: its optimizer state (%d slots) does not belong to this network (%d trainable weights): weights '
                      'loaded by name, Adam restarts' % (path, len(ms), len(order)))
        return False
    for nm, m, v in zip(order, ms, vs):
        shp = tuple(net.offsets[nm][1])
        net.view(nm, net.adam_m).copy_(torch.from_numpy(m.reshape(shp)))
        net.view(nm, net.adam_v).copy_(torch.from_numpy(v.reshape(shp)))
    net.iterations = it
    return True


def load_checkpoint(path, net, by_name=True, skip=()):
    """weights by layer name (model.load_weights(checkpoint, by_name=True), SynthSR/training.py:363) + the Adam state: from
    our own .npz checkpoints, or from a full-model Keras .h5 (what the reference's ModelCheckpoint writes and its resume
    path `models.load_model` restores, training.py:429-439)"""
    import torch
    z = read_weights(path)
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in z.items() if not k.startswith('optimizer/') and
          not any(k.startswith(s) for s in skip)}
    if not any(nm in sd for nm, _, _ in net.specs):
        raise ValueError('%s holds no weight of this network (layer prefix %r)' % (path, net.prefix))
    net.load_state_dict(sd, strict=False)
    if 'optimizer/m' in z and z['optimizer/m'].shape[0] == net.n_params and not skip:
        net.adam_m.copy_(torch.from_numpy(z['optimizer/m']))
        net.adam_v.copy_(torch.from_numpy(z['optimizer/v']))
        net.iterations = int(z['optimizer/iterations'])
    elif str(path).lower().endswith(('.h5', '.hdf5')) and not skip:
        restore_keras_optimizer(path, net)


def training(labels_dir, model_dir, prior_means, prior_stds, path_generation_labels, segmentation_label_list=None,
             segmentation_label_equivalency=None, segmentation_model_file=None, fs_header_segnet=False,
             relative_weight_segmentation=0.25, prior_distributions='normal', images_dir=None,
             path_generation_classes=None, FS_sort=True, batchsize=1, input_channels=True, output_channel=0,
             target_res=None, output_shape=None, flipping=True, padding_margin=None, scaling_bounds=0.15,
             rotation_bounds=15, shearing_bounds=0.02, translation_bounds=5, nonlin_std=4.,
             nonlin_shape_factor=0.03125, simulate_registration_error=True, data_res=None, thickness=None,
             randomise_res=None, downsample=True, blur_range=1.15, build_reliability_maps=True, bias_field_std=.3,
             bias_shape_factor=0.03125, n_levels=5, nb_conv_per_level=2, conv_size=3, unet_feat_count=24,
             feat_multiplier=2, dropout=0, activation='elu', lr=1e-4, lr_decay=0, epochs=100, steps_per_epoch=1000,
             regression_metric='l1', work_with_residual_channel=None, loss_cropping=None, checkpoint=None,
             model_file_has_different_lhood_layer=False, seed=0, verbose=True, dtype='f32', deterministic=False,
             segnet_frozen_bn='batch', reference_batch_gmm=False):
    """Parameters as documented in SynthSR/training.py:90-240 (+ `reference_batch_gmm`: at batchsize > 1 sample every item
    from the SUM of the batch's GMM means / stds like the reference's SampleConditionalGMM does (a bug; off: each
    item from its own); `seed`, `verbose`, `dtype`: 'f32' like the reference, or
    'bf16' = bf16 activations / packed weights with fp32 accumulation, BatchNorm statistics and master weights,
    BASELINE.json configs[3]; `deterministic`: bit-identical weights run after run for the same seed, ops.set_deterministic,
    1.2x (fp32) / 1.4x (bf16) slower at 160^3; `segnet_frozen_bn`: what the frozen
    segmentation network's BatchNormalization normalises with, 'batch' = the statistics of its own activations, which is what
    Keras 2.3.1 does to a trainable=False BatchNormalization under fit(), or 'inference' = its moving averages)."""
    import torch
    if deterministic:  # process-wide switch: on for the duration of this call, previous setting restored on the way out
        from . import ops as _ops
        kw = dict(locals())
        kw.pop('torch', None)
        kw.pop('_ops', None)
        kw['deterministic'] = False
        previous = _ops.set_deterministic(True)
        try:
            return training(**kw)
        finally:
            _ops.set_deterministic(previous)
    # the launcher script hands `--input_channels` over as text (scripts/training.py:36 of the reference): 'True' / 'False'
    input_channels = [{'True': True, 'False': False}.get(c, c) if isinstance(c, str) else c
                      for c in hm.reformat_to_list(input_channels)]
    n_channels = len(input_channels)
    if output_channel is not None:
        output_channel = list(hm.reformat_to_list(output_channel))
    # checks, training.py:252-271
    if (images_dir is None) & (output_channel is None):
        raise Exception('please provide a value for output_channel or image_dir')
    elif (images_dir is not None) & (output_channel is not None):
        raise Exception('please provide a value either for output_channel or image_dir, but not both at the same time')
    if output_channel is not None:
        if any(x >= n_channels for x in output_channel):
            raise Exception('indices in output_channel cannot be greater than the total number of channels')
    if work_with_residual_channel is not None:
        work_with_residual_channel = hm.reformat_to_list(work_with_residual_channel)
        if output_channel is not None:
            if len(work_with_residual_channel) != len(output_channel):
                raise Exception('The number or residual channels and output channels must be the same')
        if any(x >= n_channels for x in work_with_residual_channel):
            raise Exception('indices in work_with_residual_channel cannot be greater than the total number of channels')
        if build_reliability_maps:
            # SynthSR/training.py:270-271 repeats the python LIST here (`2 * list`) instead of doubling the
            # indices.  One residual channel c -> [c, c]: metrics_model adds image_out[..., c] twice as a 2-channel tensor
            # that broadcasts against the 1-channel prediction / target, so the loss equals the single-channel one with
            # the index taken IN THE INTERLEAVED image (c = 1 is the reliability map of channel 0, as in the reference);
            # pinned by tests/golden/unet_training_graph.npz (`tg_l1_res`).  Several residual channels [a, b] -> [a, b, a, b]
            # make keras' Add fail on shapes (4 vs 2 channels): same outcome here.
            if len(work_with_residual_channel) > 1:
                raise ValueError('Operands could not be broadcast together: work_with_residual_channel has %d entries and '
                                 'build_reliability_maps=True repeats the list (SynthSR/training.py:270-271)'
                                 % len(work_with_residual_channel))
    if regression_metric not in ('l1', 'l2', 'laplace', 'ssim'):
        raise Exception('metrics should either be "l1" or "l2" or "ssim" oro "laplace", got {}'.format(regression_metric))

    dist_on = 'RANK' in os.environ and int(os.environ.get('WORLD_SIZE', '1')) > 1
    rank, world = 0, 1
    if dist_on:
        import torch.distributed as dist
        if not dist.is_initialized():
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
            dist.init_process_group('nccl')
        rank, world = dist.get_rank(), dist.get_world_size()

    generation_labels, n_neutral_labels = volumes.get_list_labels(label_list=path_generation_labels,
                                                                  labels_dir=labels_dir, FS_sort=FS_sort)
    if rank == 0:
        os.makedirs(model_dir, exist_ok=True)
    if loss_cropping == 0:
        padding_margin = None
    elif padding_margin is None:
        padding_margin = hm.get_padding_margin(output_shape, loss_cropping)

    rng = np.random.Generator(np.random.Philox(key=(int(seed) << 20) + rank))
    brain_generator = BrainGenerator(labels_dir=labels_dir, images_dir=images_dir, generation_labels=generation_labels,
                                     n_neutral_labels=n_neutral_labels, padding_margin=padding_margin,
                                     batchsize=batchsize, input_channels=input_channels,
                                     output_channel=output_channel, target_res=target_res, output_shape=output_shape,
                                     output_div_by_n=2 ** n_levels, generation_classes=path_generation_classes,
                                     prior_means=prior_means, prior_stds=prior_stds,
                                     prior_distributions=prior_distributions, flipping=flipping,
                                     scaling_bounds=scaling_bounds, rotation_bounds=rotation_bounds,
                                     shearing_bounds=shearing_bounds, translation_bounds=translation_bounds,
                                     nonlin_std=nonlin_std, nonlin_shape_factor=nonlin_shape_factor,
                                     simulate_registration_error=simulate_registration_error,
                                     randomise_res=randomise_res, data_res=data_res, thickness=thickness,
                                     downsample=downsample, blur_range=blur_range,
                                     build_reliability_maps=build_reliability_maps, bias_field_std=bias_field_std,
                                     bias_shape_factor=bias_shape_factor, rng=rng)
    brain_generator.labels_to_image_model.seed(seed, rank)
    brain_generator.labels_to_image_model.sum_gmm_over_batch = bool(reference_batch_gmm)
    unet_input_shape = brain_generator.model_output_shape
    n_output_channels = 1 if output_channel is None else len(output_channel)
    if regression_metric == 'laplace':  # intensities + spreads (SynthSR/training.py:325-326)
        n_output_channels *= 2
    net = build_unet(nb_features=unet_feat_count, input_shape=unet_input_shape, nb_levels=n_levels,
                     conv_size=conv_size, nb_labels=n_output_channels,  # training.py:246-249, 325-326
                     feat_mult=feat_multiplier,
                     nb_conv_per_level=nb_conv_per_level, conv_dropout=dropout, final_pred_activation='linear',
                     batch_norm=-1, activation=activation, input_model=brain_generator.labels_to_image_model,
                     seed=seed, dtype=dtype)
    init_epoch = 0
    if checkpoint is not None:
        if verbose and rank == 0:
            print('loading', checkpoint)
        skip = ('%s_likelihood' % net.prefix,) if model_file_has_different_lhood_layer else ()
        load_checkpoint(checkpoint, net, skip=skip)
        try:
            init_epoch = int(os.path.basename(checkpoint)[:3])
        except ValueError:
            init_epoch = 0
    if dist_on:
        import torch.distributed as dist
        dist.broadcast(net.params, 0)  # identical initial weights on every rank
        net.repack()
        net.set_dropout_seed(int(seed) + 0x5eed + 7919 * rank)  # ... but its own dropout masks (as its own samples)
    # frozen segmentation CNN for the segmentation-regularised loss (training.py:371-409)
    seg_reg = None
    if segmentation_model_file is not None:
        from .seg_loss import SegmentationRegulariser
        segmentation_labels = np.asarray(hm.load_array_if_path(segmentation_label_list))
        seg_shape = list(unet_input_shape[:-1])
        if fs_header_segnet:  # the network sees the volume with its last two axes swapped (metrics_model.py:158-160)
            seg_shape = [seg_shape[0], seg_shape[2], seg_shape[1]]
        seg_net = build_unet(nb_features=unet_feat_count, input_shape=seg_shape + [1], nb_levels=n_levels,
                             conv_size=conv_size, nb_labels=len(segmentation_labels), feat_mult=feat_multiplier,
                             nb_conv_per_level=nb_conv_per_level, conv_dropout=dropout, final_pred_activation='softmax',
                             batch_norm=-1, activation=activation, input_model=None, seed=seed + 1)
        load_checkpoint(segmentation_model_file, seg_net)
        m = M = None
        if images_dir is not None:  # clip the synthesised images at the 2nd / 98th percentiles of the first real scan
            first = volumes.list_images_in_folder(images_dir)[0]
            im = volumes.load_volume(first, im_only=True).flatten()
            m, M = np.percentile(im, 2), np.percentile(im, 98)
        seg_reg = SegmentationRegulariser(seg_net, brain_generator.generation_labels,
                                          hm.load_array_if_path(segmentation_label_equivalency),
                                          relative_weight_segmentation, m=m, M=M, fs_header=fs_header_segnet,
                                          frozen_bn=segnet_frozen_bn)
    if loss_cropping == 0:
        loss_cropping = None
    trainer = Trainer(brain_generator, net, lr, lr_decay, work_with_residual_channel, distributed=dist_on,
                      seg_regulariser=seg_reg, regression_metric=regression_metric, loss_cropping=loss_cropping)

    log_path = os.path.join(model_dir, 'logs', 'loss.csv')
    tb = None
    if rank == 0:
        os.makedirs(os.path.dirname(log_path), exist_ok=True)
        from .tb_events import EventFileWriter
        tb = EventFileWriter(os.path.dirname(log_path))   # KC.TensorBoard(log_dir=model_dir/logs) (SynthSR/training.py:425-431)
    gc_settled = False
    for epoch in range(init_epoch, epochs):
        t0 = time.time()
        acc = torch.zeros(1, device=net.device)
        for _ in range(steps_per_epoch):
            acc += trainer.step()
            if not gc_settled:
                settle_host_gc()
                gc_settled = True
        mean_loss = float(acc.item()) / steps_per_epoch
        if not np.isfinite(mean_loss):  # tf.debugging.check_numerics in IdentityLoss (metrics_model.py:228)
            raise FloatingPointError('Loss not finite')
        if rank == 0:
            dt = time.time() - t0
            if verbose:
                print('Epoch %d/%d - %.1fs - loss: %.6f - %.2f volumes/s/GPU' % (epoch + 1, epochs, dt, mean_loss,
                                                                                  steps_per_epoch / dt))
            with open(log_path, 'a') as f:
                f.write('%d,%.8f,%.3f\n' % (epoch + 1, mean_loss, dt))
            tb.add_scalar('loss', mean_loss, epoch)   # Keras logs the epoch's mean loss at step = 0-based epoch index
            save_checkpoint(os.path.join(model_dir, '%03d.npz' % (epoch + 1)), net)
            save_checkpoint(os.path.join(model_dir, '%03d.h5' % (epoch + 1)), net)  # SynthSR/training.py:429 file name
    if tb is not None:
        tb.close()
    return net
