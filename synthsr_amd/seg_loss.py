"""Segmentation-regularised loss, SynthSR/metrics_model.py:136-215 (`add_seg_loss_to_model`) + training.py:371-409:
the predicted image is pushed through a FROZEN segmentation U-Net and the soft Dice between its (label-merged)
posteriors and the generator's label map is added to the image loss with weight `rel_weight`
(total = image_loss + rel_weight * dice, metrics_model.py:209).

Everything runs on the device: U-Net forward / data-gradient backward through the HIP conv kernels
(`UNet3D.predict_probs` / `backward_input`: no weight gradients; BatchNorm on batch statistics like Keras' learning phase, or
on the moving averages: `frozen_bn`), head softmax, Dice sums and
their backward in `csrc/unet_pointwise.hip`.  The reference's quirk is kept: the ground-truth one-hot of generation
label number i is `segmentation_target == i` (the INDEX, metrics_model.py:191), not `== generation_labels[i]`.
"""
import numpy as np

from . import ops


class SegmentationRegulariser:
    def __init__(self, seg_net, generation_labels, segmentation_label_equivalency, rel_weight, m=None, M=None,
                 fs_header=False, frozen_bn='batch'):
        """frozen_bn: what the frozen network's BatchNormalization layers normalise with.  'batch' (default): the statistics
        of the current activations -- Keras 2.3.1 semantics of a non-trainable BatchNormalization inside a model being fitted
        (learning phase 1; `trainable = False` only stops the updates), i.e. what SynthSR/training.py:371-409 +
        metrics_model.py:136-215 compute; 'inference': the moving averages stored in the segmentation model.  Both are pinned by
        the reference graph executed on the shim (tests/golden/unet_seg_loss.npz: `*_bnbatch_*`, `*_bninf_*`)."""
        import torch
        if frozen_bn not in ('batch', 'inference'):
            raise ValueError("frozen_bn should be 'batch' or 'inference'")
        self.batch_stats = frozen_bn == 'batch'
        self.torch = torch
        self.net = seg_net
        seg_net.training = False
        seg_net.enable_input_grad()
        self.rel_weight = float(rel_weight)
        self.m, self.M = (None, None) if m is None else (float(m), float(M))
        self.fs_header = bool(fs_header)
        gen = np.asarray(generation_labels).reshape(-1)
        eq = np.asarray(segmentation_label_equivalency).reshape(-1)
        if len(eq) != seg_net.nb_labels:
            raise ValueError('segmentation_label_equivalency should have one entry per label of the segmentation network, '
                             'had %d and %d' % (len(eq), seg_net.nb_labels))
        idx, gt = [], []
        for i, lab in enumerate(gen):
            j = np.where(eq == lab)[0]
            if len(j) == 0:
                continue
            if len(j) > 3:
                raise Exception("uuummm weird that you're merging so many labels...")
            idx.append(list(j) + [-1] * (3 - len(j)))
            gt.append(i)
        if not idx:
            raise ValueError('no generation label has an equivalent among the segmentation labels')
        dev = seg_net.device
        self.cls_idx = torch.tensor(idx, dtype=torch.int32, device=dev).reshape(-1)
        self.cls_gt = torch.tensor(gt, dtype=torch.int32, device=dev)
        self.K = len(gt)
        self.sums = torch.zeros(2 * self.K, dtype=torch.float32, device=dev)

    def _to_seg_frame(self, t):  # metrics_model.py:158-160: swap the last two spatial axes, then reverse the new 2nd axis
        return self.torch.flip(t.permute(0, 2, 1), dims=[1]).contiguous() if self.fs_header else t

    def _from_seg_frame(self, t):  # :162-163
        return self.torch.flip(t, dims=[1]).permute(0, 2, 1).contiguous() if self.fs_header else t

    def __call__(self, pred, seg_target, dpred, loss_cropping=None, head_channels=1):
        """pred: predicted image, device float [nvox] (or [d0,d1,d2]); seg_target: int32 [d0,d1,d2] (the generator's
        `segmentation_target`); dpred [nvox]: gradient of the image loss w.r.t. pred, incremented IN PLACE by
        rel_weight * d(dice)/d(pred).  loss_cropping: sizes of the centred box the Dice is evaluated on
        (metrics_model.py:166-183: the network still sees the whole volume, posteriors and labels are cropped).
        head_channels = 2: the Laplace head (metrics_model.py:33-49) -- pred / dpred are [nvox][2] = (intensity, spread);
        `predicted_image`, what the segmentation network sees, is the intensity channel alone (:53) and only that channel
        receives the Dice gradient.  Returns the Dice loss as a 0-d device tensor."""
        torch = self.torch
        net = self.net
        if head_channels != 1:
            pred2, dpred2 = pred.reshape(-1, head_channels), dpred.reshape(-1, head_channels)
            pred, dpred_out = pred2[:, 0].contiguous(), dpred2[:, 0]
        else:
            dpred_out = dpred
        # batchsize > 1: the volumes are stacked along the first spatial axis (UNet3D.set_batch); the frozen network runs on the
        # stack (batch-statistics BatchNorm then normalises over the whole batch, as Keras does), the Dice is evaluated volume
        # by volume and averaged (DiceLoss: mean over batch and labels)
        d0 = int(net.input_shape[0])                 # (the FreeSurfer frame swaps axes 1 and 2 only)
        nb = int(seg_target.shape[0]) // d0
        if nb * d0 != int(seg_target.shape[0]):
            raise ValueError('segmentation network built for %s, label maps are %s' % (net.input_shape[:3], list(seg_target.shape)))
        net.set_batch(nb)
        shape = tuple(seg_target.shape)
        one = (d0,) + shape[1:]                      # one volume, generator frame
        x = pred.reshape(shape)
        if self.m is not None:  # :155
            inside = (x > self.m) & (x < self.M)
            x = (torch.clamp(x, self.m, self.M) - self.m) / (self.M - self.m)
        xs = self._to_seg_frame(x)                   # (the swap / flip leave the stacking axis alone)
        seg = self._to_seg_frame(seg_target).reshape(-1)
        if [int(xs.shape[0]) // nb] + list(xs.shape[1:]) != net.input_shape[:3]:
            raise ValueError('segmentation network built for %s, prediction is %s' % (net.input_shape[:3], list(xs.shape)))
        probs = net.predict_probs(xs[..., None].contiguous(), batch_stats=self.batch_stats)
        if loss_cropping is not None:
            # outside the box: no ground-truth class (label -1) and zero posteriors, which removes those voxels from both
            # Dice sums and - the softmax Jacobian p_i (delta_ij - p_j) vanishing with p - from the gradient
            size = [int(loss_cropping)] * 3 if np.ndim(loss_cropping) == 0 else [int(v) for v in loss_cropping]
            if len(size) != 3 or any(c < 1 or c > d for c, d in zip(size, one)):
                raise ValueError('loss_cropping %s does not fit the output shape %s' % (size, list(one)))
            lo = [int((d - c) / 2) for d, c in zip(one, size)]
            mask = torch.zeros(one, dtype=torch.bool, device=probs.device)
            mask[lo[0]:lo[0] + size[0], lo[1]:lo[1] + size[1], lo[2]:lo[2] + size[2]] = True
            mask = self._to_seg_frame(mask.repeat(nb, 1, 1)).reshape(-1)
            seg = torch.where(mask, seg, torch.full_like(seg, -1))
            probs.mul_(mask[:, None])
        low, _ = net.saved['last']
        dbn = net.buf('seg_dbn', list(low.shape))
        dice = None
        for b, (pb, sb, db) in enumerate(zip(probs.chunk(nb, 0), seg.chunk(nb, 0), dbn.chunk(nb, 0))):
            ops.seg_dice_sums(pb, sb, self.cls_idx, self.cls_gt, self.sums)
            T, B = self.sums[:self.K], self.sums[self.K:]
            d = (1.0 - (T + 1e-7) / (B + 1e-7)).mean() / nb
            dice = d if dice is None else dice + d
            ops.seg_dice_bwd(pb, sb, net.view(net.head['w']), self.cls_idx, self.cls_gt, self.sums, self.rel_weight / nb, db)
        dx = net.backward_input(dbn)[..., 0]
        dx = self._from_seg_frame(dx)
        if self.m is not None:
            dx = dx * inside / (self.M - self.m)
        dpred_out.add_(dx.reshape(-1))
        return dice
