"""MI355X-native counterpart of SynthSR/labels_to_image_model.py:32-266.

`labels_to_image_model(...)` keeps the reference's name, positional order, keyword names and defaults,
and returns a `LabelsToImageModel` whose call takes the same inputs as the reference's Keras model —
`[labels int32[B,*S,1], means f32[B,L,C], stds f32[B,L,C]]` — and returns `[image, target]`.
Instead of a TF graph it runs a short sequence of fused HIP kernels (csrc/generator.hip) on the
current stream; per-volume O(1) parameters (affine, blur kernels, small SVF / bias grids) are drawn
and prepared on the host (host_math.py) and handed to the kernels as arguments.

Random draws are explicit (`Draws`).  In production they come from a per-rank numpy Philox generator
(host, O(500) values per volume) plus an in-kernel Philox4x32-10 stream for the per-voxel GMM noise;
parity tests inject the reference's recorded tape through `draws_from_tape`.

Scope: synthetic or real-image (output_channel=None, §8f-4) regression targets; fixed acquisition
resolution (non-separable or, for |sigma| > 5, separable blur) or `randomise_res=True` (SampleResolution /
DynamicGaussianBlur / MimicAcquisition, §8f-2 generator half), with or without registration error.
Batch items are generated independently; for batchsize > 1 the reference sums the GMM LUT over the
batch (F9, a bug: every item samples from the SUM of the items' means / stds) — reproduced only on request
(`model.sum_gmm_over_batch = True`, `training(..., reference_batch_gmm=True)`; host_math.batch_gmm_parameters).
"""
import ctypes
import numpy as np

from . import host_math as hm
from . import ops as _ops
from . import _lib


class Draws:
    """all random inputs of ONE generated volume (raw U[0,1) / N(0,1) values, TF's affine maps are applied later)"""

    def __init__(self):
        self.u_rot = self.u_shear = self.u_scale = self.u_trans = None
        self.u_svf_std = None
        self.n_svf = None
        self.u_crop = None
        self.u_flip = None
        self.gmm_noise = None  # np.ndarray [S..., C] (tape) or None -> in-kernel philox
        self.philox_key = (0, 0)
        self.philox_offset = 0
        self.channels = []  # per channel dict: u_bias_std, n_bias, u_bias_gate, n_gamma, u_regT(rot,trans), u_blur, u_regE


class LabelsToImageModel:
    def __init__(self, labels_shape, input_channels, output_channel, generation_labels, n_neutral_labels, atlas_res,
                 target_res, output_shape, output_div_by_n, padding_margin, flipping, aff, scaling_bounds,
                 rotation_bounds, shearing_bounds, translation_bounds, nonlin_std, nonlin_shape_factor,
                 simulate_registration_error, randomise_res, data_res, thickness, downsample,
                 build_reliability_maps, blur_range, bias_field_std, bias_shape_factor, device=None):
        import torch
        self.torch = torch
        self.lib = _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.sum_gmm_over_batch = False   # True: the reference's batch-wise LUT sum (F9), see host_math.batch_gmm_parameters

        # ---- parameter normalisation, SynthSR/labels_to_image_model.py:69-100
        input_channels = [bool(c) for c in hm.reformat_to_list(input_channels)]
        self.input_channels = input_channels
        self.n_channels = n_channels = len(input_channels)
        # output_channel=None: the regression target is a real scan fed as a 4th input (labels_to_image_model.py:71,109-113)
        self.use_real_image = output_channel is None
        self.output_channel = [] if output_channel is None else [int(c) for c in hm.reformat_to_list(output_channel)]
        self.idx_first_input_channel = int(np.argmax(input_channels))
        self.simulate_registration_error = hm.reformat_to_list(simulate_registration_error, length=n_channels)
        labels_shape = hm.reformat_to_list(labels_shape)
        if len(labels_shape) != 3:
            raise NotImplementedError('only 3-D label maps are supported')
        atlas = hm.reformat_to_n_channels_array(atlas_res, 3, n_channels)
        data_res = hm.load_array_if_path(data_res)
        thickness = hm.load_array_if_path(thickness)
        for idx in self.output_channel:
            if not input_channels[idx]:
                if data_res is not None:
                    data_res = np.insert(np.atleast_2d(np.array(data_res, dtype=np.float64)), idx, 1, axis=0)
                if thickness is not None:
                    thickness = np.insert(np.atleast_2d(np.array(thickness, dtype=np.float64)), idx, 1, axis=0)
        data_res = atlas if data_res is None else hm.reformat_to_n_channels_array(data_res, 3, n_channels)
        thickness = data_res if thickness is None else hm.reformat_to_n_channels_array(thickness, 3, n_channels)
        self.downsample = hm.reformat_to_list(downsample, n_channels) if downsample else \
            list(np.min(thickness - data_res, 1) < 0)
        self.atlas_res = list(atlas[0])
        self.target_res = self.atlas_res if target_res is None else \
            list(hm.reformat_to_n_channels_array(hm.load_array_if_path(target_res), 3)[0])
        self.data_res, self.thickness = data_res, thickness
        if isinstance(randomise_res, (bool, np.bool_)) or randomise_res is None:
            randomise_res = n_channels * [bool(randomise_res)]
        # randomise_res (labels_to_image_model.py:215-220): per-volume random acquisition resolution / slice thickness
        self.randomise_res = [bool(r) for r in randomise_res]
        self.crop_shape, self.output_shape, self.padding_margin = hm.get_shapes(
            labels_shape, output_shape, self.atlas_res, self.target_res, padding_margin, output_div_by_n)
        self.input_labels_shape = list(labels_shape)
        if self.padding_margin is not None:
            labels_shape = [labels_shape[i] + 2 * self.padding_margin[i] for i in range(3)]
        self.labels_shape = list(labels_shape)
        self.generation_labels = np.asarray(generation_labels).astype(np.int32)
        self.n_neutral_labels = n_neutral_labels
        self.flipping = flipping
        if flipping:
            assert aff is not None, 'aff should not be None if flipping is True'
            # labels_to_image_model.py:154-162 hands RandomFlip the right/left axis of `aff`, but the vendored RandomFlip
            # reverses the axis at the POSITION of that entry inside flip_axes (ext/lab2im/layers.py:398-400,424-427; SURVEY
            # F10) -- a one-element list, position 0: the reference flips axis 0 whatever the affine says, and so does the
            # kernel (pinned by the graph_nonras_* goldens, generated by the reference with a non-RAS aff)
            hm.get_ras_axes(aff, 3)   # (validates aff like the reference does)
        self.swap_lut = hm.flip_swap_lut(self.generation_labels, n_neutral_labels) if flipping else None
        # a (2n, m) array of bounds: ONE of its n two-row blocks serves the whole model, picked with numpy's global stream
        # while the graph is built (utils.draw_value_from_distribution, ext/lab2im/utils.py:1013-1016, called by
        # sample_affine_transform in the order rotation, shearing, scaling, translation: utils.py:685-738)
        self.rotation_bounds = hm.pick_bounds_block(rotation_bounds)
        self.shearing_bounds = hm.pick_bounds_block(shearing_bounds)
        self.scaling_bounds = hm.pick_bounds_block(scaling_bounds)
        self.translation_bounds = hm.pick_bounds_block(translation_bounds)
        self.apply_affine = any(b is not False for b in (self.scaling_bounds, self.rotation_bounds,
                                                         self.shearing_bounds, self.translation_bounds))
        self.nonlin_std = nonlin_std
        self.nonlin_shape_factor = nonlin_shape_factor
        self.apply_elastic = nonlin_std > 0
        self.small_shape = hm.get_resample_shape(self.labels_shape, nonlin_shape_factor) if self.apply_elastic else None
        self.half_shape = [max(int(self.labels_shape[i] / 2), self.small_shape[i]) for i in range(3)] \
            if self.apply_elastic else None
        self.build_reliability_maps = build_reliability_maps
        self.blur_range = blur_range
        self.bias_field_std = bias_field_std
        self.bias_shape_factor = bias_shape_factor
        self.small_bias_shape = hm.get_resample_shape(self.crop_shape, bias_shape_factor)
        self.resample_target = self.crop_shape != self.output_shape
        # (more than 4 synthetic channels: the fused deformation / GMM kernel runs once per group of four)

        # ---- output layout
        self.n_image_channels = sum(input_channels) * (2 if build_reliability_maps else 1)
        self.n_target_channels = 1 if self.use_real_image else len(self.output_channel)
        self.model_output_shape = list(self.output_shape) + [self.n_image_channels]

        # ---- persistent device buffers
        f32, i32 = torch.float32, torch.int32
        dev = self.device
        nin = int(np.prod(self.labels_shape))
        nc = self.ncrop = int(np.prod(self.crop_shape))
        no = self.nout = int(np.prod(self.output_shape))
        self.d_labels = torch.empty(nin, dtype=i32, device=dev)
        if self.apply_elastic:
            nh = int(np.prod(self.half_shape)) * 3
            self.d_svf = torch.empty(nh, dtype=f32, device=dev)
            self.d_svf_tmp = torch.empty(nh, dtype=f32, device=dev)
        self.d_seg = torch.empty(nc, dtype=i32, device=dev)
        self.d_chan = torch.empty(n_channels * nc, dtype=f32, device=dev)
        self.d_tmp = [torch.empty(max(nc, no), dtype=f32, device=dev) for _ in range(3)]
        self.d_minmax = torch.empty(2 * n_channels + 2, dtype=torch.int32, device=dev)  # + the real image's pair
        if self.use_real_image:
            self.d_real_in = torch.empty(nin, dtype=f32, device=dev)
            self.d_real = torch.empty(nc, dtype=f32, device=dev)
        self.d_image = torch.empty(no * self.n_image_channels, dtype=f32, device=dev)
        self.d_target = torch.empty(no * self.n_target_channels, dtype=f32, device=dev)
        self.small_cap = 1 << 16  # floats of per-volume small parameters (SVF grid, bias grids, kernels, LUTs)
        # two pinned staging buffers used alternately + an event per buffer: with device-resident labels a step has no host
        # sync, so the host could otherwise rewrite the parameters of step k+1 while the async copy of step k is pending
        self.h_small_ring = [torch.empty(self.small_cap, dtype=f32).pin_memory() for _ in range(2)]
        self.h_small_events = [None, None]
        self.h_small_idx = 0
        self.h_small = self.h_small_ring[0]
        self.d_small = torch.empty(self.small_cap, dtype=f32, device=dev)
        self.d_noise = None
        lut_size = int(self.generation_labels.max()) + 1
        self.lut_size = lut_size
        if self.swap_lut is not None:
            self.d_swap = torch.from_numpy(self.swap_lut.astype(np.int32)).to(dev)
        else:
            self.d_swap = None
        self.fuse_blur = True  # one-pass normalise -> blur -> blur kernel where it applies (generate()); False: separate kernels
        self.host_rng = np.random.Generator(np.random.Philox(key=0))
        self.philox_counter = 0
        self.seed(0)

    # ------------------------------------------------------------------ random draws
    def seed(self, seed, rank=0):
        """per-rank stream: host draws from numpy Philox(key), device noise from Philox4x32-10(key)"""
        key = (int(seed) * 0x9E3779B1 + int(rank) * 0x85EBCA77 + 1) & 0xFFFFFFFFFFFFFFFF
        self.host_rng = np.random.Generator(np.random.Philox(key=key))
        self.philox_key = (key & 0xFFFFFFFF, (key >> 32) & 0xFFFFFFFF)
        self.philox_counter = 0

    def sample_draws(self):
        r = self.host_rng
        d = Draws()

        def U(n):
            return r.random(n, dtype=np.float32)

        def N(n):
            return r.standard_normal(n, dtype=np.float32)

        if self.rotation_bounds is not False:
            d.u_rot = U(3)
        if self.shearing_bounds is not False:
            d.u_shear = U(6)
        if self.scaling_bounds is not False:
            d.u_scale = U(3)
        if self.translation_bounds is not False:
            d.u_trans = U(3)
        if self.apply_elastic:
            d.u_svf_std = U(1)[0]
            d.n_svf = N(int(np.prod(self.small_shape)) * 3)
        if self.crop_shape != self.labels_shape:
            d.u_crop = U(3)
        if self.flipping:
            d.u_flip = U(1)[0]
        d.gmm_noise = None
        d.philox_key = self.philox_key
        d.philox_offset = self.philox_counter
        self.philox_counter += 1
        for i in range(self.n_channels):
            c = {}
            if self.input_channels[i] and self.bias_field_std > 0:
                c['u_bias_std'] = U(1)[0]
                c['n_bias'] = N(int(np.prod(self.small_bias_shape)))
                c['u_bias_gate'] = U(1)[0]
            c['n_gamma'] = N(1)[0]
            if self.input_channels[i]:
                reg = bool(self.simulate_registration_error[i]) and i != self.idx_first_input_channel
                if reg:
                    c['u_regT'] = (U(3), U(3))
                if self.randomise_res[i]:  # SampleResolution draws, ext/lab2im/layers.py:609,625,626,649
                    c['u_rr'] = (U(1), U(3), U(1), U(3))
                if self.blur_range is not None and self.blur_range != 1:
                    c['u_blur'] = U(3)
                if reg:
                    c['u_regE'] = (U(3), U(3))
            d.channels.append(c)
        return d

    def draws_from_tape(self, tape):
        """tape: list of (kind, array) in the reference's call order"""
        it = iter(tape)

        def nxt(kind, size):
            k, a = next(it)
            a = np.asarray(a, dtype=np.float32).reshape(-1)
            assert k == kind and a.size == size, (k, kind, a.size, size)
            return a

        d = Draws()
        if self.rotation_bounds is not False:
            d.u_rot = nxt('u', 3)
        if self.shearing_bounds is not False:
            d.u_shear = nxt('u', 6)
        if self.scaling_bounds is not False:
            d.u_scale = nxt('u', 3)
        if self.translation_bounds is not False:
            d.u_trans = nxt('u', 3)
        if self.apply_elastic:
            d.u_svf_std = nxt('u', 1)[0]
            d.n_svf = nxt('n', int(np.prod(self.small_shape)) * 3)
        if self.crop_shape != self.labels_shape:
            d.u_crop = nxt('u', 3)
        if self.flipping:
            d.u_flip = nxt('u', 1)[0]
        d.gmm_noise = nxt('n', self.ncrop * self.n_channels)
        for i in range(self.n_channels):
            c = {}
            if self.input_channels[i] and self.bias_field_std > 0:
                c['u_bias_std'] = nxt('u', 1)[0]
                c['n_bias'] = nxt('n', int(np.prod(self.small_bias_shape)))
                c['u_bias_gate'] = nxt('u', 1)[0]
            c['n_gamma'] = nxt('n', 1)[0]
            if self.input_channels[i]:
                reg = bool(self.simulate_registration_error[i]) and i != self.idx_first_input_channel
                if reg:
                    c['u_regT'] = (nxt('u', 3), nxt('u', 3))
                if self.randomise_res[i]:
                    c['u_rr'] = (nxt('u', 1), nxt('u', 3), nxt('u', 1), nxt('u', 3))
                if self.blur_range is not None and self.blur_range != 1:
                    c['u_blur'] = nxt('u', 3)
                if reg:
                    c['u_regE'] = (nxt('u', 3), nxt('u', 3))
            d.channels.append(c)
        assert next(it, None) is None, 'tape not fully consumed'
        return d

    # ------------------------------------------------------------------ small-parameter staging
    class _Small:
        def __init__(self, model):
            self.m = model
            self.pos = 0
            model.h_small_idx ^= 1
            ev = model.h_small_events[model.h_small_idx]
            if ev is not None:
                ev.synchronize()       # the copy that last used this buffer (two generate() calls ago) has completed
            model.h_small = model.h_small_ring[model.h_small_idx]
            self.h = model.h_small.numpy()

        def put(self, arr):
            a = np.asarray(arr, dtype=np.float32).reshape(-1)
            n = a.size
            off = (self.pos + 3) & ~3
            if off + n > self.m.small_cap:
                raise ValueError('small-parameter buffer overflow')
            self.h[off:off + n] = a
            self.pos = off + n
            return off

        def flush(self):
            n = self.pos
            self.m.d_small[:n].copy_(self.m.h_small[:n], non_blocking=True)
            ev = self.m.torch.cuda.Event()
            ev.record()
            self.m.h_small_events[self.m.h_small_idx] = ev

        def dptr(self, off):
            return ctypes.c_void_p(self.m.d_small.data_ptr() + 4 * off)

    # ------------------------------------------------------------------ the generator
    def generate(self, labels, means, stds, draws=None, labels_on_device=False, real_image=None):
        """one volume.  labels: int32 [*labels_shape] (numpy, or a device tensor if labels_on_device);
        means/stds: [L, C]; real_image: float [*labels_shape] when the model was built with output_channel=None.  Returns (image [*S, Ci], target [*S, Ct], seg int32 [*S]) device tensors that are
        views of persistent buffers (valid until the next call)."""
        torch, lib = self.torch, self.lib
        st = _lib.stream()
        d = self.sample_draws() if draws is None else draws
        C = self.n_channels
        # ---- labels on device (PadAroundCentre, ext/lab2im/layers.py:1754, is done on the host copy)
        label_bytes = 4
        if labels_on_device:
            lab = labels.reshape(-1)
            assert lab.numel() == self.d_labels.numel() and lab.dtype in (torch.int32, torch.int16, torch.uint8)
            d_labels = lab
            label_bytes = lab.element_size()  # a resident pool keeps its maps in the narrowest type that holds their values
        else:
            lab = np.asarray(labels)
            if self.padding_margin is not None:
                lab = np.pad(lab.reshape(self.input_labels_shape), [(p, p) for p in self.padding_margin])
            self.d_labels.copy_(torch.from_numpy(np.ascontiguousarray(lab, dtype=np.int32).reshape(-1)),
                                non_blocking=False)
            d_labels = self.d_labels
        if self.use_real_image:
            if real_image is None:
                raise ValueError('this model was built with output_channel=None: a real image input is required')
            if torch.is_tensor(real_image):
                assert real_image.numel() == self.d_real_in.numel() and real_image.dtype == torch.float32
                self.d_real_in.copy_(real_image.reshape(-1))
            else:
                real = np.asarray(real_image, dtype=np.float32)
                if self.padding_margin is not None:  # PadAroundCentre on the real image too (:119-120)
                    real = np.pad(real.reshape(self.input_labels_shape), [(p, p) for p in self.padding_margin])
                self.d_real_in.copy_(torch.from_numpy(np.ascontiguousarray(real).reshape(-1)))

        # ---- host-side O(1) parameters
        sm = self._Small(self)
        p = _lib.DeformParams()
        p.label_bytes = label_bytes
        p.in_shape[:] = self.labels_shape
        p.out_shape[:] = self.crop_shape
        crop = [0, 0, 0]
        if self.crop_shape != self.labels_shape:  # RandomCrop, ext/lab2im/layers.py:267
            mx = np.asarray(np.array(self.labels_shape) - np.array(self.crop_shape), dtype=np.float32)
            crop = (np.asarray(d.u_crop, np.float32) * (mx - np.float32(0)) + np.float32(0)).astype(np.int32).tolist()
        p.crop[:] = crop
        flip = bool(self.flipping and np.float32(d.u_flip) < np.float32(0.5))  # layers.py:400
        p.flip = int(flip)
        p.has_affine = int(self.apply_affine)
        A = np.eye(4, dtype=np.float32)
        if self.apply_affine:
            A = hm.sample_affine(dict(rot=d.u_rot, shear=d.u_shear, scale=d.u_scale, trans=d.u_trans),
                                 self.rotation_bounds, self.scaling_bounds, self.shearing_bounds,
                                 self.translation_bounds)
        p.aff[:] = [float(v) for v in A[:3].reshape(-1)]
        p.has_field = int(self.apply_elastic)
        off_svf = None
        if self.apply_elastic:
            p.half_shape[:] = self.half_shape
            std = hm.uniform_f32(d.u_svf_std, 0., self.nonlin_std)  # layers.py:189
            off_svf = sm.put(np.asarray(d.n_svf, np.float32) * std)  # :190
        p.n_channels = C
        p.lut_size = self.lut_size
        p.swap_lut_size = 0 if self.swap_lut is None else int(self.swap_lut.shape[0])
        off_lut = sm.put(hm.gmm_luts(self.generation_labels, means, stds))
        off_bias = None
        gates = []
        bias_grids = []  # deform_gmm_kernel walks the grids back to back (boff += b0*b1*b2): ONE contiguous block
        bias_on_all, bias_elems_before = [], []   # per channel: gate (None = no bias grid), floats of the grids in front of it
        for i in range(C):
            ch = d.channels[i]
            bias_elems_before.append(sum(g_.size for g_ in bias_grids))
            if self.input_channels[i] and self.bias_field_std > 0:
                bstd = hm.uniform_f32(ch['u_bias_std'], 0., self.bias_field_std)  # layers.py:1080
                bias_grids.append((np.asarray(ch['n_bias'], np.float32) * bstd).reshape(-1))
                gate = bool(np.float32(ch['u_bias_gate']) < np.float32(0.95))  # :1090
                bias_on_all.append(gate)
                gates.append(gate)
            else:
                bias_on_all.append(None)
        if bias_grids:
            off_bias = sm.put(np.concatenate(bias_grids))
        p.clip_hi = 300.0  # IntensityAugmentation(clip=300), labels_to_image_model.py:184
        # blur kernels
        k05 = hm.gaussian_kernel([.5] * 3)
        off_k05 = sm.put(k05)
        chan_plan = []
        for i in range(C):
            ch = d.channels[i]
            plan = {}
            if self.input_channels[i] and self.randomise_res[i]:
                plan['rr'] = hm.randomise_res_plan(ch['u_rr'], ch.get('u_blur'), self.blur_range, self.atlas_res,
                                                   self.crop_shape, self.output_shape)
                plan['rr_k'] = [(sm.put(k), [len(k) if a == ax else 1 for a in range(3)])
                                for ax, k in enumerate(plan['rr']['kernels'])]
                plan['k_lr'] = None
            elif self.input_channels[i]:
                sig = hm.blurring_sigma_for_downsampling(self.atlas_res, self.data_res[i], .42, self.thickness[i])
                if hm.is_separable_sigma(sig):  # |sigma| > 5: one 1-D pass per axis (layers.py:720,747-749)
                    ks1 = hm.gaussian_kernels_separable(list(sig), ch.get('u_blur'), self.blur_range)
                    plan['k_lr'] = [(sm.put(k), [len(k) if a == ax else 1 for a in range(3)])
                                    for ax, k in enumerate(ks1) if k is not None] or None
                elif any(sig):
                    k = hm.gaussian_kernel(list(sig), ch.get('u_blur'), self.blur_range)
                    plan['k_lr'] = [(sm.put(k), list(k.shape))]
                else:
                    plan['k_lr'] = None
                if self.downsample[i] and list(self.data_res[i]) != list(self.atlas_res):
                    down = [int(self.crop_shape[k] * self.atlas_res[k] / self.data_res[i][k]) for k in range(3)]
                    plan['down'] = down
                    prof = np.concatenate([hm.reliability_profile(self.output_shape[k], down[k]) for k in range(3)])
                    plan['rel_prof'] = sm.put(prof.astype(np.float32))
            if self.input_channels[i]:
                reg = bool(self.simulate_registration_error[i]) and i != self.idx_first_input_channel
                if reg:
                    T = hm.sample_affine(dict(rot=ch['u_regT'][0], trans=ch['u_regT'][1]), rotation_bounds=5,
                                         translation_bounds=5)
                    Te = hm.sample_affine(dict(rot=ch['u_regE'][0], trans=ch['u_regE'][1]), rotation_bounds=.5,
                                          translation_bounds=.5)
                    plan['T'] = T
                    plan['Tie'] = hm.matmul4(Te, hm.invert_affine(T))
            if self.resample_target and i in self.output_channel:
                sigt = hm.blurring_sigma_for_downsampling(self.atlas_res, self.target_res)
                kt = hm.gaussian_kernel(list(sigt))
                plan['k_tgt'] = (sm.put(kt), list(kt.shape))
            plan['gexp'] = float(np.exp(np.float32(ch['n_gamma']) * np.float32(0.5)))  # layers.py:1240-1242
            chan_plan.append(plan)
        if self.use_real_image and self.resample_target:  # blur to target_res before the final resize (:251-255)
            ktr = hm.gaussian_kernel(list(hm.blurring_sigma_for_downsampling(self.atlas_res, self.target_res)))
            self.real_k_tgt = (sm.put(ktr), list(ktr.shape))
        # noise
        if d.gmm_noise is not None:
            noise = torch.from_numpy(np.ascontiguousarray(d.gmm_noise, dtype=np.float32).reshape(-1))
            if self.d_noise is None or self.d_noise.numel() != noise.numel():
                self.d_noise = torch.empty(noise.numel(), dtype=torch.float32, device=self.device)
            self.d_noise.copy_(noise)
            p.use_philox = 0
        else:
            p.use_philox = 1
            p.philox_key[0], p.philox_key[1] = int(d.philox_key[0]), int(d.philox_key[1])
            p.philox_offset = int(d.philox_offset)
        sm.flush()

        # ---- device pipeline
        i3 = _lib.i3
        tk = _ops.timed      # per-kernel HIP events while bench.py profiles (free otherwise)
        if self.apply_elastic:
            with tk('gen:svf_resize+integrate'):
                _lib.check(lib.synthsr_resize_f32(sm.dptr(off_svf), _lib.ptr(self.d_svf), 3, i3(self.small_shape),
                                                  i3(self.half_shape), 0, st), 'resize(svf)')
                _lib.check(lib.synthsr_svf_integrate(_lib.ptr(self.d_svf), _lib.ptr(self.d_svf_tmp), i3(self.half_shape),
                                                     7, st), 'svf_integrate')
        _lib.check(lib.synthsr_minmax_init(_lib.ptr(self.d_minmax), C + 1, st), 'minmax_init')
        real_mm = ctypes.c_void_p(self.d_minmax.data_ptr() + 8 * C)
        _t_dg = tk('gen:deform_gmm')
        _t_dg.__enter__()
        for c0 in range(0, C, 4):   # one launch per group of four channels (the label gather / real image ride in the first)
            p.n_channels = min(4, C - c0)
            p.chan_first, p.n_channels_total = c0, (C if C > 4 else 0)
            for j in range(4):
                on = bias_on_all[c0 + j] if c0 + j < C else None
                p.bias_on[j] = int(bool(on)) if on is not None else 0
                for k in range(3):
                    p.bias_shape[j][k] = self.small_bias_shape[k] if on is not None else 0
            first = c0 == 0
            _lib.check(lib.synthsr_deform_gmm_real(
                _lib.ptr(d_labels), _lib.ptr(self.d_svf) if self.apply_elastic else None, sm.dptr(off_lut),
                _lib.ptr(self.d_swap) if self.d_swap is not None else None,
                _lib.ptr(self.d_noise) if not p.use_philox else None,
                sm.dptr(off_bias + bias_elems_before[c0]) if off_bias is not None else None,
                _lib.ptr(self.d_seg) if first else None,
                ctypes.c_void_p(self.d_chan.data_ptr() + 4 * c0 * self.ncrop),
                ctypes.c_void_p(self.d_minmax.data_ptr() + 8 * c0),
                _lib.ptr(self.d_real_in) if (self.use_real_image and first) else None,
                _lib.ptr(self.d_real) if (self.use_real_image and first) else None,
                real_mm if (self.use_real_image and first) else None,
                ctypes.byref(p), st), 'deform_gmm')
        _t_dg.__exit__()

        nc, no = self.ncrop, self.nout
        cs, os_ = i3(self.crop_shape), i3(self.output_shape)
        k3 = i3([3, 3, 3])
        img_slot = 0
        tgt_slot = 0
        Ci, Ct = self.n_image_channels, self.n_target_channels

        def fptr(t, off=0):
            return ctypes.c_void_p(t.data_ptr() + 4 * off)

        for i in range(C):
            plan = chan_plan[i]
            x = fptr(self.d_chan, i * nc)
            mm = ctypes.c_void_p(self.d_minmax.data_ptr() + 8 * i)
            t0, t1, t2 = (fptr(t) for t in self.d_tmp)
            is_target = i in self.output_channel
            # the common case (training() defaults: the channel is the single regression target AND a network input, nothing
            # is resampled): normalise + gamma -> blur(.5) -> target -> acquisition blur -> image (+ all-ones map) in ONE pass
            if (self.fuse_blur and is_target and self.input_channels[i] and not self.resample_target and Ct == 1
                    and 'down' not in plan and 'T' not in plan and 'rr' not in plan and plan.get('k_lr') is not None
                    and len(plan['k_lr']) == 1 and list(plan['k_lr'][0][1]) == [3, 3, 3] and list(k05.shape) == [3, 3, 3]):
                fill = img_slot + 1 if self.build_reliability_maps else -1
                with tk('gen:normalise+blur(.5)+blur(lr)+map'):
                    _lib.check(lib.synthsr_normalise_blur2(x, cs, mm, plan['gexp'], sm.dptr(off_k05),
                                                           sm.dptr(plan['k_lr'][0][0]), fptr(self.d_target),
                                                           fptr(self.d_image), Ci, img_slot, fill, 1.0, st),
                               'normalise_blur2')
                tgt_slot += 1
                img_slot += 2 if self.build_reliability_maps else 1
                continue
            with tk('gen:normalise_gamma'):
                _lib.check(lib.synthsr_normalise_gamma(x, x, nc, mm, plan['gexp'], st), 'normalise_gamma')
            # GaussianBlur(sigma=.5): target tap (labels_to_image_model.py:186-196)
            if is_target and not self.resample_target and not self.input_channels[i]:
                _lib.check(lib.synthsr_blur3d(x, fptr(self.d_target), cs, sm.dptr(off_k05), k3, Ct, tgt_slot, -1, 0., st),
                           'blur(target)')
                tgt_slot += 1
                continue
            # the blurred channel IS the regression target when there is a single, unresampled target channel: blur straight
            # into the (then contiguous) target tensor and let the LR blur read it from there -- no staging copy
            # (only when nothing downstream ping-pongs into the buffer: one LR-blur pass, no resampling / registration)
            direct_target = (is_target and not self.resample_target and Ct == 1 and 'down' not in plan and 'T' not in plan
                             and 'rr' not in plan and (plan.get('k_lr') is None or len(plan['k_lr']) == 1))
            if direct_target:
                t0 = fptr(self.d_target)
            with tk('gen:blur3d(.5)'):
                _lib.check(lib.synthsr_blur3d(x, t0, cs, sm.dptr(off_k05), k3, 1, 0, -1, 0., st), 'blur(.5)')
            if direct_target:
                tgt_slot += 1
            elif is_target:
                if self.resample_target:
                    ko, ks = plan['k_tgt']
                    _lib.check(lib.synthsr_blur3d(t0, t1, cs, sm.dptr(ko), i3(ks), 1, 0, -1, 0., st), 'blur(tgt)')
                    _lib.check(lib.synthsr_resize_f32(t1, t2, 1, cs, os_, 0, st), 'resize(tgt)')
                    _lib.check(lib.synthsr_copy_strided(t2, fptr(self.d_target), no, 1, 0, Ct, tgt_slot, st), 'copy(tgt)')
                else:
                    with tk('gen:copy(target)'):
                        _lib.check(lib.synthsr_copy_strided(t0, fptr(self.d_target), nc, 1, 0, Ct, tgt_slot, st),
                                   'copy(tgt)')
                tgt_slot += 1
            if not self.input_channels[i]:
                continue
            cur, oth, oth2 = t0, t1, t2
            if 'T' in plan:  # registration error (labels_to_image_model.py:202-208)
                aff12 = _lib.F12(*[float(v) for v in plan['T'][:3].reshape(-1)])
                _lib.check(lib.synthsr_affine_resample_linear(cur, oth, 1, cs, aff12, st), 'reg T')
                cur, oth = oth, cur
            simple_out = ('down' not in plan) and (not self.resample_target) and ('T' not in plan) and ('rr' not in plan)
            if simple_out:
                # LR blur written straight into the interleaved image (+ all-ones reliability map)
                fill = img_slot + 1 if self.build_reliability_maps else -1
                if plan['k_lr'] is not None:
                    for ko, ks in plan['k_lr'][:-1]:
                        _lib.check(lib.synthsr_blur3d(cur, oth, cs, sm.dptr(ko), i3(ks), 1, 0, -1, 0., st), 'blur(lr)')
                        cur, oth = oth, cur
                    ko, ks = plan['k_lr'][-1]
                    with tk('gen:blur3d(lr)+map'):
                        _lib.check(lib.synthsr_blur3d(cur, fptr(self.d_image), cs, sm.dptr(ko), i3(ks), Ci, img_slot, fill,
                                                      1.0, st), 'blur(lr)')
                else:
                    _lib.check(lib.synthsr_copy_strided(cur, fptr(self.d_image), nc, 1, 0, Ci, img_slot, st), 'copy')
                    if fill >= 0:
                        self.d_image.view(-1, Ci)[:, fill] = 1.0
                img_slot += 2 if self.build_reliability_maps else 1
                continue
            if 'rr' in plan:  # separable dynamic blur, then acquisition mimicking with the distance map (:215-220)
                for ko, ks in plan['rr_k']:
                    _lib.check(lib.synthsr_blur3d(cur, oth, cs, sm.dptr(ko), i3(ks), 1, 0, -1, 0., st), 'blur(rr)')
                    cur, oth = oth, cur
                rr = plan['rr']
                F3 = ctypes.c_float * 3
                if 'Tie' not in plan:  # straight into the interleaved image (+ distance map)
                    dmap = img_slot + 1 if self.build_reliability_maps else -1
                    _lib.check(lib.synthsr_mimic_acquisition(cur, fptr(self.d_image), cs, os_, F3(*rr['down_zoom']),
                                                             F3(*rr['up_zoom']), F3(*rr['res']), Ci, img_slot, dmap, st),
                               'mimic_acquisition')
                    img_slot += 2 if self.build_reliability_maps else 1
                    continue
                # registration error afterwards (:231-238): volume and distance map to separate buffers, both moved by Tie
                _lib.check(lib.synthsr_mimic_acquisition(cur, oth, cs, os_, F3(*rr['down_zoom']), F3(*rr['up_zoom']),
                                                         F3(*rr['res']), 1, 0, -1, st), 'mimic_acquisition')
                _lib.check(lib.synthsr_mimic_acquisition(cur, oth2, cs, os_, F3(*rr['down_zoom']), F3(*rr['up_zoom']),
                                                         F3(*rr['res']), 1, -1, 0, st), 'mimic_acquisition(dist)')
                aff12 = _lib.F12(*[float(v) for v in plan['Tie'][:3].reshape(-1)])
                _lib.check(lib.synthsr_affine_resample_linear(oth, cur, 1, os_, aff12, st), 'reg Tie')
                _lib.check(lib.synthsr_copy_strided(cur, fptr(self.d_image), no, 1, 0, Ci, img_slot, st), 'copy(img)')
                img_slot += 1
                if self.build_reliability_maps:
                    _lib.check(lib.synthsr_affine_resample_linear(oth2, cur, 1, os_, aff12, st), 'reg Tie(dist)')
                    _lib.check(lib.synthsr_copy_strided(cur, fptr(self.d_image), no, 1, 0, Ci, img_slot, st), 'copy(dist)')
                    img_slot += 1
                continue
            for ko, ks in (plan['k_lr'] or []):
                _lib.check(lib.synthsr_blur3d(cur, oth, cs, sm.dptr(ko), i3(ks), 1, 0, -1, 0., st), 'blur(lr)')
                cur, oth = oth, cur
            cur_shape = self.crop_shape
            if 'down' in plan:  # et.resample_tensor, edit_tensors.py:295-304
                _lib.check(lib.synthsr_resize_f32(cur, oth, 1, cs, i3(plan['down']), 1, st), 'resize(nearest)')
                cur, oth = oth, cur
                cur_shape = plan['down']
            if list(cur_shape) != list(self.output_shape):
                _lib.check(lib.synthsr_resize_f32(cur, oth, 1, i3(cur_shape), os_, 0, st), 'resize(linear)')
                cur, oth = oth, cur
            rel_buf = None
            if self.build_reliability_maps or 'Tie' in plan:
                if 'down' in plan:
                    _lib.check(lib.synthsr_outer3(sm.dptr(plan['rel_prof']), oth2, os_, 1, 0, st), 'outer3')
                else:
                    self.d_tmp[2][:no] = 1.0
                rel_buf = oth2
            if 'Tie' in plan:  # labels_to_image_model.py:231-238
                aff12 = _lib.F12(*[float(v) for v in plan['Tie'][:3].reshape(-1)])
                _lib.check(lib.synthsr_affine_resample_linear(cur, oth, 1, os_, aff12, st), 'reg Tie')
                cur, oth = oth, cur
                if self.build_reliability_maps:
                    _lib.check(lib.synthsr_affine_resample_linear(rel_buf, oth, 1, os_, aff12, st), 'reg Tie(map)')
                    rel_buf = oth
            _lib.check(lib.synthsr_copy_strided(cur, fptr(self.d_image), no, 1, 0, Ci, img_slot, st), 'copy(img)')
            img_slot += 1
            if self.build_reliability_maps:
                _lib.check(lib.synthsr_copy_strided(rel_buf, fptr(self.d_image), no, 1, 0, Ci, img_slot, st), 'copy(map)')
                img_slot += 1

        if self.use_real_image:  # IntensityAugmentation(normalise=True) on the deformed scan (:250), then to target_res
            r = fptr(self.d_real)
            t0, t1, t2 = (fptr(t) for t in self.d_tmp)
            _lib.check(lib.synthsr_normalise_gamma(r, r, nc, real_mm, 0.0, st), 'normalise(real)')
            if self.resample_target:
                ko, ks = self.real_k_tgt
                _lib.check(lib.synthsr_blur3d(r, t1, cs, sm.dptr(ko), i3(ks), 1, 0, -1, 0., st), 'blur(real tgt)')
                _lib.check(lib.synthsr_resize_f32(t1, t2, 1, cs, os_, 0, st), 'resize(real tgt)')
                _lib.check(lib.synthsr_copy_strided(t2, fptr(self.d_target), no, 1, 0, 1, 0, st), 'copy(real tgt)')
            else:
                _lib.check(lib.synthsr_copy_strided(r, fptr(self.d_target), nc, 1, 0, 1, 0, st), 'copy(real tgt)')
        image = self.d_image.view(*self.output_shape, Ci)
        target = self.d_target.view(*self.output_shape, Ct)
        seg = self.d_seg.view(*self.crop_shape)
        return image, target, seg

    # ------------------------------------------------------------------ Keras-like protocol
    def __call__(self, inputs, draws=None):
        """inputs = [labels [B,*S,1], means [B,L,C], stds [B,L,C]] -> [image [B,*S',Ci], target [B,*S',Ct]] (device)"""
        labels, means, stds = inputs[:3]
        labels = np.asarray(labels)
        real = np.asarray(inputs[3]) if self.use_real_image else None  # 4th Keras input 'real_image_input'
        B = labels.shape[0]
        images, targets = [], []
        means_b, stds_b = hm.batch_gmm_parameters(means, stds, self.sum_gmm_over_batch)
        for b in range(B):
            img, tgt, _ = self.generate(labels[b, ..., 0], means_b[b], stds_b[b],
                                        None if draws is None else draws[b],
                                        real_image=None if real is None else real[b, ..., 0])
            if B > 1:
                img, tgt = img.clone(), tgt.clone()
            images.append(img)
            targets.append(tgt)
        torch = self.torch
        return [torch.stack(images, 0), torch.stack(targets, 0)]

    def predict(self, inputs, draws=None):
        image, target = self(inputs, draws)
        return [image.cpu().numpy(), target.cpu().numpy()]


def labels_to_image_model(labels_shape,
                          input_channels,
                          output_channel,
                          generation_labels,
                          n_neutral_labels,
                          atlas_res,
                          target_res,
                          output_shape=None,
                          output_div_by_n=None,
                          padding_margin=None,
                          flipping=True,
                          aff=None,
                          scaling_bounds=0.15,
                          rotation_bounds=15,
                          shearing_bounds=0.012,
                          translation_bounds=False,
                          nonlin_std=3.,
                          nonlin_shape_factor=.0625,
                          simulate_registration_error=True,
                          randomise_res=False,
                          data_res=None,
                          thickness=None,
                          downsample=False,
                          build_reliability_maps=False,
                          blur_range=1.15,
                          bias_field_std=.3,
                          bias_shape_factor=.025,
                          device=None):
    """same signature and defaults as SynthSR/labels_to_image_model.py:32-58 (+ `device`)"""
    return LabelsToImageModel(labels_shape, input_channels, output_channel, generation_labels, n_neutral_labels,
                              atlas_res, target_res, output_shape, output_div_by_n, padding_margin, flipping, aff,
                              scaling_bounds, rotation_bounds, shearing_bounds, translation_bounds, nonlin_std,
                              nonlin_shape_factor, simulate_registration_error, randomise_res, data_res, thickness,
                              downsample, build_reliability_maps, blur_range, bias_field_std, bias_shape_factor,
                              device=device)


get_shapes = hm.get_shapes
