"""CPU tests: host logic (host_math, volumes, nifti, model_inputs), C-ABI exports, 2-rank gloo gradient reducer."""
import os
import re
import sys
import ctypes
import subprocess
import numpy as np
import pytest

from conftest import load_golden, tape_from_golden, REPO
from synthsr_amd import host_math as hm
from oracle import generator_ref as R


def test_host_math_matches_goldens_and_oracle():
    g = load_golden('host_math')
    tape = tape_from_golden(g, 'affine_tape')
    for b in range(2):
        u = dict(rot=tape[0][1][b], shear=tape[1][1][b], scale=tape[2][1][b], trans=tape[3][1][b])
        T = hm.sample_affine(u, rotation_bounds=15, scaling_bounds=.15, shearing_bounds=.02, translation_bounds=5)
        np.testing.assert_array_equal(T, g['affine_T'][b])
    for name, sig in [('k050', [.5] * 3), ('k042', [.42] * 3), ('khyp', [.63, .63, 2.1])]:
        np.testing.assert_array_equal(hm.gaussian_kernel(sig), R.gaussian_kernel(sig))
        np.testing.assert_allclose(hm.gaussian_kernel(sig), g['gk_' + name], atol=1e-7)
    np.testing.assert_array_equal(hm.gaussian_kernel([.42] * 3, g['gk_rand_tape_00'], 1.15),
                                  R.gaussian_kernel([.42] * 3, g['gk_rand_tape_00'], 1.15))
    np.testing.assert_allclose(hm.blurring_sigma_for_downsampling([1.] * 3, [1.5, 1.5, 5.], .42, [1.5, 1.5, 5.]),
                               g['sigma_lr'])
    np.testing.assert_array_equal(hm.flip_swap_lut(g['swap_lut_labels'], 3), g['swap_lut'])
    assert hm.flip_swap_lut(np.arange(4), 4) is None
    np.testing.assert_array_equal(hm.reliability_profile(192, 38), R.reliability_map_1d(192, 38))


def test_get_shapes_matches_golden():
    g = load_golden('host_math')
    cases = [([160, 160, 160], None, [1.] * 3, [1.] * 3, None, 32), ([148, 187, 155], None, [1.] * 3, [1.] * 3, None, 32),
             ([148, 187, 155], 128, [1.] * 3, [1.] * 3, None, 32), ([148, 187, 155], [96, 128, 100], [1.] * 3, [1.] * 3, 8, 32),
             ([148, 187, 155], 160, [1.] * 3, [.7] * 3, None, 32), ([192, 192, 192], 192, [1.] * 3, [1.] * 3, None, None),
             ([40, 48, 36], 32, [1.] * 3, [1.] * 3, None, 32)]
    for c, ref in zip(cases, g['get_shapes_out']):
        crop, out, _ = hm.get_shapes(*c)
        assert list(crop) + list(out) == list(ref)


def test_gmm_lut_accumulates_duplicates_and_unknown_labels_are_zero():
    lut = hm.gmm_luts(np.array([0, 2, 2, 5]), np.array([[1.], [2.], [3.], [4.]]), np.ones((4, 1)))
    assert lut.shape == (2, 1, 6)
    np.testing.assert_array_equal(lut[0, 0], [1, 0, 5, 0, 0, 4])  # tf.scatter_nd adds duplicates; label 1,3,4 -> 0


def test_reformat_helpers():
    assert hm.reformat_to_list(3, length=3) == [3, 3, 3]
    assert hm.reformat_to_list(np.array([1, 2, 3]), length=3, dtype='int') == [1, 2, 3]
    with pytest.raises(ValueError):
        hm.reformat_to_list([1, 2], length=3)
    a = hm.reformat_to_n_channels_array([1.5, 1.5, 5.], 3, 2)
    assert a.shape == (2, 3) and a[1, 2] == 5
    assert hm.get_padding_margin(160, 128) == 16 and hm.get_padding_margin(None, 3) is None
    assert hm.find_closest_number_divisible_by_m(187, 32) == 160


def test_nifti_roundtrip_and_orientation(tmp_path):
    from synthsr_amd import volumes
    from synthsr_amd.nifti import write_nifti, read_nifti
    rng = np.random.default_rng(0)
    vol = rng.integers(0, 30, (12, 13, 14)).astype(np.float32)
    aff = np.array([[0, -1.5, 0, 10], [1.2, 0, 0, -4], [0, 0, 2., 3], [0, 0, 0, 1.]])  # axes swapped + one flipped
    p = str(tmp_path / 'v.nii.gz')
    write_nifti(p, vol, aff)
    d, a, h = read_nifti(p)
    np.testing.assert_array_equal(d, vol)
    np.testing.assert_allclose(a, aff, atol=1e-6)
    v2, a2, _ = volumes.load_volume(p, im_only=False, dtype='int', aff_ref=np.eye(4))
    assert v2.shape == (13, 12, 14) and v2.dtype == np.int64
    np.testing.assert_array_equal(volumes.get_ras_axes(a2), [0, 1, 2])
    assert all(np.diag(a2)[:3] > 0)
    back = volumes.align_volume_to_ref(v2, a2, aff_ref=aff)
    np.testing.assert_array_equal(back, vol.astype(np.int64))
    shape, aff_o, n_dims, n_ch, _, res = volumes.get_volume_info(p, aff_ref=np.eye(4))
    assert shape == [13, 12, 14] and n_dims == 3 and n_ch == 1
    np.testing.assert_allclose(res, [1.5, 1.2, 2.], atol=1e-6)


def test_nifti_without_sform_qform_gets_the_centred_base_affine(tmp_path):
    """sform_code = qform_code = 0: nibabel (what the reference loads volumes with) falls back to the base affine -- zooms on
    the diagonal, first axis flipped, origin at the centre voxel"""
    import gzip
    from synthsr_amd.nifti import write_nifti, read_nifti
    vol = np.arange(4 * 5 * 6, dtype=np.float32).reshape(4, 5, 6)
    p = str(tmp_path / 'v.nii.gz')
    write_nifti(p, vol, np.diag([2., 3., 4., 1.]))
    raw = bytearray(gzip.open(p, 'rb').read())
    raw[252:256] = b'\0' * 4                      # qform_code, sform_code (two int16)
    with gzip.open(p, 'wb') as f:
        f.write(bytes(raw))
    d, aff, _ = read_nifti(p)
    np.testing.assert_array_equal(d, vol)
    exp = np.diag([-2., 3., 4., 1.])
    exp[:3, 3] = [2. * 1.5, -3. * 2.0, -4. * 2.5]
    np.testing.assert_allclose(aff, exp, atol=1e-6)


def test_get_list_labels_fs_sort():
    from synthsr_amd import volumes
    from synthsr_amd.synthetic import GENERATION_LABELS
    lab, n_neutral = volumes.get_list_labels(label_list=GENERATION_LABELS[::-1].copy(), FS_sort=True)
    np.testing.assert_array_equal(lab, GENERATION_LABELS)  # neutral sorted, then left sorted
    assert n_neutral == 19  # only one hemisphere present -> all neutral (utils.py:275-278)
    lab2, n2 = volumes.get_list_labels(label_list=[0, 41, 2, 14], FS_sort=True)
    np.testing.assert_array_equal(lab2, [0, 14, 2, 41])
    assert n2 == 2
    with pytest.raises(Exception):
        volumes.get_list_labels(label_list=[0, 99999], FS_sort=True)


def test_model_inputs_generator_protocol():
    from synthsr_amd.model_inputs import build_model_inputs
    from synthsr_amd.synthetic import GENERATION_CLASSES, PRIOR_MEANS_T1_HR, PRIOR_STDS_T1_HR
    maps = [np.zeros((4, 5, 6), np.int32), np.ones((4, 5, 6), np.int32)]
    gen = build_model_inputs(None, 19, PRIOR_MEANS_T1_HR, PRIOR_STDS_T1_HR, 'normal', n_channels=1,
                             generation_classes=GENERATION_CLASSES, rng=np.random.default_rng(0), label_maps=maps)
    labels, means, stds = next(gen)
    assert labels.shape == (1, 4, 5, 6, 1) and means.shape == (1, 19, 1) and stds.shape == (1, 19, 1)
    assert (means >= 0).all() and (stds >= 0).all()
    assert means[0, 1, 0] == means[0, 2, 0]  # labels 14 and 15 share class 3
    with pytest.raises(ValueError):
        next(build_model_inputs(None, 19, PRIOR_MEANS_T1_HR, PRIOR_STDS_T1_HR, 'normal', n_channels=2,
                                generation_classes=GENERATION_CLASSES, label_maps=maps))
    gen2 = build_model_inputs(None, 19, None, None, 'uniform', n_channels=2, batchsize=2, label_maps=maps,
                              rng=np.random.default_rng(1))
    l2, m2, s2 = next(gen2)
    assert l2.shape == (2, 4, 5, 6, 1) and m2.shape == (2, 19, 2)
    assert (m2 >= 25).all() and (m2 <= 225).all() and (s2 >= 5).all() and (s2 <= 25).all()


def test_c_abi_exports_every_declared_symbol():
    """the shared library loads (no GPU needed) and exports every function include/synthsr_hip.h declares"""
    from synthsr_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from synthsr_amd import build
        build.build(verbose=False)
    import glob
    hdr = ''.join(open(h).read() for h in sorted(glob.glob(os.path.join(REPO, 'include', '*.h'))))
    declared = set(re.findall(r'\b(synthsr_[a-z0-9_]+)\s*\(', hdr))
    # the process-wide option switch / arithmetic setter / layout epoch of rounds 1-4 are gone from the ABI
    for gone in ('synthsr_conv3d_set_option', 'synthsr_set_conv_arithmetic', 'synthsr_conv_arithmetic', 'synthsr_conv3d_layout_epoch'):
        assert gone not in declared
        with pytest.raises(AttributeError):
            getattr(ctypes.CDLL(_lib.LIB_PATH), gone)
    assert len(declared) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), 'missing export %s' % name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib.synthsr_abi_version.restype = ctypes.c_int
    assert lib.synthsr_abi_version() == 2
    # argument validation happens before any HIP call: usable without a GPU
    lib = _lib.load()
    big = _lib.i3([160, 160, 160])
    # packed sizes (query mode).  NULL context = split arithmetic (the default): 3 pieces x [co-chunk][8-channel chunk][7 K steps]
    # [co tiles] fragments of 64 lanes x 8 bf16 (= 4 floats); Cout = 24: the three pieces stacked along M -- [8-channel chunk][7 K
    # steps][5 row tiles] fragments instead of 3 x 2
    assert lib.synthsr_conv3d_pack(None, None, None, big, 24, 24, 0, None) == 3 * 7 * 5 * 64 * 4
    assert lib.synthsr_conv3d_pack(None, None, None, _lib.i3([80, 80, 80]), 48, 48, 0, None) == 3 * 6 * 7 * 3 * 64 * 4
    assert lib.synthsr_conv3d_pack(None, None, None, big, 0, 24, 0, None) == -1
    bad = _lib.ConvCtx(arithmetic=3)
    assert lib.synthsr_conv3d_pack(ctypes.byref(bad), None, None, big, 24, 24, 0, None) == -1
    for kw in (dict(reserved0=1), dict(reserved=(ctypes.c_int * 2)(0, 7)), dict(workspace_bytes=16)):   # reserved fields are
        bad = _lib.ConvCtx(arithmetic=1, **kw)                                # checked; a workspace size needs a workspace
        assert lib.synthsr_conv3d_pack(ctypes.byref(bad), None, None, big, 24, 24, 0, None) == -1, kw
    assert ctypes.sizeof(_lib.ConvCtx) == 32 and int(lib.synthsr_conv_workspace_bytes()) == 16 << 20


def test_two_conv_contexts_with_different_arithmetic_coexist():
    """include/synthsr_hip.h: the arithmetic of the fp32 convolutions is a field of the caller's synthsr_conv_ctx, not state of the
    library.  Two contexts used alternately -- and from two threads at once -- each keep their own plans and
    packed-weight sizes; NULL means split.  Host-only entry points: no GPU needed."""
    import threading
    from synthsr_amd import _lib
    lib = _lib.load()
    split, split9, mfma = (_lib.ConvCtx(arithmetic=a) for a in (1, 2, 0))
    big, mid = _lib.i3([160, 160, 160]), _lib.i3([80, 80, 80])

    def sizes(ctx):
        c = None if ctx is None else ctypes.byref(ctx)
        out = (ctypes.c_int64 * 8)()
        assert lib.synthsr_conv3d_plan(c, big, 24, 24, 1, out) == 0
        return (int(lib.synthsr_conv3d_pack(c, None, None, big, 24, 24, 0, None)),
                int(lib.synthsr_conv3d_pack(c, None, None, mid, 48, 48, 0, None)), int(out[2]) <= -100,
                int(lib.synthsr_conv3d_wgrad_runs_split(c, mid, 48, 48)))
    want_split = (3 * 7 * 5 * 64 * 4, 3 * 6 * 7 * 3 * 64 * 4, True, 1)
    # fp32 MFMA: Cout = 24 uses the unpadded 4x4x1-MFMA layout, 48 -> 48 the 16x16x4 B-fragment layout
    want_mfma = (27 * 24 * 24, 2 * 27 * 3 * 3 * 128, False, 0)
    for _ in range(3):     # interleaved: no call leaves anything behind for the next one
        assert sizes(split) == want_split
        assert sizes(mfma) == want_mfma
        assert sizes(None) == want_split
        assert sizes(split9)[2:] == want_split[2:]      # same kernels and plans, nine products
    errors = []

    def worker(ctx, want):
        try:
            for _ in range(2000):
                if sizes(ctx) != want:
                    errors.append((ctx.arithmetic, sizes(ctx)))
                    return
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    threads = [threading.Thread(target=worker, args=a) for a in ((split, want_split), (mfma, want_mfma), (split, want_split), (mfma, want_mfma))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_split_tile_schedule_visits_every_tile_once_and_evenly():
    """synthsr_split_tile_schedule = the function the split kernels evaluate on the device (csrc/conv_split.hip: tile_walk_of,
    host-callable): for every launch geometry of the U-Net levels at 160^3 / 192^3 and a sweep of awkward tile counts and grid
    widths, the workgroups of one (y, z) column together visit each tile exactly once, no workgroup gets more than one tile
    above the mean, and a launch is never narrower than its tile count when it could be wider"""
    from synthsr_amd import _lib
    lib = _lib.load()
    out = (ctypes.c_int * 4)()

    def column(kernel, ntiles, ny, yz):
        assert lib.synthsr_split_tile_schedule(kernel, ntiles, ny, 0, yz, out) == 0
        gx = out[0]
        seen, loads = np.zeros(ntiles, dtype=np.int32), []
        for b in range(gx):
            assert lib.synthsr_split_tile_schedule(kernel, ntiles, ny, b, yz, out) == 0
            assert out[0] == gx and out[3] >= 1
            tiles = list(range(out[1], out[2], out[3]))
            assert all(0 <= t < ntiles for t in tiles)
            seen[tiles] += 1
            loads.append(len(tiles))
        assert np.all(seen == 1), (kernel, ntiles, ny, yz, gx)
        assert max(loads) <= -(-ntiles // gx) + (1 if gx >= 8 and gx % 8 else 0), (kernel, ntiles, ny, gx, max(loads))
        return gx, loads

    def ntiles_of(d):
        return -(-d // 4) * -(-d // 4) * -(-d // 16)

    for size in (160, 192, 80, 96, 40, 48, 44):
        nt = ntiles_of(size)
        for ny in (1, 2, 3, 4, 6, 8, 12, 24):
            for yz in (0, 1, ny - 1, 5):
                gx, loads = column(0, nt, ny, yz)
                assert gx % 8 == 0 or nt < 8
                assert gx >= min(nt, (512 // ny) // 8 * 8) or gx * ny > 512 - 8 * ny
                column(1, nt, ny, yz)
        for groups in (2, 4):
            for yz in range(groups):
                column(2, nt, groups, yz)
    # the weight-gradient widths of the network: 256 // (chunks x column groups) = 85, 42, 21, 10 are not multiples of 8
    # (round 4, 16 / 24 input channels per workgroup: 160^3 24->24 one chunk, 80^3 48->48 three, 40^3 96->96 6 x 2,
    #  20^3 192->192 12 x 4, 10^3 384->384 24 x 8 chunks x column groups)
    for nt, ny, width in ((16000, 3, 85), (2000, 6, 42), (300, 12, 21), (300, 24, 10), (16000, 1, 256), (2000, 3, 85),
                          (50, 48, 5), (9, 192, 1)):
        gx, loads = column(1, nt, ny, min(2, ny - 1))
        assert gx == width and max(loads) - min(loads) <= 1, (gx, loads)
    # awkward small cases: fewer tiles than XCDs, one tile more than a multiple of 8, primes
    for nt in (1, 2, 7, 8, 9, 50, 63, 65, 257, 300, 301, 1031):
        for ny in (1, 2, 5, 24, 64):
            for kernel in (0, 1):
                for yz in (0, 3):
                    column(kernel, nt, ny, yz)
    gx, loads = column(0, 300, 1, 0)        # 300 tiles: 304 workgroups with at most one tile each, not 296 with 8 two-tile stragglers
    assert gx == 304 and max(loads) == 1
    assert lib.synthsr_split_tile_schedule(0, 300, 1, 304, 0, out) == -1 and lib.synthsr_split_tile_schedule(3, 300, 1, 0, 0, out) == -1


def test_conv_arithmetic_switch_and_layer_plans():
    """ops.set_conv_arithmetic / conv_runs_split (host logic, no GPU): the split arithmetic is the default and covers the
    layers with >= 256 tiles of 4x4x16 voxels and channel counts that are multiples of 8"""
    from synthsr_amd import ops
    assert ops.conv_arithmetic() == 'split'
    with pytest.raises(ValueError):
        ops.set_conv_arithmetic('bf16')
    big = (160, 160, 160)
    assert ops.conv_runs_split('conv3d_fwd', big, 24, 24) and ops.conv_runs_split('conv3d_dgrad', big, 24, 24)
    assert ops.conv_runs_split('conv3d_wgrad', big, 24, 24) and ops.conv_runs_split('conv3d_wgrad', (80, 80, 80), 48, 48)
    assert ops.conv_runs_split('conv3d_fwd', (40, 40, 40), 96, 96)
    assert not ops.conv_runs_split('conv3d_fwd', big, 2, 24)              # first layer: 2 input channels
    assert ops.conv_runs_split('conv3d_fwd', (20, 20, 20), 192, 192)      # round 4: 50 tiles x 4 co-chunks = 200 workgroups
    assert not ops.conv_runs_split('conv3d_fwd', (20, 20, 20), 192, 96)   # 100 workgroups: fp32 MFMA kernels
    assert not ops.conv_runs_split('conv3d_fwd', (10, 10, 10), 384, 384)
    assert ops.conv_runs_split('conv3d_wgrad', (20, 20, 20), 192, 192)    # round 4: the split weight gradient at every size
    assert not ops.conv_runs_split('conv3d_wgrad', big, 2, 24) and not ops.conv_runs_split('conv3d_wgrad', big, 24, 16)
    assert ops.conv_runs_split('conv3d_up_fwd', (80, 80, 80), 48, 24)     # folded decoder conv, up-sampled channels (low-res grid)
    assert ops.conv_runs_split('conv3d_up_dgrad', (80, 80, 80), 48, 24)
    assert not ops.conv_runs_split('conv3d_up_fwd', (20, 20, 20), 192, 96)
    assert ops.conv_runs_split('conv3d_up_wgrad', (80, 80, 80), 48, 24)       # (round 6) eight parities over one staged x halo
    assert not ops.conv_runs_split('conv3d_up_wgrad', (80, 80, 80), 40, 24)   # 16-channel input chunks
    prev = ops.set_conv_arithmetic('fp32_mfma')
    try:
        assert prev == 'split' and ops.conv_arithmetic() == 'fp32_mfma'
        assert not ops.conv_runs_split('conv3d_fwd', big, 24, 24)
    finally:
        ops.set_conv_arithmetic(prev)
    assert ops.conv_arithmetic() == 'split'


def _repo_root():
    return os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_layout_epoch_follows_the_default_context():
    """synthsr_amd.ops keeps ONE default conv context for the host code that passes none; conv_layout_epoch moves when (and only
    when) that context is replaced -- what UNet3D / Critic3D re-pack on.  The library itself has no such state."""
    from synthsr_amd import ops
    e0 = ops.conv_layout_epoch()
    first = ops.conv_arithmetic()
    ops.set_conv_arithmetic(first)
    assert ops.conv_layout_epoch() == e0                      # same value: nothing to re-pack
    other = 'fp32_mfma' if first != 'fp32_mfma' else 'split'
    ops.set_conv_arithmetic(other)
    e1 = ops.conv_layout_epoch()
    assert e1 != e0 and ops.conv_arithmetic() == other
    ops.set_conv_arithmetic(first)
    assert ops.conv_layout_epoch() != e1 and ops.conv_arithmetic() == first
    assert not any(k.startswith('SYNTHSR_CONV') for k in open(os.path.join(_repo_root(), 'synthsr_amd', '_lib.py')).read().split("'"))


def test_refused_segmentation_loss_configurations_break_the_reference_graph_too():
    """The two segmentation-loss configurations this build refuses with a ValueError are ones the reference's own graph
    cannot be built for: tests/golden/seg_loss_limits.json records what add_seg_loss_to_model does on the shim
    (tests/golden/gen/make_unet_goldens.py seg_limits) -- the control builds and gives a loss; two regression targets fail
    where the single-channel segmentation unet is applied to `predicted_image` (metrics_model.py:165); a segmentation target
    on another grid fails the reference's OWN assertion in DiceLoss.build (ext/lab2im/layers.py:1314)."""
    import json
    with open(os.path.join(REPO, 'tests', 'golden', 'seg_loss_limits.json')) as f:
        rec = json.load(f)
    assert rec['control_one_target_same_grid']['raised'] is False and rec['control_one_target_same_grid']['total_loss'] > 0
    two = rec['two_regression_targets']
    assert two['raised'] and two['reference_frame'].startswith('SynthSR/metrics_model.py:')
    grid = rec['segmentation_target_on_another_grid']
    assert grid['raised'] and grid['reference_frame'].startswith('ext/lab2im/layers.py:') and \
        'same shape' in grid['message']
    # ours: refused before anything is built (no GPU, no files needed)
    from synthsr_amd.fine_tuning_with_adversary import training as adv_training
    with pytest.raises(ValueError, match='ONE regression target'):
        adv_training('/nonexistent/labels', None, '/nonexistent/models', None, None, '/nonexistent/gl.npy',
                     segmentation_model_file='/nonexistent/seg.npz', input_channels=[True, True], output_channel=[0, 1])


def test_product_fails_loudly_without_the_library(monkeypatch, tmp_path):
    from synthsr_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.SynthSRHipError):
        _lib.load()


_GLOO_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from synthsr_amd.training import GradBucketReducer
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
n = 10007
g = torch.arange(n, dtype=torch.float32) * (rank + 1)
red = GradBucketReducer(g, bucket_elems=1500)
red.start()
# the backward reports readiness tail-first at layer boundaries
for lo in (9000, 8800, 6000, 5999, 1200, 1000):
    red.ready(lo)
scale = red.finish()
exp = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
assert torch.equal(g, exp), (g[:5], exp[:5])
assert abs(scale - 1.0 / world) < 1e-12
# weights broadcast + identical update on both ranks
p = torch.full((5,), float(rank))
dist.broadcast(p, 0)
assert torch.equal(p, torch.zeros(5))
dist.barrier()
print('OK', rank)
'''


# world size 8 = BASELINE.json configs[2]: bucket order (tail first, contiguous, the whole buffer exactly once), the 1 / world
# gradient scale, and -- the point of data parallelism -- identical weights on every rank after the Keras-Adam update of the
# averaged gradient (oracle.unet_ref.adam_keras: the restated formula the HIP adam_kernel is tested against)
_GLOO_WORKER_8 = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from synthsr_amd.training import GradBucketReducer
from oracle import unet_ref as U
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
n = 100003
gen = torch.Generator().manual_seed(100 + rank)
g = torch.randn(n, generator=gen)                       # every rank its own gradient
mine = g.clone()
everyone = [torch.empty(n) for _ in range(world)]
dist.all_gather(everyone, mine)
red = GradBucketReducer(g, bucket_elems=16384)
red.start()
for lo in (99000, 90000, 83000, 60000, 59999, 30000, 12000, 100):
    red.ready(lo)
scale = red.finish()
assert abs(scale - 1.0 / world) < 1e-12
assert all(a[0] == b[1] for a, b in zip(red.ranges, red.ranges[1:])), red.ranges           # tail first, contiguous
assert red.ranges[0][1] == n and red.ranges[-1][0] == 0 and red.bytes_launched == 4 * n      # the whole buffer, once
assert all(hi - lo >= 16384 for lo, hi in red.ranges[:-1])                                   # full buckets but the last
exp = torch.stack(everyone).double().sum(0)
assert float((g.double() - exp).abs().max()) < 1e-5, float((g.double() - exp).abs().max())
p = torch.full((n,), float(rank))                        # weights: rank 0's, broadcast
dist.broadcast(p, 0)
p2, m, v = U.adam_keras(p, g * scale, torch.zeros(n), torch.zeros(n), 1, lr=1e-3)
digest = torch.stack([p2.double().sum(), p2.double().abs().max(), m.double().sum(), v.double().sum()])
lo_, hi_ = digest.clone(), digest.clone()
dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
assert torch.equal(lo_, hi_), (lo_, hi_)                 # bit-identical weights and moments on all ranks
dist.barrier()
print('OK', rank)
'''


def _run_gloo_worker(tmp_path, text, world):
    script = tmp_path / 'worker.py'
    script.write_text(text)
    port = 29500 + (os.getpid() % 2000) + world
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % world, '--master-addr',
           '127.0.0.1', '--master-port', str(port), str(script), REPO]
    env = dict(os.environ, OMP_NUM_THREADS='1')
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count('OK') == world


def test_grad_bucket_reducer_two_ranks_gloo(tmp_path):
    _run_gloo_worker(tmp_path, _GLOO_WORKER, 2)


def test_grad_bucket_reducer_eight_ranks_gloo(tmp_path):
    _run_gloo_worker(tmp_path, _GLOO_WORKER_8, 8)


@pytest.mark.parametrize('script', ['training', 'predict_command_line', 'predict_command_line_hyperfine'])
def test_command_line_interfaces_match_reference(script):
    """every flag of the reference's launcher scripts (tests/golden/cli.json, extracted from their `add_argument` calls)
    exists here with the same destination, default and kind, and parses to the same value"""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, 'tests', 'golden', 'cli.json')) as f:
        ref = json.load(f)[script]
    spec = importlib.util.spec_from_file_location('cli_' + script, os.path.join(root, 'scripts', script + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    parser = mod.build_parser()
    actions = {a.option_strings[0] if a.option_strings else a.dest: a for a in parser._actions if a.dest != 'help'}
    positional = [r['name'] for r in ref if not r['name'].startswith('-')]
    assert [a.dest for a in parser._actions if not a.option_strings] == positional
    for r in ref:
        a = actions[r['name']]
        assert a.dest == r['dest'], r
        if r['action'] in ('store_true', 'store_false'):
            assert a.nargs == 0 and a.const == (r['action'] == 'store_true') and a.default == (not a.const), r
        elif r['name'].startswith('-'):
            assert a.default == r['default'], r
            probe = {'int': ('7', 7), 'float': ('0.25', 0.25), 'str': ('abc', 'abc'), 'infer': ('True', True)}.get(r['type'])
            if probe is not None:
                assert a.type(probe[0]) == probe[1] if a.type is not None else probe[0] == probe[1], r
    if script == 'training':   # the parsed namespace is passed as keywords: every destination must be a training() parameter
        import inspect
        from synthsr_amd.training import training
        params = inspect.signature(training).parameters
        assert all(a.dest in params for a in parser._actions if a.dest != 'help')


def test_public_signatures_match_reference():
    """names, order and defaults of the reference's public callables (tests/golden/api.json, read from its source with
    ast): ours start with exactly those parameters; additional ones (device, rng, seed, ...) only after them"""
    import inspect
    import json
    from synthsr_amd import training as T, brain_generator as B, labels_to_image_model as L, model_inputs as M
    from synthsr_amd import unet as U, estimate_priors as E, volumes as V
    from synthsr_amd import fine_tuning_with_adversary as A
    here = {'SynthSR/training.py:training': T.training, 'SynthSR/fine_tuning_with_adversary.py:training': A.training,
            'SynthSR/fine_tuning_with_adversary.py:make_discriminator': A.make_discriminator, 'SynthSR/brain_generator.py:BrainGenerator.__init__': B.BrainGenerator.__init__,
            'SynthSR/labels_to_image_model.py:labels_to_image_model': L.labels_to_image_model,
            'SynthSR/model_inputs.py:build_model_inputs': M.build_model_inputs, 'ext/neuron/models.py:unet': U.unet,
            'SynthSR/estimate_priors.py:build_intensity_stats': E.build_intensity_stats,
            'SynthSR/estimate_priors.py:sample_intensity_stats_from_image': E.sample_intensity_stats_from_image,
            'SynthSR/estimate_priors.py:sample_intensity_stats_from_single_dataset': E.sample_intensity_stats_from_single_dataset,
            'ext/lab2im/utils.py:load_volume': V.load_volume, 'ext/lab2im/utils.py:save_volume': V.save_volume,
            'ext/lab2im/utils.py:get_volume_info': V.get_volume_info, 'ext/lab2im/utils.py:get_list_labels': V.get_list_labels,
            'ext/lab2im/edit_volumes.py:align_volume_to_ref': V.align_volume_to_ref,
            'ext/lab2im/edit_volumes.py:resample_volume': V.resample_volume,
            'ext/lab2im/edit_volumes.py:resample_volume_like': V.resample_volume_like,
            'ext/lab2im/edit_volumes.py:rescale_volume': V.rescale_volume}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, 'tests', 'golden', 'api.json')) as f:
        api = json.load(f)
    assert set(api) == set(here)
    problems = []
    for key, ref in api.items():
        params = [p for p in inspect.signature(here[key]).parameters.values() if p.name != 'self']
        if [p.name for p in params[:len(ref)]] != [r[0] for r in ref]:
            problems.append((key, 'names', [p.name for p in params[:len(ref)]], [r[0] for r in ref]))
            continue
        for p, (name, default) in zip(params, ref):
            if default == '<required>':
                ok = p.default is inspect.Parameter.empty
            elif default == '<expr>':
                ok = p.default is not inspect.Parameter.empty
            else:
                ok = p.default is not inspect.Parameter.empty and repr(p.default) == default
            if not ok:
                problems.append((key, name, repr(p.default), default))
        problems += [(key, 'extra parameter without default', p.name) for p in params[len(ref):]
                     if p.default is inspect.Parameter.empty]
    assert not problems, problems


def test_model_inputs_and_draws_match_reference_under_seeded_numpy():
    """H1: SynthSR/model_inputs.py:build_model_inputs and utils.draw_value_from_distribution (numpy branch) of the
    REFERENCE, run under a seeded global numpy state (tests/golden/gen/make_model_inputs_golden.py), against the product's
    sampler fed from the same global state (rng=None): identical call order => identical draws, bit for bit"""
    from conftest import load_golden
    from synthsr_amd import host_math as hm
    from synthsr_amd.model_inputs import build_model_inputs
    g = load_golden('model_inputs')
    pm, ps, pm2, ps2, classes = g['pm'], g['ps'], g['pm2'], g['ps2'], g['classes']
    cases = [('none_u', None, 4, 'uniform', 125., 100., True), ('num_n', 7.5, 3, 'normal', 10., 2., False),
             ('pair_u', [2., 9.], 5, 'uniform', 0., 10., False), ('pair_n', (3., .5), 5, 'normal', 0., 10., True),
             ('arr_n', pm, 1, 'normal', 125., 100., True), ('arr2_u', np.abs(pm2), 1, 'uniform', 125., 100., False),
             ('neg_n', np.stack([np.full(8, -1.), np.full(8, 3.)]), 1, 'normal', 0., 1., True)]
    for tag, hp, size, dist, centre, rg, pos in cases:
        np.random.seed(int(g['dv_' + tag + '_seed']))
        vals = np.stack([hm.draw_value_from_distribution(hp, size, dist, centre, rg, positive_only=pos) for _ in range(3)])
        np.testing.assert_array_equal(vals, g['dv_' + tag], err_msg=tag)
    assert hm.draw_value_from_distribution(False) is None
    labs = [g['lab%d' % i] for i in range(3)]
    ims = [g['im%d' % i] for i in range(3)]
    runs = [('a', pm, ps, 'normal', None, 1, 1, classes), ('b', pm2, ps2, 'normal', None, 1, 2, classes),
            ('c', None, None, 'uniform', None, 1, 1, None), ('d', [30., 150.], 12., 'uniform', ims, 1, 3, classes),
            ('e', pm, ps, 'uniform', ims, 2, 1, classes)]
    for tag, m, s, dist, images, bs, nch, cls in runs:
        np.random.seed(int(g['mi_%s_seed' % tag]))
        gen = build_model_inputs(None, 9, m, s, dist, path_images=images, batchsize=bs, n_channels=nch,
                                 generation_classes=cls, label_maps=labs)
        for it in range(3):
            items = next(gen)
            assert len(items) == (4 if images is not None else 3)
            for j, a in enumerate(items):
                ref = g['mi_%s_%d_%d' % (tag, it, j)]
                assert np.asarray(a).shape == ref.shape, (tag, it, j)
                np.testing.assert_array_equal(np.asarray(a), ref, err_msg='%s %d %d' % (tag, it, j))


def test_multi_modality_affine_bounds_pick_one_block_at_build_time():
    """(2n, m) bounds (ext/lab2im/utils.py:1011-1016): ONE block, drawn with np.random.randint like the reference's
    draw_value_from_distribution (whose numpy branch is pinned by model_inputs.npz `arr2_u`), then used as min / max rows"""
    b = np.arange(18, dtype=np.float64).reshape(6, 3)
    np.random.seed(11)
    want = 2 * np.random.randint(3)
    np.random.seed(11)
    got = hm.pick_bounds_block(b)
    assert got.shape == (2, 3) and np.array_equal(got, b[want:want + 2])
    lo, hi = hm.bounds_pair(got, 0., 3)
    assert np.array_equal(lo, b[want]) and np.array_equal(hi, b[want + 1])
    assert hm.pick_bounds_block(False) is False and hm.pick_bounds_block(.15) == .15
    np.random.seed(11)
    T = hm.sample_affine(dict(rot=np.full(3, .5, np.float32)), rotation_bounds=b)   # direct callers: picked on the spot
    lo, hi = b[want], b[want + 1]
    assert T.shape == (4, 4) and abs(np.linalg.det(T[:3, :3]) - 1) < 1e-5
    with pytest.raises(AssertionError):
        hm.pick_bounds_block(np.zeros((3, 3)))


def test_trace_ranges_are_noops_unless_enabled(monkeypatch):
    """ops.trace_range: nothing without SYNTHSR_ROCTX=1; with it, roctx ranges through the marker library if the image has one
    (no GPU needed: the calls only record markers for an attached profiler)"""
    from synthsr_amd import ops
    monkeypatch.setattr(ops, '_roctx', None)
    monkeypatch.delenv('SYNTHSR_ROCTX', raising=False)
    with ops.trace_range('x') as r:
        assert not r.on
    monkeypatch.setattr(ops, '_roctx', None)
    monkeypatch.setenv('SYNTHSR_ROCTX', '1')
    with ops.trace_range('step') as r:
        with ops.trace_range('nested'):
            pass
    assert r.on == bool(ops._roctx)
    monkeypatch.setattr(ops, '_roctx', None)


def test_bench_self_launch_rendezvous_cpu():
    """bench.py --gpus 2 launched plainly re-executes itself as two ranks (torch.distributed.run, 127.0.0.1); the rendezvous-only
    mode runs that path without device work: both ranks meet (gloo) and are counted.  Under a launcher whose world size differs
    from --gpus it refuses instead of measuring one GPU under a two-GPU command line (missing 1)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--rendezvous-only'], capture_output=True,
                       text=True, timeout=600, env=env, cwd=repo)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 2 and d['n_ranks_seen'] == 2 and d['world_size'] == 2
    # configs[2]: eight ranks; the record carries what one step's gradient all-reduce consists of
    r = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '8', '--rendezvous-only'], capture_output=True,
                       text=True, timeout=900, env=dict(env, OMP_NUM_THREADS='1'), cwd=repo)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 8 and d['n_ranks_seen'] == 8 and d['world_size'] == 8 and d['backend'] == 'gloo'
    assert d['rccl_version'] is None                                     # only filled when the collectives run over RCCL
    assert d['allreduce_bytes_per_step'] == 4 * 13240489                 # 52.96 MB: the flat gradient buffer, once
    assert d['allreduce_covers_buffer_once'] and d['allreduce_tail_first'] and d['allreduce_identical_on_every_rank']
    assert d['allreduce_buckets_per_step'] >= 2 and abs(d['gradient_scale'] - 0.125) < 1e-12
    r = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--rendezvous-only'], capture_output=True,
                       text=True, timeout=600, env=dict(env, WORLD_SIZE='1', RANK='0'), cwd=repo)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stdout + r.stderr)


def test_public_headers_are_plain_c(tmp_path):
    """the drop-in boundary is a C ABI: include/*.h compile as C99 (no C++, no torch / HIP types in any signature) and a caller can
    fill the conv context as a plain aggregate"""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    src = tmp_path / 'hdr.c'
    src.write_text('#include "synthsr_hip.h"\n#include "synthsr_hip_tuning.h"\n'
                   'int main(void) { synthsr_conv_ctx c = { SYNTHSR_ARITH_SPLIT9, 0, 0, 0, {0, 0} }; synthsr_stream_t s = 0; (void)s;\n'
                   '  return c.arithmetic == 2 && c.workspace == 0 && sizeof(c) == 32 && SYNTHSR_EWORKSPACE == -3 ? 0 : 1; }\n')
    r = subprocess.run([gcc, '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I', os.path.join(REPO, 'include'), str(src), '-o',
                        str(tmp_path / 'hdr')], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert subprocess.run([str(tmp_path / 'hdr')]).returncode == 0
