"""Inference path (reference scripts/predict_command_line.py): host pre/post-processing against
goldens produced by the reference's own functions, and the full predict() through the HIP U-Net against the oracle."""
import os
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden', 'inference.npz')


@pytest.mark.parametrize('case', ['down', 'up', 'mixed'])
def test_resample_volume_matches_reference(case):
    from synthsr_amd import volumes as V
    z = np.load(GOLD)
    v2, a2 = V.resample_volume(z[case + '_vol'], z[case + '_aff'], [1.0, 1.0, 1.0])
    assert v2.shape == z[case + '_res_vol'].shape
    np.testing.assert_allclose(v2, z[case + '_res_vol'], rtol=0, atol=1e-10)  # float64 re-association only
    np.testing.assert_allclose(a2, z[case + '_res_aff'], rtol=0, atol=1e-12)
    v3, a3 = V.align_volume_to_ref(v2, a2, aff_ref=np.eye(4), return_aff=True, n_dims=3)
    np.testing.assert_allclose(v3, z[case + '_ras_vol'], rtol=0, atol=1e-10)
    np.testing.assert_allclose(a3, z[case + '_ras_aff'], rtol=0, atol=1e-12)


def test_resample_volume_like_matches_reference():
    from synthsr_amd import volumes as V
    z = np.load(GOLD)
    out = V.resample_volume_like(z['mixed_ras_vol'], z['mixed_ras_aff'], z['like_flo'], z['like_flo_aff'])
    assert 0.2 < (z['like_out'] == 0).mean() < 0.8  # the golden covers inside and outside (fill value 0) of the grid
    np.testing.assert_allclose(out, z['like_out'], rtol=0, atol=1e-10)
    near = V.resample_volume_like(z['mixed_ras_vol'], z['mixed_ras_aff'], z['like_flo'], z['like_flo_aff'], 'nearest')
    assert set(np.unique(near)) <= set(np.unique(z['like_flo'])) | {0.0}


def test_resample_volume_nearest_and_errors():
    from synthsr_amd import volumes as V
    vol = np.arange(4 * 3 * 2, dtype=float).reshape(4, 3, 2)
    aff = np.diag([2.0, 1.0, 1.0, 1.0])
    out, aff2 = V.resample_volume(vol, aff, [1.0, 1.0, 1.0], interpolation='nearest', blur=False)
    assert out.shape == (8, 3, 2)
    assert set(np.unique(out)) <= set(np.unique(vol))  # nearest never invents values
    np.testing.assert_allclose(np.sqrt((aff2[:3, :3] ** 2).sum(0)), [1.0, 1.0, 1.0])
    # world position of the first voxel's corner is preserved: centre moves by -0.5*(f-1) new voxels
    np.testing.assert_allclose(aff2[:3, 3], [-0.5, 0.0, 0.0])
    with pytest.raises(ValueError):
        V.resample_volume(vol, aff, [1.0, 1.0, 1.0], interpolation='cubic')


def test_prepare_and_postprocess():
    from synthsr_amd.predict import prepare_volume, postprocess
    rng = np.random.RandomState(0)
    im = rng.rand(20, 35, 33) * 90 - 10
    S, idx, shape, aff2 = prepare_volume(im, np.eye(4), ct=True)
    assert S.shape == (32, 64, 64) and list(shape) == [20, 35, 33] and list(idx) == [6, 14, 15]
    inner = S[idx[0]:idx[0] + 20, idx[1]:idx[1] + 35, idx[2]:idx[2] + 33]
    assert inner.min() == 0.0 and inner.max() == 1.0 and S.sum() == inner.sum()
    from scipy.ndimage import gaussian_filter
    ct = gaussian_filter(np.clip(im, 0, 80), 0.25)  # the reference pre-blurs with sigma .25/zoom unless up-sampling (zoom 1 too)
    np.testing.assert_allclose(inner, (ct - ct.min()) / (ct - ct.min()).max(), atol=1e-12)
    out = rng.randn(1, 32, 64, 64, 1)
    pred = postprocess(out, idx, shape)
    assert pred.shape == (20, 35, 33) and pred.min() >= 0 and pred.max() <= 128
    ref = np.clip(255 * out[0, ..., 0], 0, 128)[6:26, 14:49, 15:48]
    np.testing.assert_allclose(pred, ref)


def test_predict_argument_errors(tmp_path):
    from synthsr_amd.predict import predict, Predictor
    bad = tmp_path / 'scan.txt'
    bad.write_text('x')
    with pytest.raises(Exception, match='extension not supported'):
        predict(str(bad), str(tmp_path / 'out.nii.gz'), predictor=object())
    with pytest.raises(AssertionError):
        predict(str(tmp_path / 'missing.nii.gz'), str(tmp_path / 'out.nii.gz'), predictor=object())
    with pytest.raises(FileNotFoundError, match='model file'):
        Predictor(str(tmp_path / 'nope.npz'))


@pytest.mark.gpu
@pytest.mark.parametrize('flip', [True, False])
def test_predict_end_to_end_vs_oracle(tmp_path, flip):
    """synthetic anisotropic NIfTI -> predict() with a random-weight checkpoint; the U-Net output must equal the oracle's
    inference-mode forward (moving statistics) on the same prepared volume: 2e-4 of the output range (fp32 convs)"""
    import torch
    from synthsr_amd import volumes as V
    from synthsr_amd.predict import predict, prepare_volume, postprocess
    from synthsr_amd.training import save_checkpoint
    from synthsr_amd.unet import unet
    from oracle import unet_ref as U
    rng = np.random.RandomState(3)
    vol = rng.rand(30, 22, 26) * 1000
    aff = np.array([[-1.2, 0, 0, 40.], [0, 1.0, 0, -30.], [0, 0, 1.4, 5.], [0, 0, 0, 1.]])
    pin = str(tmp_path / 'scan.nii.gz')
    V.save_volume(vol, aff, None, pin)
    # a random network with non-trivial moving statistics, saved in the checkpoint format
    net = unet(24, [32, 32, 32, 1], 5, 3, 1, feat_mult=2, nb_conv_per_level=2, batch_norm=-1, activation='elu',
               final_pred_activation='linear', seed=1)
    g = torch.Generator().manual_seed(0)
    net.bn_moving.copy_((torch.rand(net.bn_moving.shape, generator=g) * 0.5 + 0.25).to(net.bn_moving.device))
    ck = str(tmp_path / 'model.npz')
    save_checkpoint(ck, net)
    pout = str(tmp_path / 'sub' / 'pred.nii.gz')
    predict(pin, pout, path_model=ck, disable_flipping=not flip, verbose=False)
    got, aff_out, _ = V.load_volume(pout, im_only=False)
    # oracle on the same prepared input
    im, aff_in, _ = V.load_volume(pin, im_only=False, dtype='float')
    S, idx, shape, aff2 = prepare_volume(im, aff_in)
    assert S.shape == (64, 32, 64)
    sd = net.state_dict()
    P = {k: v.float() for k, v in sd.items()}
    x = torch.from_numpy(S.astype(np.float32))[..., None]
    with torch.no_grad():
        out = U.unet_forward(x, P, net.prefix, 5, 2, training=False, moving=P)[..., 0]
        if flip:
            outf = U.unet_forward(torch.flip(x, dims=[0]), P, net.prefix, 5, 2, training=False, moving=P)[..., 0]
            out = 0.5 * out + 0.5 * torch.flip(outf, dims=[0])
    ref = postprocess(out.numpy(), idx, shape)
    assert got.shape == ref.shape == tuple(shape)
    np.testing.assert_allclose(aff_out, aff2, atol=1e-5)
    scale = max(np.abs(255 * out.numpy()).max(), 1.0)
    assert np.abs(got - ref).max() / scale < 2e-4
    assert ref.std() > 0  # not a clipped-flat image


@pytest.mark.gpu
def test_predict_hyperfine_end_to_end_vs_oracle(tmp_path):
    """two-input (T1, T2) variant, scripts/predict_command_line_hyperfine.py: residual prediction on the rescaled T1"""
    import torch
    from synthsr_amd import volumes as V
    from synthsr_amd.predict import predict_hyperfine, prepare_hyperfine, postprocess_hyperfine
    from synthsr_amd.training import save_checkpoint
    from synthsr_amd.unet import unet
    from oracle import unet_ref as U
    rng = np.random.RandomState(5)
    aff1 = np.array([[1.5, 0, 0, -20.], [0, 1.5, 0, -18.], [0, 0, 5.0, -30.], [0, 0, 0, 1.]])
    aff2 = aff1 @ np.array([[1, 0.02, 0, 0.5], [-0.02, 1, 0, -0.4], [0, 0, 1, 0.3], [0, 0, 0, 1.]])
    p1, p2 = str(tmp_path / 't1.nii.gz'), str(tmp_path / 't2.nii.gz')
    V.save_volume(rng.rand(24, 22, 8) * 400 + 30, aff1, None, p1)
    V.save_volume(rng.rand(24, 22, 8) * 900, aff2, None, p2)
    net = unet(24, [32, 32, 32, 2], 5, 3, 1, feat_mult=2, nb_conv_per_level=2, batch_norm=-1, activation='elu',
               final_pred_activation='linear', seed=2)
    g = torch.Generator().manual_seed(1)
    net.bn_moving.copy_((torch.rand(net.bn_moving.shape, generator=g) * 0.5 + 0.25).to(net.bn_moving.device))
    ck = str(tmp_path / 'model_hf.npz')
    save_checkpoint(ck, net)
    pout = str(tmp_path / 'out.nii.gz')
    predict_hyperfine(p1, p2, pout, path_model=ck, verbose=False)
    got, aff_out, _ = V.load_volume(pout, im_only=False)
    im1, a1, _ = V.load_volume(p1, im_only=False, dtype='float')
    im2, a2, _ = V.load_volume(p2, im_only=False, dtype='float')
    S, idx, shape, aff_mod, scaling, t1n = prepare_hyperfine(im1, a1, im2, a2)
    assert S.shape[:3] == (64, 64, 64) and S.shape[3] == 2
    P = {k: v.float() for k, v in net.state_dict().items()}
    with torch.no_grad():
        out = U.unet_forward(torch.from_numpy(S.astype(np.float32)), P, net.prefix, 5, 2, training=False, moving=P)[..., 0]
    ref = postprocess_hyperfine(out.numpy(), idx, shape, scaling, t1n)
    assert got.shape == ref.shape
    np.testing.assert_allclose(aff_out, aff_mod, atol=1e-5)
    assert np.abs(got - ref).max() / max(np.abs(ref).max(), 1.0) < 2e-4


def test_mgz_volumes_layout_and_round_trip(tmp_path):
    """synthsr_amd/mgh.py (PARITY UNPINNED: no nibabel / no .mgz sample in the image): the byte layout follows the
    published MGH header offsets, write -> read is exact for every supported type, and the affine agrees with NIfTI's"""
    import gzip
    import struct
    from synthsr_amd import mgh, volumes as V
    from synthsr_amd.nifti import write_nifti, read_nifti
    rng = np.random.RandomState(3)
    vol = rng.uniform(0, 200, (5, 7, 6)).astype(np.float32)
    aff = np.array([[0., -1.5, 0., 12.], [0., 0., 2.5, -30.], [-1., 0., 0., 7.], [0., 0., 0., 1.]])
    p = str(tmp_path / 'a.mgz')
    V.save_volume(vol, aff, None, p)
    raw = gzip.open(p, 'rb').read()
    assert struct.unpack('>7i', raw[:28]) == (1, 5, 7, 6, 1, 3, 0) and struct.unpack('>h', raw[28:30]) == (1,)
    assert np.allclose(struct.unpack('>3f', raw[30:42]), [1.0, 1.5, 2.5])
    assert np.allclose(struct.unpack('>9f', raw[42:78]), [0, 0, -1, -1, 0, 0, 0, 1, 0])       # x_ras, y_ras, z_ras
    centre = aff[:3, :3] @ (np.array([5, 7, 6]) / 2) + aff[:3, 3]
    assert np.allclose(struct.unpack('>3f', raw[78:90]), centre, atol=1e-5)
    assert len(raw) == 284 + vol.size * 4
    assert struct.unpack('>f', raw[284 + 4:284 + 8])[0] == vol[1, 0, 0]                          # x runs fastest
    back, aff2, hdr = V.load_volume(p, im_only=False)
    assert back.dtype == np.float64 and np.array_equal(back, vol) and np.allclose(aff2, aff, atol=1e-5)
    assert np.allclose(hdr['pixdim'][1:4], [1.0, 1.5, 2.5])
    # voxel size of a .mgz from the header's delta (ext/lab2im/utils.py:185), in the axis order of the reference frame
    _, _, _, _, _, res = V.get_volume_info(p, max_channels=3)           # (5, 7, 6): the last axis is spatial
    assert np.allclose(res, [1.0, 1.5, 2.5])
    _, _, _, _, _, res_ras = V.get_volume_info(p, aff_ref=np.eye(4), max_channels=3)
    assert np.allclose(res_ras, [1.5, 2.5, 1.0])
    # same volume through NIfTI: identical data and affine, so predict/training treat both alike
    write_nifti(str(tmp_path / 'a.nii.gz'), vol, aff)
    d_n, a_n, _ = read_nifti(str(tmp_path / 'a.nii.gz'))
    assert np.array_equal(d_n, back) and np.allclose(a_n, aff2, atol=1e-5)
    for dt in (np.uint8, np.int16, np.int32, np.float32):
        x = rng.randint(0, 120, (4, 3, 5, 2)).astype(dt)
        q = str(tmp_path / ('t_%s.mgh' % np.dtype(dt).name))
        mgh.write_mgh(q, x, aff)
        y, a, h = mgh.read_mgh(q)
        assert y.dtype == np.dtype(dt) and np.array_equal(x, y) and h['dims'] == (4, 3, 5, 2)
    mgh.write_mgh(str(tmp_path / 'f64.mgz'), vol.astype(np.float64), None)
    y, a, _ = mgh.read_mgh(str(tmp_path / 'f64.mgz'))
    assert y.dtype == np.float32 and np.array_equal(a[:3, :3], np.eye(3)) and np.allclose(a[:3, 3], 0)
    # goodRASFlag = 0 -> FreeSurfer's default coronal orientation
    bad = bytearray(raw)
    bad[28:30] = struct.pack('>h', 0)
    (tmp_path / 'b.mgh').write_bytes(bytes(bad))
    _, a0, _ = mgh.read_mgh(str(tmp_path / 'b.mgh'))
    assert np.array_equal(a0[:3, :3], [[-1, 0, 0], [0, 0, 1], [0, -1, 0]])
    (tmp_path / 'c.mgh').write_bytes(b'\0' * 300)
    with pytest.raises(ValueError):
        mgh.read_mgh(str(tmp_path / 'c.mgh'))
