"""synthsr_amd/keras_h5.py (library-free HDF5 reader/writer for Keras weight files) against fixtures written by the
real HDF5 library (tests/golden/gen/make_keras_h5.py, h5py 3.3.0 / HDF5 1.10.6) - SURVEY §8f row 1, the `.h5` side of
`load_weights(..., by_name=True)` (SynthSR/training.py:363, scripts/predict_command_line.py:79-82)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from synthsr_amd import keras_h5

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H5PY_PYTHON = '/opt/conda/bin/python3.9'   # an interpreter with the real library, when the image has one


def expected():
    z = np.load(os.path.join(GOLD, 'keras_h5_expected.npz'))
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize('name,n', [('keras_weights_tiny.h5', 42), ('keras_model_tiny.h5', 42),
                                    ('keras_weights_variants.h5', 42), ('keras_weights_latest.h5', 8)])
def test_reader_matches_real_library_files(name, n):
    """save_weights layout, model.save layout (/model_weights), chunked/gzip/shuffle/fletcher32 storage, and the
    libver='latest' object headers: every array bit-exact"""
    exp = expected()
    sd = keras_h5.load_keras_weights(os.path.join(GOLD, name))
    assert len(sd) == n
    for k, v in sd.items():
        assert v.dtype == np.float32 and v.shape == exp[k].shape
        assert np.array_equal(v, exp[k]), k
    if n == 42:
        assert list(sd)[:2] == ['unet_conv_downarm_0_0/kernel', 'unet_conv_downarm_0_0/bias']   # model order kept
        assert sd['unet_likelihood/kernel'].shape == (1, 1, 1, 4, 1)


def test_reader_format_coverage():
    f = keras_h5.H5File(os.path.join(GOLD, 'keras_weights_variants.h5'))
    a = f.root.attrs
    assert a['keras_version'] == b'2.4.0'                       # variable-length string in the global heap
    assert a['a_float'] == 2.5 and np.array_equal(a['ints'], np.arange(6).reshape(2, 3))
    names = keras_h5._string_list(a, 'long_list')               # Keras' chunked long_list0, long_list1
    assert 'long_list' not in a and len(names) == 400 and names[399].startswith('layer_with_a_long_name_0399_')
    x = f['extras']
    assert np.array_equal(x['f64'].read(), np.linspace(0, 1, 7))
    assert np.array_equal(x['be_f32'].read(), np.arange(5, dtype=np.float32)) and x['be_f32'].dtype == np.dtype('>f4')
    assert np.array_equal(x['i16'].read(), np.arange(-3, 3).reshape(2, 3)) and x['i16'].dtype == np.int16
    assert np.array_equal(x['u8'].read(), np.arange(250, 256))
    assert x['scalar'].shape == () and float(x['scalar'].read()) == 3.25
    assert np.array_equal(x['never_written'].read(), np.zeros(4, np.float32))
    assert list(x['strings'].read()) == [b'ab', b'cde', b'']
    many = x['many']                                            # several symbol-table nodes, two-level B-tree
    assert many.keys() == ['d%04d' % j for j in range(300)]
    assert all(float(many['d%04d' % j].read()) == j for j in range(0, 300, 7))
    assert 'nope' not in x and 'extras/many/d0001' in f.root
    with pytest.raises(KeyError):
        f['extras/nope']
    m = keras_h5.H5File(os.path.join(GOLD, 'keras_model_tiny.h5'))
    assert m.root.attrs['model_config'].startswith(b'{"class_name": "Model"')
    ow = m['optimizer_weights']
    assert keras_h5._string_list(ow.attrs, 'weight_names')[0] == 'training/Adam/iterations:0'
    assert int(ow['training/Adam/iterations:0'].read()) == 1234


def test_not_hdf5_and_not_keras(tmp_path):
    p = tmp_path / 'x.h5'
    p.write_bytes(b'not an hdf5 file' * 100)
    with pytest.raises(keras_h5.H5FormatError):
        keras_h5.load_keras_weights(str(p))
    keras_h5.save_keras_weights(str(p), {'a/kernel': np.zeros((2, 2), np.float32)})
    f = keras_h5.H5File(str(p))
    assert f.root.keys() == ['a']
    with pytest.raises(keras_h5.H5FormatError):                 # a group without weight_names/layer_names
        keras_h5._string_list(f['a/a'].attrs, 'layer_names')


def test_writer_round_trip(tmp_path):
    exp = expected()
    path = str(tmp_path / 'w.h5')
    # BN weights handed over in this package's order (beta, gamma): the file must hold Keras' order
    shuffled = {}
    for k in exp:
        if k.endswith('/gamma'):
            shuffled[k[:-5] + 'beta'] = exp[k[:-5] + 'beta']
        if k not in shuffled:
            shuffled[k] = exp[k]
    shuffled['optimizer/m'] = np.zeros(3, np.float32)            # never written
    keras_h5.save_keras_weights(path, shuffled)
    back = keras_h5.load_keras_weights(path)
    assert set(back) == set(exp) and all(np.array_equal(back[k], exp[k]) for k in exp)
    f = keras_h5.H5File(path)
    assert keras_h5._string_list(f['unet_bn_up_1'].attrs, 'weight_names') == [
        'unet_bn_up_1/' + w + ':0' for w in ('gamma', 'beta', 'moving_mean', 'moving_variance')]
    assert f.root.attrs['backend'] == b'tensorflow'
    # a network with more layers than one default symbol-table node holds, and a long name list
    big = {'layer_%03d_%s/kernel' % (i, 'n' * 150): np.full((2, 3), i, np.float32) for i in range(500)}
    keras_h5.save_keras_weights(path, big)
    f = keras_h5.H5File(path)
    assert 'layer_names' not in f.root.attrs and 'layer_names1' in f.root.attrs
    back = keras_h5.load_keras_weights(path)
    assert list(back) == list(big) and all(np.array_equal(back[k], big[k]) for k in big)


@pytest.mark.skipif(not os.path.exists(H5PY_PYTHON), reason='no interpreter with the real HDF5 library in this image')
def test_written_file_is_read_by_the_real_library(tmp_path):
    """what Keras' load_weights_from_hdf5_group_by_name does with the file, through genuine h5py"""
    exp = expected()
    path = str(tmp_path / 'w.h5')
    keras_h5.save_keras_weights(path, exp)
    np.savez(str(tmp_path / 'e.npz'), **exp)
    code = ("import h5py, numpy as np, sys\n"
            "f = h5py.File(sys.argv[1], 'r'); exp = np.load(sys.argv[2]); n = 0\n"
            "assert f.attrs['keras_version'].decode('utf8') == '2.3.1'\n"
            "for ln in [x.decode('utf8') for x in f.attrs['layer_names']]:\n"
            "    g = f[ln]\n"
            "    for wn in [x.decode('utf8') for x in g.attrs['weight_names']]:\n"
            "        assert np.array_equal(np.asarray(g[wn]), exp[wn[:-2]]), wn; n += 1\n"
            "print('OK', n)\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith('PYTHON')}
    r = subprocess.run([H5PY_PYTHON, '-c', code, path, str(tmp_path / 'e.npz')], capture_output=True, text=True, env=env)
    if 'No module named' in r.stderr:
        pytest.skip('h5py not importable there')
    assert r.returncode == 0 and 'OK 42' in r.stdout, r.stderr[-2000:]


def test_convert_cli(tmp_path):
    npz, h5 = str(tmp_path / 'a.npz'), str(tmp_path / 'b.h5')
    script = os.path.join(ROOT, 'scripts', 'convert_keras_h5.py')
    subprocess.run([sys.executable, script, os.path.join(GOLD, 'keras_model_tiny.h5'), npz], check=True)
    subprocess.run([sys.executable, script, npz, h5], check=True)
    exp, back = expected(), keras_h5.load_keras_weights(h5)
    z = np.load(npz)
    assert set(z.files) == set(exp) == set(back)
    assert all(np.array_equal(z[k], exp[k]) and np.array_equal(back[k], exp[k]) for k in exp)


@pytest.mark.gpu
def test_unet_loads_and_saves_keras_h5(tmp_path):
    """load_weights(by_name=True) semantics on the device network: every parameter and BN moving statistic of a U-Net
    with the fixture's architecture comes from the .h5 (strict), the forward pass equals the one from the same weights
    handed over as arrays, and save_checkpoint('.h5') writes them back bit-exact"""
    import torch
    from synthsr_amd.unet import unet
    from synthsr_amd.training import load_checkpoint, save_checkpoint
    exp = expected()

    def make(seed):
        return unet(nb_features=4, input_shape=[16, 16, 16, 1], nb_levels=3, conv_size=3, nb_labels=1, feat_mult=2,
                    nb_conv_per_level=2, batch_norm=-1, activation='elu', device='cuda', seed=seed,
                    final_pred_activation='linear')
    net = make(1)
    load_checkpoint(os.path.join(GOLD, 'keras_model_tiny.h5'), net)
    sd = net.state_dict()
    assert set(sd) == set(exp)
    for k in exp:
        assert np.array_equal(sd[k].numpy().reshape(exp[k].shape), exp[k]), k
    ref = make(2)
    # moving_variance of the fixture is random (also negative): keep it positive for a finite inference pass
    pos = {k: (np.abs(v) + 0.5 if k.endswith('moving_variance') else v) for k, v in exp.items()}
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in pos.items()})
    path = str(tmp_path / '001.h5')
    save_checkpoint(path, ref)
    back = keras_h5.load_keras_weights(path)
    assert back['unet_likelihood/kernel'].shape == (1, 1, 1, 4, 1)
    assert set(back) == set(pos) and all(np.array_equal(back[k], pos[k]) for k in pos)
    load_checkpoint(path, net)
    x = torch.rand(16, 16, 16, 1, device='cuda')
    assert torch.equal(net.predict(x), ref.predict(x))
    with pytest.raises(ValueError):                              # a file of another network (layer prefix)
        other = unet(nb_features=4, input_shape=[16, 16, 16, 1], nb_levels=3, conv_size=3, nb_labels=1, feat_mult=2,
                     nb_conv_per_level=2, batch_norm=-1, activation='elu', device='cuda', name='segunet',
                     final_pred_activation='linear')
        load_checkpoint(path, other)


def test_adam_slots_of_a_full_model_file():
    """load_keras_optimizer on a `model.save()`-layout file written by the real HDF5 library with the slot list Keras 2.3.1's
    Adam serialises ([iterations] + ms + vs + vhats, tests/golden/gen/make_keras_h5.py): slots come back by position, and
    training.keras_trainable_order maps them onto this build's parameter table (BatchNormalization: gamma before beta in
    Keras, beta before gamma here).  Reference: SynthSR/training.py:429-439 (ModelCheckpoint + models.load_model)."""
    from synthsr_amd.training import keras_trainable_order
    from synthsr_amd.unet import UNet3D
    it, ms, vs = keras_h5.load_keras_optimizer(os.path.join(GOLD, 'keras_model_tiny_opt.h5'))
    exp = np.load(os.path.join(GOLD, 'keras_h5_opt_expected.npz'))
    assert it == 4321 == int(exp['iterations'])
    net = UNet3D(4, [16, 16, 16, 1], 3, 3, 1, feat_mult=2, nb_conv_per_level=2, batch_norm=-1, table_only=True)
    order = keras_trainable_order(net)
    assert len(order) == len(ms) == len(vs) == len(net.specs)
    assert order[4:6] == ['unet_bn_down_0/gamma', 'unet_bn_down_0/beta']
    shapes = {nm: shp for nm, shp, _ in net.specs}
    for nm, m, v in zip(order, ms, vs):
        want = (1, 1, 1) + tuple(shapes[nm]) if nm == 'unet_likelihood/kernel' else tuple(shapes[nm])
        assert tuple(m.shape) == want == tuple(v.shape), nm
        assert np.array_equal(m, exp['m/' + nm]) and np.array_equal(v, exp['v/' + nm]), nm
    assert keras_h5.load_keras_optimizer(os.path.join(GOLD, 'keras_weights_tiny.h5')) is None   # save_weights(): no slots


def test_adam_slot_layout_is_decided_from_the_data():
    """[ms, vs] vs [ms, vs, vhats] by shapes / names, not by divisibility of the count (9 weights without vhats = 18 arrays,
    which is also 3 x 6)"""
    f = keras_h5._adam_slot_count
    w9 = [np.zeros((3, 3, 3, 2, 4)), np.zeros(4)] * 4 + [np.zeros((1, 1, 1, 4, 1))]
    assert f(['a'] * 18, w9 + w9) == 9                                         # two slots, count divisible by 3
    w6 = w9[:6]
    assert f(['a'] * 18, w6 + w6 + [np.zeros(1)] * 6) == 6                     # amsgrad=False placeholders
    names = ['m'] * 6 + ['v'] * 6 + ['Adam/vhat_%d' % i for i in range(6)]
    assert f(names, w6 + w6 + w6) == 6                                         # amsgrad=True, named
    assert f(['a'] * 18, w6 + w6 + w6) == 6                                    # ... unnamed: three alike thirds
    assert f(['a'] * 4, [np.zeros(3), np.zeros(2), np.zeros(3), np.zeros(2)]) == 2
    assert f(['a'] * 4, [np.zeros(3), np.zeros(2), np.zeros(2), np.zeros(3)]) is None
    assert f([], []) is None


@pytest.mark.gpu
def test_resume_from_a_full_model_h5_restores_the_adam_state():
    """training(checkpoint='NNN.h5') on a file the reference's ModelCheckpoint wrote (full model): weights by layer name AND the
    Adam moments / iteration count, like the reference's resume path `models.load_model` (SynthSR/training.py:434-439:
    "momentum is comprised in checkpoints")"""
    import torch
    from synthsr_amd.training import load_checkpoint
    from synthsr_amd.unet import unet
    net = unet(4, [16, 16, 16, 1], 3, 3, 1, feat_mult=2, nb_conv_per_level=2, final_pred_activation='linear', batch_norm=-1)
    load_checkpoint(os.path.join(GOLD, 'keras_model_tiny_opt.h5'), net)
    exp = np.load(os.path.join(GOLD, 'keras_h5_opt_expected.npz'))
    w = np.load(os.path.join(GOLD, 'keras_h5_expected.npz'))
    assert net.iterations == 4321
    for nm, shp, _ in net.specs:
        assert np.array_equal(net.view(nm, net.adam_m).cpu().numpy(), exp['m/' + nm].reshape(shp)), nm
        assert np.array_equal(net.view(nm, net.adam_v).cpu().numpy(), exp['v/' + nm].reshape(shp)), nm
        assert np.array_equal(net.view(nm).cpu().numpy(), w[nm].reshape(shp)), nm
    net.grads.fill_(1e-3)
    net.adam_step(1e-4)
    assert net.iterations == 4322 and torch.isfinite(net.params).all()
