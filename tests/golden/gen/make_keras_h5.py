"""Writes the Keras-layout HDF5 fixtures of tests/test_keras_h5.py with the REAL HDF5 library (h5py), so that the
library-free reader in synthsr_amd/keras_h5.py is pinned against files it did not write itself.

Run with an interpreter that has h5py (this image: /opt/conda/bin/python3.9, h5py 3.3.0 / HDF5 1.10.6):

    /opt/conda/bin/python3.9 tests/golden/gen/make_keras_h5.py

The file layout restates what Keras 2.x writes (keras/engine/saving.py, `save_weights_to_hdf5_group` /
`save_attributes_to_hdf5_group` and `_serialize_model`; Keras is a third-party dependency of the reference, pinned to
keras 2.3.1 in /root/reference/requirements.txt, and is not vendored there):

  save_weights():  /            attrs layer_names (fixed-length byte strings, one per layer INCLUDING weight-less ones,
                                chunked into layer_names0, layer_names1.. beyond 64512 bytes), backend, keras_version
                   /<layer>     attrs weight_names;  datasets /<layer>/<weight name>, e.g.
                                /unet_conv_downarm_0_0/unet_conv_downarm_0_0/kernel:0   (contiguous float32)
  model.save():    the same tree under /model_weights, plus /optimizer_weights and the attrs model_config,
                   training_config (what KC.ModelCheckpoint(save_file_name) writes, SynthSR/training.py:430)

The layer list is the 3-D U-Net of ext/neuron/models.py:304-494 as instantiated by SynthSR/training.py:330-341, at a
reduced size (3 levels, 4 features) so that the fixtures stay small; values are seeded random numbers which are also
stored in keras_h5_expected.npz."""
import json
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '..')
HDF5_OBJECT_HEADER_LIMIT = 64512


def unet_layers(nb_features=4, nb_levels=3, nb_conv_per_level=2, feat_mult=2, cin=1, nb_labels=1, prefix='unet'):
    """[(layer name, [(weight name, shape)])] in model order, weight-less layers included"""
    L = nb_levels
    feats = [int(np.round(nb_features * feat_mult ** l)) for l in range(L)]
    layers = [('%s_input' % prefix, [])]
    c = cin

    def conv(name, ci, co, k=3):
        return (name, [(name + '/kernel:0', (k, k, k, ci, co)), (name + '/bias:0', (co,))])

    def bn(name, C):
        return (name, [(name + '/' + w + ':0', (C,)) for w in ('gamma', 'beta', 'moving_mean', 'moving_variance')])

    for l in range(L):
        for k in range(nb_conv_per_level):
            layers.append(conv('%s_conv_downarm_%d_%d' % (prefix, l, k), c, feats[l]))
            c = feats[l]
        layers.append(bn('%s_bn_down_%d' % (prefix, l), c))
        if l < L - 1:
            layers.append(('%s_maxpool_%d' % (prefix, l), []))
    for k in range(L - 1):
        l = L - 2 - k
        layers.append(('%s_up_%d' % (prefix, L + k), []))
        layers.append(('%s_merge_%d' % (prefix, L + k), []))
        ci = feats[l] + c
        for j in range(nb_conv_per_level):
            layers.append(conv('%s_conv_uparm_%d_%d' % (prefix, L + k, j), ci, feats[l]))
            ci = feats[l]
        c = feats[l]
        layers.append(bn('%s_bn_up_%d' % (prefix, k), c))
    layers.append(conv('%s_likelihood' % prefix, c, nb_labels, k=1))
    layers.append(('%s_prediction' % prefix, []))
    return layers


def save_attributes(group, name, data):
    """Keras' chunking of long string-list attributes"""
    data = [d.encode('utf8') for d in data]
    bad = [x for x in data if len(x) > HDF5_OBJECT_HEADER_LIMIT]
    assert not bad
    arr = np.asarray(data)
    n = 1
    chunks = np.array_split(arr, n)
    while any(x.nbytes > HDF5_OBJECT_HEADER_LIMIT for x in chunks):
        n += 1
        chunks = np.array_split(arr, n)
    if n > 1:
        for i, ch in enumerate(chunks):
            group.attrs['%s%d' % (name, i)] = ch
    else:
        group.attrs[name] = data


def save_weights_group(f, layers, values, **dset_kw):
    save_attributes(f, 'layer_names', [ln for ln, _ in layers])
    f.attrs['backend'] = 'tensorflow'.encode('utf8')
    f.attrs['keras_version'] = '2.3.1'.encode('utf8')
    for ln, ws in layers:
        g = f.create_group(ln)
        save_attributes(g, 'weight_names', [wn for wn, _ in ws])
        for wn, shp in ws:
            val = values[wn]
            d = g.create_dataset(wn, val.shape, dtype=val.dtype, **dset_kw)
            if not val.shape:
                d[()] = val
            else:
                d[:] = val


def main():
    rng = np.random.RandomState(20210712)
    layers = unet_layers()
    values = {}
    for ln, ws in layers:
        for wn, shp in ws:
            values[wn] = rng.standard_normal(shp).astype(np.float32)
    np.savez(os.path.join(OUT, 'keras_h5_expected.npz'), **{k[:-2]: v for k, v in values.items()})

    # 1) model.save_weights(path)
    with h5py.File(os.path.join(OUT, 'keras_weights_tiny.h5'), 'w') as f:
        save_weights_group(f, layers, values)

    # 2) model.save(path) / ModelCheckpoint: weights under /model_weights, + config attrs + optimizer slots
    with h5py.File(os.path.join(OUT, 'keras_model_tiny.h5'), 'w') as f:
        f.attrs['keras_version'] = '2.3.1'.encode('utf8')
        f.attrs['backend'] = 'tensorflow'.encode('utf8')
        f.attrs['model_config'] = json.dumps({'class_name': 'Model', 'config': {'name': 'unet', 'layers': [
            {'name': ln, 'class_name': 'Layer'} for ln, _ in layers]}}).encode('utf8')
        f.attrs['training_config'] = json.dumps({'optimizer_config': {'class_name': 'Adam', 'config': {'lr': 1e-4}},
                                                 'loss': 'l1'}).encode('utf8')
        save_weights_group(f.create_group('model_weights'), layers, values)
        og = f.create_group('optimizer_weights')
        names = ['training/Adam/iterations:0'] + ['training/Adam/m_%d:0' % i for i in range(3)]
        save_attributes(og, 'weight_names', names)
        og.create_dataset(names[0], (), dtype='int64')[()] = 1234
        for n_ in names[1:]:
            og.create_dataset(n_, (5,), dtype='float32')[:] = 0.5

    # 2b) a full-model file whose /optimizer_weights holds what Keras 2.3.1's Adam serialises (keras/optimizers.py, Adam.weights
    #     = [iterations] + ms + vs + vhats; one m / v per entry of model.trainable_weights -- Conv3D: kernel, bias;
    #     BatchNormalization: gamma, beta -- and a (1,)-shaped vhat placeholder each when amsgrad is off); variable names as
    #     the TensorFlow backend of that version spells them ('training/Adam/m_<i>:0').  Expected slots by OUR weight names.
    with h5py.File(os.path.join(OUT, 'keras_model_tiny_opt.h5'), 'w') as f:
        f.attrs['keras_version'] = '2.3.1'.encode('utf8')
        f.attrs['backend'] = 'tensorflow'.encode('utf8')
        f.attrs['model_config'] = json.dumps({'class_name': 'Model', 'config': {'name': 'unet', 'layers': [
            {'name': ln, 'class_name': 'Layer'} for ln, _ in layers]}}).encode('utf8')
        f.attrs['training_config'] = json.dumps({'optimizer_config': {'class_name': 'Adam', 'config': {'lr': 1e-4}},
                                                 'loss': 'l1'}).encode('utf8')
        save_weights_group(f.create_group('model_weights'), layers, values)
        trainable = [(wn, shp) for ln, ws in layers for wn, shp in ws if 'moving' not in wn]
        og = f.create_group('optimizer_weights')
        n = len(trainable)
        names = (['training/Adam/iterations:0'] + ['training/Adam/m_%d:0' % i for i in range(n)] +
                 ['training/Adam/v_%d:0' % i for i in range(n)] + ['training/Adam/vhat_%d:0' % i for i in range(n)])
        save_attributes(og, 'weight_names', names)
        og.create_dataset(names[0], (), dtype='int64')[()] = 4321
        expect = {'iterations': np.array(4321)}
        for i, (wn, shp) in enumerate(trainable):
            m = rng.standard_normal(shp).astype(np.float32) * 1e-3
            v = (rng.random_sample(shp).astype(np.float32) + .1) * 1e-6
            og.create_dataset(names[1 + i], data=m)
            og.create_dataset(names[1 + n + i], data=v)
            og.create_dataset(names[1 + 2 * n + i], data=np.zeros(1, np.float32))
            expect['m/' + wn[:-2]] = m
            expect['v/' + wn[:-2]] = v
        np.savez(os.path.join(OUT, 'keras_h5_opt_expected.npz'), **expect)

    # 3) the same weights through the storage features Keras does not use by default but other writers do:
    #    chunked + shuffle + gzip, chunked without filters, fletcher32, float64 / big-endian / int datasets, a str
    #    (variable-length UTF-8) attribute, a scalar numeric attribute, and chunked attribute lists
    with h5py.File(os.path.join(OUT, 'keras_weights_variants.h5'), 'w') as f:
        long_names = ['layer_with_a_long_name_%04d_' % i + 'x' * 180 for i in range(400)]  # > 64512 bytes -> chunks
        save_attributes(f, 'layer_names', [ln for ln, ws in layers if ws])
        save_attributes(f, 'long_list', long_names)
        f.attrs['keras_version'] = '2.4.0'              # str -> variable-length string (global heap)
        f.attrs['a_float'] = np.float64(2.5)
        f.attrs['ints'] = np.arange(6, dtype=np.int32).reshape(2, 3)
        kinds = [dict(chunks=True, compression='gzip', shuffle=True), dict(chunks=(2,)), dict(chunks=True, fletcher32=True),
                 dict(chunks=True, compression='gzip', compression_opts=9), {}]
        i = 0
        for ln, ws in layers:
            if not ws:
                continue
            g = f.create_group(ln)
            save_attributes(g, 'weight_names', [wn for wn, _ in ws])
            for wn, shp in ws:
                kw = dict(kinds[i % len(kinds)])
                if kw.get('chunks') == (2,):
                    kw['chunks'] = tuple(max(1, (s + 1) // 2) for s in shp)   # ragged edge chunks
                g.create_dataset(wn, data=values[wn], **kw)
                i += 1
        x = f.create_group('extras')
        x.create_dataset('f64', data=np.linspace(0, 1, 7))
        x.create_dataset('be_f32', data=np.arange(5, dtype='>f4'))
        x.create_dataset('i16', data=np.arange(-3, 3, dtype=np.int16).reshape(2, 3))
        x.create_dataset('u8', data=np.arange(250, 256, dtype=np.uint8))
        x.create_dataset('scalar', data=np.float32(3.25))
        x.create_dataset('never_written', (4,), dtype='float32')
        x.create_dataset('strings', data=np.array([b'ab', b'cde', b''], dtype='S4'))
        big = x.create_group('many')       # > 2*K entries: several symbol-table nodes and a two-level group B-tree
        for j in range(300):
            big.create_dataset('d%04d' % j, data=np.float32(j))

    # 4) a file written with the newest object-header format (libver='latest'): compact link messages, v2 headers
    with h5py.File(os.path.join(OUT, 'keras_weights_latest.h5'), 'w', libver='latest') as f:
        few = [lw for lw in layers if lw[1]][:3]
        save_weights_group(f, few, values)
    print('written', [n for n in sorted(os.listdir(OUT)) if n.startswith('keras_')])


if __name__ == '__main__':
    sys.exit(main())
