"""Golden vectors for the host input sampler, produced by running the reference's own
SynthSR/model_inputs.py:build_model_inputs and ext/lab2im/utils.py:draw_value_from_distribution (numpy branch) under a
SEEDED global numpy state.  Development container only:   python tests/golden/gen/make_model_inputs_golden.py
Writes tests/golden/model_inputs.npz (data only)."""
import os
import sys
import tempfile
import importlib.util
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.dirname(HERE)
REF = '/root/reference'
sys.path.insert(0, HERE)
import tf_numpy_shim as shim  # noqa: E402

shim.install([], REF)
from ext.lab2im import utils as l2i_utils  # noqa: E402

_spec = importlib.util.spec_from_file_location('ref_model_inputs', os.path.join(REF, 'SynthSR', 'model_inputs.py'))
ref_mi = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ref_mi)

out = {}
rng = np.random.default_rng(3)
K, L = 6, 9
classes = np.array([0, 1, 1, 2, 3, 3, 4, 5, 2])
pm = np.stack([rng.uniform(20, 200, K), rng.uniform(5, 40, K)])
ps = np.stack([rng.uniform(5, 20, K), rng.uniform(1, 6, K)])
pm2 = np.concatenate([pm, pm[:, ::-1] * .5])
ps2 = np.concatenate([ps, ps[:, ::-1] * 2.])
out.update(classes=classes, pm=pm, ps=ps, pm2=pm2, ps2=ps2)

# ---- draw_value_from_distribution, every hyper-parameter form
cases = [('none_u', None, 4, 'uniform', 125., 100., True), ('num_n', 7.5, 3, 'normal', 10., 2., False),
         ('pair_u', [2., 9.], 5, 'uniform', 0., 10., False), ('pair_n', (3., .5), 5, 'normal', 0., 10., True),
         ('arr_n', pm, 1, 'normal', 125., 100., True), ('arr2_u', np.abs(pm2), 1, 'uniform', 125., 100., False),
         ('neg_n', np.stack([np.full(8, -1.), np.full(8, 3.)]), 1, 'normal', 0., 1., True)]
for tag, hp, size, dist, centre, rg, pos in cases:
    np.random.seed(100 + len(tag))
    vals = [l2i_utils.draw_value_from_distribution(hp, size, dist, centre, rg, positive_only=pos) for _ in range(3)]
    out['dv_' + tag] = np.stack(vals)
    out['dv_' + tag + '_seed'] = np.int64(100 + len(tag))
assert l2i_utils.draw_value_from_distribution(False) is None

# ---- build_model_inputs on three tiny .npz label maps (+ matching scans)
tmp = tempfile.mkdtemp()
lab_paths, im_paths = [], []
for i in range(3):
    lab = rng.integers(0, L, (6, 5, 4)).astype(np.int32)
    im = rng.normal(100, 20, (6, 5, 4)).astype(np.float32)
    lab_paths.append(os.path.join(tmp, 'lab%d.npz' % i))
    im_paths.append(os.path.join(tmp, 'im%d.npz' % i))
    np.savez_compressed(lab_paths[-1], vol_data=lab)
    np.savez_compressed(im_paths[-1], vol_data=im)
    out['lab%d' % i], out['im%d' % i] = lab, im
runs = [('a', pm, ps, 'normal', None, 1, 1, classes), ('b', pm2, ps2, 'normal', None, 1, 2, classes),
        ('c', None, None, 'uniform', None, 1, 1, None), ('d', [30., 150.], 12., 'uniform', im_paths, 1, 3, classes),
        ('e', pm, ps, 'uniform', im_paths, 2, 1, classes)]
for tag, m, s, dist, ims, bs, nch, cls in runs:
    np.random.seed(7 + ord(tag))
    gen = ref_mi.build_model_inputs(lab_paths, L, m, s, dist, path_images=ims, batchsize=bs, n_channels=nch,
                                    generation_classes=cls)
    for it in range(3):
        items = next(gen)
        for j, a in enumerate(items):
            out['mi_%s_%d_%d' % (tag, it, j)] = np.asarray(a)
    out['mi_%s_seed' % tag] = np.int64(7 + ord(tag))
np.savez_compressed(os.path.join(OUT, 'model_inputs.npz'), **out)
print({k: v.shape for k, v in out.items() if k.startswith('mi_a')})
