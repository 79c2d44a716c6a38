"""synthsr_amd/estimate_priors.py against the reference's own functions (SynthSR/estimate_priors.py:76-310 executed by
tests/golden/gen/make_goldens.py::golden_estimate_priors on synthetic datasets) - SURVEY §8f row 4."""
import os

import numpy as np
import pytest

from synthsr_amd import estimate_priors as EP
from synthsr_amd import volumes as V

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'estimate_priors.npz'))
TOL = dict(rtol=1e-12, atol=1e-12)   # float64 medians of the same values: only the order of summation differs


def test_single_image_stats_with_nans():
    labels, classes = G['labels_list'], G['classes_list']
    got = EP.sample_intensity_stats_from_image(G['single_image'], G['d0_labels_0'], labels, classes)
    assert got.shape == (2, 5) and np.allclose(got, G['single_stats'], **TOL)
    assert got[0, 4] == 0 and got[1, 4] == 0                 # class of a label that never occurs
    got0 = EP.sample_intensity_stats_from_image(G['single_image'], G['d0_labels_0'], labels, classes,
                                                keep_strictly_positive=False)
    assert np.allclose(got0, G['single_stats_keep0'], **TOL) and not np.allclose(got0, got)
    # a label listed twice counts once per listing: same median/MAD as listing it once in that class
    twice = EP.sample_intensity_stats_from_image(G['single_image'], G['d0_labels_0'], list(labels) + [2], list(classes) + [1])
    assert np.allclose(twice, got, **TOL)
    with pytest.raises(ValueError):
        EP.sample_intensity_stats_from_image(G['single_image'], G['d0_labels_0'], labels, classes * 2)
    with pytest.raises(AssertionError):
        EP.sample_intensity_stats_from_image(G['single_image'], G['d0_labels_0'], labels, classes[:-1])


def test_rescale_volume():
    x = G['d0_image_1'][:12]
    assert np.allclose(V.rescale_volume(x), G['rescaled'], **TOL)
    assert np.allclose(V.rescale_volume(x, new_min=-1, new_max=1, min_percentile=0, max_percentile=99,
                                        use_positive_only=True), G['rescaled_pos'], **TOL)
    assert not V.rescale_volume(np.full((3, 3, 3), 7.0)).any()


def _write_datasets(tmp_path):
    dirs = []
    for d, n_sub in enumerate([3, 2]):
        im_dir, la_dir = tmp_path / ('im%d' % d), tmp_path / ('la%d' % d)
        im_dir.mkdir(), la_dir.mkdir()
        for i in range(n_sub):
            np.savez_compressed(str(im_dir / ('s%d.npz' % i)), vol_data=G['d%d_image_%d' % (d, i)])
            np.savez_compressed(str(la_dir / ('s%d.npz' % i)), vol_data=G['d%d_labels_%d' % (d, i)])
        dirs.append((str(im_dir), str(la_dir)))
    return dirs


def test_build_intensity_stats(tmp_path):
    dirs = _write_datasets(tmp_path)
    res = str(tmp_path / 'res')
    np.save(str(tmp_path / 'labels.npy'), G['labels_list'])
    pm, ps = EP.build_intensity_stats([d[0] for d in dirs], [d[1] for d in dirs], res, str(tmp_path / 'labels.npy'),
                                      G['classes_list'])
    assert pm.shape == (6, 5)                                # 1 channel + 2 channels, two rows each
    assert np.allclose(pm, G['prior_means'], **TOL) and np.allclose(ps, G['prior_stds'], **TOL)
    assert np.array_equal(np.load(os.path.join(res, 'prior_means.npy')), pm)
    assert np.array_equal(np.load(os.path.join(res, 'prior_stds.npy')), ps)
    pm1, ps1 = EP.build_intensity_stats(dirs[0][0], dirs[0][1], res, G['labels_list'], None, rescale=False)
    assert np.allclose(pm1, G['prior_means_noclasses_norescale'], **TOL)
    assert np.allclose(ps1, G['prior_stds_noclasses_norescale'], **TOL)
    with pytest.raises(AssertionError):                      # folders of different sizes
        EP.build_intensity_stats(dirs[0][0], dirs[1][1], res, G['labels_list'])
