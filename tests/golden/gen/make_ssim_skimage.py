"""Independent known answers for the oracle's restatement of tf.image.ssim (oracle/unet_ref.py:tf_image_ssim): the
Gaussian-weighted SSIM of scikit-image, an unrelated implementation of the same published definition (Wang et al. 2004:
11x11 window, sigma 1.5, K1 .01, K2 .03, population covariances, borders cropped by the window radius = 'VALID').

    /opt/conda/bin/python3.9 tests/golden/gen/make_ssim_skimage.py       (scikit-image 0.18.3 in this image)

Writes tests/golden/ssim_skimage.npz: image pairs [slices, H, W] in [0, 1] and the per-slice mean SSIM."""
import os

import numpy as np
import skimage
from skimage.metrics import structural_similarity

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'ssim_skimage.npz')
rng = np.random.RandomState(2004)
x = rng.uniform(0, 1, (12, 24, 19)).astype(np.float32)
y = np.clip(0.7 * x + 0.3 * rng.uniform(0, 1, x.shape) + 0.05 * rng.standard_normal(x.shape), 0, 1).astype(np.float32)
y[4] = x[4]                      # identical slice -> 1
y[5] = 1.0 - x[5]                # anti-correlated slice -> negative structure term
vals = [structural_similarity(x[i].astype(np.float64), y[i].astype(np.float64), gaussian_weights=True, sigma=1.5,
                              use_sample_covariance=False, data_range=1.0) for i in range(x.shape[0])]
np.savez_compressed(OUT, x=x, y=y, ssim=np.array(vals), skimage_version=skimage.__version__)
print(skimage.__version__, vals)
