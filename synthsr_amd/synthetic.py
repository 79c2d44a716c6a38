"""Synthetic benchmark inputs: head-sized label maps made of nested ellipsoidal shells and
blobs with the 19 generation labels / 14 classes and the T1 priors that ship with the reference
(values of data/labels_classes_priors/{generation_labels,generation_classes,prior_*_t1_hr}.npy, data only).
There is no network on the benchmark box and the reference's data directory does not travel."""
import numpy as np

GENERATION_LABELS = np.array([0, 14, 15, 16, 2, 3, 4, 5, 7, 8, 10, 11, 12, 13, 17, 18, 26, 28, 31], dtype=np.int32)
GENERATION_CLASSES = np.array([0, 3, 3, 4, 1, 2, 3, 3, 1, 2, 5, 6, 7, 8, 9, 10, 11, 12, 13], dtype=np.int32)
PRIOR_MEANS_T1_HR = np.array(
    [[0., 226.39749756, 130.1802597, 58.78450394, 180.03954468, 188.77471466, 152.42570953, 179.17560577,
      224.64091797, 129.22801056, 134.12723236, 150.08264465, 209.16018677, 114.48826218],
     [0., 3.69382072, 10.10750663, 13.70010701, 3.37157044, 7.07953396, 8.76154213, 6.9904825, 9.48875256,
      9.93875443, 8.56745962, 6.81452129, 5.07955181, 17.64642318]])
PRIOR_STDS_T1_HR = np.array(
    [[0., 15.3967035, 18.45478487, 28.59792772, 10.89657271, 15.44379147, 10.45586083, 10.67849573, 6.77283017,
      10.47950663, 8.00639687, 5.70405782, 12.40567971, 30.38147852],
     [0., 4.40367605, 2.85733914, 20.92950483, 2.43298443, 2.81659789, 2.8678922, 1.65150387, 1.90385766,
      1.94075614, 2.247009, 2.39205922, 2.64180287, 5.62032277]])


def synthetic_label_map(shape=(160, 160, 160), seed=1234):
    """int32 label map, ~27 % foreground (the shipped maps are 71-75 % background)"""
    rng = np.random.default_rng(seed)
    shape = tuple(int(s) for s in shape)
    g = np.meshgrid(*[np.arange(s, dtype=np.float32) for s in shape], indexing='ij')
    c = [(s - 1) / 2 + rng.uniform(-.02, .02) * s for s in shape]
    ax = [.38 * shape[0], .45 * shape[1], .38 * shape[2]]
    r = np.sqrt(sum(((g[d] - c[d]) / ax[d]) ** 2 for d in range(3)))
    lab = np.zeros(shape, dtype=np.int32)
    inside = r < 1
    # shells: CSF-like rim, cortex, white matter
    lab[inside] = 3   # cortex
    lab[r < .82] = 2  # white matter
    lab[(r >= .93) & inside] = 14
    # deep structures as small ellipsoids (sided labels on one hemisphere, mirrored copies keep the same label)
    others = [4, 5, 7, 8, 10, 11, 12, 13, 15, 16, 17, 18, 26, 28, 31]
    for k, la in enumerate(others):
        ctr = [c[d] + rng.uniform(-.45, .45) * ax[d] for d in range(3)]
        rad = [rng.uniform(.05, .14) * ax[d] for d in range(3)]
        rr = sum(((g[d] - ctr[d]) / rad[d]) ** 2 for d in range(3))
        lab[(rr < 1) & (r < .8)] = la
    return lab


def synthetic_label_pool(n=8, shape=(160, 160, 160), seed=1234):
    return [synthetic_label_map(shape, seed + i) for i in range(n)]
