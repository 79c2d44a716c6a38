"""Keras .h5 <-> .npz checkpoint conversion.

    python scripts/convert_keras_h5.py models/SynthSR_v10_210712.h5 models/SynthSR_v10_210712.npz
    python scripts/convert_keras_h5.py model_dir/020.npz model_dir/020.h5

.h5 -> .npz reads a Keras `save_weights()` / `model.save()` / `ModelCheckpoint` file without an HDF5 library;
.npz -> .h5 writes the `save_weights()` layout that the reference's `load_weights(path, by_name=True)` reads."""
import os
import sys
from argparse import ArgumentParser

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synthsr_amd import keras_h5  # noqa: E402


def main(argv=None):
    parser = ArgumentParser(description=__doc__.split('\n')[0])
    parser.add_argument('src', help='.h5 (Keras) or .npz (synthsr_amd.training.save_checkpoint)')
    parser.add_argument('dst', help='.npz or .h5')
    args = parser.parse_args(argv)
    src_h5 = args.src.lower().endswith(('.h5', '.hdf5'))
    dst_h5 = args.dst.lower().endswith(('.h5', '.hdf5'))
    if src_h5:
        sd = keras_h5.load_keras_weights(args.src)
    else:
        z = np.load(args.src)
        sd = {k: z[k] for k in z.files if not k.startswith('optimizer/')}
    if not sd:
        sys.exit('no weights found in %s' % args.src)
    if dst_h5:
        keras_h5.save_keras_weights(args.dst, {k: (v.reshape((1, 1, 1) + v.shape) if k.endswith('/kernel') and
                                                   v.ndim == 2 else v) for k, v in sd.items()})
    else:
        np.savez(args.dst, **sd)
    print('%s: %d arrays, %d parameters -> %s' % (args.src, len(sd), sum(int(v.size) for v in sd.values()), args.dst))


if __name__ == '__main__':
    main()
