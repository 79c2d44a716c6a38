"""FreeSurfer MGH / MGZ volumes (the `.mgz` branch of ext/lab2im/utils.py:76-160 `load_volume` / `save_volume`, which
the reference delegates to nibabel).  Written from the published MGH format description (FreeSurfer wiki
"FsTutorial/MghFormat"): a 284-byte big-endian header {int32 version = 1, width, height, depth, nframes, type, dof;
int16 goodRASFlag; float32 spacing[3]; float32 x_ras[3], y_ras[3], z_ras[3]; float32 c_ras[3]}, then the voxels
big-endian with x fastest (frames last); `.mgz` is the gzip of that.  vox2ras = [Mdc*spacing | c_ras - Mdc*spacing*(dims/2)].

PARITY UNPINNED: the image holds neither nibabel nor a single .mgz/.mgh sample, so this module is only checked for
internal consistency (byte layout against the offsets above, write -> read round trips, NIfTI <-> MGZ affine agreement)
in tests/test_inference.py; it has not been compared with files produced by FreeSurfer or nibabel."""
import gzip
import struct

import numpy as np

_TYPES = {0: np.dtype('>u1'), 1: np.dtype('>i4'), 3: np.dtype('>f4'), 4: np.dtype('>i2')}
_CODES = {np.dtype('uint8'): 0, np.dtype('int32'): 1, np.dtype('float32'): 3, np.dtype('int16'): 4}
HEADER_SIZE = 284


def _open(path, mode):
    return gzip.open(path, mode) if path.endswith('.mgz') else open(path, mode)


def read_mgh(path):
    """-> (data [w,h,d(,frames)] in its on-disk type (native byte order), vox2ras affine 4x4 float64, header dict)"""
    with _open(path, 'rb') as f:
        raw = f.read()
    if len(raw) < HEADER_SIZE:
        raise ValueError('%s is too short for an MGH header' % path)
    version, w, h, d, nframes, mtype, dof = struct.unpack('>7i', raw[:28])
    if version != 1:
        raise ValueError('%s is not an MGH file (version %d)' % (path, version))
    if mtype not in _TYPES:
        raise ValueError('unsupported MGH data type %d' % mtype)
    good_ras, = struct.unpack('>h', raw[28:30])
    if good_ras > 0:
        delta = np.array(struct.unpack('>3f', raw[30:42]), dtype=np.float64)
        mdc = np.array(struct.unpack('>9f', raw[42:78]), dtype=np.float64).reshape(3, 3).T   # columns x_ras, y_ras, z_ras
        c_ras = np.array(struct.unpack('>3f', raw[78:90]), dtype=np.float64)
    else:   # FreeSurfer's defaults when the orientation fields are not valid: coronal, 1 mm
        delta = np.ones(3)
        mdc = np.array([[-1., 0., 0.], [0., 0., 1.], [0., -1., 0.]])
        c_ras = np.zeros(3)
    shape = (w, h, d) if nframes == 1 else (w, h, d, nframes)
    n = int(np.prod(shape))
    dt = _TYPES[mtype]
    if len(raw) < HEADER_SIZE + n * dt.itemsize:
        raise ValueError('%s is truncated' % path)
    data = np.frombuffer(raw, dtype=dt, count=n, offset=HEADER_SIZE).reshape(shape, order='F')
    data = np.ascontiguousarray(data.astype(dt.newbyteorder('=')))
    m = mdc * delta[None, :]
    aff = np.eye(4)
    aff[:3, :3] = m
    aff[:3, 3] = c_ras - m @ (np.array([w, h, d], dtype=np.float64) / 2.0)
    hdr = dict(dims=(w, h, d, nframes), type=mtype, dof=dof, goodRASFlag=good_ras, delta=tuple(delta), Mdc=mdc.T.copy(),
               Pxyz_c=tuple(c_ras), pixdim=(1.0,) + tuple(float(v) for v in delta) + (1.0,) * 4)
    return data, aff, hdr


def write_mgh(path, data, affine=None, dtype=None):
    """data [w,h,d] or [w,h,d,frames]; dtype None keeps uint8 / int16 / int32 / float32 and stores anything else as
    float32 (integers as int32)"""
    data = np.asarray(data)
    if data.ndim not in (3, 4):
        raise ValueError('MGH volumes are 3-D (+ frames), had shape %s' % (data.shape,))
    dt = np.dtype(dtype) if dtype is not None else data.dtype
    if dt not in _CODES:
        dt = np.dtype('int32') if dt.kind in 'iub' else np.dtype('float32')
    affine = np.eye(4) if affine is None else np.asarray(affine, dtype=np.float64)
    m = affine[:3, :3]
    delta = np.sqrt((m * m).sum(0))
    delta[delta == 0] = 1.0
    mdc = m / delta[None, :]
    dims = np.array(data.shape[:3], dtype=np.float64)
    c_ras = affine[:3, 3] + m @ (dims / 2.0)
    hdr = struct.pack('>7i', 1, data.shape[0], data.shape[1], data.shape[2], data.shape[3] if data.ndim == 4 else 1,
                      _CODES[dt], 0)
    hdr += struct.pack('>h', 1) + struct.pack('>3f', *delta) + struct.pack('>9f', *mdc.T.reshape(-1))
    hdr += struct.pack('>3f', *c_ras)
    hdr += bytes(HEADER_SIZE - len(hdr))
    with _open(path, 'wb') as f:
        f.write(hdr)
        f.write(np.asfortranarray(data.astype(dt.newbyteorder('>'))).tobytes(order='F'))
