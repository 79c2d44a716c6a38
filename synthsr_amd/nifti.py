"""Minimal NIfTI-1 (.nii / .nii.gz) reader/writer (nibabel is not available on the target image).

Covers what the hot path's host side needs (reference: ext/lab2im/utils.py:76-160 `load_volume`,
:122 `save_volume`, :163 `get_volume_info`): single-file NIfTI-1, scl_slope/inter, sform/qform
affine.  Not a general neuro-imaging IO library.
"""
import gzip
import struct
import numpy as np

_DT = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64,
       256: np.int8, 512: np.uint16, 768: np.uint32, 1024: np.int64, 1280: np.uint64}
_DT_INV = {np.dtype(v).name: k for k, v in _DT.items()}


def _open(path, mode='rb'):
    return gzip.open(path, mode) if str(path).endswith('.gz') else open(path, mode)


def _quat_to_affine(hdr):
    b, c, d = hdr['quatern_b'], hdr['quatern_c'], hdr['quatern_d']
    a2 = 1.0 - (b * b + c * c + d * d)
    a = np.sqrt(max(a2, 0.0))
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
    qfac = -1.0 if hdr['pixdim'][0] < 0 else 1.0
    S = np.diag([hdr['pixdim'][1], hdr['pixdim'][2], hdr['pixdim'][3] * qfac])
    aff = np.eye(4)
    aff[:3, :3] = R @ S
    aff[:3, 3] = [hdr['qoffset_x'], hdr['qoffset_y'], hdr['qoffset_z']]
    return aff


def read_nifti(path):
    """-> (data ndarray in on-disk dtype scaled by slope/inter if set, affine 4x4 float64, header dict)"""
    with _open(path) as f:
        raw = f.read()
    if struct.unpack('<i', raw[:4])[0] == 348:
        e = '<'
    elif struct.unpack('>i', raw[:4])[0] == 348:
        e = '>'
    else:
        raise ValueError('%s is not a NIfTI-1 file' % path)
    dim = struct.unpack(e + '8h', raw[40:56])
    datatype, bitpix = struct.unpack(e + '2h', raw[70:74])
    pixdim = struct.unpack(e + '8f', raw[76:108])
    vox_offset, scl_slope, scl_inter = struct.unpack(e + '3f', raw[108:120])
    qform_code, sform_code = struct.unpack(e + '2h', raw[252:256])
    qb, qc, qd, qx, qy, qz = struct.unpack(e + '6f', raw[256:280])
    srow = np.array(struct.unpack(e + '12f', raw[280:328]), dtype=np.float64).reshape(3, 4)
    hdr = dict(dim=dim, datatype=datatype, bitpix=bitpix, pixdim=pixdim, vox_offset=vox_offset,
               scl_slope=scl_slope, scl_inter=scl_inter, qform_code=qform_code, sform_code=sform_code,
               quatern_b=qb, quatern_c=qc, quatern_d=qd, qoffset_x=qx, qoffset_y=qy, qoffset_z=qz,
               srow=srow, endian=e)
    if datatype not in _DT:
        raise ValueError('unsupported NIfTI datatype %d' % datatype)
    ndim = dim[0]
    shape = tuple(int(d) for d in dim[1:1 + ndim])
    while len(shape) > 3 and shape[-1] == 1:
        shape = shape[:-1]
    dt = np.dtype(_DT[datatype]).newbyteorder(e)
    n = int(np.prod(shape))
    off = int(vox_offset) if vox_offset >= 352 else 352
    data = np.frombuffer(raw, dtype=dt, count=n, offset=off).reshape(shape, order='F')
    if scl_slope not in (0.0,) and not np.isnan(scl_slope) and (scl_slope != 1.0 or scl_inter != 0.0):
        data = data.astype(np.float64) * scl_slope + scl_inter
    if sform_code > 0:
        aff = np.eye(4)
        aff[:3, :] = srow
    elif qform_code > 0:
        aff = _quat_to_affine(hdr)
    else:
        # neither sform nor qform: nibabel's base affine (what the reference gets from nib.load): zooms on the diagonal with
        # the first axis flipped, origin at the centre voxel
        zooms = np.array([pixdim[1], pixdim[2], pixdim[3]], dtype=np.float64)
        zooms[zooms == 0] = 1.0
        aff = np.diag([-zooms[0], zooms[1], zooms[2], 1.0])
        aff[:3, 3] = -aff[:3, :3] @ ((np.array(shape[:3], dtype=np.float64) - 1) / 2.0)
    return np.ascontiguousarray(data), aff, hdr


def write_nifti(path, data, affine=None, dtype=None):
    """writes a single-file NIfTI-1 with an sform affine (default identity)"""
    data = np.asarray(data)
    if dtype is not None:
        data = data.astype(dtype)
    if data.dtype == np.bool_:
        data = data.astype(np.uint8)
    if data.dtype.name not in _DT_INV:
        data = data.astype(np.float32)
    affine = np.eye(4) if affine is None else np.asarray(affine, dtype=np.float64)
    shape = data.shape
    dim = [len(shape)] + list(shape) + [1] * (7 - len(shape))
    vox = np.sqrt(np.sum(affine[:3, :3] ** 2, axis=0))
    pixdim = [1.0] + [float(v) for v in vox] + [1.0] * 4
    h = bytearray(348)
    struct.pack_into('<i', h, 0, 348)
    struct.pack_into('<8h', h, 40, *dim)
    struct.pack_into('<2h', h, 70, _DT_INV[data.dtype.name], data.dtype.itemsize * 8)
    struct.pack_into('<8f', h, 76, *pixdim)
    struct.pack_into('<3f', h, 108, 352.0, 1.0, 0.0)
    struct.pack_into('<B', h, 123, 2)  # xyzt_units: mm
    struct.pack_into('<2h', h, 252, 0, 1)  # qform_code 0, sform_code 1
    struct.pack_into('<12f', h, 280, *affine[:3, :].reshape(-1))
    h[344:348] = b'n+1\x00'
    with _open(path, 'wb') as f:
        f.write(bytes(h) + b'\x00' * 4)
        f.write(np.asfortranarray(data).tobytes(order='F'))
