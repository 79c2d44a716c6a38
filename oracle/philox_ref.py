"""ORACLE (test infrastructure): numpy restatement of the in-kernel noise stream of
csrc/generator.hip — Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11; the published algorithm, also the
generator family TF's `tf.random.normal` uses) followed by Box-Muller.  The reference never seeds TF
(SURVEY F4), so the stream itself is OUR convention; this file pins the kernel to it."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3)]
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)) & MASK, lo1, (hi0 ^ c3 ^ np.uint64(k1)) & MASK, lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32)


def box_muller(r0, r1):
    u1 = ((r0 >> np.uint32(8)).astype(np.float32) + np.float32(1)) * np.float32(2.0 ** -24)
    u2 = (r1 >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    rad = np.sqrt(np.float32(-2.0) * np.log(u1))
    ang = np.float32(6.283185307179586) * u2
    return rad * np.cos(ang), rad * np.sin(ang)


def normals(n_vox, n_channels, key, offset):
    """noise [n_vox, n_channels] exactly as deform_gmm_kernel draws it: counter = (voxel, offset); channels 4 g .. 4 g + 3 (one
    kernel launch per group of four) come from the counter whose last word carries g in its bits 16 and up"""
    v = np.arange(n_vox, dtype=np.uint64)
    c0, c1 = v & MASK, v >> np.uint64(32)
    c2 = np.full(n_vox, np.uint64(offset) & MASK)
    out = []
    for grp in range((n_channels + 3) // 4):
        c3 = np.full(n_vox, ((np.uint64(offset) >> np.uint64(32)) + np.uint64(grp << 16)) & MASK)
        r = philox4x32_10(c0, c1, c2, c3, int(key[0]), int(key[1]))
        n0, n1 = box_muller(r[0], r[1])
        n2, n3 = box_muller(r[2], r[3])
        out += [n0, n1, n2, n3]
    return np.stack(out[:n_channels], -1).astype(np.float32)
