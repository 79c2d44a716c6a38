#!/usr/bin/env python
"""Empirical HBM bandwidth of the box ("re-measure an empirical peak ... and report both"): device-to-device
copy (read + write bytes / time) with torch's copy kernel, hipMemcpyDtoD, the library's own scalar copy kernel and a
read-only reduction, on buffers far larger than the 256 MB Infinity Cache.   python tools/ubench/hbm_peak.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from synthsr_amd import _lib  # noqa: E402


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


def main():
    n = 1 << 29  # 2 GiB of float32 per buffer
    x = torch.rand(n, device='cuda')
    y = torch.empty_like(x)
    lib = _lib.load()
    gb = n * 4 / 1e9
    t = timed(lambda: y.copy_(x))
    print('torch copy_            : %7.1f GB/s (read + write)' % (2 * gb / t))
    t = timed(lambda: _lib.check(lib.synthsr_copy_strided(_lib.ptr(x), _lib.ptr(y), n, 1, 0, 1, 0, _lib.stream()), 'copy'))
    print('synthsr_copy_strided   : %7.1f GB/s (read + write, 4-byte accesses)' % (2 * gb / t))
    t = timed(lambda: torch.sum(x))
    print('torch sum (read only)  : %7.1f GB/s' % (gb / t))
    t = timed(lambda: y.zero_())
    print('torch zero_ (write only): %7.1f GB/s' % (gb / t))


if __name__ == '__main__':
    main()
