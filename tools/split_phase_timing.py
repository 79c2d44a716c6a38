"""per-phase shader cycles (s_memtime) of conv3d_split_fwd_kernel, wave 0 of workgroups 0 and 300, from an instrumented build:

    cd synthsr_amd/csrc && hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSYN_SPLIT_TIMING -c conv_split.hip -o /tmp/cs_t.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/libsynthsr_hip_timing.so generator.o unet_pointwise.o \
          ssim.o critic.o conv_bf16.o /tmp/cs_t.o conv3d.o
    python tools/split_phase_timing.py        (on the GPU box)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthsr_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'scratch', 'libsynthsr_hip_timing.so')
from synthsr_amd import ops
import numpy as np
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.synthsr_split_timing_buffer.argtypes = [ctypes.c_void_p]
ACT = int(os.environ.get('SYN_TIMING_ACT', '1'))
for D, ci, co in ((160, 24, 24), (80, 48, 48)):
    x = torch.randn(D, D, D, ci, device='cuda')
    w = torch.randn(3, 3, 3, ci, co, device='cuda') * .05
    b = torch.zeros(co, device='cuda')
    wp = ops.pack_conv_weights(w, (D, D, D), 0)
    out = torch.empty(D, D, D, co, device='cuda')
    for _ in range(3):
        ops.conv3d(x, wp, b, co, ACT, out=out)
    tm = torch.zeros(2 * 120 * 8, dtype=torch.int64, device='cuda')
    torch.cuda.synchronize()
    raw.synthsr_split_timing_buffer(ctypes.c_void_p(tm.data_ptr()))
    ops.conv3d(x, wp, b, co, ACT, out=out)
    torch.cuda.synchronize()
    raw.synthsr_split_timing_buffer(ctypes.c_void_p(0))
    t = tm.cpu().numpy().reshape(2, 120, 8)
    ncc = ci // 8
    print('== %d^3 %d->%d (chunks per tile %d)' % (D, ci, co, ncc))
    for wg in range(2):
        rows = t[wg]
        n = int((rows[:, 0] > 0).sum())
        names = ['barrier', 'issue-halo-loads', 'K-loop', 'wait-for-loads', 'convert+lds-write', 'to-next (epilogue on last chunk)']
        d, last = [], []
        for j in range(2, n - 1):
            r = rows[j]
            nxt = rows[j + 1][0]
            row = [r[1] - r[0], r[2] - r[1], r[3] - r[2], r[6] - r[3], r[4] - r[6], nxt - r[4]]
            (last if (j % ncc) == ncc - 1 else d).append(row)
        for nm, arr in (('inner chunks', d), ('last chunk of a tile', last)):
            arr = np.array(arr, dtype=np.float64)
            if len(arr):
                print(' wg%d %-22s n %3d  median cycles: ' % (wg, nm, len(arr)) +
                      '  '.join('%s %.0f' % (k, v) for k, v in zip(names, np.median(arr, 0))), ' total %.0f' % np.median(arr.sum(1)))
