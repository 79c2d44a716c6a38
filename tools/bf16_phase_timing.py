"""per-phase cycle counts (clock64) of conv3d_bf16_fwd_kernel, wave 0 of workgroups 0 and 300, from an instrumented build:

    cd synthsr_amd/csrc && hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSYN_BF16_TIMING -c conv_bf16.hip -o /tmp/cb_t.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/libsynthsr_hip_timing.so generator.o unet_pointwise.o \
          ssim.o critic.o /tmp/cb_t.o conv3d.o
    python tools/bf16_phase_timing.py        (on the GPU box)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthsr_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'scratch', os.environ.get('SYN_TIMING_LIB', 'libsynthsr_hip_timing.so'))
from synthsr_amd import ops
import numpy as np
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.synthsr_bf16_timing_buffer.argtypes = [ctypes.c_void_p]
ACT = int(os.environ.get('SYN_TIMING_ACT', '1'))
for D, ci, co, stats in ((160, 24, 24, False), (160, 8, 24, False), (160, 72, 24, False), (80, 48, 48, False)):
    x = torch.randn(D, D, D, ci, device='cuda').bfloat16()
    w = torch.randn(3, 3, 3, ci, co, device='cuda') * .05
    b = torch.zeros(co, device='cuda')
    wp = ops.pack_conv_weights_bf16(w, 0)
    out = torch.empty(D, D, D, co, device='cuda', dtype=torch.bfloat16)
    for _ in range(3):
        ops.conv3d_bf16(x, wp, b, co, ACT, out=out)
    tm = torch.zeros(2 * 40 * 8, dtype=torch.int64, device='cuda')
    torch.cuda.synchronize()
    raw.synthsr_bf16_timing_buffer(ctypes.c_void_p(tm.data_ptr()))
    ops.conv3d_bf16(x, wp, b, co, ACT, out=out)
    torch.cuda.synchronize()
    raw.synthsr_bf16_timing_buffer(ctypes.c_void_p(0))
    t = tm.cpu().numpy().reshape(2, 40, 8)
    print('== %d^3 %d->%d' % (D, ci, co))
    for wg in range(2):
        rows = t[wg]
        n = int((rows[:, 0] > 0).sum())
        names = ['barrierA', 'wait+ldswrite', 'barrierB', 'issue-loads', 'K-loop', 'epilogue', 'loop-back']
        d = []
        for j in range(1, n - 1):
            r = rows[j]
            nxt = rows[j + 1][0]
            d.append([r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[6] - r[5], nxt - r[6]])
        d = np.array(d, dtype=np.float64)
        if len(d):
            print(' wg%d tiles %d  per-tile cycles (median): ' % (wg, n) + '  '.join('%s %.0f' % (nm, v) for nm, v in zip(names, np.median(d, 0))),
                  ' total %.0f' % np.median(d.sum(1)))
