"""per-phase shader cycles of conv3d_split_fwd2_kernel (thread 0 of two workgroups) from a -DSYN_SPLIT_TIMING build:
    bash tools/build_variant.sh timing -DSYN_SPLIT_TIMING ; python tools/fwd2_phase_timing.py   (on the GPU box)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthsr_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'scratch', 'lib_timing.so')
from synthsr_amd import ops
import numpy as np
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.synthsr_split_timing_buffer.argtypes = [ctypes.c_void_p]
NAMES = {0: 'start', 1: 'prologue done', 2: 'chunk top', 3: 'after barrier', 4: 'K loop done', 5: 'epilogue start', 6: 'epilogue done'}
for D, ci, co in ((20, 192, 192), (40, 96, 48), (160, 24, 24)):
    x = torch.randn(D, D, D, ci, device='cuda')
    w = torch.randn(3, 3, 3, ci, co, device='cuda') * .05
    b = torch.zeros(co, device='cuda')
    wp = ops.pack_conv_weights(w, (D, D, D), 0)
    out = torch.empty(D, D, D, co, device='cuda')
    for _ in range(3):
        ops.conv3d(x, wp, b, co, 1, out=out)
    tm = torch.zeros(2 * 400 * 2, dtype=torch.int64, device='cuda')
    torch.cuda.synchronize()
    raw.synthsr_split_timing_buffer(ctypes.c_void_p(tm.data_ptr()))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.conv3d(x, wp, b, co, 1, out=out)
    e.record()
    torch.cuda.synchronize()
    raw.synthsr_split_timing_buffer(ctypes.c_void_p(0))
    t = tm.cpu().numpy().reshape(2, 400, 2)
    print('== %d^3 %d->%d: launch %.1f us' % (D, ci, co, 1e3 * s.elapsed_time(e)))
    for wg in range(2):
        rows = t[wg]
        n = int((rows[:, 1] > 0).sum())
        if n < 3:
            continue
        t0 = rows[0, 1]
        print(' workgroup %d: %d stamps, total %.0f cycles' % (wg, n, rows[n - 1, 1] - t0))
        agg = {}
        for j in range(1, n):
            key = '%s -> %s' % (NAMES[int(rows[j - 1, 0])], NAMES[int(rows[j, 0])])
            agg.setdefault(key, []).append(rows[j, 1] - rows[j - 1, 1])
        for key, v in agg.items():
            v = np.array(v, dtype=np.float64)
            print('   %-34s n %3d  median %8.0f  min %8.0f  max %8.0f  sum %9.0f' % (key, len(v), np.median(v), v.min(), v.max(), v.sum()))
