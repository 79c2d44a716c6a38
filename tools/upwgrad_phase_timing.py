"""per-phase shader cycles of conv3d_split_upwgrad_kernel (thread 0 of two workgroups) from a -DSYN_SPLIT_TIMING build:
    bash tools/build_variant.sh timing -DSYN_SPLIT_TIMING ; python tools/upwgrad_phase_timing.py   (on the GPU box)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthsr_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scratch", "lib_%s.so" % os.environ.get("UW_LIB", "timing"))
from synthsr_amd import ops
import numpy as np
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.synthsr_split_timing_buffer.argtypes = [ctypes.c_void_p]
NAMES = {0: 'start', 1: 'prologue done', 2: 'stage top', 3: 'MFMAs done', 4: 'next stage stored + requested', 5: 'after barrier'}
for D, cl, co in ((80, 48, 24), (40, 96, 48)):
    lo = torch.randn(D, D, D, cl, device='cuda')
    dz = torch.randn(2 * D, 2 * D, 2 * D, co, device='cuda')
    dw = torch.zeros(3, 3, 3, cl, co, device='cuda')
    dwc = torch.empty(8, 27, cl, co, device='cuda')
    for _ in range(3):
        ops.conv3d_up_wgrad(lo, dz, dwc, dw, 0)
    tm = torch.zeros(2 * 400 * 2 + 8 * 48 * 2, dtype=torch.int64, device='cuda')
    torch.cuda.synchronize()
    raw.synthsr_split_timing_buffer(ctypes.c_void_p(tm.data_ptr()))
    ops.conv3d_up_wgrad(lo, dz, dwc, dw, 0)
    torch.cuda.synchronize()
    raw.synthsr_split_timing_buffer(ctypes.c_void_p(0))
    full = tm.cpu().numpy()
    t = full[:1600].reshape(2, 400, 2)
    pw = full[1600:].reshape(8, 48, 2)
    print('== low-res %d^3 %d->%d' % (D, cl, co))
    for wg in range(2):
        rows = t[wg]
        n = int((rows[:, 1] > 0).sum())
        if n < 3:
            continue
        print(' workgroup %d: %d stamps, total %.0f cycles' % (wg, n, rows[n - 1, 1] - rows[0, 1]))
        agg = {}
        for j in range(1, n):
            key = '%s -> %s' % (NAMES[int(rows[j - 1, 0])], NAMES[int(rows[j, 0])])
            agg.setdefault(key, []).append(rows[j, 1] - rows[j - 1, 1])
        for key, v in agg.items():
            v = np.array(v, dtype=np.float64)
            print('   %-36s n %3d  median %8.0f  min %8.0f  max %8.0f  sum %9.0f' % (key, len(v), np.median(v), v.min(), v.max(), v.sum()))
    # per wave: start of the MFMA phase relative to wave 0's, and its duration (workgroup (7, 0), stages 8..40)
    for w in range(8):
        st = pw[w, 8:40, 0] - pw[0, 8:40, 0]
        du = pw[w, 8:40, 1] - pw[w, 8:40, 0]
        print('   wave %d: MFMA phase starts %+5.0f cycles after wave 0 (median), lasts median %6.0f  min %6.0f  max %6.0f'
              % (w, np.median(st), np.median(du), du.min(), du.max()))
