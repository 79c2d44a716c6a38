#!/usr/bin/env python
"""A/B of two BUILDS of libsynthsr_hip.so on one GPU box (the library has no run-time switches: a kernel experiment is a second
build with a -D macro, selected through SYNTHSR_HIP_LIB):

    python tools/lib_ab.py synthsr_amd/libsynthsr_hip.so tools/ubench/libsynthsr_hip_b.so [reps]

Each library runs in its own process, alternately (A B A B); per process: the split forward / forward + statistics / data gradient
with the ELU' epilogue / weight gradient of the layer shapes below (time per launch, torch.cuda.Event over `reps` launches) and an
MD5 of every result -- two builds that only re-schedule instructions must agree bit for bit -- then `bench.py --steps 30` once per
library."""
import hashlib
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(160, 24, 24), (80, 24, 48), (80, 48, 48), (40, 96, 96)]


def worker(reps):
    sys.path.insert(0, REPO)
    import torch
    from synthsr_amd import ops

    def t(fn):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps

    def md5(x):
        return hashlib.md5(x.detach().cpu().numpy().tobytes()).hexdigest()[:12]
    out = {}
    for D, ci, co in SHAPES:
        g = torch.Generator(device='cpu').manual_seed(D + ci + co)
        x = torch.randn(D, D, D, ci, generator=g).cuda()
        dy = torch.randn(D, D, D, co, generator=g).cuda()
        below = torch.nn.functional.elu(torch.randn(D, D, D, ci, generator=g)).cuda()
        w = (torch.randn(3, 3, 3, ci, co, generator=g) * 0.05).cuda()
        b = (torch.randn(co, generator=g) * 0.1).cuda()
        shape = (D, D, D)
        wp, wpd = ops.pack_conv_weights(w, shape, 0), ops.pack_conv_weights(w, shape, 1)
        y, dx = torch.empty(D, D, D, co, device='cuda'), torch.empty(D, D, D, ci, device='cuda')
        stats, ws = torch.zeros(2 * co, device='cuda'), torch.zeros(2 * co, dtype=torch.float64, device='cuda')
        dw, db = torch.zeros_like(w), torch.zeros_like(b)
        r = {}
        r['fwd'] = t(lambda: ops.conv3d(x, wp, b, co, 1, out=y))
        r['fwd_md5'] = md5(y)
        r['fwd+stats'] = t(lambda: ops.conv3d_stats(x, wp, b, co, stats, ws, 1, out=y))
        r['stats_md5'] = md5(stats)
        r['dgrad*elu\''] = t(lambda: ops.conv3d_add(dy, wpd, None, below, ci, 2, out=dx))
        r['dgrad_md5'] = md5(dx)
        r['wgrad'] = t(lambda: ops.conv3d_wgrad(x, dy, dw, db))
        prev = ops.set_deterministic(True)     # ordered sums: the weight gradient's bits are comparable between builds
        dwd, dbd = torch.zeros_like(w), torch.zeros_like(b)
        ops.conv3d_wgrad(x, dy, dwd, dbd)
        ops.set_deterministic(prev)
        r['wgrad_md5'] = md5(torch.cat([dwd.reshape(-1), dbd]))
        out['%d^3 %d->%d' % (D, ci, co)] = r
        del x, dy, below, y, dx
    print('LIBAB ' + json.dumps(out))


def main():
    if sys.argv[1] == '--worker':
        return worker(int(sys.argv[2]))
    libs = [os.path.abspath(p) for p in sys.argv[1:3]]
    reps = sys.argv[3] if len(sys.argv) > 3 else '20'
    res = {p: [] for p in libs}
    for _ in range(2):
        for p in libs:
            o = subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', reps], env=dict(os.environ, SYNTHSR_HIP_LIB=p),
                               capture_output=True, text=True, timeout=900)
            line = [ln for ln in o.stdout.splitlines() if ln.startswith('LIBAB ')]
            assert line, o.stdout[-2000:] + o.stderr[-2000:]
            res[p].append(json.loads(line[-1][6:]))
    names = [os.path.basename(p) for p in libs]
    print('# A = %s, B = %s; ms per launch (two passes each, alternating processes); md5 of the results A | B' % tuple(names))
    for layer in res[libs[0]][0]:
        for k in ('fwd', 'fwd+stats', "dgrad*elu'", 'wgrad'):
            a = [r[layer][k] for r in res[libs[0]]]
            b = [r[layer][k] for r in res[libs[1]]]
            print('%-16s %-12s A %.4f %.4f   B %.4f %.4f   B/A %.3f' % (layer, k, a[0], a[1], b[0], b[1], min(b) / min(a)))
        for k in ('fwd_md5', 'stats_md5', 'dgrad_md5', 'wgrad_md5'):
            a, b = res[libs[0]][0][layer][k], res[libs[1]][0][layer][k]
            print('%-16s %-12s %s | %s %s' % (layer, k, a, b, 'identical' if a == b else 'DIFFERENT'))
    for p in libs + libs:
        o = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--steps', '30', '--warmup', '5', '--no-cpu-baseline',
                            '--no-arith-compare'], env=dict(os.environ, SYNTHSR_HIP_LIB=p), capture_output=True, text=True, timeout=900)
        try:
            d = json.loads([ln for ln in o.stdout.splitlines() if ln.startswith('{')][-1])
            print('bench.py %-28s %.2f volumes/s  %.3f ms/step (median %.3f)' % (os.path.basename(p), d['value'], d['ms_per_step'],
                                                                                 d['step_ms']['median']))
        except Exception:  # noqa: BLE001
            print('bench.py failed under %s: %s' % (p, (o.stdout + o.stderr)[-800:]))


if __name__ == '__main__':
    main()
