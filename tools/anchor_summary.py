"""Summary of gpurun_out/parity_anchor.jsonl (written by tests/conftest.py: assert_grads_anchored during a `pytest -m gpu` run):
for every whole-network parity test and every parameter tensor, d = |device - float64| / range and o = |fp32 oracle - float64| /
range (deterministic run), a = |atomics run - deterministic run| / range.  Prints the distribution per kind of tensor and the
worst records -- the evidence behind GRAD_K / GRAD_FLOOR in tests/conftest.py.

    python tools/anchor_summary.py [gpurun_out/parity_anchor.jsonl] > profiles/r05_parity_anchor_distribution.txt"""
import json
import sys
import numpy as np

path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/parity_anchor.jsonl'
det, atm, tests = [], [], set()
for line in open(path):
    r = json.loads(line)
    t = r['test'].split('::')[-1].split(' ')[0]
    tests.add(t)
    for nm, d, o, a in r['rows']:
        kind = 'kernel' if nm.endswith('/kernel') else 'sum (bias / beta / gamma)'
        (atm if a is not None else det).append((kind, t, nm, a if a is not None else d, o))
print('# %s: %d whole-network parity tests, %d tensor records of the deterministic run, %d of the atomics run' % (path, len(tests), len(det), len(atm)))
q = [.5, .9, .99, 1.0]
for title, rows in (('deterministic run vs float64 (d)', det), ('atomics run vs deterministic run (a)', atm)):
    print('\n## %s; o = fp32 oracle vs float64 of the same tensor; quantiles 50 %% / 90 %% / 99 %% / max' % title)
    for kind in ('kernel', 'sum (bias / beta / gamma)'):
        v = np.array([r[3] for r in rows if r[0] == kind])
        o = np.array([r[4] for r in rows if r[0] == kind])
        print('%-26s n = %5d   device %s   fp32 oracle %s   device / max(oracle, 1e-6) %s' % (
            kind, len(v), ' '.join('%.2e' % x for x in np.quantile(v, q)), ' '.join('%.2e' % x for x in np.quantile(o, q)),
            ' '.join('%.2f' % x for x in np.quantile(v / np.maximum(o, 1e-6), q))))
    print('# the ten largest:')
    for r in sorted(rows, key=lambda r: -r[3])[:10]:
        print('  %.2e (oracle %.2e)  %s  %s' % (r[3], r[4], r[1], r[2]))
