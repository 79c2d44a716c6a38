#!/usr/bin/env python
"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes of bench.py (FETCH_SIZE, WRITE_SIZE; csv):

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_f -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_w -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic_json.py gpurun_out/pmc_f/p_counter_collection.csv gpurun_out/pmc_w/p_counter_collection.csv > profiles/pmc_traffic.json

bench.py reads `roofline.traffic` of the dominant kernel from this file.  Counter units are KB per dispatch; on gfx950
FETCH_SIZE reports half the bytes of wide (16 B / lane) coalesced reads (MI355X_MICROARCH.md, HBM): the conv kernels stage
their tiles with 16-byte loads, so their fetch is doubled; the generator kernels read 4 bytes per lane (uncalibrated width):
both the raw and the doubled figure are recorded and the raw one is used."""
import collections
import csv
import json
import re
import sys


def per_kernel(path):
    rows = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
            name = re.sub(r'^void ', '', name)
            name = re.sub(r'\(.*', '', name)
            rows[name].append(float(r['Counter_Value']))
    return rows


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
out = {'_note': __doc__.split('bench.py reads')[1].strip().replace('\n', ' '),
       '_source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 3 --warmup 1 --no-cpu-baseline`, round %s'
                  % (sys.argv[3] if len(sys.argv) > 3 else '?')}
# the 160^3 24 -> 24 layers are the LARGEST dispatches of their kernels (the same kernels also run smaller layers)
# (split arithmetic, the default: conv_split.hip kernels; fp32_mfma: the 4x4x1-MFMA kernels of conv3d.hip)
# (kernel symbols are matched by prefix: the template argument lists grew -- <24> became <24, 6>: COW, partial products)
# (round 4: forward / data gradient of the Cout = 24 layers = conv3d_split_fwd2_kernel<2, ST, EPI, true>: EPI 1 / 4 forward,
#  EPI 2 data gradient)
for key, kerns, wide in (('conv3d_wgrad 160x160x160 Cin=24 Cout=24', ('conv3d_split_wgrad_kernel<24', 'conv3d_wgrad_p4_kernel'), True),
                         # (round 6: <MT, ST, EPI, STK, KS> -- the stacked (STK = true) kernels are the 160^3 ones; MT = 2 without STK is now
                         #  the re-planned 40^3 layer)
                         ('conv3d_fwd 160x160x160 Cin=24 Cout=24', ('conv3d_split_fwd2_kernel<2, true, 1, true', 'conv3d_split_fwd2_kernel<2, false, 1, true',
                                                                    'conv3d_split_fwd_kernel<2, false', 'conv3d_fwd_p4_kernel'), True),
                         ('conv3d_dgrad 160x160x160 Cin=24 Cout=24', ('conv3d_split_fwd2_kernel<2, false, 2, true', 'conv3d_split_fwd_kernel<2, false',
                                                                      'conv3d_fwd_p4_kernel'), True)):
    kern = next((k for pre in kerns for k in sorted(fetch) if k.startswith(pre) and k in write), None)
    if kern is None:
        continue
    f, w = max(fetch[kern]), max(write[kern])
    out[key] = {'kernel': kern, 'fetch_kb': round(f, 1), 'write_kb': round(w, 1), 'bytes': int((2 if wide else 1) * f * 1024 + w * 1024)}
nvox = 160 ** 3
for kern, alg in (('deform_gmm_kernel', 9), ('normalise_blur2_kernel', 16), ('normalise_gamma_kernel', 8), ('blur3d_kernel', None),
                  ('copy_strided_kernel', 8), ('svf_step_kernel', None), ('resize_kernel', None)):
    if kern not in fetch and kern not in write:
        continue
    f = sum(fetch.get(kern, [0])) / max(len(fetch.get(kern, [0])), 1)
    w = sum(write.get(kern, [0])) / max(len(write.get(kern, [0])), 1)
    e = {'kernel': kern, 'launches_per_volume': len(fetch.get(kern, [])) // 4, 'fetch_kb': round(f, 1), 'write_kb': round(w, 1),
         'bytes_raw': int((f + w) * 1024), 'bytes_if_wide_reads': int((2 * f + w) * 1024),
         'bytes_per_voxel_raw': round((f + w) * 1024 / nvox, 1)}
    if alg:
        e['algorithmic_bytes_per_voxel'] = alg
    out['generator ' + kern] = e
gen = [v for k, v in out.items() if k.startswith('generator ')]
out['generator total'] = {'bytes_raw_per_volume': int(sum(v['bytes_raw'] * max(v['launches_per_volume'], 1) for v in gen)),
                          'compulsory_bytes': 13 * nvox, 'two_pass_floor_bytes': 21 * nvox,
                          'compulsory_note': 'labels uint8 in (resident pool, round 4) + image, reliability map and target float32 out'}
out['generator total']['bytes_per_voxel_raw'] = round(out['generator total']['bytes_raw_per_volume'] / nvox, 1)
print(json.dumps(out, indent=1))
