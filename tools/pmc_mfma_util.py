#!/usr/bin/env python
"""MFMA-pipe utilisation and effective clock per kernel from one rocprofv3 --pmc pass (csv) that collected
SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE (optionally SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY):

    python tools/pmc_mfma_util.py gpurun_out/pmc_util/*_counter_collection.csv [min_total_ms]

  clock    = GRBM_GUI_ACTIVE / 8 XCDs / dispatch wall time                    (the chip clocks to its power budget)
  mfma %   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) (share of the kernel's cycles the matrix pipes work)
  wait/issue/active = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY as shares of SQ_WAVE_CYCLES"""
import collections
import csv
import re
import sys

SIMDS = 256 * 4
XCDS = 8  # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (a light kernel then shows the 2.4 GHz maximum clock)
rows = collections.defaultdict(lambda: collections.defaultdict(list))
wall = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        name = re.sub(r'^void ', '', name)
        name = re.sub(r'\(.*', '', name)
        key = (name, r['Grid_Size'])
        rows[key][r['Counter_Name']].append((r['Dispatch_Id'], float(r['Counter_Value'])))
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            wall[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
out = []
for key, c in rows.items():
    if 'GRBM_GUI_ACTIVE' not in c or 'SQ_VALU_MFMA_BUSY_CYCLES' not in c:
        continue
    n = len(c['GRBM_GUI_ACTIVE'])
    gui = sum(v for _, v in c['GRBM_GUI_ACTIVE'])
    busy = sum(v for _, v in c['SQ_VALU_MFMA_BUSY_CYCLES'])
    us = sum(wall[key])
    if us * 1e-3 < min_ms or busy == 0:
        continue
    wave = sum(v for _, v in c.get('SQ_WAVE_CYCLES', [])) or float('nan')
    sh = [sum(v for _, v in c.get(k, [])) / wave * 100 if wave == wave else float('nan')
          for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY')]
    out.append((us, key[0], key[1], n, us / n, gui / XCDS / us * 1e-3, busy / (gui / XCDS * SIMDS) * 100, sh))
print('%-58s %9s %5s %9s %7s %7s %6s %6s %6s' % ('kernel', 'grid', 'calls', 'avg us', 'GHz', 'mfma %', 'wait%', 'issue%',
                                                   'activ%'))
for us, name, grid, n, avg, ghz, util, sh in sorted(out, reverse=True):
    print('%-58s %9s %5d %9.1f %7.2f %7.1f %6.1f %6.1f %6.1f' % (name[:58], grid, n, avg, ghz, util, sh[0], sh[1], sh[2]))
