#!/usr/bin/env python
"""Turn the scratch output of tools/round_profiles.sh (gpurun_out/<round>/) into the tracked summaries under profiles/:

    python tools/collect_profiles.py r03

  <round>_bench_default.json.log, _bench_sustained_1000.json.log (clock log summarised), _kernel_stats_final.txt
  (rocprofv3 --kernel-trace --stats), _pmc_mfma_util.txt, _pmc_fetch_size.txt / _pmc_write_size.txt + pmc_traffic.json,
  the bf16 / Hyperfine / adversarial bench lines, the deterministic-mode timings."""
import csv
import json
import os
import re
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = sys.argv[1]
O = os.path.join(R, 'gpurun_out', T)
P = os.path.join(R, 'profiles')


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name[:110]


def kernel_stats(src, dst, header):
    rows = list(csv.DictReader(open(src)))
    total = sum(float(r['TotalDurationNs']) for r in rows)
    with open(dst, 'w') as f:
        f.write(header + '\n# total kernel time %.3f ms over %d dispatches\n' % (total / 1e6, sum(int(r['Calls']) for r in rows)))
        f.write('%-112s %8s %12s %12s %12s %12s %7s\n' % ('kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', 'pct'))
        for r in rows:
            f.write('%-112s %8d %12.3f %12.2f %12.2f %12.2f %6.2f%%\n' % (
                short(r['Name']), int(r['Calls']), float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3,
                float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, 100.0 * float(r['TotalDurationNs']) / total))


def pmc_table(src, dst, header):
    agg = {}
    for r in csv.DictReader(open(src)):
        k = (short(r['Kernel_Name']).split('(')[0], r['Counter_Name'])
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += float(r['Counter_Value'])
        a[2] = max(a[2], float(r['Counter_Value']))
    with open(dst, 'w') as f:
        f.write(header + '\n%-70s %-14s %8s %16s %16s\n' % ('kernel', 'counter', 'calls', 'mean', 'max'))
        for (k, c), (n, s, m) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write('%-70s %-14s %8d %16.1f %16.1f\n' % (k[:70], c, n, s / n, m))


def copy_json_line(name, dst):
    src = os.path.join(O, name)
    if not os.path.exists(src):
        return None
    line = [l for l in open(src).read().splitlines() if l.startswith('{')]
    if not line:
        return None
    d = json.loads(line[-1])
    if 'clock_log' in d:   # keep a digest of the rocm-smi samples, not every sample
        cl = d.pop('clock_log')
        d['clock_log_digest'] = {'samples': len(cl), 'first': cl[0] if cl else None, 'middle': cl[len(cl) // 2] if cl else None,
                                 'last': cl[-1] if cl else None}
    with open(os.path.join(P, dst), 'w') as f:
        f.write(json.dumps(d) + '\n')
    return d


def main():
    os.makedirs(P, exist_ok=True)
    kernel_stats(os.path.join(O, 'kstats', 'p_kernel_stats.csv'), os.path.join(P, T + '_kernel_stats_final.txt'),
                 '# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-arith-compare (1x MI355X, round %s; '
                 '23 steps + set-up)' % T)
    if os.path.exists(os.path.join(O, 'kstats_fp32_mfma', 'p_kernel_stats.csv')):
        kernel_stats(os.path.join(O, 'kstats_fp32_mfma', 'p_kernel_stats.csv'), os.path.join(P, T + '_kernel_stats_fp32_mfma_same_box.txt'),
                     '# rocprofv3 --kernel-trace --stats -- python bench.py --conv-arith fp32_mfma --steps 20 --warmup 3 --no-cpu-baseline '
                     '(1x MI355X, round %s; 23 steps + set-up)' % T)
    if os.path.exists(os.path.join(O, 'pmc_lds', 'p_counter_collection.csv')):
        lds = subprocess.run([sys.executable, os.path.join(R, 'tools', 'pmc_raw.py'),
                              os.path.join(O, 'pmc_lds', 'p_counter_collection.csv'), 'conv3d'], capture_output=True, text=True).stdout
        with open(os.path.join(P, T + '_pmc_lds_valu.txt'), 'w') as f:
            f.write('# rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE -- python bench.py '
                    '--steps 3 --warmup 1 --no-cpu-baseline (tools/pmc_raw.py: per call and per CU-cycle; round %s)\n' % T + lds)
    out = subprocess.run([sys.executable, os.path.join(R, 'tools', 'pmc_mfma_util.py'),
                          os.path.join(O, 'pmc_util', 'p_counter_collection.csv')], capture_output=True, text=True).stdout
    with open(os.path.join(P, T + '_pmc_mfma_util.txt'), 'w') as f:
        f.write('# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE '
                '-- python bench.py --steps 3 --warmup 1 --no-cpu-baseline\n# (tools/pmc_mfma_util.py; 1x MI355X, round %s; GHz = '
                'GRBM_GUI_ACTIVE / 8 XCDs / dispatch time; mfma %% = matrix-pipe busy cycles / (cycles x 1024 SIMDs))\n' % T + out)
    pmc_table(os.path.join(O, 'pmc_f', 'p_counter_collection.csv'), os.path.join(P, T + '_pmc_fetch_size.txt'),
              '# rocprofv3 --pmc FETCH_SIZE -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline (KB per dispatch; round %s)' % T)
    pmc_table(os.path.join(O, 'pmc_w', 'p_counter_collection.csv'), os.path.join(P, T + '_pmc_write_size.txt'),
              '# rocprofv3 --pmc WRITE_SIZE -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline (KB per dispatch; round %s)' % T)
    tj = subprocess.run([sys.executable, os.path.join(R, 'tools', 'pmc_traffic_json.py'),
                         os.path.join(O, 'pmc_f', 'p_counter_collection.csv'),
                         os.path.join(O, 'pmc_w', 'p_counter_collection.csv'), T], capture_output=True, text=True)
    if tj.returncode == 0 and tj.stdout.strip().startswith('{'):
        with open(os.path.join(P, 'pmc_traffic.json'), 'w') as f:
            f.write(tj.stdout)
    else:
        print('pmc_traffic_json failed:', tj.stderr[-2000:])
    for src, dst in (('bench_default.json', T + '_bench_default.json.log'),
                     ('bench_sustained_1000.json', T + '_bench_sustained_1000.json.log'),
                     ('bench_fp32_mfma.json', T + '_bench_fp32_mfma_same_box.json.log'),
                     ('bf16_c1_bench.json', T + '_bf16_c1_bench.json.log'), ('bf16_hf_bench.json', T + '_bf16_hf_bench.json.log'),
                     ('f32_hf_bench.json', T + '_f32_hf_bench.json.log'),
                     ('adversarial_bf16.json', T + '_adversarial_bf16_bench.json.log'),
                     ('adversarial_f32.json', T + '_adversarial_f32_bench.json.log')):
        d = copy_json_line(src, dst)
        if d:
            print('%-40s %s %s  %s ms/step' % (dst, d.get('value'), d.get('unit'), d.get('ms_per_step')))
    for src, dst in (('conv_bf16_bench.txt', T + '_conv_bf16_bench.txt'), ('det_f32.txt', T + '_deterministic_mode_f32.txt'),
                     ('split_check.txt', T + '_split_vs_fp32_mfma_accuracy_and_time.txt'),
                     ('predict_bench.txt', T + '_predict_bench.txt'), ('layer_table.txt', T + '_layer_table.txt'),
                     ('split_fwd_variants.txt', T + '_split_fwd_variants.txt'), ('split_stacked24_ab.txt', T + '_split_stacked24_ab.txt'),
                     ('mfma_mix.txt', T + '_mfma_mix_ubench.txt'),
                     ('det_bf16.txt', T + '_deterministic_mode_bf16.txt')):
        if os.path.exists(os.path.join(O, src)):
            shutil.copy(os.path.join(O, src), os.path.join(P, dst))


if __name__ == '__main__':
    main()
