#!/bin/bash
# Every measurement artefact of a round in one GPU call: bash tools/round_profiles.sh r05 [quick]
# (quick: without the fp32_mfma kernel trace, the fp32 Hyperfine / adversarial benches and the bf16 deterministic-mode bench)
# writes gpurun_out/<round>/...; copy what is to be judged into profiles/ afterwards (tools/collect_profiles.py).
set -x
export TMPDIR=/tmp
R=$(cd "$(dirname "$0")/.." && pwd)
T=${1:-rXX}
Q=${2:-full}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
# counter passes first: profiles/pmc_traffic.json (read by bench.py for roofline.traffic) is rebuilt from THIS code's kernels
cd /tmp
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-arith-compare > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-arith-compare > $O/pmc_w.log 2>&1
python $R/tools/pmc_traffic_json.py $O/pmc_f/p_counter_collection.csv $O/pmc_w/p_counter_collection.csv $T > $O/pmc_traffic.json && cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
rocprofv3 --kernel-trace --stats -d $O/kstats -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-arith-compare > $O/kstats.log 2>&1
[ $Q = quick ] || rocprofv3 --kernel-trace --stats -d $O/kstats_fp32_mfma -o p --output-format csv -- python $R/bench.py --conv-arith fp32_mfma --steps 20 --warmup 3 --no-cpu-baseline --no-arith-compare > $O/kstats_fp32_mfma.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_util -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-arith-compare > $O/pmc_util.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $O/pmc_lds -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-arith-compare > $O/pmc_lds.log 2>&1
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 1000 --warmup 10 --no-cpu-baseline --no-arith-compare --log-clocks > $O/bench_sustained_1000.json 2> $O/bench_sustained_1000.err
python bench.py --conv-arith fp32_mfma --steps 50 --warmup 5 --no-cpu-baseline --no-arith-compare > $O/bench_fp32_mfma.json 2> $O/bench_fp32_mfma.err
python tools/hyperfine_bench.py --dtype bf16 --config c1 --size 160 --steps 50 --warmup 5 > $O/bf16_c1_bench.json 2> $O/bf16_c1_bench.err
python tools/hyperfine_bench.py --dtype bf16 --steps 50 --warmup 5 > $O/bf16_hf_bench.json 2> $O/bf16_hf_bench.err
[ $Q = quick ] || python tools/hyperfine_bench.py --dtype f32 --steps 20 --warmup 3 > $O/f32_hf_bench.json 2> $O/f32_hf_bench.err
python tools/adversarial_bench.py --dtype bf16 --steps 10 > $O/adversarial_bf16.json 2> $O/adversarial_bf16.err
[ $Q = quick ] || python tools/adversarial_bench.py --dtype f32 --steps 5 > $O/adversarial_f32.json 2> $O/adversarial_f32.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-arith-compare --layer-table $O/layer_table.txt > $O/bench_layer_table.json 2> $O/bench_layer_table.err
python tools/conv_bf16_bench.py 160 > $O/conv_bf16_bench.txt 2>&1
python tools/split_check.py --acc --time > $O/split_check.txt 2>&1
python tools/predict_bench.py 160 > $O/predict_bench.txt 2>&1
python tools/det_bench.py --dtype f32 > $O/det_f32.txt 2>&1
[ $Q = quick ] || python tools/det_bench.py --dtype bf16 > $O/det_bf16.txt 2>&1
ls $O
