set -x
cd /root/repo
mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_default.json 2> gpurun_out/bench_err.log; echo "bench rc=$?"
export TMPDIR=/tmp
rm -rf gpurun_out/prof gpurun_out/pmc_f gpurun_out/pmc_w
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o p --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof.log 2>&1
S=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)
python tools/rocprof_summary.py $S > gpurun_out/r02_kernel_stats_final.txt 2>/dev/null || cp $S gpurun_out/r02_kernel_stats_final.csv
timeout 400 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_f -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_f.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_w -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_w.log 2>&1
F=$(find gpurun_out/pmc_f -name '*counter_collection.csv' | head -1); W=$(find gpurun_out/pmc_w -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic_json.py $F $W > gpurun_out/pmc_traffic.json
python tools/pmc_summary.py $F > gpurun_out/r02_pmc_fetch_size.txt 2>/dev/null
python tools/pmc_summary.py $W > gpurun_out/r02_pmc_write_size.txt 2>/dev/null
timeout 300 python tools/hyperfine_bench.py --dtype bf16 --config c1 --size 160 --steps 20 --warmup 5 > gpurun_out/r02_bf16_c1_bench.json 2>/dev/null
timeout 300 python tools/hyperfine_bench.py --dtype bf16 --steps 10 --warmup 3 > gpurun_out/r02_bf16_hf_bench.json 2>/dev/null
timeout 300 python tools/hyperfine_bench.py --dtype f32 --steps 10 --warmup 3 > gpurun_out/r02_f32_hf_bench.json 2>/dev/null
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/prof
cat gpurun_out/r02_bench_default.json | cut -c1-400; cat gpurun_out/r02_bf16_c1_bench.json | cut -c1-200; cat gpurun_out/r02_bf16_hf_bench.json | cut -c1-200; cat gpurun_out/r02_f32_hf_bench.json | cut -c1-200
