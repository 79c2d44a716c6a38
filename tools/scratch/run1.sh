set -x
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dropout_gpu.py tests/test_generator_gpu.py -x -q > gpurun_out/t_dropout.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_dropout.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_default.json 2> gpurun_out/bench_err.log; echo "bench rc=$?"
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
timeout 400 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_f -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_f.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_w -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_w.log 2>&1
F=$(find gpurun_out/pmc_f -name '*counter_collection.csv' | head -1); W=$(find gpurun_out/pmc_w -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic_json.py $F $W > gpurun_out/pmc_traffic.json
python tools/pmc_summary.py $F > gpurun_out/r02_pmc_fetch_size.txt 2>/dev/null
python tools/pmc_summary.py $W > gpurun_out/r02_pmc_write_size.txt 2>/dev/null
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
tail -5 gpurun_out/t_dropout.log; cat gpurun_out/r02_bench_default.json
