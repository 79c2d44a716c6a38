cd /root/repo
export TMPDIR=/tmp
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
rm -rf gpurun_out/prof
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o p --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof.log 2>&1
S=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)
grep "pack_all\|adam_kernel\|fillBuffer\|FillFunctor" $S | cut -c1-60,100-220
rm -rf gpurun_out/prof
