cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/t_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_all.log
tail -15 gpurun_out/t_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
