cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_batch_gpu.py tests/test_dropout_gpu.py tests/test_unet_gpu.py tests/test_bf16_gpu.py -q -x > gpurun_out/t_batch.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_batch.log
tail -40 gpurun_out/t_batch.log
