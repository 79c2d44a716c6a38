set -x
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_dropout_gpu.py tests/test_deterministic_gpu.py -q > gpurun_out/t_det.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_det.log
tail -30 gpurun_out/t_det.log
python - <<'PY' > gpurun_out/det_timing.txt 2>&1
import time, torch, numpy as np
from synthsr_amd import ops
from synthsr_amd.unet import unet
for dtype in ('f32', 'bf16'):
    for det in (False, True):
        ops.set_deterministic(det)
        net = unet(24, [160, 160, 160, 2], 5, 3, 1, feat_mult=2, nb_conv_per_level=2, final_pred_activation='linear', batch_norm=-1, seed=0, dtype=dtype)
        x = torch.rand(160, 160, 160, 2).cuda(); t = torch.rand(160 ** 3).cuda()
        for i in range(3):
            net.loss(x, t); net.backward(); net.adam_step()
        torch.cuda.synchronize(); t0 = time.time()
        for i in range(5):
            net.loss(x, t); net.backward(); net.adam_step()
        torch.cuda.synchronize()
        print(dtype, 'deterministic' if det else 'default', '%.2f ms per U-Net step (160^3)' % ((time.time() - t0) / 5 * 1e3), 'status', ops.deterministic_status())
        del net
ops.set_deterministic(False)
PY
cat gpurun_out/det_timing.txt
