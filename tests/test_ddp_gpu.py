"""Two-rank data-parallel training step: `Trainer.step` on 2 ranks yields the SUM of the two
single-rank gradients in the flat buffer (the optimizer applies 1/world), identical gradients on both ranks, and
bit-identical weights afterwards -- starting from deliberately different initial weights that the rank-0 broadcast
must override.  The hook offsets that `unet._backward_body` hands to GradBucketReducer are exercised with several
buckets in flight.  On a box with >= 2 GPUs each rank takes its own GPU and the collectives run over RCCL; on the one-GPU
test box both ranks share cuda:0 and gloo carries them (tests/_ddp_worker.py: _init picks automatically; the backend used
is printed).  RCCL at world size 1: test_training_gpu.py::test_bench_under_torchrun_with_forced_allreduce."""
import os
import subprocess
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_trainer_step_averages_gradients_and_keeps_weights_identical(tmp_path):
    port = 29700 + (os.getpid() % 200)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(REPO, 'tests', '_ddp_worker.py'), str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    a, b = np.load(tmp_path / 'rank0.npz'), np.load(tmp_path / 'rank1.npz')
    print('collectives over', str(a['backend']))
    np.testing.assert_array_equal(a['w0'], b['w0'])                     # broadcast from rank 0
    np.testing.assert_array_equal(a['g_sum'], b['g_sum'])               # same reduced gradients everywhere
    np.testing.assert_array_equal(a['w1'], b['w1'])                     # ... hence bit-identical weights after Adam
    np.testing.assert_array_equal(a['adam_m'], b['adam_m'])
    assert not np.array_equal(a['w1'], a['w0'])
    assert abs(a['loss'] - b['loss']) > 1e-6                            # the ranks really trained on different samples
    # all-reduced buffer = sum of the two single-rank gradients (float atomics: the re-computed gradients differ in
    # the last bits)
    expect = a['g_solo'].astype(np.float64) + b['g_solo'].astype(np.float64)
    scale = np.abs(expect).max()
    assert np.abs(a["g_sum"] - expect).max() < 5e-4 * scale      # measured 1.2e-4
    # first moment of Adam after one step = (1 - beta1) * mean gradient
    np.testing.assert_allclose(a['adam_m'], 0.1 * 0.5 * a['g_sum'], rtol=1e-5, atol=1e-7 * scale)
    assert int(a['n_buckets']) >= 3


def test_two_rank_adversarial_fine_tuning(tmp_path):
    """fine_tuning_with_adversary.training() data-parallel: both networks stay identical across the
    ranks through critic and generator updates (separate reducers; the critic's flat gradient buffer goes out in buckets),
    the ranks train on different samples, rank 0 alone writes logs and checkpoints"""
    from synthsr_amd.nifti import write_nifti
    from synthsr_amd.synthetic import GENERATION_LABELS, synthetic_label_map
    shape = (40, 36, 48)
    (tmp_path / 'labels').mkdir()
    (tmp_path / 'images').mkdir()
    rng = np.random.RandomState(0)
    lut = rng.uniform(30, 220, 64)
    for i in range(2):
        lab = synthetic_label_map(shape, 10 + i)
        write_nifti(str(tmp_path / 'labels' / ('brain%d_labels.nii.gz' % i)), lab.astype(np.float32))
        write_nifti(str(tmp_path / 'images' / ('brain%d.nii.gz' % i)), (lut[lab % 64] + rng.randn(*lab.shape)).astype(np.float32))
    np.save(tmp_path / 'gl.npy', GENERATION_LABELS)
    port = 29400 + (os.getpid() % 200)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(REPO, 'tests', '_ddp_worker.py'), str(tmp_path),
           'adversarial']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    a, b = np.load(tmp_path / 'adv_rank0.npz'), np.load(tmp_path / 'adv_rank1.npz')
    np.testing.assert_array_equal(a['gen'], b['gen'])
    np.testing.assert_array_equal(a['critic'], b['critic'])
    assert int(a['gen_iter']) == 2 and int(a['critic_iter']) == 3         # 2 generator updates, 2 + 1 critic updates
    assert not np.array_equal(a['bn'], b['bn'])                           # per-replica BatchNorm statistics: own samples
    files = sorted(os.listdir(tmp_path / 'models'))
    assert 'generator_1.h5' in files and 'discriminator_1.h5' in files
    assert np.load(tmp_path / 'models' / 'logs' / 'generator_loss.npy').shape == (1,)
