"""Worker of tests/test_ddp_gpu.py: one rank of a 2-rank data-parallel Trainer.step.  On a box with at least as many GPUs as
ranks every rank takes its own GPU and the collectives run over RCCL (backend `nccl`) -- the production path; on the
one-GPU test box both ranks share cuda:0 and `gloo` carries the collectives (two processes on one device cannot form an
RCCL communicator).  The choice is automatic (`_init`) and recorded in the output files.  Saves, per rank, the single-rank gradient of ITS sample, then the
all-reduced gradient sum and the weights after the distributed step."""
import os
import sys
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def _init():
    """backend / device of this rank: RCCL with one GPU per rank whenever the box has enough GPUs, else gloo on cuda:0"""
    import torch
    import torch.distributed as dist
    world, local = int(os.environ['WORLD_SIZE']), int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.device_count() >= world and os.environ.get('SYNTHSR_TEST_BACKEND', '') != 'gloo':
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        return 'nccl'
    torch.cuda.set_device(0)
    dist.init_process_group('gloo')
    return 'gloo'


def adversarial(out_dir):
    """fine_tuning_with_adversary.training() data-parallel on 2 ranks (gloo, both on cuda:0)"""
    import torch
    import torch.distributed as dist
    from synthsr_amd.fine_tuning_with_adversary import training
    rank = int(os.environ['RANK'])
    backend = _init()
    gen, critic = training(os.path.join(out_dir, 'labels'), os.path.join(out_dir, 'images'), os.path.join(out_dir, 'models'),
                           None, None, os.path.join(out_dir, 'gl.npy'), output_shape=32, n_levels=3,
                           nonlin_shape_factor=.125, bias_shape_factor=.125, epochs=1, steps_per_epoch=2,
                           first_training_ratio=2, training_ratio=1, lr_generator=1e-3, lr_discriminator=1e-3,
                           verbose=False)
    np.savez(os.path.join(out_dir, 'adv_rank%d.npz' % rank), gen=gen.params.detach().cpu().numpy(),
             critic=critic.params.detach().cpu().numpy(), gen_iter=gen.iterations, critic_iter=critic.iterations,
             bn=gen.bn_moving.detach().cpu().numpy(), backend=backend)
    dist.barrier()
    dist.destroy_process_group()


def main():
    out_dir = sys.argv[1]
    if len(sys.argv) > 2 and sys.argv[2] == 'adversarial':
        return adversarial(out_dir)
    import torch
    import torch.distributed as dist
    from synthsr_amd.brain_generator import BrainGenerator
    from synthsr_amd.training import Trainer
    from synthsr_amd.unet import unet
    from synthsr_amd.synthetic import (synthetic_label_pool, GENERATION_LABELS, GENERATION_CLASSES, PRIOR_MEANS_T1_HR,
                                       PRIOR_STDS_T1_HR)
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    backend = _init()
    S = 32
    pool = synthetic_label_pool(2, (S, S, S), 5)
    bg = BrainGenerator(None, PRIOR_MEANS_T1_HR, PRIOR_STDS_T1_HR, 'normal', GENERATION_LABELS,
                        generation_classes=GENERATION_CLASSES, output_shape=S, output_div_by_n=8, nonlin_std=4.,
                        nonlin_shape_factor=.125, bias_shape_factor=.125, build_reliability_maps=True, downsample=True,
                        shearing_bounds=.02, label_maps=pool, rng=np.random.Generator(np.random.Philox(key=100 + rank)))
    bg.labels_to_image_model.seed(0, rank)                    # per-rank Philox stream of the in-kernel noise
    net = unet(24, bg.model_output_shape, 3, 3, 1, feat_mult=2, nb_conv_per_level=2, final_pred_activation='linear',
               batch_norm=-1, seed=1 + rank)                  # DIFFERENT initial weights: the broadcast must fix that
    dist.broadcast(net.params, 0)
    net.repack()
    inputs = next(bg.model_inputs_generator)
    draws = bg.labels_to_image_model.sample_draws()
    # (1) the gradient this rank computes alone on its own sample
    solo = Trainer(bg, net, lr=1e-3)
    gen = bg.labels_to_image_model
    image, target, _ = gen.generate(np.asarray(inputs[0])[0, ..., 0], np.asarray(inputs[1])[0], np.asarray(inputs[2])[0], draws)
    net.loss(image, target.reshape(-1), 'l1')
    net.backward()
    g_solo = net.grads.detach().cpu().numpy().copy()
    w0 = net.params.detach().cpu().numpy().copy()
    # (2) the distributed step on the same sample (tiny buckets: several all-reduces interleaved with the backward)
    tr = Trainer(bg, net, lr=1e-3, distributed=True, bucket_elems=50000)
    assert tr.reducer.world == world
    loss = tr.step(inputs, draws)
    n_buckets = getattr(tr.reducer, 'n_launched', -1)
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), g_solo=g_solo, g_sum=net.grads.detach().cpu().numpy(), w0=w0,
             w1=net.params.detach().cpu().numpy(), loss=float(loss.item()), n_buckets=n_buckets,
             adam_m=net.adam_m.detach().cpu().numpy(), backend=backend)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
