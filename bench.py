#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): training volumes/sec for 160^3 fp32 volumes —
on-the-fly synthetic brain generator + 5-level 3-D U-Net forward/backward + Keras-Adam, batch 1 per GPU,
pure data parallel over N GPUs (weak scaling).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (largest total time inside the timed
region, measured with HIP events on the launch stream); `cpu_baseline` times the oracle (numpy generator +
PyTorch-CPU U-Net, "port" — NOT TensorFlow) on a bounded sample on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
# the host driver only supports dmabuf IPC: without this RCCL's multi-process setup fails (hipIpcGetMemHandle)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense, exact f32
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_16x16x32_bf16, dense
HBM_PEAK_GBS = 8000.0


def physical_cores():
    """sockets x cores per socket from lscpu (SURVEY 8d asks for the physical count, not the thread count)"""
    import subprocess
    try:
        txt = subprocess.run(['lscpu'], capture_output=True, text=True, timeout=10).stdout
        kv = {l.split(':')[0].strip(): l.split(':', 1)[1].strip() for l in txt.splitlines() if ':' in l}
        return int(kv['Socket(s)']) * int(kv['Core(s) per socket'])
    except Exception:
        return os.cpu_count()


def usable_cpus():
    """CPUs this process may actually run on: the affinity mask, capped by a cgroup v2 / v1 CPU quota if one is set"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def _oracle_step_factory(size, seed=0):
    """one oracle training step (numpy generator + PyTorch-CPU U-Net fwd / bwd / Keras-Adam with live moments) on a
    size^3 volume of the benchmark configuration; returns a callable -> seconds"""
    import torch
    from oracle import generator_ref as R
    from oracle import unet_ref as U
    from synthsr_amd.synthetic import (synthetic_label_map, GENERATION_LABELS, GENERATION_CLASSES, PRIOR_MEANS_T1_HR,
                                       PRIOR_STDS_T1_HR)
    rng = np.random.default_rng(seed)
    labels = synthetic_label_map((size,) * 3, 1)
    small = R.get_resample_shape([size] * 3, .03125)
    u = lambda *s: rng.random(s, dtype=np.float32)
    n = lambda *s: rng.standard_normal(s, dtype=np.float32)
    # random-init U-Net with the benchmark architecture
    g = torch.Generator().manual_seed(0)
    P = {}
    feats = [24, 48, 96, 192, 384]

    def conv(name, ci, co):
        lim = float(np.sqrt(6.0 / (27 * ci + 27 * co)))
        P[name + '/kernel'] = ((torch.rand(3, 3, 3, ci, co, generator=g) * 2 - 1) * lim).requires_grad_(True)
        P[name + '/bias'] = torch.zeros(co, requires_grad=True)

    def bn(name, c):
        P[name + '/gamma'] = torch.ones(c, requires_grad=True)
        P[name + '/beta'] = torch.zeros(c, requires_grad=True)
    c = 2
    for l in range(5):
        for k in range(2):
            conv('unet_conv_downarm_%d_%d' % (l, k), c, feats[l])
            c = feats[l]
        bn('unet_bn_down_%d' % l, c)
    for k in range(4):
        l = 3 - k
        ci = feats[l] + c
        for j in range(2):
            conv('unet_conv_uparm_%d_%d' % (5 + k, j), ci, feats[l])
            ci = feats[l]
        c = feats[l]
        bn('unet_bn_up_%d' % k, c)
    P['unet_likelihood/kernel'] = ((torch.rand(24, 1, generator=g) * 2 - 1) * .4).requires_grad_(True)
    P['unet_likelihood/bias'] = torch.zeros(1, requires_grad=True)
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    state = {'t': 0}

    def step():
        t0 = time.time()
        means = np.clip(rng.normal(PRIOR_MEANS_T1_HR[0], PRIOR_MEANS_T1_HR[1]), 0, None)[GENERATION_CLASSES][:, None]
        stds = np.clip(rng.normal(PRIOR_STDS_T1_HR[0], PRIOR_STDS_T1_HR[1]), 0, None)[GENERATION_CLASSES][:, None]
        tape = [('u', u(3)), ('u', u(6)), ('u', u(3)), ('u', u(3)), ('u', u(1)), ('n', n(*small, 3)), ('u', u(1)),
                ('n', n(size, size, size, 1)), ('u', u(1)), ('n', n(*small)), ('u', u(1)), ('n', n(1)), ('u', u(3))]
        out = R.labels_to_image(labels, means, stds, tape, GENERATION_LABELS, 19, input_channels=[True],
                                output_channel=[0], output_shape=size, output_div_by_n=32, scaling_bounds=.15,
                                rotation_bounds=15, shearing_bounds=.02, translation_bounds=5, nonlin_std=4.,
                                nonlin_shape_factor=.03125, downsample=True, build_reliability_maps=True, blur_range=1.15,
                                bias_field_std=.3, bias_shape_factor=.03125)
        t_gen = time.time() - t0
        for p in P.values():
            p.grad = None
        pred = U.unet_forward(torch.from_numpy(out['image']), P, 'unet', 5, 2, training=True)
        loss = U.l1_loss(pred, torch.from_numpy(out['target']))
        loss.backward()
        state['t'] += 1
        with torch.no_grad():
            for k, p in P.items():
                pn, M[k], V[k] = U.adam_keras(p, p.grad, M[k], V[k], state['t'])
                p.copy_(pn)
        return time.time() - t0, t_gen
    return step


def cpu_baseline(size_all=160, size_one=64, timed=3):
    """the oracle (CPU restatement of the identical graph: numpy generator + PyTorch-CPU fp32 U-Net fwd / bwd /
    Adam; NOT TensorFlow) timed on the host cores: 1 warm-up + `timed` steps with all threads at size_all^3 and with ONE
    thread at size_one^3 (scaled by voxel count to 160^3 volumes/s)."""
    import torch
    cores = physical_cores()
    nthreads_default = torch.get_num_threads()
    usable = usable_cpus()
    res = {}
    for tag, size, threads in (('all', size_all, min(nthreads_default, usable)), ('one', size_one, 1)):
        if threads == 1:
            # OpenMP's thread count is a per-thread setting and autograd runs the backward on its own thread: the only
            # reliable way to pin the whole step to ONE core's worth of threads is a fresh process with OMP_NUM_THREADS=1
            import subprocess
            env = dict(os.environ, OMP_NUM_THREADS='1', MKL_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1')
            out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-worker', str(size), str(timed)],
                                 capture_output=True, text=True, env=env, timeout=1800)
            ts = json.loads([l for l in out.stdout.splitlines() if l.startswith('[')][-1])
        else:
            torch.set_num_threads(threads)
            step = _oracle_step_factory(size)
            step()                                               # warm-up (oneDNN primitive creation, page faults)
            ts = [step() for _ in range(timed)]
        scale = (size / 160.0) ** 3
        res[tag] = dict(volumes_per_s=round(scale / float(np.mean([t[0] for t in ts])), 5), threads=threads, size=size,
                        step_s=[round(t[0], 2) for t in ts], generator_s=round(float(np.mean([t[1] for t in ts])), 2))
    torch.set_num_threads(nthreads_default)
    par = [l.strip() for l in torch.__config__.parallel_info().splitlines() if l.strip() and ('threads' in l.lower() or
           'openmp' in l.lower() or 'mkl' in l.lower() or 'ATen parallel backend' in l)]
    aff = sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else []
    runs, i = [], 0                      # "0-15,32-47": the mask as ranges
    while i < len(aff):
        j = i
        while j + 1 < len(aff) and aff[j + 1] == aff[j] + 1:
            j += 1
        runs.append('%d' % aff[i] if i == j else '%d-%d' % (aff[i], aff[j]))
        i = j + 1
    aff = ','.join(runs)
    return {'value': res['all']['volumes_per_s'], 'unit': 'volumes/s', 'cores': res['all']['threads'], 'kind': 'port',
            'threads': res['all']['threads'], 'usable_cpus': usable, 'physical_cores': cores, 'value_one_thread': res['one']['volumes_per_s'],
            'affinity_mask': aff, 'torch_parallel_info': par,
            'omp_env': {k: os.environ.get(k) for k in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'GOMP_CPU_AFFINITY')},
            'sample': 'oracle (numpy generator + PyTorch-CPU U-Net fwd/bwd/Keras-Adam; CPU restatement, NOT TensorFlow): '
                      '1 warm-up + %d timed steps. all usable CPUs (%d threads; the host has %d physical cores, affinity / cgroup quota caps this process): %d^3 volumes, %s s per step '
                      '(generator %.2f s of it). one thread: %d^3 volumes (%.1f%% of the voxels of 160^3, volumes/s scaled '
                      'by voxel count), %s s per step'
                      % (timed, res['all']['threads'], cores, size_all, res['all']['step_s'], res['all']['generator_s'],
                         size_one, 100 * (size_one / 160.0) ** 3, res['one']['step_s'])}


def conv_flops(kind, shape, cin, cout):
    """flops a conv launch EXECUTES.  Folded decoder convs ('conv3d_up_*': recorded with the low-resolution shape) run 8 parity
    convs of 2x2x2 taps per low-resolution voxel = 64 tap products (the 3x3x3 conv on the up-sampled tensor they replace would
    be 8 x 27 = 216: the 3.4x saving of the folding is not counted as throughput)"""
    taps = 64 if kind.startswith('conv3d_up_') else 27
    return 2.0 * taps * cin * cout * float(np.prod(shape))


def cpu_worker(size, timed):
    """one-thread leg of cpu_baseline (run with OMP_NUM_THREADS=1): prints [[step_s, generator_s], ...]"""
    import torch
    torch.set_num_threads(1)
    step = _oracle_step_factory(size)
    step()
    print(json.dumps([list(step()) for _ in range(timed)]))


CONV_ARITH_NOTE = {
    'split': 'fp32 tensors, weights, accumulation, BatchNorm statistics and optimizer throughout; inside the 3x3x3 convs of the '
             'levels with >= 256 tiles every fp32 operand is split EXACTLY into three bf16 numbers and a product is accumulated as '
             'six exact partial products on v_mfma_f32_16x16x32_bf16 (omitted terms <= 2^-24 |a b|, 2^-27 rms: within the rounding of an fp32 multiply-add): as accurate against a float64 '
             'convolution as the fp32 matrix instructions (tests/test_split_gpu.py); other layers on fp32 MFMA',
    'split9': 'as split, with all NINE partial products: every fp32 product is reproduced exactly (no term dropped), 1.5x the '
              'matrix instructions of split',
    'fp32_mfma': 'every convolution on the fp32 matrix instructions (v_mfma_f32_4x4x1 / 16x16x4)'}


def self_launch(n):
    """re-executes this command line as n ranks of one node under torch.distributed.run (127.0.0.1, a free port); the ranks'
    output is passed through, the exit code is theirs"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC (RCCL between processes on this driver)
    return subprocess.call(cmd, env=env)


def rccl_version(backend):
    """RCCL's version string when the collectives run over it (backend 'nccl' IS RCCL on ROCm), else None"""
    if backend != 'nccl':
        return None
    try:
        import torch
        return '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception as ex:  # noqa: BLE001 -- a missing version query must not lose the measurement
        return 'unknown (%r)' % (ex,)


N_PARAMS_C1 = 13240489   # trainable parameters of the configs[1] network = floats of the flat gradient buffer


def rendezvous_only(args, world, rank):
    """--rendezvous-only: the ranks meet over gloo (CPU), count themselves with an all-reduce and push a buffer of the flat
    gradient's size through the product's GradBucketReducer (same bucket size as the training step, readiness reported tail
    first); rank 0 prints the record -- n_ranks_seen, the bytes and buckets one step's all-reduce consists of, and whether
    every rank ended with the same sum"""
    import torch
    import torch.distributed as dist
    from synthsr_amd.training import GradBucketReducer
    if world > 1:
        dist.init_process_group('gloo')
    seen = torch.ones(1)
    rec = {}
    if world > 1:
        dist.all_reduce(seen)
        g = torch.full((N_PARAMS_C1,), float(rank + 1))
        red = GradBucketReducer(g)
        red.start()
        for lo in range(N_PARAMS_C1 - 700001, 0, -1300003):   # "layer boundaries": the reducer cuts buckets of >= its size
            red.ready(lo)
        scale = red.finish()
        ok = torch.tensor([float(bool((g == world * (world + 1) / 2).all()) and abs(scale * world - 1.0) < 1e-12)])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        covered = sorted(red.ranges)
        rec = {'allreduce_bytes_per_step': int(red.bytes_launched), 'allreduce_buckets_per_step': int(red.n_launched),
               'allreduce_covers_buffer_once': bool(covered[0][0] == 0 and covered[-1][1] == N_PARAMS_C1 and
                                                    all(a[1] == b[0] for a, b in zip(covered, covered[1:]))),
               'allreduce_tail_first': bool(all(a[0] == b[1] for a, b in zip(red.ranges, red.ranges[1:]))),
               'allreduce_identical_on_every_rank': bool(ok.item() == 1.0), 'gradient_scale': scale}
    if rank == 0:
        print(json.dumps(dict({'metric': 'rendezvous only', 'n_gpus': args.gpus, 'n_ranks_seen': int(seen.item()),
                               'world_size': dist.get_world_size() if world > 1 else 1,
                               'backend': 'gloo' if world > 1 else None, 'rccl_version': rccl_version('gloo')}, **rec)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--cpu-worker':
        return cpu_worker(int(sys.argv[2]), int(sys.argv[3]))
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--size', type=int, default=160)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-size', type=int, default=160, help='volume size of the all-threads CPU baseline')
    ap.add_argument('--overlap', action='store_true', help='weight gradients on a second stream (A/B switch; slower)')
    ap.add_argument('--fold', default='auto', help="nearest-upsample folding of the decoder convs: auto | all | none")
    ap.add_argument('--log-clocks', action='store_true',
                    help='sample rocm-smi clocks / power every 2 s during the timed region (sustained runs: --steps 1000)')
    ap.add_argument('--no-fuse-pool-bwd', action='store_true', help='A/B switch: separate max-pool / BatchNorm+ELU backward kernels')
    ap.add_argument('--no-fuse-head-bwd', action='store_true', help='A/B switch: separate head backward pass')
    ap.add_argument('--conv-arith', default='split', choices=['split', 'split9', 'fp32_mfma'],
                    help='arithmetic of the fp32 convolutions (ops.set_conv_arithmetic): split = three bf16 pieces per fp32 '
                         'operand, six exact partial products on the bf16 matrix cores, fp32 accumulation; fp32_mfma = fp32 '
                         'matrix instructions everywhere')
    ap.add_argument('--layer-table', default='', help='write the per-layer conv timings of the profiled steps to this file')
    ap.add_argument('--no-arith-compare', action='store_true',
                    help='skip the short second measurement under the other conv arithmetic (N = 1 only, 20 steps)')
    ap.add_argument('--force-allreduce', action='store_true',
                    help='initialise RCCL and run the bucketed gradient all-reduce even at world size 1 (path test)')
    ap.add_argument('--rendezvous-only', action='store_true',
                    help='launch / rendezvous path only: N ranks meet, all-reduce a counter, rank 0 prints n_gpus and '
                         'n_ranks_seen (no device work: what the CPU test of the self-launch runs)')
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error('--gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` launched plainly: start the N ranks ourselves (one process per GPU, the way the driver's
        # torch.distributed.run command line does) instead of silently measuring one GPU under an N-GPU command
        return self_launch(args.gpus)

    import torch
    import torch.distributed as dist
    from synthsr_amd import ops
    from synthsr_amd.brain_generator import BrainGenerator
    from synthsr_amd.training import Trainer
    from synthsr_amd.unet import unet
    from synthsr_amd.synthetic import (synthetic_label_pool, GENERATION_LABELS, GENERATION_CLASSES, PRIOR_MEANS_T1_HR,
                                       PRIOR_STDS_T1_HR)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d runs under WORLD_SIZE=%d: launch it plainly (it starts its own ranks) or with '
                         'torch.distributed.run --nproc-per-node %d' % (args.gpus, world, args.gpus))
    if args.rendezvous_only:
        return rendezvous_only(args, world, rank)
    # one GPU per rank over RCCL; a box with fewer GPUs than ranks (the one-GPU test box) shares its devices and lets gloo
    # carry the collectives (two processes on one device cannot form an RCCL communicator) -- recorded as `backend`
    ndev = torch.cuda.device_count()
    backend = 'nccl' if ndev >= world else 'gloo'
    torch.cuda.set_device(local % max(1, ndev))
    if world > 1 or args.force_allreduce:
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group('gloo')
        if dist.get_world_size() != args.gpus:
            raise SystemExit('rendezvous gave %d ranks for --gpus %d' % (dist.get_world_size(), args.gpus))

    ops.set_conv_arithmetic(args.conv_arith)
    S = args.size
    pool = synthetic_label_pool(8, (S, S, S), 1234)
    rng = np.random.Generator(np.random.Philox(key=1000 + rank))
    bg = BrainGenerator(None, PRIOR_MEANS_T1_HR, PRIOR_STDS_T1_HR, 'normal', GENERATION_LABELS,
                        generation_classes=GENERATION_CLASSES, n_neutral_labels=19, output_shape=S, output_div_by_n=32,
                        flipping=True, scaling_bounds=.15, rotation_bounds=15, shearing_bounds=.02, translation_bounds=5,
                        nonlin_std=4., nonlin_shape_factor=.03125, randomise_res=False, downsample=True,
                        blur_range=1.15, build_reliability_maps=True, bias_field_std=.3, bias_shape_factor=.03125,
                        label_maps=pool, rng=rng)  # = training() defaults, SynthSR/training.py:57-73
    bg.labels_to_image_model.seed(0, rank)
    net = unet(24, bg.model_output_shape, 5, 3, 1, feat_mult=2, nb_conv_per_level=2, final_pred_activation='linear',
               batch_norm=-1, activation='elu', seed=0,
               fold_upsample={'auto': 'auto', 'all': True, 'none': False}[args.fold])
    net.overlap_wgrad = args.overlap
    net.fuse_pool_bwd = not args.no_fuse_pool_bwd
    if world > 1:
        dist.broadcast(net.params, 0)
        net.repack()
    tr = Trainer(bg, net, lr=1e-4, distributed=world > 1 or args.force_allreduce, force_allreduce=args.force_allreduce)
    tr.make_labels_resident(pool)
    tr.fuse_head_bwd = not args.no_fuse_head_bwd
    pick = np.random.default_rng(rank)

    def one_step():
        return tr.step(label_index=int(pick.integers(len(pool))))

    for _ in range(args.warmup):
        one_step()
    if tr.reducer is not None:
        tr.comm_events = []
    clocks, stop_clocks = [], None
    if args.log_clocks and rank == 0:
        import subprocess
        import threading
        stop_clocks = threading.Event()

        def sample():
            while not stop_clocks.is_set():
                try:
                    txt = subprocess.run(['rocm-smi', '-d', str(local), '-c', '-P', '--json'], capture_output=True, text=True,
                                         timeout=10).stdout
                    clocks.append((round(time.perf_counter(), 2), json.loads(txt[txt.index('{'):])))
                except Exception as ex:  # noqa: BLE001 -- the log is a diagnostic
                    clocks.append((round(time.perf_counter(), 2), repr(ex)))
                stop_clocks.wait(2.0)
        threading.Thread(target=sample, daemon=True).start()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # per-launch HIP events on the LAST 3 timed steps only (0.5 ms of host / event overhead each): by then the host runs ahead of
    # the GPU, so a region's time is what its kernels cost in a full queue, not the host's launch cadence right after a sync
    nprof = min(3, args.steps)
    # one HIP event per step boundary on the launch stream (= torch's current stream, which every kernel of the step is
    # launched on): per-step device times without a host sync inside the timed region
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    loss = None
    marks[0].record()
    for i in range(args.steps):
        if i == args.steps - nprof:
            ops.profile_start()
        loss = one_step()
        marks[i + 1].record()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = ops.profile_stop()
    if stop_clocks is not None:
        stop_clocks.set()
    step_ms = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)])
    comm_ms = np.array([a.elapsed_time(b) for a, b in (tr.comm_events or [])]) if tr.comm_events else np.zeros(0)
    dt_rank = dt
    per_rank = None
    if world > 1:
        tmax = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # every rank's own clock: wall time of the K steps, mean / median step (HIP events) and the time its compute
        # stream spent waiting for the gradient all-reduce at the end of each backward -- what a scaling loss is made of
        mine = torch.tensor([dt_rank * 1e3 / args.steps, float(np.mean(step_ms)), float(np.median(step_ms)),
                             float(np.mean(comm_ms)) if comm_ms.size else 0.0,
                             float(np.max(comm_ms)) if comm_ms.size else 0.0], device='cuda', dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [dict(rank=r, wall_ms_per_step=round(float(v[0]), 3), step_ms_mean=round(float(v[1]), 3),
                         step_ms_median=round(float(v[2]), 3), allreduce_wait_ms_mean=round(float(v[3]), 3),
                         allreduce_wait_ms_max=round(float(v[4]), 3)) for r, v in enumerate(allr)]
    final_loss = float(loss.item())
    # the same steps under the OTHER conv arithmetic, same process / box / clocks (N = 1): what the split arithmetic buys
    other = None
    if world == 1 and not args.no_arith_compare:
        other_name = 'fp32_mfma' if args.conv_arith != 'fp32_mfma' else 'split'
        ops.set_conv_arithmetic(other_name)
        net.repack()   # re-plans and re-packs every weight set for the new arithmetic
        for _ in range(3):
            one_step()
        k2 = min(20, args.steps)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(k2 + 1)]
        ev[0].record()
        for i in range(k2):
            one_step()
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms2 = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(k2)])
        other = {'conv_arithmetic': other_name, 'steps': k2, 'ms_per_step_median': round(float(np.median(ms2)), 3),
                 'value_median_step': round(1e3 / float(np.median(ms2)), 4)}
        ops.set_conv_arithmetic(args.conv_arith)
    if not np.isfinite(final_loss):  # tf.debugging.check_numerics of the reference's IdentityLoss
        raise FloatingPointError('non-finite loss after %d steps: the measurement is invalid' % args.steps)

    if rank == 0:
        # per-kernel aggregation of the conv launches recorded inside the timed region
        agg, gen_agg = {}, {}
        for kind, shape, cin, cout, s, e in prof:
            key = (kind, shape, cin, cout)
            (gen_agg if kind.startswith('gen') else agg).setdefault(key, []).append(s.elapsed_time(e))
        rows = []
        for (kind, shape, cin, cout), samples in agg.items():
            fl = conv_flops(kind, shape, cin, cout)
            ms, cnt = float(sum(samples)), len(samples)
            # `rank_ms` orders the kernels by median x launches: a single stalled launch (first touch under a profiler)
            # must not promote a tiny kernel to "dominant"; the reported figures stay plain averages
            rows.append(dict(kernel=kind, shape=list(shape), cin=cin, cout=cout, launches=cnt,
                             avg_ms=ms / cnt, tflops=fl / (ms / cnt * 1e-3) / 1e12, total_ms=ms,
                             rank_ms=float(np.median(samples)) * cnt))
        rows.sort(key=lambda r: -r['rank_ms'])
        conv_total = sum(r['total_ms'] for r in rows)
        dom = rows[0]
        for r in rows[:6]:   # `top_kernels` of the line: each against ITS matrix peak (split: dense bf16 / partial products)
            sp = args.conv_arith != 'fp32_mfma' and ops.conv_runs_split(r['kernel'], tuple(r['shape']), r['cin'], r['cout'])
            pk = BF16_MFMA_PEAK_TFLOPS / (9.0 if args.conv_arith == 'split9' else 6.0) if sp else FP32_MFMA_PEAK_TFLOPS
            r['frac'] = r['tflops'] / pk
            r['arithmetic'] = args.conv_arith if sp else 'fp32_mfma'

        if args.layer_table:
            with open(args.layer_table, 'w') as f:
                f.write('# %s: every conv launch of the first %d timed steps (HIP events on the launch stream), per step\n'
                        % (' '.join(sys.argv), min(3, args.steps)))
                f.write('%-16s %-14s %4s %4s %9s %8s %8s %7s %5s %s\n' % ('kernel', 'volume', 'cin', 'cout', 'launches', 'avg_ms',
                                                                          'ms/step', 'TFLOP/s', 'frac', 'arithmetic (frac = '
                                                                          'TFLOP/s / its MFMA peak: 2500 / products, or 157.3)'))
                for r in rows:
                    sp = args.conv_arith != 'fp32_mfma' and ops.conv_runs_split(r['kernel'], tuple(r['shape']), r['cin'], r['cout'])
                    pk = BF16_MFMA_PEAK_TFLOPS / (9.0 if args.conv_arith == 'split9' else 6.0) if sp else FP32_MFMA_PEAK_TFLOPS
                    f.write('%-16s %-14s %4d %4d %9.1f %8.4f %8.4f %7.1f %5.2f %s\n' % (
                        r['kernel'], 'x'.join(map(str, r['shape'])), r['cin'], r['cout'], r['launches'] / min(3, args.steps),
                        r['avg_ms'], r['total_ms'] / min(3, args.steps), r['tflops'], r['tflops'] / pk,
                        args.conv_arith if sp else 'fp32_mfma'))
        flops_launch = conv_flops(dom['kernel'], dom['shape'], dom['cin'], dom['cout'])
        # the peak the dominant kernel is priced against: layers on the split arithmetic issue 6 bf16 MFMAs per fp32 MFMA's
        # worth of algorithmic work -> dense bf16 peak / 6; layers on the fp32 matrix instructions -> the fp32 MFMA peak
        dom_split = args.conv_arith != 'fp32_mfma' and ops.conv_runs_split(dom['kernel'], dom['shape'], dom['cin'], dom['cout'])
        nprod = 9.0 if args.conv_arith == 'split9' else 6.0
        peak = BF16_MFMA_PEAK_TFLOPS / nprod if dom_split else FP32_MFMA_PEAK_TFLOPS
        roofline = {'bound': 'mfma', 'kernel': '%s %s Cin=%d Cout=%d' % (dom['kernel'], 'x'.join(map(str, dom['shape'])),
                                                                          dom['cin'], dom['cout']),
                    'achieved': round(dom['tflops'], 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                    'frac': round(dom['tflops'] / peak, 4),
                    'peak_basis': ('dense bf16 MFMA peak %.0f TFLOP/s / %d partial products per fp32 product (split arithmetic); '
                                   'achieved = ALGORITHMIC fp32 flops / time' % (BF16_MFMA_PEAK_TFLOPS, nprod)) if dom_split else
                                  'dense fp32 MFMA peak',
                    'frac_of_fp32_mfma_peak': round(dom['tflops'] / FP32_MFMA_PEAK_TFLOPS, 4), 'traffic': None,
                    'flops_per_launch': flops_launch, 'avg_launch_ms': round(dom['avg_ms'], 4),
                    'all_conv_tflops': round(sum(conv_flops(r['kernel'], r['shape'], r['cin'], r['cout']) * r['launches']
                                                 for r in rows) / (conv_total * 1e-3) / 1e12, 2),
                    'conv_ms_per_step': round(conv_total / min(3, args.steps), 3), 'profiled_steps': min(3, args.steps)}
        # HBM traffic of the dominant kernel: measured in separate rocprofv3 --pmc passes (bench.py cannot host the
        # profiler), committed under profiles/pmc_traffic.json; algorithmic bytes = the tensors one launch must touch
        roofline['traffic_source'] = None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'pmc_traffic.json')) as f:
                pmc_all = json.load(f)
            pmc = pmc_all.get(roofline['kernel'])
            if pmc and S == 160:
                roofline['traffic'] = pmc['bytes']
                roofline['traffic_source'] = ('NOT measured in this run: read from profiles/pmc_traffic.json (%s); FETCH_SIZE '
                                              'doubled for 16-byte loads per the gfx950 note' % pmc_all.get('_source', '?'))
        except (OSError, ValueError):
            pass
        vox = dom['shape'][0] * dom['shape'][1] * dom['shape'][2]
        roofline['algorithmic_bytes'] = 4 * vox * (dom['cin'] + dom['cout'])
        # the generator against ITS roofline (HBM): compulsory bytes = labels in (uint8 in the resident pool), image
        # (C_in(+maps)) and target out (float32), each touched once = 13 B/voxel at configs[1]; per-kernel times from the same HIP events
        nsteps_prof = min(3, args.steps)
        gen_ms = [v for k, v in gen_agg.items() if k[0] == 'generator']
        roofline_generator = None
        if gen_ms:
            ms = float(np.mean(gen_ms[0]))
            key = [k for k in gen_agg if k[0] == 'generator'][0]
            nvox = S ** 3
            label_bytes = tr.resident_labels[0].element_size()   # the resident pool keeps its maps in the narrowest integer type
            comp = nvox * (label_bytes + 4 * (key[2] + key[3]))
            roofline_generator = {'bound': 'hbm', 'achieved': round(comp / (ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS,
                                  'unit': 'GB/s', 'frac': round(comp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  'compulsory_bytes': comp, 'label_bytes_per_voxel': label_bytes, 'two_pass_floor_bytes': comp + 8 * nvox,
                                  'ms_per_volume': round(ms, 4),
                                  'kernels_ms': {k[0][4:]: round(float(np.sum(v)) / nsteps_prof, 4)
                                                 for k, v in gen_agg.items() if k[0].startswith('gen:')},
                                  'kernel_ms_sum': round(float(sum(np.sum(v) for k, v in gen_agg.items()
                                                                   if k[0].startswith('gen:'))) / nsteps_prof, 4),
                                  'note': 'ms_per_volume = HIP events around the whole generator call (11 launches incl. dispatch gaps) '
                                          'on the last profiled steps; kernel_ms_sum = the kernels alone; traffic per kernel: '
                                          'profiles/pmc_traffic.json'}
        out = {'metric': 'training volumes/sec (160^3 fp32, 5-level U-Net)', 'value': round(world * args.steps / dt, 4),
               'unit': 'volumes/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(1e3 * dt / args.steps, 3),
               # per-step device times of rank 0 (HIP events at the step boundaries; the last `profiled_steps` steps carry the
               # per-launch events of the roofline measurement): value stays K steps / wall time of the whole region
               'step_ms': {'mean': round(float(step_ms.mean()), 3), 'median': round(float(np.median(step_ms)), 3),
                           'min': round(float(step_ms.min()), 3), 'max': round(float(step_ms.max()), 3),
                           'p95': round(float(np.percentile(step_ms, 95)), 3),
                           'first_tenth_median': round(float(np.median(step_ms[:max(1, len(step_ms) // 10)])), 3),
                           'last_tenth_median': round(float(np.median(step_ms[-max(1, len(step_ms) // 10):])), 3)},
               'value_median_step': round(world * 1e3 / float(np.median(step_ms)), 4),
               'higher_is_better': True, 'scaling': 'weak',
               'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': 'configs[1]: brain_generator %d^3 batch=1 (training() defaults) + 5-level 3-D U-Net '
                                      '(24..384 features, Cin=2) fwd/bwd + Adam, fp32, random-init' % S,
                          'global_batch': world, 'parallelism': 'dp%d' % world, 'volume': [S, S, S],
                          'conv_arithmetic': args.conv_arith,
                          'conv_arithmetic_note': CONV_ARITH_NOTE[args.conv_arith]},
               'other_conv_arithmetic': other,
               'roofline': roofline, 'roofline_generator': roofline_generator, 'final_loss': round(final_loss, 6),
               'n_ranks_seen': dist.get_world_size() if dist.is_initialized() else 1,
               'backend': (backend if dist.is_initialized() else None),
               'rccl_version': rccl_version(backend) if dist.is_initialized() else None,
               # payload of the gradient all-reduce of the LAST step as the reducer issued it (expected: the whole flat
               # gradient buffer, 13 240 489 floats = 52.96 MB, in tail-first buckets)
               'allreduce_bytes_per_step': (int(tr.reducer.bytes_launched) if tr.reducer is not None and (world > 1 or args.force_allreduce) else 0),
               'allreduce_buckets_per_step': (int(tr.reducer.n_launched) if tr.reducer is not None and (world > 1 or args.force_allreduce) else 0),
               'top_kernels': [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in rows[:6]]}
        if per_rank is not None:
            out['per_rank'] = per_rank
        elif comm_ms.size:
            out['allreduce_wait_ms'] = {'mean': round(float(comm_ms.mean()), 3), 'max': round(float(comm_ms.max()), 3)}
        if clocks:
            out['clock_log'] = clocks
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(args.cpu_size)
            except Exception as ex:  # the baseline is a reported number, never a reason to lose the GPU measurement
                out['cpu_baseline'] = {'value': None, 'error': repr(ex)}
        print(json.dumps(out))
    if world > 1 or args.force_allreduce:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    sys.exit(main() or 0)
