#!/usr/bin/env python
"""Time of one critic update (3 forward passes, 3 backward passes, penalty pass, Adam) and of the critic's share of a
generator update (forward + input gradient) at a given volume size:  python tools/adversarial_bench.py [size] [reps] [f32|bf16]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from synthsr_amd.critic import Critic3D  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dtype = sys.argv[3] if len(sys.argv) > 3 else 'f32'
    critic = Critic3D([S, S, S, 1], seed=0, dtype=dtype)
    real, fake = torch.rand(S, S, S, 1, device='cuda'), torch.rand(S, S, S, 1, device='cuda')
    print('critic: %.1f M parameters (Dense %d x %d)' % (critic.n_params / 1e6, critic.dense[0]['n_in'], critic.dense[0]['n_out']))

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def critic_update():
        critic.critic_loss_and_grads(real, fake, 0.4, 10.0)
        critic.adam_step(1e-4)
    print('%d^3 %s: critic update %.1f ms; critic part of a generator update %.1f ms'
          % (S, dtype, timed(critic_update), timed(lambda: critic.input_gradient(fake, -0.01))))


if __name__ == '__main__':
    main()
