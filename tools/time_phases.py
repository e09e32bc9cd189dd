#!/usr/bin/env python3
"""Developer tool: where does one PPO iteration of the bench workload go? Events around collect / advantages / minibatch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tarl-simulator_amd")]
import torch  # noqa: E402

sys.argv = [sys.argv[0]] + sys.argv[1:]
import bench  # noqa: E402

args = bench.parse()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
net, engine, trainer = bench.build_trainer(args, 0, dev)
for _ in range(2):
    trainer.train_iteration()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
acc = [0.0] * 4
N = 4
for _ in range(N):
    ev[0].record()
    trainer.collect()
    ev[1].record()
    adv, tgt = trainer.advantages()
    ev[2].record()
    trainer.minibatch_step(adv, tgt)
    ev[3].record()
    torch.cuda.synchronize()
    for k in range(3):
        acc[k] += ev[k].elapsed_time(ev[k + 1])
t0 = time.perf_counter()
for _ in range(N):
    trainer.train_iteration()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N * 1e3
print(f"collect {acc[0] / N:.2f} ms  advantages {acc[1] / N:.2f} ms  minibatch_step {acc[2] / N:.2f} ms  | train_iteration wall {wall:.2f} ms")
