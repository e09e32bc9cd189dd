#!/usr/bin/env python3
"""Developer tool: does splitting the environments over G concurrent HIP streams hide the latency-bound kernels of a frame
(insert, the event tails) behind the other groups' throughput kernels? G engines of B / G environments, one stream each."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tarl-simulator_amd")]
import torch  # noqa: E402
from tarl_hip import synth  # noqa: E402
from tarl_hip.engine import SimEngine  # noqa: E402

B, T = 16384, 256
net = synth.torus_network(25, 25)
N = net.num_roads
emb = torch.randn(N, generator=torch.Generator().manual_seed(1)).cuda()
for G in (1, 2, 4):
    b = B // G
    engs, bufs, streams = [], [], []
    for g in range(G):
        pops = synth.population_batch(16384, N, b, seed=g, device="cuda")
        e = SimEngine(net.x.cuda().unsqueeze(0).repeat(b, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax, pops,
                      congestion_constant=net.congestion_constant, seed=3, env_base=g * b)
        e.prepare_policy(emb)
        engs.append(e)
        bufs.append((torch.zeros((T, N, b), dtype=torch.uint8, device="cuda"), torch.zeros((T + 1, N, b), dtype=torch.uint8, device="cuda"),
                     torch.zeros((T, b), device="cuda"), torch.zeros((T, b), device="cuda")))
        streams.append(torch.cuda.Stream())
    torch.cuda.synchronize()
    for rep in range(3):
        for e in engs:
            e.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for e, (ch, ct, lp, rw), st in zip(engs, bufs, streams):
            with torch.cuda.stream(st):
                e.rollout_fused(T, choice=ch, log_prob=lp, reward=rw, counts=ct, check=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"G = {G}: {dt * 1e3:.1f} ms per rollout of {T} frames x {B} environments -> {B * T / dt / 1e6:.2f} M env-steps/s (rollout only)")
    del engs, bufs
    torch.cuda.empty_cache()
