#!/usr/bin/env python3
"""Developer tool: time the GraphDistribution rollout sampler (k_graphdist_rollout_reg) at the state-dependent-policy lines'
size (config 4: 2 500 roads, 10 000 edges, --envs samples, device noise, choice8 + sel8 + log_prob outputs)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tarl-simulator_amd")]
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=2048)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
from tarl_hip import ops, synth  # noqa: E402

net = synth.torus_network(25, 25)
N, E = net.num_roads, net.edge_index.size(1)
plan = ops.Plan(net.edge_index, N)
g = torch.Generator().manual_seed(1)
logits = (torch.randn((a.envs, E), generator=g) * 50.0).cuda()
choice8 = torch.zeros((a.envs, N), dtype=torch.uint8, device="cuda")
sel8 = torch.zeros((N, a.envs), dtype=torch.uint8, device="cuda")
lp = torch.empty(a.envs, device="cuda")
scratch = None
for _ in range(3):
    ops.graphdist_rollout(plan, logits, 2000.0, seed=5, counter=1, choice8=choice8, sel8=sel8, log_prob=lp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for r in range(a.reps):
    ops.graphdist_rollout(plan, logits, 2000.0, seed=5, counter=2 + r, choice8=choice8, sel8=sel8, log_prob=lp)
e1.record()
torch.cuda.synchronize()
print(f"sampler {e0.elapsed_time(e1) / a.reps * 1e3:8.1f} us per {a.envs} environments   (sum of choice bytes {int(choice8.sum())}, mean log-prob {float(lp.mean()):.3f})")
