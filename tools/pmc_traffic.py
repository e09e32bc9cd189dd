#!/usr/bin/env python3
"""Developer tool: run F fused frames (device noise) at a BASELINE config size so that rocprofv3 --pmc can attribute HBM
traffic to each kernel. Usage on the GPU box (separate passes, FETCH_SIZE and WRITE_SIZE do not fit one pass):
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -- python tools/pmc_traffic.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -- python tools/pmc_traffic.py
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tarl-simulator_amd")]
import torch  # noqa: E402
from tarl_hip import synth  # noqa: E402
from tarl_hip.engine import SimEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--edges", type=int, default=10000)
ap.add_argument("--agents", type=int, default=16384)
ap.add_argument("--envs", type=int, default=2048)
ap.add_argument("--frames", type=int, default=40)
args = ap.parse_args()
W, H = synth.torus_for_edges(args.edges)
net = synth.torus_network(W, H)
N = net.num_roads
pops = torch.stack([synth.population(args.agents, N, seed=b) for b in range(args.envs)])
eng = SimEngine(net.x.cuda().unsqueeze(0).repeat(args.envs, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                pops.cuda(), congestion_constant=net.congestion_constant, seed=0)
emb = torch.randn(N, device="cuda")
eng.reset()
eng.prepare_policy(emb)
choice = torch.empty((N, args.envs), dtype=torch.int32, device="cuda")
counts = torch.empty((N, args.envs), device="cuda")
lp = torch.empty(args.envs, device="cuda")
for _ in range(args.frames):
    eng.frame_fused(choice=choice, log_prob=lp, counts=counts)
torch.cuda.synchronize()
print("frames", args.frames, "on_way", float(eng.agents[:, :, 7].sum()))
