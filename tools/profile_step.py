#!/usr/bin/env python3
"""Developer tool: run S environment steps (policy logits -> distribution -> env step) for B environments at a BASELINE
config size, with device-side Philox noise, and print per-phase HIP-event timings. Used under rocprofv3."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tarl-simulator_amd")]
import torch  # noqa: E402
from tarl_hip import ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--edges", type=int, default=10000)
ap.add_argument("--agents", type=int, default=16384)
ap.add_argument("--envs", type=int, default=256)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--warmup", type=int, default=10)
args = ap.parse_args()

W, H = synth.torus_for_edges(args.edges)
net = synth.torus_network(W, H)
N, Nmax, E, B, A = net.num_roads, net.Nmax, net.edge_index.size(1), args.envs, args.agents + 1
plan = ops.Plan(net.edge_index, N)
ec = ops.EdgeConst(net.edge_attr, "cuda")
cc = net.congestion_constant.cuda()
x = net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous()
pop = synth.population(args.agents, N, seed=0, t0=21540, t1=21540 + 600)
ag = pop.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous()
emb = torch.randn(N, device="cuda")
chosen = torch.empty((B, N), device="cuda")
popped = torch.empty((B, N), dtype=torch.uint8, device="cuda")
scratch = torch.empty((B, 2 * A), dtype=torch.int32, device="cuda")
reward = torch.empty(B, device="cuda")
counts = torch.empty((B, N), device="cuda")
print(f"N={N} E={E} B={B} A={A} state={x.numel()*4/2**20:.0f} MiB agents={ag.numel()*4/2**20:.0f} MiB")

phases = ["logits", "softmax", "sample", "logprob", "apply", "core", "withdraw", "insert"]
acc = {p: 0.0 for p in phases}


def step(t, k, timed):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(phases) + 1)] if timed else None
    def mark(i):
        if timed:
            ev[i].record()
    mark(0)
    logits = ops.policy_edge_logits(plan, x[:, :, 3 * Nmax:], emb); mark(1)
    p = ops.graphdist_softmax(plan, logits); mark(2)
    _, choice = ops.graphdist_sample(plan, p, seed=1, counter=2 * k, want_onehot=False, want_choice=True); mark(3)
    lp, ent = ops.graphdist_logprob_entropy(plan, p, choice=choice); mark(4)
    ops.apply_action(plan, x, Nmax, choice=choice); mark(5)
    ops.core_step(plan, x, Nmax, ec, t, congestion_constant=cc, seed=1, counter=2 * k + 1, want_dtt=False,
                  chosen=chosen, popped=popped); mark(6)
    ops.withdraw_step(plan, x, Nmax, ag, t, want_mask=False); mark(7)
    ops.insert_step(x, Nmax, ag, t, congestion_constant=cc, scratch=scratch, reward=reward, counts=counts); mark(8)
    return ev


t = 21540
for k in range(args.warmup):
    step(t, k, False); t += 1
torch.cuda.synchronize()
t0 = time.perf_counter()
evs = []
for k in range(args.steps):
    evs.append(step(t, args.warmup + k, True)); t += 1
torch.cuda.synchronize()
wall = time.perf_counter() - t0
for ev in evs:
    for i, p in enumerate(phases):
        acc[p] += ev[i].elapsed_time(ev[i + 1])
tot = sum(acc.values())
for p in phases:
    print(f"{p:10s} {acc[p]/args.steps*1e3:9.1f} us/step")
print(f"sum {tot/args.steps*1e3:.1f} us/step; wall {wall/args.steps*1e6:.1f} us/step; env-steps/s {B*args.steps/wall:,.0f}; "
      f"msgpass edges/s {B*E*args.steps/wall:,.0f}; on_way={ag[:, :, 7].sum().item():.0f} done={ag[:, :, 8].sum().item():.0f}")
