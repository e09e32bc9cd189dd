#!/usr/bin/env python3
"""Developer tool: who are the event rows of the row pass? Runs the bench workload (config 4) for --frames frames and, at the
last frame, classifies every (row, environment): empty / idle / arrival / pop / withdraw / due-but-blocked."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tarl-simulator_amd")]
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=2048)
ap.add_argument("--frames", type=int, default=230)
ap.add_argument("--departure-window", type=int, default=0)
a = ap.parse_args()
from tarl_hip import synth  # noqa: E402
from tarl_hip.engine import SimEngine  # noqa: E402

net = synth.torus_network(25, 25)
N, B = net.num_roads, a.envs
pops = synth.population_batch(16384, N, B, seed=1, device="cuda",
                              t1=synth.EPISODE_START + a.departure_window if a.departure_window else synth.EPISODE_END)
eng = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax, pops,
                congestion_constant=net.congestion_constant, seed=3)
eng.reset()
eng.prepare_policy(torch.randn(N, generator=torch.Generator().manual_seed(1)).cuda())
T = a.frames
ch = torch.zeros((T, N, B), dtype=torch.uint8, device="cuda")
ct = torch.zeros((T + 1, N, B), dtype=torch.uint8, device="cuda")
rw = torch.zeros((T, B), device="cuda")
eng.rollout_fused(T, choice=ch, log_prob=None, reward=rw, counts=ct)
fs = eng.fs
n0 = fs.count.clone()                      # (N, B)
dep0 = fs.head_dep.clone()
t = float(eng.time)
popped = torch.zeros((B, N), dtype=torch.uint8, device="cuda")
wd = torch.zeros((B, N), dtype=torch.uint8, device="cuda")
cf = torch.zeros((N, B), device="cuda")
eng.frame_fused(popped=popped, withdrawn=wd, counts=cf)
post = fs.post.clone()
arrived = (post & 1) != 0
pop, w = popped.t() != 0, wd.t() != 0
nonempty = n0 > 0
due = nonempty & (dep0 <= t)
moved = pop | w
blocked = due & ~moved
tot = N * B
f = lambda m: f"{100.0 * float(m.sum()) / tot:6.2f} %"
print(f"frame {T}, window {a.departure_window}: non-empty {f(nonempty)}  head due {f(due)}  arrival {f(arrived)}  pop {f(pop)}  withdraw {f(w)}")
print(f"  due but neither popped nor withdrawn (re-examined every frame) {f(blocked)};  events of any kind {f(arrived | pop | w | blocked)}"
      f"  without the blocked ones {f(arrived | pop | w)}")
print(f"  agents on the way per env {float(eng.agents[:, :, 7].sum()) / B:.0f}, mean count of non-empty rows {float(n0[nonempty].mean()):.2f}")
