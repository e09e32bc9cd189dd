import os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "tarl-simulator_amd")]
import torch
from tarl_hip import synth
from tarl_hip.engine import SimEngine
net = synth.torus_network(25, 25)
N = net.num_roads
B, A = 64, 16384
pops = torch.stack([synth.population(A, N, seed=b) for b in range(B)])
eng = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax, pops.cuda(),
                congestion_constant=net.congestion_constant, device=torch.device("cuda"), seed=0)
eng.reset()
emb = torch.zeros(N, device="cuda")
eng.prepare_policy(emb, 1.0)
T = 256
choice = torch.empty((T, N, B), dtype=torch.int32, device="cuda")
reward = torch.empty((T, B), device="cuda"); counts = torch.zeros((T + 1, N, B), device="cuda")
for chunk in range(4):
    Tc = 64
    eng.rollout_fused(Tc, choice=choice[:Tc], log_prob=None, reward=reward[:Tc], counts=counts[:Tc + 1])
    fs = eng.fs
    t = eng.time
    dep = fs.a_dep[0]; st = fs.a_status[0]
    due = dep <= t
    order = fs.a_order[0].long()
    cur = int(fs.cur_lo[0])
    ndue = int(due.sum()); nwait = int((due & (st == 0)).sum())
    first_notdue = int((~due[order]).float().argmax()) if bool((~due).any()) else A
    print(f"t={t}: cur_lo={cur} first_not_due_pos={first_notdue} window={first_notdue - cur} due={ndue} waiting&due={nwait} "
          f"on_way={int((st == 1).sum())} done={int((st == 2).sum())} reward={float(reward[Tc-1, 0])}", flush=True)
