#!/usr/bin/env python3
"""Developer tool: time the per-edge MLP forward kernels (fp32 MFMA, bf16, bf16x3) at the bench's state-dependent-policy
size (config 4: 10 000 edges, --envs samples) and report their logit error against an fp64 evaluation of the head."""
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
ec = ops.EdgeConst(net.edge_attr, "cuda")
g = torch.Generator().manual_seed(1)
obs = torch.randn((a.envs, N, 16), generator=g)
obs[..., 7:] = obs[..., 7:].abs() * 2.0e4          # clock-time sized columns
obs = obs.cuda()
ws = [torch.randn(s, generator=g) * 0.2 for s in ((64, 33), (64,), (32, 64), (32,), (1, 32), (1,))]
w = ops.EdgeMlpWeights(*[t.cuda() for t in ws])
out = torch.empty((a.envs, E), device="cuda")
res = {}
obs_bf = obs.to(torch.bfloat16)
for prec in ("fp32", "x3", "bf16", "bf16 obs"):
    if prec == "bf16 obs":      # the rollout's form: observations already bf16 (tarl_fused_obs16_bf16)
        x, prec_kw = obs_bf, None
    else:
        x, prec_kw = obs, prec
    ops.policy_edge_mlp(plan, x, ec, w, precision=prec_kw, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        ops.policy_edge_mlp(plan, x, ec, w, precision=prec_kw, out=out)
    e1.record()
    torch.cuda.synchronize()
    res[prec] = (e0.elapsed_time(e1) / a.reps * 1e3, out[:4].clone())
from oracle import nets  # noqa: E402
ref = nets.edge_mlp_logits(obs[:4].cpu().double(), net.edge_index, net.edge_attr.double().expand(4, -1, -1),
                           *[t.double() for t in ws])
scale = float(ref.abs().max())
for prec, (us, o) in res.items():
    err = float((o.cpu().double() - ref).abs().max())
    print(f"{prec:8s} {us:8.1f} us per {a.envs * E / 1e6:.1f} M edges   max |err| vs fp64 {err:.3e} = {err / scale:.2e} of scale")
