#!/usr/bin/env python3
"""Developer probe: which small congested scenarios drive FIFO counts to Nmax - 1 (the clean -> dirty transition of the
fused path) without reaching Nmax (outside the reference's domain)?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tarl-simulator_amd")]
import torch  # noqa: E402
from tarl_hip import synth  # noqa: E402
from tarl_hip.engine import SimEngine  # noqa: E402

for (W, H, A, win, frames, het) in [(2, 2, 400, 10, 200, False), (2, 2, 250, 5, 300, False), (3, 2, 500, 10, 300, True),
                                    (2, 2, 300, 10, 300, True), (2, 3, 600, 20, 400, False), (2, 2, 220, 5, 400, False)]:
    net = synth.torus_network(W, H, heterogeneous=het, seed=3)
    N, B = net.num_roads, 4
    pops = torch.stack([synth.population(A, N, seed=b, t0=21540, t1=21540 + win) for b in range(B)])
    eng = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                    pops.cuda(), congestion_constant=net.congestion_constant, seed=5)
    eng.reset()
    eng.prepare_policy(torch.randn(N, generator=torch.Generator().manual_seed(1)).cuda())
    mx, flagged = 0, False
    for f in range(frames):
        eng.frame_fused()
        mx = max(mx, int(eng.fs.count.max()))
        if int(eng.fs.flags.item()) != 0:
            flagged = True
            break
    print(f"torus {W}x{H} het={het} A={A} window={win}s: Nmax={net.Nmax}, max count reached {mx}, dirty rows {int(((eng.fs.hdp[..., 0] & 0x80) != 0).sum())} of {N * B}, flagged={flagged} (frame {f})")
