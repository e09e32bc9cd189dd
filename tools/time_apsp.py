"""Time tarl_apsp (all-pairs next-hop + distance tables) on BASELINE config 1's and config 4's graph sizes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tarl-simulator_amd")]
import torch  # noqa: E402

from tarl_hip import ops, synth  # noqa: E402

for name, (W, H), het in (("config-1 size (5x6 torus, N=120)", (5, 6), False), ("config-4 (25x25 torus, N=2500)", (25, 25), False),
                          ("config-4 heterogeneous", (25, 25), True)):
    net = synth.torus_network(W, H, heterogeneous=het, seed=1)
    plan = ops.Plan(net.edge_index, net.num_roads)
    w = net.x[:, 3 * net.Nmax + 2][net.edge_index[0]].cuda()
    ops.all_pairs_shortest_paths(plan, w, want_dist=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        nh, d = ops.all_pairs_shortest_paths(plan, w, want_dist=True)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per all-pairs table "
          f"(reachable pairs {int((nh >= 0).sum())}, max dist {float(d[torch.isfinite(d)].max()):.1f})", flush=True)
