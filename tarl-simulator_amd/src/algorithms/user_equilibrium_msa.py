"""Method of Successive Averages on the device (reference: src/algorithms/user_equilibrium_msa.py:65-165).

``run_msa(graph, agents)`` post-processes a simulation snapshot into deterministic user-equilibrium link flows: build
the OD demand from the agents' trips, then iterate {all-or-nothing assignment on the current shortest paths, MSA
averaging of the flows, BPR update of the link costs} until the L1 change of the flows drops below ``tol``.

Where the reference walks ``nx.shortest_path`` per OD pair on the host, this build computes the all-pairs next-hop
table once per iteration with ``tarl_apsp_f64`` (float64 costs, as the reference keeps them) and assigns every OD pair
along it with ``tarl_msa_assign`` (one thread per pair, fp64 atomics). Shortest-path COSTS are unique; when two paths
tie exactly the reference's bidirectional Dijkstra and the all-pairs order may pick different ones (regular grids at
free flow) — the flows then differ by how the tied volume is routed, the equilibrium cost does not.
"""
from __future__ import annotations

from typing import Dict

import torch

from .._compat import cached_plan, require_cuda
from ..feature_helpers import FeatureHelpers

ALPHA, BETA = 0.15, 4.0    # BPR parameters of the reference


def build_demand(agents, num_nodes: int):
    """Distinct (origin, destination) pairs of the trips (row 0 = dummy agent is skipped) and their volumes."""
    feats = agents.agent_features
    dev = feats.device if feats is not None else "cuda"
    if feats is None or feats.size(0) <= 1:
        e = torch.zeros(0, dtype=torch.int64, device=dev)
        return e, e.clone(), torch.zeros(0, dtype=torch.float64, device=dev)
    flat = feats[1:, agents.ORIGIN].to(torch.int64) * num_nodes + feats[1:, agents.DESTINATION].to(torch.int64)
    pairs, counts = torch.unique(flat, return_counts=True)
    return (torch.div(pairs, num_nodes, rounding_mode="floor").contiguous(), (pairs % num_nodes).contiguous(),
            counts.to(torch.float64).contiguous())


def run_msa(graph, agents, tol: float = 1e-5, max_iter: int = 1000) -> Dict[int, float]:
    from tarl_hip import ops
    x = graph.x
    require_cuda(x, "graph.x")
    N = int(x.size(0))
    h = FeatureHelpers(Nmax=(int(x.size(1)) - 7) // 3)
    num_roads = int(getattr(graph, "num_roads", N))
    free_flow = x[:, h.FREE_FLOW_TIME_TRAVEL].to(torch.float64)
    capacity = x[:, h.MAX_FLOW].to(torch.float64).clamp(min=1e-8)
    is_road = x[:, h.ROAD_INDEX] >= 0
    road_u8 = is_road.to(torch.uint8).contiguous()
    od_o, od_d, od_vol = build_demand(agents, N)
    od_o, od_d, od_vol = od_o.to(x.device), od_d.to(x.device), od_vol.to(x.device)
    flow = torch.zeros(N, dtype=torch.float64, device=x.device)
    cost = torch.where(is_road, free_flow, torch.zeros_like(free_flow))
    plan = cached_plan(graph.edge_index, N)
    enter = graph.edge_index[1]                       # an edge costs what its head node costs
    for it in range(1, max_iter + 1):
        next_hop = ops.all_pairs_shortest_paths(plan, cost[enter].contiguous())[0][0]
        aux = torch.zeros_like(flow)
        ops.msa_assign(next_hop, od_o, od_d, od_vol, road_u8, aux)
        prev = flow.clone()
        flow += (1.0 / it) * (aux - flow)
        cost = torch.where(is_road, free_flow * (1.0 + ALPHA * (flow / capacity) ** BETA), cost)
        if float((flow - prev).abs().sum()) < tol:
            break
    out = flow[:num_roads].cpu().tolist()
    return {i: float(v) for i, v in enumerate(out)}
