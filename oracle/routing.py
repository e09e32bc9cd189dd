"""CPU restatement of the reference's shortest-path routing (TEST INFRASTRUCTURE — see oracle/__init__.py).

The reference delegates to networkx (requirements.txt pins no version; 3.4.2 in the build container):
``nx.all_pairs_dijkstra_path`` in DijkstraAgents.choice (src/agents/base.py:556-570) and ``nx.shortest_path_length`` in
MPNNPolicyNet.refresh_dijkstra (src/agents/mpnn_agent.py:53-79). Restated here: networkx's ``_dijkstra_multisource`` —
binary heap keyed by (distance, push counter), a node's path fixed by the first settled predecessor that reaches it at
a strictly smaller distance, successors visited in adjacency (edge insertion) order, Python-float (double) sums.
Pinned by tests/golden/routing.npz, generated from the reference running on real networkx."""
from __future__ import annotations

import heapq
from itertools import count

import torch


def edge_travel_time(x, edge_index, congestion_constant, Nmax):
    """src/agents/base.py:541-550."""
    xu = x[edge_index[0]]
    tc = congestion_constant[edge_index[1]] / (xu[:, 3 * Nmax] + 10 - xu[:, 3 * Nmax + 1])
    return torch.max(torch.stack((xu[:, 3 * Nmax + 2], tc)), dim=0).values


def _adjacency(edge_index, weights, N):
    succ = [dict() for _ in range(N)]
    w = weights.reshape(-1).tolist()
    for e, (u, v) in enumerate(zip(edge_index[0].tolist(), edge_index[1].tolist())):
        succ[u][v] = w[e]          # DiGraph: a repeated (u, v) keeps its first position, last weight
    return succ


def all_pairs(edge_index, weights, N):
    """-> next_hop int64 (N, N) (src/agents/base.py:556-570 conventions), dist float32 (N, N) (inf / 0 diagonal)."""
    succ = _adjacency(edge_index, weights, N)
    next_hop = torch.full((N, N), -1, dtype=torch.int64)
    dist_m = torch.full((N, N), float("inf"), dtype=torch.float64)
    for s in range(N):
        dist, seen, hop = {}, {s: 0.0}, {s: s}
        c = count()
        fringe = [(0.0, next(c), s)]
        while fringe:
            d, _, v = heapq.heappop(fringe)
            if v in dist:
                continue
            dist[v] = d
            for u, cost in succ[v].items():
                vu = d + cost
                if u in dist:
                    continue
                if u not in seen or vu < seen[u]:
                    seen[u] = vu
                    heapq.heappush(fringe, (vu, next(c), u))
                    hop[u] = u if v == s else hop[v]
        for t, d in dist.items():
            next_hop[s, t] = hop[t]
            dist_m[s, t] = d
    return next_hop, dist_m.to(torch.float32)


def dijkstra_choice(x, agents, edge_index, congestion_constant, Nmax, next_hop=None):
    """DijkstraAgents.choice (src/agents/base.py:527-584) -> (x with SELECTED_ROAD rewritten, next_hop used)."""
    N = x.size(0)
    if next_hop is None:
        next_hop, _ = all_pairs(edge_index, edge_travel_time(x, edge_index, congestion_constant, Nmax), N)
    head = x[:, 0].to(torch.int64)
    dest = agents[head, 1].to(torch.int64)
    x = x.clone()
    x[:, 3 * Nmax + 5] = next_hop[torch.arange(N), dest].to(x.dtype)
    return x, next_hop
