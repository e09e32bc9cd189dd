"""Seeded synthetic dual graphs and populations (no reference scenario data ships).

Topology: directed ``W x H`` torus of intersections, 4 outgoing links each => ``R = 4WH`` roads; dual-graph edges are every
(in-link, out-link) pair at an intersection incl. the U-turn, built the way the reference's ``config_network`` does
(``src/transportation_simulator.py:150-171``): for every upstream link, one edge per outgoing link of its head
node, ``edge_attr = capacity / sum(capacity)``. This is a *pure road graph* (no SRC/DEST pseudo-nodes): ``N = R``,
``edge_index == edge_index_routes``, every node has out-degree 4 — the domain on which the reference's mpnn path is
well-defined.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

EPISODE_START = 6 * 3600 - 60   # src/reinforcement_learning.py:203
EPISODE_END = 7 * 3600          # src/reinforcement_learning.py:273


@dataclass
class SynthNetwork:
    x: torch.Tensor                 # (R, F) fp32
    edge_index: torch.Tensor        # (2, E) int64
    edge_attr: torch.Tensor         # (E, 1) fp32
    Nmax: int
    num_roads: int
    critical_number: torch.Tensor   # (R,)
    congestion_constant: torch.Tensor  # (R,)

    @property
    def F(self) -> int:
        return 3 * self.Nmax + 7

    def dense_adjacency(self) -> torch.Tensor:
        """``graph.adj_matrix`` of the reference (``src/transportation_simulator.py:196-198``) — O(N^2), tests only."""
        n = self.x.size(0)
        adj = torch.zeros((n, n), dtype=torch.bool)
        adj[self.edge_index[0], self.edge_index[1]] = True
        return adj


def torus_network(W: int, H: int, *, length: float = 100.0, lanes: float = 1.0, freespeed: float = 10.0,
                  capacity: float = 10.0, cell: float = 7.5, heterogeneous: bool = False, seed: int = 0,
                  Nmax: int | None = None) -> SynthNetwork:
    """Build the torus network. Homogeneous links reproduce the reference's test link
    (``tests/conftest.py:98-101``): MAX_NUMBER_OF_AGENT = 14, Nmax = 15, F = 52, free-flow 10 s.
    ``heterogeneous=True`` draws per-link length / capacity / lanes from a seeded generator (parity stress)."""
    g = torch.Generator().manual_seed(seed)
    V = W * H
    R = 4 * V
    node = torch.arange(V)
    vx, vy = node % W, node // W
    to = torch.stack([((vx + 1) % W) + vy * W, ((vx - 1) % W) + vy * W,
                      vx + ((vy + 1) % H) * W, vx + ((vy - 1) % H) * W], dim=1).reshape(-1)  # head node of link 4v+k
    if heterogeneous:
        lengths = 60.0 + 90.0 * torch.rand(R, generator=g)
        caps = 5.0 + torch.randint(0, 4, (R,), generator=g).float() * 5.0
        lanes_t = 1.0 + torch.randint(0, 2, (R,), generator=g).float()
    else:
        lengths = torch.full((R,), float(length))
        caps = torch.full((R,), float(capacity))
        lanes_t = torch.full((R,), float(lanes))
    maxn = torch.floor(lengths * lanes_t / cell) + 1            # int(len*lanes/cell) + 1
    nmax = int(maxn.max().item()) + 1 if Nmax is None else Nmax
    F = 3 * nmax + 7
    x = torch.zeros((R, F), dtype=torch.float32)
    x[:, 3 * nmax + 0] = maxn
    x[:, 3 * nmax + 2] = lengths / freespeed
    x[:, 3 * nmax + 3] = lengths
    x[:, 3 * nmax + 4] = caps
    x[:, 3 * nmax + 6] = torch.arange(R, dtype=torch.float32)
    src = torch.arange(R).repeat_interleave(4)
    dst = (4 * to).repeat_interleave(4) + torch.arange(4).repeat(R)
    edge_index = torch.stack([src, dst]).to(torch.int64)
    if heterogeneous:   # arbitrary turn probabilities, normalised per upstream link (edge_attr is an input of the path)
        w = 0.5 + torch.rand((R, 4), generator=g)
        edge_attr = (w / w.sum(dim=1, keepdim=True)).to(torch.float32).view(-1, 1)
    else:               # the reference divides the *upstream* capacity by its own multiple => 1/4
        edge_attr = (caps[src] / (4.0 * caps[src])).to(torch.float32).view(-1, 1)
    critical = x[:, 3 * nmax + 4] * x[:, 3 * nmax + 2] / 3600
    cong = x[:, 3 * nmax + 2] * (x[:, 3 * nmax + 0] + 10 - critical)
    return SynthNetwork(x=x, edge_index=edge_index, edge_attr=edge_attr, Nmax=nmax, num_roads=R,
                        critical_number=critical, congestion_constant=cong)


def population(num_agents: int, num_roads: int, *, seed: int = 0, t0: int = EPISODE_START, t1: int = EPISODE_END,
               dummy_departure: float = 48 * 3600.0) -> torch.Tensor:
    """``agent_features`` ``(num_agents + 1, 9)``; row 0 is the reference's never-departing dummy
    (``src/agents/base.py:132-133,444``). ORIGIN / DESTINATION iid uniform road ids, DEPARTURE_TIME ~ U{t0..t1}."""
    g = torch.Generator().manual_seed(seed)
    a = torch.zeros((num_agents + 1, 9), dtype=torch.float32)
    a[1:, 0] = torch.randint(0, num_roads, (num_agents,), generator=g).float()
    a[1:, 1] = torch.randint(0, num_roads, (num_agents,), generator=g).float()
    a[1:, 2] = torch.randint(t0, t1 + 1, (num_agents,), generator=g).float()
    a[0, 2] = dummy_departure
    return a


def population_batch(num_agents: int, num_roads: int, num_envs: int, *, seed: int = 0, device="cuda",
                     t0: int = EPISODE_START, t1: int = EPISODE_END, dummy_departure: float = 48 * 3600.0) -> torch.Tensor:
    """``num_envs`` independent populations ``(B, num_agents + 1, 9)`` drawn directly on ``device`` from one seeded device
    generator (same distributions as :func:`population`): set-up takes milliseconds instead of a host loop over the
    environments, which matters when 8 ranks share the host cores."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(int(seed))
    a = torch.zeros((num_envs, num_agents + 1, 9), dtype=torch.float32, device=dev)
    shape = (num_envs, num_agents)
    a[:, 1:, 0] = torch.randint(0, num_roads, shape, generator=g, device=dev).float()
    a[:, 1:, 1] = torch.randint(0, num_roads, shape, generator=g, device=dev).float()
    a[:, 1:, 2] = torch.randint(t0, t1 + 1, shape, generator=g, device=dev).float()
    a[:, 0, 2] = dummy_departure
    return a


def random_state(net: SynthNetwork, *, seed: int = 0, t: float = 100.0, fill: float = 0.5,
                 num_agents: int | None = None) -> torch.Tensor:
    """A random but *consistent* mid-simulation state for kernel parity tests: FIFO prefixes hold distinct agent ids,
    arrival <= t, departures scattered around ``t`` so both admissibility branches of DirectionMPNN fire, and
    SELECTED_ROAD points at a random out-neighbour (sometimes at a non-neighbour)."""
    g = torch.Generator().manual_seed(seed)
    x = net.x.clone()
    R, nmax = x.size(0), net.Nmax
    maxn = x[:, 3 * nmax].to(torch.int64)
    u = torch.rand(R, generator=g)
    n = torch.where(u < 0.15, torch.zeros_like(maxn),
                    torch.where(u > 0.85, maxn - torch.randint(0, 4, (R,), generator=g),
                                (torch.rand(R, generator=g) * fill * 2 * maxn.float()).to(torch.int64)))
    n = n.clamp(min=0)
    n = torch.minimum(n, maxn - 1)   # a FIFO at MAX that receives a gridlock-relief move leaves the reference's domain
    total = int(n.sum().item())
    pool = num_agents if num_agents is not None else max(total, 1)
    ids = (torch.randperm(max(pool, total), generator=g)[:total] + 1).float()
    slot = torch.arange(nmax).unsqueeze(0)
    occ = slot < n.unsqueeze(1)
    x[:, 0:nmax][occ] = ids
    arr = t - torch.randint(0, 40, (R, nmax), generator=g).float()
    dep = t + torch.randint(-30, 12, (R, nmax), generator=g).float()
    x[:, nmax:2 * nmax] = torch.where(occ, arr, torch.zeros_like(arr))
    x[:, 2 * nmax:3 * nmax] = torch.where(occ, dep, torch.zeros_like(dep))
    # stale garbage beyond the occupied prefix in ~1/4 of the rows (the reference leaves such values behind)
    stale = (torch.rand(R, generator=g) < 0.25).unsqueeze(1) & ~occ
    x[:, 0:nmax] = torch.where(stale, torch.randint(1, 50, (R, nmax), generator=g).float(), x[:, 0:nmax])
    x[:, 3 * nmax + 1] = n.float()
    # selected road: a real out-neighbour for ~90 % of the roads
    out_dst = net.edge_index[1].view(R, 4)
    pick = out_dst[torch.arange(R), torch.randint(0, 4, (R,), generator=g)]
    rnd = torch.randint(0, R, (R,), generator=g)
    x[:, 3 * nmax + 5] = torch.where(torch.rand(R, generator=g) < 0.9, pick, rnd).float()
    return x


def torus_for_edges(num_edges: int) -> tuple[int, int]:
    """(W, H) with 16*W*H == num_edges for the BASELINE configs: 1 024 -> 8x8, 10 000 -> 25x25, 100 000 -> 25x250."""
    v = num_edges // 16
    w = int(math.isqrt(v))
    while v % w:
        w -= 1
    return w, v // w


def parse_scenario(name: str):
    """``synthetic-<edges>-<agents>[-<seed>]`` -> dict, anything else -> None."""
    parts = str(name).split("-")
    if len(parts) < 3 or parts[0] != "synthetic":
        return None
    return {"edges": int(parts[1]), "agents": int(parts[2]), "seed": int(parts[3]) if len(parts) > 3 else 0}


# ---- MATSim-format writers (inputs for the network / population builders; no reference scenario data ships) ------------
def write_matsim_network_xml(path: str, W: int, H: int, *, seed: int = 0, heterogeneous: bool = True,
                             effectivecellsize: float | None = 7.5) -> None:
    """A ``W x H`` torus of intersections ("n<k>" ids, coordinates on a 100 m grid) with 4 outgoing links each, written
    as a MATSim ``network.xml`` (nodes + links with from/to/length/capacity/freespeed/permlanes)."""
    g = torch.Generator().manual_seed(seed)
    V = W * H
    lines = ['<?xml version="1.0" encoding="utf-8"?>', '<network name="synthetic torus">', "  <nodes>"]
    for v in range(V):
        lines.append(f'    <node id="n{v}" x="{(v % W) * 100.0}" y="{(v // W) * 100.0}"/>')
    cs = "" if effectivecellsize is None else f' effectivecellsize="{effectivecellsize}"'
    lines += ["  </nodes>", f'  <links capperiod="01:00:00"{cs}>']
    lid = 0
    for v in range(V):
        vx, vy = v % W, v // W
        for to in (((vx + 1) % W) + vy * W, ((vx - 1) % W) + vy * W, vx + ((vy + 1) % H) * W, vx + ((vy - 1) % H) * W):
            if heterogeneous:
                length = round(60.0 + 90.0 * float(torch.rand(1, generator=g)), 2)
                cap = 300 + 100 * int(torch.randint(0, 6, (1,), generator=g))
                speed = [8.33, 13.89, 16.67][int(torch.randint(0, 3, (1,), generator=g))]
                lanes = 1 + int(torch.randint(0, 2, (1,), generator=g))
            else:
                length, cap, speed, lanes = 100, 10, 10, 1
            lines.append(f'    <link id="{lid}" from="n{v}" to="n{to}" length="{length}" capacity="{cap}" '
                         f'freespeed="{speed}" permlanes="{lanes}"/>')
            lid += 1
    lines += ["  </links>", "</network>"]
    with open(path, "w", encoding="utf-8") as f:
        f.write("\n".join(lines) + "\n")


def write_matsim_grid_xml(path: str, W: int, H: int, *, seed: int = 0, heterogeneous: bool = False) -> None:
    """A ``W x H`` NON-torus grid with a link in both directions between 4-neighbours (4 x 6 -> 24 nodes / 76 links, the
    size of Sioux Falls: BASELINE config 1), same node naming as the torus writer."""
    g = torch.Generator().manual_seed(seed)
    lines = ['<?xml version="1.0" encoding="utf-8"?>', '<network name="synthetic grid">', "  <nodes>"]
    for v in range(W * H):
        lines.append(f'    <node id="n{v}" x="{(v % W) * 100.0}" y="{(v // W) * 100.0}"/>')
    lines += ["  </nodes>", '  <links capperiod="01:00:00" effectivecellsize="7.5">']
    lid = 0
    for v in range(W * H):
        vx, vy = v % W, v // W
        for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            ux, uy = vx + dx, vy + dy
            if not (0 <= ux < W and 0 <= uy < H):
                continue
            if heterogeneous:
                length = round(60.0 + 90.0 * float(torch.rand(1, generator=g)), 2)
                cap = 300 + 100 * int(torch.randint(0, 6, (1,), generator=g))
                speed = [8.33, 13.89, 16.67][int(torch.randint(0, 3, (1,), generator=g))]
            else:
                length, cap, speed = 100, 10, 10
            lines.append(f'    <link id="{lid}" from="n{v}" to="n{ux + uy * W}" length="{length}" capacity="{cap}" '
                         f'freespeed="{speed}" permlanes="1"/>')
            lid += 1
    lines += ["  </links>", "</network>"]
    with open(path, "w", encoding="utf-8") as f:
        f.write("\n".join(lines) + "\n")


def write_matsim_population_xml(path: str, W: int, H: int, persons: int, *, seed: int = 0,
                                first_departure: int = 6 * 3600, spread: int = 3600) -> None:
    """Persons with 2-4 activities whose ``link`` attribute names an intersection id (the convention the reference's
    parser uses), some with an unknown link + coordinates (nearest-intersection fallback), some without a car, some with
    explicit attributes, ``HH:MM`` and ``HH:MM:SS`` end times."""
    g = torch.Generator().manual_seed(seed)
    V = W * H
    r = lambda n: int(torch.randint(0, n, (1,), generator=g))
    lines = ["<?xml version='1.0' encoding='utf-8'?>", "<population>"]
    for p in range(persons):
        attrs = ""
        if r(5) == 0:
            attrs += ' car_avail="never"'
        if r(3) == 0:
            attrs += f' sex="{"f" if r(2) else "m"}" age="{18 + r(60)}" employed="{"yes" if r(2) else "no"}"'
        lines.append(f'  <person id="p{p}"{attrs}>')
        if r(4) == 0:
            lines.append(f'    <attributes><attribute name="age">{20 + r(50)}</attribute></attributes>')
        lines.append("    <plan>")
        t = first_departure + r(spread)
        for a in range(2 + r(3) if r(12) else 1):
            v = r(V)
            if r(6) == 0:   # unknown link id, coordinates near intersection v
                where = f'x="{(v % W) * 100.0 + 3.0}" y="{(v // W) * 100.0 - 2.0}" link="zz{r(99)}"'
            else:
                where = f'x="{(v % W) * 100.0}" y="{(v // W) * 100.0}" link="n{v}"'
            hh, mm, ss = t // 3600, (t % 3600) // 60, t % 60
            end = f"{hh:02d}:{mm:02d}:{ss:02d}" if r(2) else f"{hh:02d}:{mm:02d}"
            lines.append(f'      <act type="{"h" if a % 2 == 0 else "w"}" {where} end_time="{end}"/>')
            t += 600 + r(3000)
        lines += ["    </plan>", "  </person>"]
    lines.append("</population>")
    with open(path, "w", encoding="utf-8") as f:
        f.write("\n".join(lines) + "\n")
