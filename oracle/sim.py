"""CPU restatement of the traffic-flow step (TEST INFRASTRUCTURE — see oracle/__init__.py).

State layout (reference ``src/feature_helpers.py:38-54``), ``x`` fp32 ``(R, F)`` with ``F = 3*Nmax + 7``:
``[0,Nmax)`` FIFO agent ids (head = col 0) | ``[Nmax,2Nmax)`` arrival times | ``[2Nmax,3Nmax)`` departure times |
``3Nmax+0`` MAX_NUMBER_OF_AGENT | ``+1`` NUMBER_OF_AGENT | ``+2`` FREE_FLOW_TIME_TRAVEL | ``+3`` LENGHT_OF_ROAD |
``+4`` MAX_FLOW | ``+5`` SELECTED_ROAD | ``+6`` ROAD_INDEX.
Agent table (``src/feature_helpers.py:59-71``), fp32 ``(A, 9)``: ORIGIN, DESTINATION, DEPARTURE_TIME, ARRIVAL_TIME,
AGE, SEX, EMPLOYMENT_STATUS, ON_WAY, DONE.

All functions mutate ``x`` (and ``agent_features``) in place like the reference does.
"""
from __future__ import annotations

import torch

CONGESTION_FILE = 3  # src/feature_helpers.py:54
EPS = 1e-12          # src/direction_mpnn.py:136

# agent columns
ORIGIN, DESTINATION, DEPARTURE_TIME, ARRIVAL_TIME, AGE, SEX, EMPLOYMENT_STATUS, ON_WAY, DONE = range(9)


class Cols:
    """Column indices for a given Nmax (src/feature_helpers.py:38-54)."""

    def __init__(self, Nmax: int):
        self.Nmax = Nmax
        self.F = 3 * Nmax + 7
        self.HEAD, self.HEAD_ARR, self.HEAD_DEP = 0, Nmax, 2 * Nmax
        self.MAXN = 3 * Nmax
        self.N = 3 * Nmax + 1
        self.FF = 3 * Nmax + 2
        self.LEN = 3 * Nmax + 3
        self.MAXFLOW = 3 * Nmax + 4
        self.SEL = 3 * Nmax + 5
        self.ROAD = 3 * Nmax + 6


def gumbel_from_uniform(u: torch.Tensor) -> torch.Tensor:
    """src/direction_mpnn.py:137 — ``-log(-log(u))`` in fp32."""
    return -torch.log(-torch.log(u))


def congestion_constants(x: torch.Tensor, Nmax: int):
    """src/simulation_core_model.py:55-67 and src/transportation_simulator.py:207-210."""
    c = Cols(Nmax)
    critical = x[:, c.MAXFLOW] * x[:, c.FF] / 3600
    cong = x[:, c.FF] * (x[:, c.MAXN] + 10 - critical)
    return critical, cong


def direction_message(x, edge_index, edge_attr, t, Nmax):
    """src/direction_mpnn.py:74-100. Edge e: j=edge_index[0] (upstream) -> i=edge_index[1] (downstream).

    Returns (agent_id (E,), prob (E,), delta_travel_time (E,)).
    """
    c = Cols(Nmax)
    x_j = x.index_select(0, edge_index[0])  # full-row gathers, as PyG's collect does
    x_i = x.index_select(0, edge_index[1])
    dep = x_j[:, c.HEAD_DEP]
    arr = x_j[:, c.HEAD_ARR]
    agent_id = x_j[:, c.HEAD]
    m1 = (dep <= t) & (x_i[:, c.N] < x_i[:, c.MAXN] - CONGESTION_FILE)
    m1 = m1 & (x_j[:, c.SEL] == x_i[:, c.ROAD]) & (x_j[:, c.N] > 0)
    m2 = (dep - t < -10) & (x_j[:, c.MAXN] - CONGESTION_FILE <= x_j[:, c.N])
    m2 = m2 & (x_j[:, c.MAXN] - x_j[:, c.N] <= x_i[:, c.MAXN] - x_i[:, c.N])
    m2 = m2 & (x_j[:, c.SEL] == x_i[:, c.ROAD])
    prob = edge_attr.reshape(-1) * (m1 | m2).float()
    dtt = torch.clamp((dep - arr) - x_j[:, c.FF], min=0)
    return agent_id, prob, dtt


def segment_argmax_first(scores: torch.Tensor, index: torch.Tensor, n: int) -> torch.Tensor:
    """torch-scatter 2.1.2 ``scatter_max`` arg on CPU: strict ``>`` scan => lowest element index among the maxima;
    groups without elements get ``len(scores)``."""
    E = scores.numel()
    mx = scores.new_full((n,), float("-inf")).scatter_reduce_(0, index, scores, reduce="amax", include_self=True)
    is_max = scores == mx[index]
    eid = torch.arange(E, dtype=torch.long)
    cand = torch.where(is_max, eid, torch.full_like(eid, E))
    return torch.full((n,), E, dtype=torch.long).scatter_reduce_(0, index, cand, reduce="amin", include_self=True)


def direction_aggregate(agent_id, prob, index, n, gumbel=None, uniform=None):
    """src/direction_mpnn.py:103-146. Noise: pass ``gumbel`` (already transformed) or ``uniform``; otherwise
    ``E`` uniforms are drawn from the global generator exactly as ``torch.rand_like`` does there."""
    P = torch.zeros(n, dtype=prob.dtype).index_add_(0, index, prob)  # sequential fp32, edge order
    if gumbel is None:
        if uniform is None:
            uniform = torch.rand_like(prob + EPS)
        gumbel = gumbel_from_uniform(uniform)
    scores = torch.log(prob + EPS) + gumbel
    arg = segment_argmax_first(scores, index, n)
    chosen = torch.zeros(n, dtype=prob.dtype)
    has = P > 0
    chosen[has] = agent_id[arg[has]]
    return chosen


def direction_update(x, chosen, t, Nmax, congestion_constant=None):
    """src/direction_mpnn.py:171-196 — in place, every row (also when nothing was chosen)."""
    c = Cols(Nmax)
    R = x.size(0)
    rows = torch.arange(R)
    q = x[:, c.N].to(torch.int64)
    n0 = x[:, c.N]                       # a live VIEW, exactly as the reference's ``start_counts`` (``:173``)
    x[rows, q] = chosen
    x[rows, Nmax + q] = float(t)
    if congestion_constant is None:      # ``:178-183`` — evaluated inside update, after the first two writes
        critical = x[rows, c.MAXFLOW] * x[rows, c.FF] / 3600
        congestion_constant = x[rows, c.FF] * (x[rows, c.MAXN] + 10 - critical)
    t_cong = congestion_constant / (x[rows, c.MAXN] + 10 - n0)
    tt = torch.maximum(x[rows, c.FF], t_cong)
    x[rows, 2 * Nmax + q] = t + tt
    is_agent = chosen != 0
    x[is_agent, c.N] = n0[is_agent] + 1
    return x


def direction_step(x, edge_index, edge_attr, t, Nmax, *, gumbel=None, uniform=None, congestion_constant=None):
    """DirectionMPNN.forward == propagate (src/direction_mpnn.py:210-236). Returns (x, delta_travel_time)."""
    agent_id, prob, dtt = direction_message(x, edge_index, edge_attr, t, Nmax)
    chosen = direction_aggregate(agent_id, prob, edge_index[1], x.size(0), gumbel=gumbel, uniform=uniform)
    direction_update(x, chosen, t, Nmax, congestion_constant)
    return x, dtt


def response_message(x, edge_index, Nmax):
    """src/response_mpnn.py:66-83. flow=target_to_source: x_i = x[edge_index[0]] upstream, x_j = x[edge_index[1]]."""
    c = Cols(Nmax)
    x_i = x.index_select(0, edge_index[0])
    x_j = x.index_select(0, edge_index[1])
    E = x_i.size(0)
    cnt_up = x_i[:, c.N].to(torch.int64)
    cnt_dn = x_j[:, c.N].to(torch.int64)
    head = x_i[:, c.HEAD].to(torch.int64)
    tail_idx = torch.clamp(cnt_dn - 1, min=0)
    tail_all = x_j[torch.arange(E), tail_idx].to(torch.int64)
    tail = torch.where(cnt_dn > 0, tail_all, torch.full_like(tail_all, -1))
    return ((cnt_up > 0) & (cnt_dn > 0) & (tail == head)).to(x.dtype)


def response_step(x, edge_index, Nmax):
    """ResponseMPNN.forward (src/response_mpnn.py:25-127). Returns (x, popped mask (R,) bool)."""
    c = Cols(Nmax)
    R = x.size(0)
    msg = response_message(x, edge_index, Nmax)
    aggr = torch.zeros(R, dtype=x.dtype).scatter_reduce_(0, edge_index[0], msg, reduce="amax", include_self=False)
    mask = aggr > 0
    if bool(mask.any()):
        for base in (0, Nmax, 2 * Nmax):
            x[mask, base:base + Nmax - 1] = x[mask, base + 1:base + Nmax]  # last slot keeps its stale value
        x[mask, c.N] = x[mask, c.N] - 1
    return x, mask


def core_step(x, edge_index, edge_attr, t, Nmax, *, gumbel=None, uniform=None, congestion_constant=None):
    """SimulationCoreModel.forward on the road rows (src/simulation_core_model.py:41-83).
    Returns (x, delta_travel_time, popped)."""
    if congestion_constant is None:  # ``:55-67``: computed from the pre-round state when the graph lacks the attributes
        _, congestion_constant = congestion_constants(x, Nmax)
    _, dtt = direction_step(x, edge_index, edge_attr, t, Nmax, gumbel=gumbel, uniform=uniform,
                            congestion_constant=congestion_constant)
    _, popped = response_step(x, edge_index, Nmax)
    return x, dtt, popped


def withdraw(x, agent_features, adj, t, Nmax):
    """Agents.withdraw_agent_from_network (src/agents/base.py:348-403). ``adj`` dense bool (N, N).
    Returns (x, withdrawn mask over rows)."""
    c = Cols(Nmax)
    roads = x[:, c.ROAD].to(torch.long)
    ids = x[:, 0:Nmax].to(torch.long)
    dest = agent_features[ids, DESTINATION].to(torch.long)
    connectivity = adj[roads.unsqueeze(1), dest] > 0
    depart_ok = x[:, 2 * Nmax:3 * Nmax] <= t
    active = torch.arange(Nmax) < x[:, c.N].unsqueeze(1)
    eligible = connectivity & depart_ok & active
    lead = torch.cumprod(eligible.long(), dim=1).bool()
    cnt = lead.sum(dim=1)
    if bool(cnt.any()):
        gone = ids[lead]
        shift = torch.arange(Nmax).unsqueeze(0) + cnt.unsqueeze(1)
        valid = shift < Nmax
        shift = shift.clamp(max=Nmax - 1)
        for base in (0, Nmax, 2 * Nmax):
            blk = x[:, base:base + Nmax].gather(1, shift)
            blk[~valid] = 0
            x[:, base:base + Nmax] = blk
        x[:, c.N] -= cnt
        agent_features[gone, DONE] = 1
        agent_features[gone, ON_WAY] = 0
        agent_features[gone, ARRIVAL_TIME] = t
    return x, cnt > 0


def insert(x, agent_features, t, Nmax, congestion_constant=None):
    """Agents.insert_agent_into_network (src/agents/base.py:247-331). The reference's Python loop over roads
    (``:289-291``) keeps the first ``min(count, capacity)`` ready agents per road in argsort order; restated as a
    rank-within-road test with a *stable* argsort (the reference's unstable call is stable on CPU in practice)."""
    c = Cols(Nmax)
    ready = (agent_features[:, DEPARTURE_TIME] <= t) & (agent_features[:, ON_WAY] == 0) & (agent_features[:, DONE] == 0)
    if not bool(ready.any()):
        return x
    origins = agent_features[ready, ORIGIN].to(torch.long)
    road = x[origins, c.SEL].to(torch.long)
    cap = (x[road, c.MAXN] - CONGESTION_FILE - x[road, c.N]).to(torch.long)
    ok = cap > 0
    if not bool(ok.any()):
        return x
    agent_idx = torch.nonzero(ready).squeeze(1)[ok]
    road, cap = road[ok], cap[ok]
    order = torch.argsort(road, stable=True)
    road_s, agent_s, cap_s = road[order], agent_idx[order], cap[order]
    first = torch.ones_like(road_s, dtype=torch.bool)
    first[1:] = road_s[1:] != road_s[:-1]
    start = torch.cummax(torch.where(first, torch.arange(road_s.numel()), torch.zeros_like(road_s)), 0).values
    rank = torch.arange(road_s.numel()) - start
    keep = rank < cap_s
    road_s, agent_s, rank = road_s[keep], agent_s[keep], rank[keep]
    if agent_s.numel() == 0:
        return x
    n0 = x[road_s, c.N].to(torch.long)
    pos = n0 + rank
    x[road_s, pos] = agent_s.to(x.dtype)
    x[road_s, Nmax + pos] = float(t)
    if congestion_constant is not None:
        t_cong = congestion_constant[road_s].to(x.dtype) / (x[road_s, c.MAXN] + 10 - n0.to(x.dtype))
    else:
        t_cong = torch.zeros_like(n0, dtype=x.dtype)
    tt = torch.maximum(x[road_s, c.FF], t_cong)
    x[road_s, 2 * Nmax + pos] = float(t) + tt
    x[:, c.N] += torch.zeros(x.size(0), dtype=x.dtype).index_add_(0, road_s, torch.ones_like(tt))
    agent_features[agent_s, ON_WAY] = 1.0
    return x


def apply_action(x, edge_index, action, Nmax):
    """SimulatorEnv._step choice phase (src/reinforcement_learning.py:223-231)."""
    c = Cols(Nmax)
    m = action.to(torch.bool)
    x[edge_index[0][m], c.SEL] = edge_index[1][m].to(x.dtype)
    return x


def env_step(x, agent_features, edge_index, edge_attr, adj, action, t, Nmax, *, timestep=1, gumbel=None,
             uniform=None, congestion_constant=None, num_roads=None):
    """SimulatorEnv._step on a pure road graph (src/reinforcement_learning.py:222-309).

    ``t`` is the simulator clock when the step starts (core, withdraw and insert all use it).
    Returns dict(reward (1,), time (new clock), done, delta_travel_time, popped, withdrawn).
    """
    c = Cols(Nmax)
    R = x.size(0) if num_roads is None else num_roads
    apply_action(x, edge_index, action, Nmax)
    cc = None if congestion_constant is None else congestion_constant[:R]
    _, dtt, popped = core_step(x[:R], edge_index, edge_attr, t, Nmax, gumbel=gumbel, uniform=uniform,
                               congestion_constant=cc)
    _, withdrawn = withdraw(x, agent_features, adj, t, Nmax)
    insert(x, agent_features, t, Nmax, congestion_constant)
    reward = (-torch.sum(x[:, c.N])).flatten()
    t_new = t + timestep  # old_state is a view of the live column => always advances
    return {"reward": reward, "time": t_new, "done": t_new > 7 * 3600, "delta_travel_time": dtt,
            "popped": popped, "withdrawn": withdrawn}


def observe(x, Nmax):
    """TransportationSimulator.state (src/transportation_simulator.py:360-366): (node_features (N,7), agent_index (N,))."""
    return x[:, 3 * Nmax:], x[:, 0].to(torch.int64)
