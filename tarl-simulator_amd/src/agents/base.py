"""Agents — the population store and the two per-step population kernels (reference: src/agents/base.py).

``insert_agent_into_network`` / ``withdraw_agent_from_network`` keep the reference's signatures and in-place
conventions and run as ``tarl_insert_step`` / ``tarl_withdraw_step``: no host sync, no Python loop over roads, no dense
N x N adjacency (the plan's CSR rows answer "is there an edge road -> destination"). ``config_agents_from_xml`` parses a
MATSim population (src/matsim_io.py). The Dijkstra baseline is outside this build's scope.
"""
from __future__ import annotations

import os

import torch

from .._compat import cached_plan, cached_routing_plan, require_cuda
from ..feature_helpers import AgentFeatureHelpers, FeatureHelpers


class Agents(AgentFeatureHelpers):
    def __init__(self, device):
        super().__init__()
        self.agent_features = None
        self.time = 0
        self.device = device
        self.withdraw_history: list = []      # (time, uint8/Bool mask over the road rows), kept on the device

    # -- population from MATSim XML ----------------------------------------------------------------------------------------
    def config_agents_from_xml(self, scenario: str, *, verbose: bool = True) -> None:
        """``data/<scenario>/population.xml[.gz]`` + ``data/<scenario>/network.xml[.gz]`` -> ``agent_features`` with one
        row per trip (src/agents/base.py:36-242). An absolute ``scenario`` path is used as is (os.path.join)."""
        from ..matsim_io import build_population
        rows, stats = build_population(os.path.join("data", scenario, "population"),
                                       os.path.join("data", scenario, "network"), log=print if verbose else None)
        self.agent_features = rows.to(self.device)
        print(f"Population: {stats['selected']}/{stats['total']} persons selected, {rows.size(0) - 1} trips")
        if verbose:
            excl = {k: stats[k] for k in ("car_avail_not_always", "no_plan", "too_few_activities", "no_valid_trip")}
            print(f"  exclusion reasons: {excl}")
            if stats["trips"]:
                t = stats["trips"]
                print(f"  trips per person: min {min(t)} max {max(t)} mean {sum(t) / len(t):.2f}")

    # -- per-step kernels -------------------------------------------------------------------------------------------------
    def insert_agent_into_network(self, graph, h: FeatureHelpers) -> torch.Tensor:
        """Every ready agent (departure time reached, not on its way, not done) enters the road its origin currently
        selects, first-come (agent id) first-served up to ``MAX - 3 - count`` per road."""
        from tarl_hip import ops
        x = graph.x
        require_cuda(x, "graph.x")
        cc = getattr(graph, "congestion_constant", None)
        if cc is not None:
            cc = cc.to(torch.float32).contiguous()
        ops.insert_step(x, h.Nmax, self.agent_features, self.time, congestion_constant=cc)
        return x

    def withdraw_agent_from_network(self, graph, h: FeatureHelpers) -> torch.Tensor:
        """Pop the leading run of agents whose destination is adjacent to the road they head and whose departure time
        has come; mark them DONE with ARRIVAL_TIME = now."""
        from tarl_hip import ops
        x = graph.x
        require_cuda(x, "graph.x")
        plan = cached_plan(graph.edge_index, x.size(0))
        mask = ops.withdraw_step(plan, x, h.Nmax, self.agent_features, self.time)
        num_roads = int(getattr(graph, "num_roads", x.size(0)))
        self.withdraw_history.append((self.time, mask.view(-1)[:num_roads].bool()))
        return x

    @torch.no_grad()
    def choice(self, graph, h: FeatureHelpers):
        """Random routing: every node with a road among its successors selects one of them uniformly (the classical
        ``random`` agent; road -> DEST edges are not candidates and nodes without a candidate keep their selection)."""
        from tarl_hip import ops
        x = graph.x
        plan = cached_routing_plan(graph.edge_index, x.size(0), int(getattr(graph, "num_roads", x.size(0))))
        logits = torch.zeros(plan.num_edges, dtype=torch.float32, device=x.device)
        proba = ops.graphdist_softmax(plan, logits)
        self._choice_counter = getattr(self, "_choice_counter", 0) + 1
        _, ch = ops.graphdist_sample(plan, proba, seed=torch.initial_seed() & 0x7FFFFFFF, counter=self._choice_counter,
                                     want_onehot=False, want_choice=True)
        ops.apply_action(plan, x, h.Nmax, choice=ch)
        return graph

    # -- bookkeeping ------------------------------------------------------------------------------------------------------
    def reset(self):
        self.agent_features[:, self.ON_WAY] = 0.0
        self.agent_features[:, self.DONE] = 0.0
        self.withdraw_history = []

    def set_time(self, time):
        self.time = time

    def save(self, file_path: str) -> None:
        os.makedirs(os.path.dirname(file_path), exist_ok=True)
        torch.save(self.agent_features.cpu(), file_path)

    def load(self, scenario: str) -> None:
        """``save/<scenario>/population.pt`` (a bare tensor, as the reference writes it) or a synthetic scenario name
        ``synthetic-<edges>-<agents>[-seed]`` (no reference scenario data ships)."""
        from tarl_hip import synth
        spec = synth.parse_scenario(scenario)
        if spec is not None:
            W, H = synth.torus_for_edges(spec["edges"])
            self.agent_features = synth.population(spec["agents"], 4 * W * H, seed=spec["seed"]).to(self.device)
        else:
            path = os.path.join("save", scenario, "population.pt")
            if not os.path.exists(path):   # src/agents/base.py load(): no cache -> parse the XML population, cache it
                self.config_agents_from_xml(scenario)
                self.save(path)
            else:
                obj = torch.load(path, weights_only=True, map_location="cpu")
                if not torch.is_tensor(obj):
                    raise TypeError(f"expected a Tensor in {path}, got {type(obj)}")
                self.agent_features = obj.to(self.device, torch.float32).contiguous()
        self.agent_features[0, self.DEPARTURE_TIME] = 48 * 3600   # agent 0 never joins the network


class DijkstraAgents(Agents):
    """Shortest-path routing (reference: src/agents/base.py:519-584): every ``refresh_rate`` calls the all-pairs next-hop
    table is rebuilt from the current travel times, and every row selects the next hop towards its head agent's
    destination. The table comes from ``tarl_apsp`` — one wave per source node, networkx's tie order — instead of
    ``nx.all_pairs_dijkstra_path`` on the host."""

    def __init__(self, device):
        super().__init__(device)
        self.count = 0
        self.refresh_rate = 10
        self.next_hop_tensor = None

    @torch.no_grad()
    def choice(self, graph, h: FeatureHelpers):
        from tarl_hip import ops
        x = graph.x
        require_cuda(x, "graph.x")
        if self.count % self.refresh_rate == 0:
            plan = cached_plan(graph.edge_index, x.size(0))
            w = ops.edge_travel_time(plan, x, h.Nmax, graph.congestion_constant)
            self.next_hop_tensor = ops.all_pairs_shortest_paths(plan, w)[0][0]
        ops.select_next_hop(x, h.Nmax, self.agent_features, self.next_hop_tensor)
        self.count += 1
        return graph
