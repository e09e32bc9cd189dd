"""TransportationSimulator — the graph container the env and the runner talk to (reference:
src/transportation_simulator.py). In scope here: the graph layout contract (``graph.x``, ``edge_index``,
``edge_index_routes``, ``edge_attr*``, ``num_roads``, ``congestion_constant``), ``config_network`` (MATSim XML ->
graph, src/matsim_io.py), ``load_network`` for synthetic scenarios, the reference's ``save/<scenario>/network.pt`` cache
and ``data/<scenario>/network.xml[.gz]``, ``config_parameters / configure_core / set_time / reset / state / run``.
The metric tables and the analysis figures (``plot_*``, drawn by src/figures.py from the series kept on the device) follow
the reference's (src/transportation_simulator.py:387-745).
"""
from __future__ import annotations

import os
import time

import torch

from ._compat import Data
from .agents.base import Agents
from .feature_helpers import FeatureHelpers
from .simulation_core_model import SimulationCoreModel


class TransportationSimulator:
    def __init__(self, device: str, torch_compile: bool = False):
        self.model_core = None
        self.agent = Agents(device)
        self.device = device
        self.torch_compile = torch_compile
        self.graph = None
        self.time = 0
        self.inserting_time = self.core_time = self.withdraw_time = self.choice_time = 0
        self.timestep = 1
        self.leg_histogram_values = []
        self.road_optimality_values = []
        self.on_way_before = 0
        self.done_before = 0

    # -- network ------------------------------------------------------------------------------------------------------
    def config_network(self, file_path: str) -> None:
        """Road graph (+ SRC/DEST pseudo-nodes) from ``<file_path>.xml[.gz]`` (src/transportation_simulator.py:61-228)."""
        from .matsim_io import build_network
        t0 = time.time()
        self.graph, self.Nmax = build_network(file_path)
        self.graph = self.graph.to(self.device)
        self.h = FeatureHelpers(Nmax=self.Nmax)
        print(f"Network configured from {file_path} in {time.time() - t0:.2f} seconds")

    def load_network(self, scenario: str) -> None:
        from tarl_hip import synth
        spec = synth.parse_scenario(scenario)
        if spec is None and not os.path.exists(os.path.join("save", scenario, "network.pt")):
            # src/transportation_simulator.py:255-259: no cache -> build from data/<scenario>/network.xml[.gz], cache it
            self.config_network(os.path.join("data", scenario, "network"))
            self.save_network(os.path.join("save", scenario, "network.pt"))
            return
        if spec is not None:
            W, H = synth.torus_for_edges(spec["edges"])
            net = synth.torus_network(W, H)
            self.graph = Data(x=net.x, edge_index=net.edge_index, edge_attr=net.edge_attr,
                              edge_index_routes=net.edge_index, edge_attr_routes=net.edge_attr,
                              num_roads=net.num_roads, critical_number=net.critical_number,
                              congestion_constant=net.congestion_constant)
            self.Nmax = net.Nmax
        else:
            path = os.path.join("save", scenario, "network.pt")
            d = torch.load(path, weights_only=False)
            self.graph, self.Nmax = d["graph"], d["Nmax"]
        for k in list(vars(self.graph)) if hasattr(self.graph, "__dict__") else []:
            v = getattr(self.graph, k)
            if torch.is_tensor(v) and k != "adj_matrix":      # the dense N x N adjacency is never needed here
                setattr(self.graph, k, v.to(self.device))
        self.h = FeatureHelpers(Nmax=self.Nmax)

    def save_network(self, file_path: str) -> None:
        os.makedirs(os.path.dirname(file_path), exist_ok=True)
        torch.save({"graph": self.graph.clone().to("cpu"), "Nmax": self.Nmax}, file_path)

    def configure_core(self):
        self.model_core = SimulationCoreModel(self.Nmax, self.device, self.time, torch_compile=self.torch_compile)

    def config_parameters(self, timestep_size: float = 1, start_time: int = 0):
        self.timestep = timestep_size
        self.time = start_time
        self.configure_core()

    def set_time(self, time):
        self.time = time
        self.agent.set_time(time)
        self.model_core.set_time(time)

    # -- classical step (insert -> withdraw -> choice -> core), as ``run()`` of the reference ------------------------------
    def run(self):
        h = self.h
        b = time.time()
        self.graph.x = self.agent.insert_agent_into_network(self.graph, h)
        e = time.time(); self.inserting_time += e - b; b = e
        self.graph.x = self.agent.withdraw_agent_from_network(self.graph, h)
        e = time.time(); self.withdraw_time += e - b; b = e
        self.graph = self.agent.choice(self.graph, h)
        e = time.time(); self.choice_time += e - b; b = e
        self.graph = self.model_core(self.graph)
        self.core_time += time.time() - b
        self.set_time(self.time + self.timestep)
        self._log_step()

    def _log_step(self):
        """Per-step records kept on the device (the reference syncs and copies to the host every step)."""
        on_way = torch.sum(self.agent.agent_features[:, self.agent.ON_WAY])
        done = torch.sum(self.agent.agent_features[:, self.agent.DONE])
        self.leg_histogram_values.append([on_way - self.on_way_before + done - self.done_before,
                                          done - self.done_before, on_way, self.time])
        self.on_way_before, self.done_before = on_way, done
        self.road_optimality_values.append(
            (self.time, self.model_core.direction_mpnn.road_optimality_data["delta_travel_time"]))

    # -- metrics: tables and figures ----------------------------------------------------------------------------------------
    def compute_node_metrics(self, output_dir: str | None = "data/outputs"):
        """Hourly departures per road (Response pops + withdrawals), their V/C ratio against MAX_FLOW, mean and std over
        the hours; ``node_metrics.csv`` with the reference's columns (src/transportation_simulator.py:563-670). The
        per-step masks stay on the device; one (H, T) x (T, R) product and a single copy to the host."""
        import csv
        hist = list(getattr(self.model_core.response_mpnn, "update_history", [])) + \
            list(getattr(self.agent, "withdraw_history", []))
        if not hist:
            print("No update history available for computing node metrics.")
            return {}
        dev = hist[0][1].device
        times = torch.tensor([int(t) for t, _ in hist], dtype=torch.long, device=dev)
        masks = torch.stack([m.view(-1) for _, m in hist]).to(torch.float32)             # (T, R)
        hours = (times // 3600).clamp(min=0)
        H = int(hours.max().item()) + 1
        onehot = torch.nn.functional.one_hot(hours, num_classes=H).to(torch.float32)       # (T, H); counts < 2^24: exact
        counts = (onehot.t() @ masks).t().contiguous()                                     # (R, H)
        R = counts.size(0)
        cap = self.graph.x[:R, self.h.MAX_FLOW].clone()
        cap[cap == 0] = float("nan")
        vc = counts / cap.unsqueeze(1)
        avg, std = torch.nanmean(vc, dim=1), torch.std(vc, dim=1, unbiased=False)
        counts_h, avg_h, std_h = counts.to(torch.long).cpu(), avg.cpu(), std.cpu()
        if output_dir is not None:
            os.makedirs(output_dir, exist_ok=True)
            with open(os.path.join(output_dir, "node_metrics.csv"), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["node_id", "avg_vc", "std_vc"] + [f"count_{k}h" for k in range(H)])
                for n in range(R):
                    w.writerow([n, float(avg_h[n]), float(std_h[n])] + counts_h[n].tolist())
        return {n: {"avg_vc": float(avg_h[n]), "std_vc": float(std_h[n]), "hourly_counts": counts_h[n].tolist()}
                for n in range(R)}

    def leg_histogram(self):
        """(T, 4) tensor [departures, arrivals, on the way, time] per step — the series behind the reference's
        ``plot_leg_histogram`` (src/transportation_simulator.py:387-451), one host copy."""
        if not self.leg_histogram_values:
            return torch.zeros((0, 4))
        return torch.stack([torch.stack([torch.as_tensor(v, dtype=torch.float32).reshape(()).cpu() for v in row])
                            for row in self.leg_histogram_values])

    def _history(self):
        return list(getattr(self.model_core.response_mpnn, "update_history", [])) + \
            list(getattr(self.agent, "withdraw_history", []))

    def plot_leg_histogram(self, output_dir: str | None = "data/outputs"):
        """Departures / arrivals / agents on the way in bins of ``18 // timestep`` steps
        (src/transportation_simulator.py:387-451)."""
        from . import figures
        return figures.leg_histogram_figure(self.leg_histogram().numpy(), self.timestep, output_dir)

    def plot_road_optimality(self, output_dir: str | None = "data/outputs", road_ids: list = []):
        """delta_travel_time summed over every road's outgoing route edges, one line per road over the clock
        (src/transportation_simulator.py:453-517)."""
        from . import figures
        return figures.road_optimality_figure(self.road_optimality_values, self.graph.edge_index_routes[0],
                                              self.graph.num_roads, road_ids, output_dir)

    def plot_computation_time(self, output_dir: str = "data/outputs"):
        """Pie of the four phase timers (src/transportation_simulator.py:519-561); here they are host enqueue times."""
        from . import figures
        return figures.computation_time_figure(self.inserting_time, self.choice_time, self.core_time,
                                               self.withdraw_time, output_dir)

    def plot_daily_counts(self, expected_counts: dict, output_dir: str | None = "data/outputs"):
        """Simulated departures per road against the MSA assignment's expected flows, scatter + ``daily_counts.csv``
        (src/transportation_simulator.py:672-745)."""
        from . import figures
        return figures.daily_counts_figure(self._history(), expected_counts, output_dir)

    def reset(self):
        from tarl_hip import ops
        ops.reset_state(self.graph.x, self.Nmax)

    def state(self):
        h = self.h
        x = self.graph.x[:, h.MAX_NUMBER_OF_AGENT:]
        agent_index = self.graph.x[:, h.HEAD_FIFO].to(torch.int64)
        return x, self.graph.edge_attr, self.graph.edge_index, agent_index
