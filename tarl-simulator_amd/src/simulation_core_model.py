"""SimulationCoreModel — DirectionMPNN then ResponseMPNN on the road rows of the graph
(reference: src/simulation_core_model.py:41-88)."""
from __future__ import annotations

import torch.nn as nn

from .direction_mpnn import DirectionMPNN
from .response_mpnn import ResponseMPNN


class SimulationCoreModel(nn.Module):
    def __init__(self, Nmax: int, device: str, time: int, torch_compile: bool = False):
        super().__init__()
        self.direction_mpnn = DirectionMPNN(Nmax=Nmax, time=time)
        self.response_mpnn = ResponseMPNN(Nmax=Nmax, time=time)
        # ``torch_compile`` is accepted for signature compatibility; there is nothing to trace: the rounds are
        # hand-written kernels already.
        self.time = time
        self.Nmax = Nmax

    def forward(self, graph):
        R = graph.num_roads
        x_roads = graph.x[:R]                       # a view: the kernels update graph.x in place
        cc = getattr(graph, "congestion_constant", None)
        if cc is not None:
            cc = cc[:R]
        # without precomputed constants the update kernel derives them from the row itself, which is what
        # src/simulation_core_model.py:55-67 computes from the same (pre-round) columns
        self.direction_mpnn(x_roads, graph.edge_index_routes, graph.edge_attr_routes, congestion_constant=cc)
        self.response_mpnn(x_roads, graph.edge_index_routes, graph.edge_attr_routes)
        return graph

    def set_time(self, time):
        self.time = time
        self.direction_mpnn.set_time(time)
        self.response_mpnn.set_time(time)
