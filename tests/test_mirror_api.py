"""CPU: the drop-in mirror keeps the reference's module paths, names, constants, CLI dispatch order and state-dict
keys. No GPU compute here."""
import pytest
import torch

from conftest import load_golden


def test_feature_helpers_layout():
    from src.feature_helpers import FeatureHelpers, AgentFeatureHelpers, ObservationFeatureHelpers
    for Nmax in (5, 15, 100):
        h = FeatureHelpers(Nmax=Nmax)
        assert (h.AGENT_POSITION, h.AGENT_TIME_ARRIVAL, h.AGENT_TIME_DEPARTURE) == (
            slice(0, Nmax), slice(Nmax, 2 * Nmax), slice(2 * Nmax, 3 * Nmax))
        assert [h.MAX_NUMBER_OF_AGENT, h.NUMBER_OF_AGENT, h.FREE_FLOW_TIME_TRAVEL, h.LENGHT_OF_ROAD, h.MAX_FLOW,
                h.SELECTED_ROAD, h.ROAD_INDEX, h.NODE_TYPE] == [3 * Nmax + k for k in range(8)]
        assert (h.HEAD_FIFO, h.HEAD_FIFO_ARRIVAL_TIME, h.HEAD_FIFO_DEPARTURE_TIME, h.CONGESTION_FILE) == (0, Nmax, 2 * Nmax, 3)
    a = AgentFeatureHelpers()
    assert len(a) == 9     # tests/agents_test.py:9-10 of the reference
    assert [a.ORIGIN, a.DESTINATION, a.DEPARTURE_TIME, a.ARRIVAL_TIME, a.AGE, a.SEX, a.EMPLOYMENT_STATUS, a.ON_WAY,
            a.DONE] == list(range(9))
    o = ObservationFeatureHelpers()
    assert (o.NUMBER_OF_AGENT, o.ROAD_INDEX, o.ORIGIN, o.DONE) == (1, 6, 7, 15)
    # the golden fixtures were produced with the reference's own helpers: F = 3*Nmax + 7
    g = load_golden("core_hom")
    assert g["x0"].size(1) == 3 * g["Nmax"] + 7


def test_module_surface():
    import torch.nn as nn
    from src.direction_mpnn import DirectionMPNN
    from src.response_mpnn import ResponseMPNN
    from src.simulation_core_model import SimulationCoreModel
    d, r = DirectionMPNN(), ResponseMPNN()
    assert isinstance(d, nn.Module) and isinstance(r, nn.Module)
    assert d.Nmax == 100 and d.time == 0 and d.NUMBER_OF_AGENT == 301 and r.NUMBER_OF_AGENT == 301
    assert d.road_optimality_data is None and r.update_history == []
    core = SimulationCoreModel(Nmax=15, device="cpu", time=7)
    core.set_time(9)
    assert core.direction_mpnn.time == 9 and core.response_mpnn.time == 9 and core.Nmax == 15


def test_state_dict_keys_match_reference_modules():
    """Keys recorded from the reference's own MPNNPolicyNet / MPNNValueNetSimple (tests/golden/nets.npz)."""
    from src.agents.mpnn_agent import MPNNPolicyNet, MPNNValueNetSimple
    g = load_golden("nets")
    ei = g["edge_index"]
    N = g["node_features"].size(0)
    pol = MPNNPolicyNet(ei, N, torch.ones(ei.size(1)), device="cpu")
    val = MPNNValueNetSimple(ei, N, device="cpu")
    ref_pol = sorted(k[len("pol__"):].replace("__", ".") for k in g if k.startswith("pol__"))
    ref_val = sorted(k[len("val__"):].replace("__", ".") for k in g if k.startswith("val__"))
    assert sorted(pol.state_dict()) == ref_pol and sorted(val.state_dict()) == ref_val
    for k, v in pol.state_dict().items():
        assert tuple(v.shape) == tuple(g["pol__" + k.replace(".", "__")].shape)
    assert len(pol) == 9 and pol.agent_features is None     # it is also the population store (Agents)
    # the checkpoint written by the reference loads into the mirror
    pol.load_state_dict({k: g["pol__" + k.replace(".", "__")] for k in pol.state_dict()})
    val.load_state_dict({k: g["val__" + k.replace(".", "__")] for k in val.state_dict()})


def test_cli_dispatch_order(monkeypatch, tmp_path):
    """tests/main_cli_test.py of the reference: setup -> [train] -> eval."""
    import importlib
    import os
    import sys
    from conftest import PKG
    sys.path.insert(0, PKG)
    main = importlib.import_module("main").main
    from src.runner import Runner
    calls = []
    monkeypatch.setattr(Runner, "setup", lambda self: calls.append("setup"))
    monkeypatch.setattr(Runner, "train", lambda self: calls.append("train"))
    monkeypatch.setattr(Runner, "eval", lambda self: calls.append("eval"))
    main(["--algo", "dijkstra", "--mode", "eval"])
    assert calls == ["setup", "eval"]
    calls.clear()
    main(["--algo", "mpnn+ppo", "--mode", "train", "--output-dir", str(tmp_path)])
    assert calls == ["setup", "train", "eval"]
    calls.clear()
    main(["--algo", "mpnn", "--mode", "eval", "--steps", "10"])
    assert calls == ["setup", "eval"]


def test_cpu_graph_is_refused_loudly():
    """The product path never computes on the CPU: handing it host tensors raises."""
    from src.direction_mpnn import DirectionMPNN
    from tarl_hip.lib import TarlError
    g = load_golden("core_hom")
    with pytest.raises(TarlError):
        DirectionMPNN(Nmax=g["Nmax"])(g["x0"].clone(), g["edge_index"], g["edge_attr"])


def test_synthetic_scenarios():
    from tarl_hip import synth
    assert synth.parse_scenario("Easy") is None
    assert synth.parse_scenario("synthetic-10000-16384") == {"edges": 10000, "agents": 16384, "seed": 0}
    for edges, (r, e) in {1024: (256, 1024), 10000: (2500, 10000), 100000: (25000, 100000)}.items():
        W, H = synth.torus_for_edges(edges)
        assert 16 * W * H == e and 4 * W * H == r
    net = synth.torus_network(8, 8)
    assert net.Nmax == 15 and net.F == 52 and float(net.x[0, 45]) == 14 and float(net.x[0, 47]) == 10
    assert torch.equal(net.edge_index[0], torch.arange(256).repeat_interleave(4))          # sorted by source
    indeg = torch.zeros(256).index_add_(0, net.edge_index[1], torch.ones(1024))
    assert bool((indeg == 4).all()) and bool((net.edge_attr == 0.25).all())
    pop = synth.population(100, 256, seed=1)
    assert pop.shape == (101, 9) and float(pop[0, 2]) == 48 * 3600


def test_fused_path_supported_is_decided_from_the_topology():
    """ppo_train picks the packed path or the unfused entry points from the graph alone (every rank takes the same
    branch): Nmax <= 127, out-degree <= 126, no parallel dual edges."""
    import torch
    from tarl_hip import ops
    ring = torch.tensor([[0, 1, 2, 3], [1, 2, 3, 0]])
    assert ops.fused_path_supported(ring, 15) and not ops.fused_path_supported(ring, 128)
    parallel = torch.tensor([[0, 0, 1], [1, 1, 0]])                  # two dual edges 0 -> 1
    assert not ops.fused_path_supported(parallel, 15)
    star = torch.stack([torch.zeros(127, dtype=torch.long), torch.arange(1, 128)])      # out-degree 127
    assert not ops.fused_path_supported(star, 15)
    assert ops.fused_path_supported(star[:, :126], 15)
