"""GPU: shortest-path routing (csrc/routing.hip) — edge travel times, the all-pairs next-hop / distance tables with
networkx's tie order, and the classical ``dijkstra`` loop of the mirror — against the reference's goldens (real
networkx, tests/golden/routing.npz) and the oracle's heap replay on further graphs. All integer outputs bit-exact;
distances are double sums rounded once to fp32, also exact."""
import os

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tarl_hip import ops as _ops
    return _ops


def test_choice_tables_and_prior_golden(ops):
    from src.agents.base import DijkstraAgents
    from src.agents.mpnn_agent import MPNNPolicyNet
    from src._compat import Data
    from src.feature_helpers import FeatureHelpers
    from tarl_hip import synth
    g = load_golden("routing")
    net = synth.torus_network(3, 3, heterogeneous=True, seed=int(g["torus__seed"]))
    Nmax, N = net.Nmax, net.num_roads
    plan = ops.Plan(net.edge_index, N)
    x = g["torus__x0"].clone().cuda().unsqueeze(0)
    w = ops.edge_travel_time(plan, x, Nmax, net.congestion_constant.cuda())
    from oracle import routing
    assert torch.equal(w[0].cpu(), routing.edge_travel_time(g["torus__x0"], net.edge_index, net.congestion_constant, Nmax))
    nh, dist = ops.all_pairs_shortest_paths(plan, w, want_dist=True)
    assert torch.equal(nh[0].cpu().to(torch.int16), g["torus__next_hop"])
    # the mirror class, driven like the reference's
    ag = DijkstraAgents("cuda")
    ag.agent_features = g["torus__agents"].clone().cuda()
    graph = Data(x=g["torus__x0"].clone().cuda(), edge_index=net.edge_index.cuda(), num_roads=N,
                 congestion_constant=net.congestion_constant.cuda())
    out = ag.choice(graph, FeatureHelpers(Nmax=Nmax))
    assert torch.equal(out.x.cpu(), g["torus__x1"]) and ag.count == 1
    assert torch.equal(ag.next_hop_tensor.cpu().to(torch.int16), g["torus__next_hop"])
    # free-flow prior of the policy
    pol = MPNNPolicyNet(net.edge_index.cuda(), N, g["torus__ff_edges"].cuda(), device="cuda")
    assert torch.equal(pol.dist_matrix.cpu(), g["torus__dist_matrix"])
    prior = pol.compute_dijkstra_logits(g["torus__prior_dest"].cuda(), g["torus__ff_edges"].cuda())
    assert torch.equal(prior.cpu(), g["torus__prior_logits"])


@pytest.mark.parametrize("tag,het", [("grid", False), ("gridhet", True)])
def test_classical_dijkstra_run_golden(ops, tmp_path, tag, het):
    """BASELINE config 1's shape (4 x 6 MATSim grid, 76 links, SRC/DEST pseudo-nodes, N = 124): the mirror's
    TransportationSimulator.run() with DijkstraAgents, fed the reference's per-step uniforms, reproduces the reference's
    state and agent table after every step and its next-hop tables at every refresh."""
    from src.agents.base import DijkstraAgents
    from src.transportation_simulator import TransportationSimulator
    from tarl_hip import synth
    g = load_golden("routing")
    synth.write_matsim_grid_xml(str(tmp_path / "network.xml"), 4, 6, seed=3, heterogeneous=het)
    synth.write_matsim_population_xml(str(tmp_path / "population.xml"), 4, 6, 260, seed=4, first_departure=21600,
                                      spread=60)
    sim = TransportationSimulator("cuda")
    sim.config_network(str(tmp_path / "network"))
    ag = DijkstraAgents("cuda")
    ag.config_agents_from_xml(str(tmp_path), verbose=False)
    ag.agent_features[0, ag.DEPARTURE_TIME] = 48 * 3600
    sim.agent = ag
    sim.config_parameters(timestep_size=1, start_time=21600)
    ag.set_time(21600)
    E_r = sim.graph.edge_index_routes.size(1)
    for s in range(int(g[f"{tag}__steps"])):
        u = torch.rand(E_r, generator=torch.Generator().manual_seed(900 + s))
        sim.model_core.direction_mpnn.inject_uniform(u)
        sim.run()
        assert torch.equal(sim.graph.x.cpu(), g[f"{tag}__x"][s]), f"state after step {s}"
        assert torch.equal(ag.agent_features.cpu(), g[f"{tag}__agents"][s]), f"agents after step {s}"
        if s in (0, 10, 40):
            assert torch.equal(ag.next_hop_tensor.cpu().to(torch.int16), g[f"{tag}__next_hop_{s}"]), f"table {s}"
    assert float(ag.agent_features[:, ag.DONE].sum()) > 0
    # metric tables of the same run (src/transportation_simulator.py:563-670, :387-451)
    nm = sim.compute_node_metrics(output_dir=str(tmp_path / "out"))
    R = int(sim.graph.num_roads)
    assert len(nm) == R and os.path.exists(tmp_path / "out" / "node_metrics.csv")
    assert torch.equal(torch.tensor([nm[n]["hourly_counts"] for n in range(R)]), g[f"{tag}__nm_counts"])
    assert torch.allclose(torch.tensor([nm[n]["avg_vc"] for n in range(R)]), g[f"{tag}__nm_avg_vc"], rtol=1e-6, atol=0)
    assert torch.allclose(torch.tensor([nm[n]["std_vc"] for n in range(R)]), g[f"{tag}__nm_std_vc"], rtol=1e-5, atol=1e-7)
    assert torch.equal(sim.leg_histogram(), g[f"{tag}__leg_hist"])
    # run_msa on the same end state (src/algorithms/user_equilibrium_msa.py:65-165): all-pairs next-hop table per
    # iteration (float64 costs) + one-thread-per-OD-pair assignment instead of nx.shortest_path per pair
    from src.algorithms.user_equilibrium_msa import run_msa
    for iters in (1, 3, 25):
        fl = run_msa(sim.graph, ag, max_iter=iters)
        got = torch.tensor([fl[i] for i in range(R)], dtype=torch.float64)
        want = g[f"{tag}__msa_{iters}"]
        if het:     # untied shortest paths: the reference's flows, to fp64 rounding
            assert torch.allclose(got, want, rtol=1e-9, atol=1e-9), f"MSA flows after {iters} iterations"
        elif iters == 1:
            # homogeneous grid at free flow: every OD pair has many equal-cost paths and the reference's bidirectional
            # Dijkstra may pick another one — each carries the same number of roads, so the total assigned volume agrees
            assert abs(float(got.sum()) - float(want.sum())) < 1e-9 and float(got.min()) >= 0.0


@pytest.mark.parametrize("kind,scratch", [("torus_hom", False), ("torus_het", False), ("grid_srcdest", False),
                                          ("torus_hom", True), ("batched", False)])
def test_apsp_vs_oracle(ops, tmp_path, monkeypatch, kind, scratch):
    """Heavily tied (homogeneous torus), untied (heterogeneous) and partly unreachable (SRC/DEST) graphs; LDS and
    global-scratch variants; several weight sets in one launch."""
    from oracle import routing
    from tarl_hip import synth
    if scratch:
        monkeypatch.setenv("TARL_APSP_LDS_MAX", "0")
    gen = torch.Generator().manual_seed(11)
    if kind == "grid_srcdest":
        from src.matsim_io import build_network
        synth.write_matsim_grid_xml(str(tmp_path / "network.xml"), 5, 4, seed=2, heterogeneous=False)
        graph, _ = build_network(str(tmp_path / "network"))
        ei, N = graph.edge_index, graph.x.size(0)
        ws = [torch.where(torch.rand(ei.size(1), generator=gen) < 0.5, 10.0, 12.5)]
    else:
        net = synth.torus_network(6, 5, heterogeneous=(kind != "torus_hom"), seed=9)
        ei, N = net.edge_index, net.num_roads
        if kind == "torus_hom":
            ws = [torch.full((ei.size(1),), 10.0)]
        elif kind == "batched":
            ws = [torch.rand(ei.size(1), generator=gen) * 20 + 1 for _ in range(3)]
            ws[1] = torch.round(ws[1])                      # integer weights: many ties
        else:
            ws = [torch.rand(ei.size(1), generator=gen) * 20 + 1]
    plan = ops.Plan(ei, N)
    nh, dist = ops.all_pairs_shortest_paths(plan, torch.stack(ws).cuda(), want_dist=True)
    for b, w in enumerate(ws):
        nh_o, dist_o = routing.all_pairs(ei, w, N)
        assert torch.equal(nh[b].cpu(), nh_o), f"next hop, weight set {b}"
        assert torch.equal(dist[b].cpu(), dist_o), f"distances, weight set {b}"
    if kind == "grid_srcdest":
        assert bool((nh.cpu() == -1).any()) and bool(torch.isinf(dist).any())


def test_main_cli_dijkstra(tmp_path, monkeypatch, capsys):
    """``main.py --algo dijkstra --mode eval`` end to end on a MATSim scenario directory (BASELINE config 1's command
    line shape), caches written like the reference's."""
    import sys
    from tarl_hip import synth
    monkeypatch.chdir(tmp_path)
    os.makedirs("data/grid")
    synth.write_matsim_grid_xml("data/grid/network.xml", 4, 6, seed=3)
    synth.write_matsim_population_xml("data/grid/population.xml", 4, 6, 120, seed=4, first_departure=21600, spread=30)
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tarl-simulator_amd"))
    import main as cli
    cli.main(["--algo", "dijkstra", "--mode", "eval", "--scenario", "grid", "--start-end-time", "21600", "21720",
              "--device", "cuda", "--output-dir", str(tmp_path / "runs")])
    out = capsys.readouterr().out
    assert "Average travel time" in out and os.path.exists("save/grid/network.pt")
    assert os.path.exists("save/grid/population.pt")
