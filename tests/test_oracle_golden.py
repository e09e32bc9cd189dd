"""CPU: pin the oracle (oracle/*.py) to the golden vectors produced by the reference's own source
(oracle/make_golden.py) and to the integer facts of the reference's tests. Integer state bit-exact; fp32 <= 1e-4
(in practice bit-exact too, same torch CPU kernels)."""
import math

import pytest
import torch

from conftest import load_golden
from oracle import sim, dist, nets, ppo


def eq(a, b):
    if not torch.is_tensor(b):
        b = torch.as_tensor(b, dtype=a.dtype).reshape(a.shape)
    return torch.equal(a, b)


@pytest.mark.parametrize("name", ["core_hom", "core_het", "core_noconst"])
def test_core_steps_bit_exact(name):
    g = load_golden(name)
    Nmax, ei, ea = g["Nmax"], g["edge_index"], g["edge_attr"]
    cc = g["congestion_constant"] if g["with_const"] else None
    x = g["x0"].clone()
    pops = 0
    for s in range(g["steps"]):
        t = g[f"t{s}"]
        _, dtt = sim.direction_step(x, ei, ea, t, Nmax, uniform=g[f"u{s}"], congestion_constant=cc)
        assert eq(x, g[f"xd{s}"]), f"direction state differs at step {s}"
        assert eq(dtt, g[f"dtt{s}"])
        _, popped = sim.response_step(x, ei, Nmax)
        assert eq(x, g[f"xr{s}"]), f"response state differs at step {s}"
        assert eq(popped, g[f"pop{s}"])
        pops += int(popped.sum())
    assert pops > 0, "fixture must exercise the pop path"


def test_braess_reference_fixture():
    """tests/conftest.py:45-91 + tests/simulation_core_model_test.py / response_mpnn_test.py facts."""
    g = load_golden("braess")
    x = g["x0"].clone()
    _, dtt, popped = sim.core_step(x, g["edge_index"], g["edge_attr"], 0, g["Nmax"], uniform=g["u0"])
    assert eq(x, g["x1"]) and eq(dtt, g["dtt"])
    assert x.shape == g["x0"].shape                       # shape preserved (the reference's assertion)
    assert int(popped.sum()) == 0 and g["n_history"] == 0  # update_history stays empty
    assert x[:, 3 * g["Nmax"] + 1].tolist() == [1.0, 1.0, 2.0]


def test_agents_tiny_reference_facts():
    """tests/agents_test.py:12-73: insert 2 -> count 2 / ON_WAY; no withdraw before departure; withdraw at t=10;
    capacity limit 5-3=2 admits exactly the first two of four."""
    g = load_golden("agents_tiny")
    Nmax, adj = g["Nmax"], g["adj"]
    c = sim.Cols(Nmax)
    x, ag = g["x0"].clone(), g["agents0"].clone()
    sim.insert(x, ag, 0, Nmax)
    assert eq(x, g["x_ins"]) and eq(ag, g["agents_ins"])
    assert x[0, c.N] == 2 and torch.all(ag[:2, sim.ON_WAY] == 1)
    sim.withdraw(x, ag, adj, 0, Nmax)
    assert eq(x, g["x_w0"]) and x[0, c.N] == 2
    sim.withdraw(x, ag, adj, 10, Nmax)
    assert eq(x, g["x_w10"]) and eq(ag, g["agents_w10"])
    assert x[0, c.N] == 0 and torch.all(ag[:2, sim.DONE] == 1)
    x2, ag2 = g["x0"].clone(), g["cap_agents0"].clone()
    sim.insert(x2, ag2, 0, Nmax)
    assert eq(x2, g["cap_x"]) and eq(ag2, g["cap_agents"])
    assert x2[0, c.N] == 2 and torch.all(ag2[:2, sim.ON_WAY] == 1) and torch.all(ag2[2:, sim.ON_WAY] == 0)


def test_agents_torus_bit_exact():
    g = load_golden("agents_torus")
    Nmax = g["Nmax"]
    x, ag = g["x0"].clone(), g["agents0"].clone()
    n = x.size(0)
    adj = torch.zeros((n, n), dtype=torch.bool)
    adj[g["edge_index"][0], g["edge_index"][1]] = True
    moved = 0
    for s in range(g["steps"]):
        t = g[f"t{s}"]
        _, wmask = sim.withdraw(x, ag, adj, t, Nmax)
        assert eq(x, g[f"xw{s}"]) and eq(ag, g[f"aw{s}"]) and eq(wmask, g[f"wmask{s}"])
        before = ag[:, sim.ON_WAY].sum()
        sim.insert(x, ag, t, Nmax, g["congestion_constant"])
        assert eq(x, g[f"xi{s}"]) and eq(ag, g[f"ai{s}"])
        moved += int(wmask.sum()) + int(ag[:, sim.ON_WAY].sum() - before)
    assert moved > 0


@pytest.mark.parametrize("name", ["dist_small", "dist_mid"])
def test_graphdist(name):
    g = load_golden(name)
    ei = g["edge_index"]
    d = dist.GraphDist(g["logits"], ei)
    assert d.nb_nodes == g["nb_nodes"]
    assert eq(d.proba, g["proba"]) and eq(d.cumsum, g["cumsum_sorted"])
    assert eq(d.entropy(), g["entropy"]) and eq(d.mode, g["mode"])
    for k in range(4):
        a = d.sample(g[f"u{k}"])
        assert a.dtype == torch.int64 and eq(a, g[f"a{k}"])            # int64 one-hot, bit-exact
        assert eq(d.log_prob(a), g[f"lp{k}"])
    assert d.log_prob(g["bad"]).item() == -math.inf and float(g["lp_bad"]) == -math.inf
    lb = g["logits_b"].clone().requires_grad_(True)
    db = dist.GraphDist(lb, ei)
    lp, ent = db.log_prob(g["acts_b"]), db.entropy()
    assert torch.allclose(lp, g["lp_b"], atol=1e-6, rtol=0) and torch.allclose(ent, g["ent_b"], atol=1e-6, rtol=0)
    (lp * g["w_b"]).sum().backward(retain_graph=True)
    assert torch.allclose(lb.grad, g["grad_lp_b"], atol=1e-6, rtol=0)
    lb.grad = None
    (ent * g["w_b"]).sum().backward()
    assert torch.allclose(lb.grad, g["grad_ent_b"], atol=1e-6, rtol=0)


@pytest.mark.parametrize("name", ["env_hom", "env_het"])
def test_env_rollout_bit_exact(name):
    g = load_golden(name)
    Nmax, ei, ea, cc = g["Nmax"], g["edge_index"], g["edge_attr"], g["congestion_constant"]
    x = g["x_init"].clone()
    ag = g["agents0"].clone()
    # _reset: FIFO blocks + counter zeroed, agents' ON_WAY / DONE cleared, clock = 6h - 60s
    x[:, :3 * Nmax] = 0
    x[:, 3 * Nmax + 1] = 0
    ag[:, sim.ON_WAY] = 0
    ag[:, sim.DONE] = 0
    t = g["time0"]
    assert t == 6 * 3600 - 60
    nf, ai = sim.observe(x, Nmax)
    assert eq(nf, g["obs0_node"]) and eq(ai, g["obs0_agent_index"])
    n = x.size(0)
    adj = torch.zeros((n, n), dtype=torch.bool)
    adj[ei[0], ei[1]] = True
    for s in range(g["T"]):
        logits = nets.policy_logits(sim.observe(x, Nmax)[0], ei, g["w_emb"])
        d = dist.GraphDist(logits, ei)
        a = d.sample(g["u_sample"][s])
        assert eq(a, g["action"][s]), f"action differs at step {s}"
        assert torch.allclose(d.log_prob(a), g["log_prob"][s], atol=1e-6, rtol=0)
        out = sim.env_step(x, ag, ei, ea, adj, a, t, Nmax, uniform=g["u_dir"][s], congestion_constant=cc)
        t = out["time"]
        assert eq(x, g["x"][s]), f"state differs at step {s}"
        assert eq(ag, g["agents"][s]), f"agents differ at step {s}"
        assert eq(out["reward"], g["reward"][s]) and t == int(g["time"][s])
        assert eq(out["delta_travel_time"], g["dtt"][s])
    assert float(ag[:, sim.DONE].sum()) == g["done_total"] > 0


def test_nets_forward():
    g = load_golden("nets")
    ei = g["edge_index"]
    logits = nets.policy_logits(g["node_features"], ei, g["pol__nodes_embedding__weight"])
    assert eq(logits, g["logits"])
    assert eq(nets.policy_logits(g["node_features_b"], ei, g["pol__nodes_embedding__weight"]), g["logits_b"])
    w = [g[f"val__final_mlp__{i}__{p}"] for i in (0, 2, 4) for p in ("weight", "bias")]
    v = nets.critic_value(g["node_features"], g["time"], *w)
    assert torch.allclose(v, g["value"], atol=1e-5, rtol=1e-6)
    vb = nets.critic_value(g["node_features_b"], g["time_b"], *w)
    assert torch.allclose(vb, g["value_b"], atol=1e-5, rtol=1e-6)
    # state-dict key contract of the reference modules
    for k in ["pol__nodes_embedding__weight", "pol__edge_mlp__0__weight", "pol__edge_mlp__4__bias",
              "pol__edge_mlp_test__0__weight", "pol__edge_mlp_test__2__bias", "val__final_mlp__0__weight"]:
        assert k in g


def test_adam_matches_torch_optim():
    """torch.optim.Adam(lr=1e-3) is present in this image, so this part of a17 IS pinned."""
    torch.manual_seed(0)
    p = torch.randn(257, requires_grad=True)
    q = p.detach().clone()
    m, v = torch.zeros_like(q), torch.zeros_like(q)
    opt = torch.optim.Adam([p], lr=1e-3)
    for step in range(1, 6):
        grad = torch.randn(257)
        p.grad = grad.clone()
        opt.step()
        ppo.adam_step(q, grad, m, v, step)
        assert torch.allclose(p.detach(), q, atol=1e-7, rtol=1e-6)


def test_gae_and_loss_formulas():
    """Formula-level self-checks (parity unpinned by the reference, see oracle/ppo.py)."""
    torch.manual_seed(1)
    T = 17
    r, v, nv = torch.randn(T, 1), torch.randn(T, 1), torch.randn(T, 1)
    done = torch.zeros(T, 1, dtype=torch.bool)
    done[9] = True
    adv, tgt = ppo.gae(r, v, nv, done, done, average_gae=False)
    # brute force
    delta = r + 0.99 * nv * (~done).float() - v
    for t in range(T):
        acc, coef = 0.0, 1.0
        for k in range(t, T):
            acc += coef * delta[k, 0].item()
            if done[k, 0]:
                break
            coef *= 0.99 * 0.95
        assert abs(adv[t, 0].item() - acc) < 1e-4
    assert torch.allclose(tgt, adv + v)
    advn, _ = ppo.gae(r, v, nv, done, done, average_gae=True)
    assert abs(advn.mean().item()) < 1e-6 and abs(advn.std().item() - 1) < 1e-5
    out = ppo.clip_ppo_loss(torch.zeros(8), torch.zeros(8), torch.ones(8), torch.zeros(8), torch.ones(8) * 3, torch.ones(8))
    assert abs(out["loss_objective"].item() + 1.0) < 1e-6      # ratio 1 => -mean(A)
    assert abs(out["loss_critic"].item() - 2.5) < 1e-6          # smooth_l1(3) = 2.5
    assert abs(out["loss_entropy"].item() + 0.01) < 1e-7


# ---- shortest-path routing -----------------------------------------------------------------------------
def _matsim_grid(tmp_path, het):
    """The same MATSim documents the golden generator fed to the reference (tarl_hip.synth writers, seeded), parsed by
    the build's own builders (themselves pinned by tests/test_builders.py)."""
    from tarl_hip import synth
    from src.matsim_io import build_network, build_population
    synth.write_matsim_grid_xml(str(tmp_path / "network.xml"), 4, 6, seed=3, heterogeneous=het)
    synth.write_matsim_population_xml(str(tmp_path / "population.xml"), 4, 6, 260, seed=4, first_departure=21600,
                                      spread=60)
    graph, Nmax = build_network(str(tmp_path / "network"))
    agents, _ = build_population(str(tmp_path / "population"), str(tmp_path / "network"))
    agents[0, 2] = 48 * 3600
    return graph, Nmax, agents


@pytest.mark.parametrize("tag,het", [("grid", False), ("gridhet", True)])
def test_dijkstra_classical_run_vs_reference(tmp_path, tag, het):
    """insert -> withdraw -> DijkstraAgents.choice (all-pairs refresh every 10 calls) -> core, replayed by the oracle
    with the reference's per-step uniforms: state, agents and the next-hop tables equal the reference's (networkx)."""
    from oracle import routing, sim
    g = load_golden("routing")
    graph, Nmax, agents = _matsim_grid(tmp_path, het)
    assert agents.size(0) == int(g[f"{tag}__agents0_n"])
    x, R = graph.x.clone(), int(graph.num_roads)
    adj = graph.adj_matrix
    next_hop = None
    for s in range(int(g[f"{tag}__steps"])):
        t = 21600 + s
        sim.insert(x, agents, t, Nmax, graph.congestion_constant)
        sim.withdraw(x, agents, adj, t, Nmax)
        if s % 10 == 0:
            next_hop = None
        x, next_hop = routing.dijkstra_choice(x, agents, graph.edge_index, graph.congestion_constant, Nmax, next_hop)
        if s in (0, 10, 40):
            assert torch.equal(next_hop.to(torch.int16), g[f"{tag}__next_hop_{s}"]), f"next-hop table at step {s}"
        u = torch.rand(graph.edge_index_routes.size(1), generator=torch.Generator().manual_seed(900 + s))
        sim.core_step(x[:R], graph.edge_index_routes, graph.edge_attr_routes, t, Nmax, uniform=u,
                      congestion_constant=graph.congestion_constant[:R])
        assert torch.equal(x, g[f"{tag}__x"][s]), f"state after step {s}"
        assert torch.equal(agents, g[f"{tag}__agents"][s]), f"agents after step {s}"
    assert float(agents[:, 8].sum()) > 0


def test_dijkstra_choice_and_prior_on_torus_vs_reference():
    from oracle import routing
    from tarl_hip import synth
    g = load_golden("routing")
    net = synth.torus_network(3, 3, heterogeneous=True, seed=int(g["torus__seed"]))
    x1, nh = routing.dijkstra_choice(g["torus__x0"], g["torus__agents"], net.edge_index, net.congestion_constant,
                                     net.Nmax)
    assert torch.equal(nh.to(torch.int16), g["torus__next_hop"]) and torch.equal(x1, g["torus__x1"])
    _, dist = routing.all_pairs(net.edge_index, g["torus__ff_edges"], net.num_roads)
    assert torch.equal(dist, g["torus__dist_matrix"])
    prior = -dist[net.edge_index[1], g["torus__prior_dest"]] - g["torus__ff_edges"]
    assert torch.equal(prior, g["torus__prior_logits"])


def test_value_mpnn_oracle_vs_reference():
    """MPNNValueNet (the reference's dormant message-passing critic, eval mode): oracle restatement vs the reference's
    own class with the same weights — unbatched and batched."""
    from oracle import nets
    g = load_golden("value_mpnn")
    sd = {k.replace("__", "."): v for k, v in g.items() if "__" in k}
    v = nets.value_mpnn(sd, g["edge_index"], g["agent_features"], g["node_features"].unsqueeze(0),
                        g["edge_attr"].view(1, -1), g["agent_index"].unsqueeze(0), g["time"].view(-1))
    assert torch.allclose(v, g["value"], rtol=1e-6, atol=1e-7)
    vb = nets.value_mpnn(sd, g["edge_index"], g["agent_features"], g["node_features_b"], g["edge_attr_b"].squeeze(-1),
                         g["agent_index_b"], g["time_b"].view(-1))
    assert torch.allclose(vb, g["value_b"].view(-1), rtol=1e-6, atol=1e-7)


def test_edge_mlp_oracle_matches_reference_module():
    """oracle.nets.edge_mlp_logits == the reference's own edge_mlp module applied as src/agents/mpnn_agent.py:227-231
    specifies (fixture generated by oracle/make_golden.py:gen_edge_mlp), and autograd through it gives the recorded
    parameter gradients."""
    from oracle import nets
    g = load_golden("edge_mlp")
    x16 = torch.cat((g["node_features"], g["agent_features"][g["agent_index"]]), dim=-1)
    for tag in ("ref", "biased"):
        ws = [g[f"{tag}__{k}"].clone().requires_grad_(True) for k in ("0__weight", "0__bias", "2__weight", "2__bias",
                                                                        "4__weight", "4__bias")]
        logits = nets.edge_mlp_logits(x16, g["edge_index"], g["edge_attr"].expand(3, -1, -1), *ws)
        assert torch.equal(logits, g[f"{tag}__logits"])
        (logits * g["coef"]).sum().backward()
        for w_, k in zip(ws, ("0__weight", "0__bias", "2__weight", "2__bias", "4__weight", "4__bias")):
            assert torch.allclose(w_.grad, g[f"{tag}__grad__{k}"], rtol=1e-6, atol=1e-3)
