#!/usr/bin/env python3
"""Generate tests/golden/*.npz by executing the REFERENCE's own source (TEST INFRASTRUCTURE, build container only).

Usage (from the repo root, in the container that has /root/reference):

    python oracle/make_golden.py

The reference's hot-path modules are imported unchanged from ``/root/reference`` with ``oracle/refshim`` first on
``sys.path`` (stand-ins for the absent torch-geometric / torch-scatter / torchrl / tensordict / lxml wheels — see
``oracle/refshim/README.md``). Inputs come from the build's seeded synthetic generator; all randomness the reference
draws from torch's global generator is pinned by re-seeding immediately before each call and recording the very same
draws, so every fixture carries its noise explicitly. Only data (inputs + the reference's outputs) is written; no
reference source text goes into the fixtures. Nothing here runs on the GPU box (``/root/reference`` is absent there).
"""
from __future__ import annotations

import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("TARL_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(ROOT, "oracle", "refshim"), REF, os.path.join(ROOT, "tarl-simulator_amd")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

warnings.filterwarnings("ignore")

# --- environment pin: CPU sort stability -------------------------------------------------------------------------
# The reference calls ``torch.sort`` / ``torch.argsort`` without ``stable=True`` where the order of equal keys matters
# (src/reinforcement_learning.py:21, src/agents/base.py:275). Its pinned torch 2.5.1 (requirements.txt:2) has a single
# CPU sort kernel that is always stable; this image's torch 2.10 routes unstable calls to x86-simd-sort, which is not
# (observed: 64 already-sorted keys with duplicates come back permuted). To reproduce the reference *in its pinned
# environment* the generator makes stable the default for the two function-form calls the reference uses. The
# reference source itself is executed unchanged.
_sort, _argsort = torch.sort, torch.argsort


def _stable_sort(input, dim=-1, descending=False, stable=True, **kw):
    return _sort(input, dim=dim, descending=descending, stable=stable, **kw)


def _stable_argsort(input, dim=-1, descending=False, stable=True):
    return _argsort(input, dim=dim, descending=descending, stable=stable)


torch.sort, torch.argsort = _stable_sort, _stable_argsort

from torch_geometric.data import Data  # noqa: E402  (stand-in)
from src.direction_mpnn import DirectionMPNN  # noqa: E402  (reference)
from src.response_mpnn import ResponseMPNN  # noqa: E402
from src.simulation_core_model import SimulationCoreModel  # noqa: E402
from src.feature_helpers import FeatureHelpers  # noqa: E402
from src.agents.base import Agents  # noqa: E402
from src.transportation_simulator import TransportationSimulator  # noqa: E402
from src.reinforcement_learning import GraphDistribution, SimulatorEnv  # noqa: E402
from tensordict import TensorDict  # noqa: E402  (stand-in)

from tarl_hip import synth  # noqa: E402  (the build's own input generator)

OUT = os.path.join(ROOT, "tests", "golden")


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrays.items():
        conv[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"{name}: {os.path.getsize(path)} bytes")


def draws(seed, n):
    """The n uniforms the reference will draw from the global generator after ``torch.manual_seed(seed)``."""
    torch.manual_seed(seed)
    u = torch.rand(n)
    torch.manual_seed(seed)
    return u


# ----------------------------------------------------------------------------------------------------------------
def gen_core_steps():
    """DirectionMPNN / ResponseMPNN / SimulationCoreModel on random mid-simulation states (a2-a8)."""
    # NOTE: states in which gridlock relief over-fills a FIFO (count reaches Nmax) are outside the reference's defined
    # domain: its unconditional slot writes then overwrite MAX_NUMBER_OF_AGENT / NUMBER_OF_AGENT and it raises
    # IndexError one or two steps later (observed here on an 8x8 torus). No fixture can pin that corner.
    for tag, (W, H, het, with_const, sseed) in {"core_hom": (2, 3, False, True, 5), "core_het": (3, 2, True, True, 5),
                                                "core_noconst": (2, 2, True, False, 5)}.items():
        net = synth.torus_network(W, H, heterogeneous=het, seed=11)
        x = synth.random_state(net, seed=sseed, t=100.0)
        E = net.edge_index.size(1)
        g = Data(x=x.clone(), edge_index=net.edge_index, edge_attr=net.edge_attr,
                 edge_index_routes=net.edge_index, edge_attr_routes=net.edge_attr, num_roads=net.num_roads)
        if with_const:
            g.critical_number = net.critical_number
            g.congestion_constant = net.congestion_constant
        core = SimulationCoreModel(Nmax=net.Nmax, device="cpu", time=100)
        rec = {"x0": x, "edge_index": net.edge_index, "edge_attr": net.edge_attr, "Nmax": net.Nmax,
               "congestion_constant": net.congestion_constant, "with_const": int(with_const)}
        steps = 6
        for s in range(steps):
            t = 100 + s
            core.set_time(t)
            rec[f"u{s}"] = draws(1000 + s, E)
            hist_before = len(core.response_mpnn.update_history)
            # run the two rounds separately so the intermediate state is recorded too
            xr = g.x[:net.num_roads]
            kw = {}
            if with_const:
                kw = dict(critical_number=g.critical_number, congestion_constant=g.congestion_constant)
            else:
                h = core.direction_mpnn
                crit = g.x[:, h.MAX_FLOW] * g.x[:, h.FREE_FLOW_TIME_TRAVEL] / 3600
                kw = dict(critical_number=crit,
                          congestion_constant=g.x[:, h.FREE_FLOW_TIME_TRAVEL] * (g.x[:, h.MAX_NUMBER_OF_AGENT] + 10 - crit))
            xd = core.direction_mpnn(xr, g.edge_index_routes, g.edge_attr_routes, **kw)
            rec[f"xd{s}"] = xd.clone()
            rec[f"dtt{s}"] = core.direction_mpnn.road_optimality_data["delta_travel_time"].clone()
            xo = core.response_mpnn(xd, g.edge_index_routes, g.edge_attr_routes)
            rec[f"xr{s}"] = xo.clone()
            hist = core.response_mpnn.update_history
            rec[f"pop{s}"] = hist[-1][1].clone() if len(hist) > hist_before else torch.zeros(net.num_roads, dtype=torch.bool)
            rec[f"t{s}"] = t
        rec["steps"] = steps
        rec["max_count"] = float(g.x[:, 3 * net.Nmax + 1].max())
        save(tag, **rec)


def gen_braess():
    """The reference's own Braess fixture (tests/conftest.py:45-91) through SimulationCoreModel — its tests pin the
    output shape and an empty update_history; SURVEY §8c adds counts [1,1,2]."""
    Nmax = 100
    F = 3 * Nmax + 7
    x = torch.zeros(3, F)
    vals = [(2, 1, 3.0, 100.0, 10.0, 1, 0), (2, 1, 1.0, 100.0, 10.0, 2, 1), (2, 2, 1.0, 100.0, 10.0, 0, 2)]
    for r, v in enumerate(vals):
        x[r, 3 * Nmax:3 * Nmax + 7] = torch.tensor(v)
    x[0, 0], x[1, 0], x[2, 0], x[2, 1] = 1.0, 2.0, 3.0, 4.0
    x[2, 2 * Nmax + 1] = 1.0
    ei = torch.tensor([[0, 1, 2], [1, 2, 0]])
    torch.manual_seed(3)
    ea = torch.rand(3, 1)
    g = Data(x=x.clone(), edge_index=ei, edge_attr=ea, edge_index_routes=ei, edge_attr_routes=ea, num_roads=3)
    core = SimulationCoreModel(Nmax=Nmax, device="cpu", time=0)
    u = draws(77, 3)
    out = core(g)
    save("braess", x0=x, edge_index=ei, edge_attr=ea, Nmax=Nmax, u0=u, x1=out.x.clone(),
         n_history=len(core.response_mpnn.update_history),
         dtt=core.direction_mpnn.road_optimality_data["delta_travel_time"])


def gen_agents_tests():
    """Inputs/outputs of the reference's tests/agents_test.py:12-73 plus a random torus case for insert/withdraw."""
    h = FeatureHelpers(Nmax=5)

    def tiny_graph():
        x = torch.zeros((2, 3 * h.Nmax + 7))
        x[0, h.MAX_NUMBER_OF_AGENT] = 5
        x[0, h.ROAD_INDEX] = 0
        x[0, h.FREE_FLOW_TIME_TRAVEL] = 10
        ei = torch.tensor([[1, 0], [0, 0]])
        adj = torch.zeros((2, 2), dtype=torch.bool)
        adj[ei[0], ei[1]] = 1
        return Data(x=x, edge_index=ei, edge_index_routes=torch.empty((2, 0), dtype=torch.long),
                    edge_attr_routes=torch.empty((0, 1)), num_roads=1, adj_matrix=adj)

    ag = Agents("cpu")
    ag.agent_features = torch.tensor([[1.0, 0, 0, 0, 30.0, 0, 1.0, 0, 0], [1.0, 0, 0, 0, 25.0, 1.0, 0, 0, 0]])
    g = tiny_graph()
    rec = {"Nmax": 5, "x0": g.x.clone(), "agents0": ag.agent_features.clone(), "adj": g.adj_matrix}
    ag.time = 0
    g.x = ag.insert_agent_into_network(g, h)
    rec["x_ins"], rec["agents_ins"] = g.x.clone(), ag.agent_features.clone()
    g.x = ag.withdraw_agent_from_network(g, h)
    rec["x_w0"] = g.x.clone()
    ag.time = 10
    g.x = ag.withdraw_agent_from_network(g, h)
    rec["x_w10"], rec["agents_w10"] = g.x.clone(), ag.agent_features.clone()
    # capacity limit (4 ready agents, room for 5-3 = 2)
    ag2 = Agents("cpu")
    ag2.agent_features = torch.zeros(4, 9)
    ag2.agent_features[:, 0] = 1.0
    g2 = tiny_graph()
    ag2.time = 0
    rec["cap_agents0"] = ag2.agent_features.clone()
    g2.x = ag2.insert_agent_into_network(g2, h)
    rec["cap_x"], rec["cap_agents"] = g2.x.clone(), ag2.agent_features.clone()
    save("agents_tiny", **rec)

    # random torus case: several insert/withdraw rounds interleaved with time
    net = synth.torus_network(3, 3, heterogeneous=True, seed=2)
    x = synth.random_state(net, seed=9, t=50.0, num_agents=400)
    pop = synth.population(400, net.num_roads, seed=4, t0=30, t1=70)
    # agents that are already queued must be marked ON_WAY
    ids = x[:, :net.Nmax][torch.arange(net.Nmax).unsqueeze(0) < x[:, 3 * net.Nmax + 1].unsqueeze(1)].long()
    pop[ids, 7] = 1.0
    hh = FeatureHelpers(Nmax=net.Nmax)
    g = Data(x=x.clone(), edge_index=net.edge_index, edge_attr=net.edge_attr, edge_index_routes=net.edge_index,
             edge_attr_routes=net.edge_attr, num_roads=net.num_roads, adj_matrix=net.dense_adjacency(),
             congestion_constant=net.congestion_constant, critical_number=net.critical_number)
    ag = Agents("cpu")
    ag.agent_features = pop.clone()
    rec = {"Nmax": net.Nmax, "x0": x, "agents0": pop, "edge_index": net.edge_index,
           "congestion_constant": net.congestion_constant}
    for s, t in enumerate([50, 55, 60, 70, 80]):
        ag.time = t
        g.x = ag.withdraw_agent_from_network(g, hh)
        rec[f"xw{s}"], rec[f"aw{s}"] = g.x.clone(), ag.agent_features.clone()
        rec[f"wmask{s}"] = ag.withdraw_history[-1][1].clone()
        g.x = ag.insert_agent_into_network(g, hh)
        rec[f"xi{s}"], rec[f"ai{s}"] = g.x.clone(), ag.agent_features.clone()
        rec[f"t{s}"] = t
    rec["steps"] = 5
    save("agents_torus", **rec)


def gen_graphdist():
    """GraphDistribution (a11-a13) on torus topologies, unbatched and batched."""
    for tag, (W, H, scale) in {"dist_small": (2, 2, 1.0), "dist_mid": (5, 4, 3.0)}.items():
        net = synth.torus_network(W, H)
        E = net.edge_index.size(1)
        perm = torch.randperm(E, generator=torch.Generator().manual_seed(8))
        ei = net.edge_index[:, perm]          # edges NOT sorted by source: exercises the sort/inverse path
        torch.manual_seed(21)
        logits = torch.randn(E) * scale
        d = GraphDistribution(logits, ei)
        rec = {"edge_index": ei, "logits": logits, "proba": d.proba, "cumsum_sorted": d.cumsum,
               "entropy": d.entropy(), "mode": d.mode, "nb_nodes": d.nb_nodes}
        for k in range(4):
            u = draws(300 + k, d.nb_nodes)
            a = d.sample()
            rec[f"u{k}"], rec[f"a{k}"], rec[f"lp{k}"] = u, a, d.log_prob(a)
        # batched log_prob / entropy with gradients
        torch.manual_seed(22)
        lb = (torch.randn(3, E) * scale).requires_grad_(True)
        db = GraphDistribution(lb, ei)
        acts = torch.stack([rec["a0"], rec["a1"], rec["a2"]])
        lp = db.log_prob(acts)
        ent = db.entropy()
        w = torch.tensor([0.3, -1.1, 0.7])
        (lp * w).sum().backward(retain_graph=True)
        g_lp = lb.grad.clone()
        lb.grad = None
        (ent * w).sum().backward()
        rec.update(logits_b=lb.detach(), acts_b=acts, lp_b=lp.detach(), ent_b=ent.detach(), w_b=w,
                   grad_lp_b=g_lp, grad_ent_b=lb.grad.clone(), proba_b=db.proba.detach())
        bad = rec["a0"].clone()
        bad[ei[0] == 0] = 0                   # node 0 selects nothing -> infeasible
        rec["bad"], rec["lp_bad"] = bad, d.log_prob(bad)
        save(tag, **rec)


def gen_env_rollout():
    """SimulatorEnv._reset/_step (a14-a15) + insert/withdraw on a pure road graph, actions sampled from the reference's
    GraphDistribution. The scenario loader is replaced by the synthetic network exactly like the reference's own
    tests/rl_metrics_test.py:10-13 replaces it with a test network."""
    for tag, (W, H, het, n_agents, T) in {"env_hom": (2, 2, False, 150, 90), "env_het": (3, 2, True, 300, 90)}.items():
        net = synth.torus_network(W, H, heterogeneous=het, seed=6)

        def fake_load(self, scenario, net=net):
            self.graph = Data(x=net.x.clone(), edge_index=net.edge_index, edge_attr=net.edge_attr,
                              edge_index_routes=net.edge_index, edge_attr_routes=net.edge_attr,
                              num_roads=net.num_roads, adj_matrix=net.dense_adjacency(),
                              critical_number=net.critical_number, congestion_constant=net.congestion_constant)
            self.Nmax = net.Nmax
            self.h = FeatureHelpers(Nmax=net.Nmax)

        orig = TransportationSimulator.load_network
        TransportationSimulator.load_network = fake_load
        try:
            env = SimulatorEnv(device="cpu", timestep_size=1, start_time=0, scenario="synthetic")
        finally:
            TransportationSimulator.load_network = orig
        pop = synth.population(n_agents, net.num_roads, seed=3, t0=synth.EPISODE_START, t1=synth.EPISODE_START + 50)
        ag = Agents("cpu")
        ag.agent_features = pop.clone()
        env.simulator.agent = ag
        td = env._reset()
        E = net.edge_index.size(1)
        rec = {"Nmax": net.Nmax, "x_init": net.x, "agents0": pop, "edge_index": net.edge_index,
               "edge_attr": net.edge_attr, "congestion_constant": net.congestion_constant, "T": T,
               "time0": env.simulator.time, "obs0_node": td["node_features"].clone(),
               "obs0_agent_index": td["agent_index"].clone()}
        torch.manual_seed(5)
        W_emb = torch.randn(net.num_roads)     # stands for nn.Embedding(num_nodes, 1).weight[:, 0]
        rec["w_emb"] = W_emb
        xs, ags, acts, us, udir, rew, times, dtts, lps = [], [], [], [], [], [], [], [], []
        for s in range(T):
            logits = W_emb[env.simulator.graph.x[:, env.simulator.h.ROAD_INDEX].long()][net.edge_index[1]]
            d = GraphDistribution(logits, net.edge_index)
            u = draws(5000 + s, d.nb_nodes)
            a = d.sample()
            lps.append(d.log_prob(a))
            ud = draws(9000 + s, E)
            out = env._step(TensorDict({"action": a}, batch_size=[]))
            us.append(u); acts.append(a); udir.append(ud)
            xs.append(env.simulator.graph.x.clone()); ags.append(ag.agent_features.clone())
            rew.append(out["reward"].clone()); times.append(env.simulator.time)
            dtts.append(env.simulator.model_core.direction_mpnn.road_optimality_data["delta_travel_time"].clone())
        rec.update(x=torch.stack(xs), agents=torch.stack(ags), action=torch.stack(acts), u_sample=torch.stack(us),
                   u_dir=torch.stack(udir), reward=torch.stack(rew), time=torch.tensor(times), dtt=torch.stack(dtts),
                   log_prob=torch.stack(lps), n_pop_events=len(env.simulator.model_core.response_mpnn.update_history),
                   done_total=float(ag.agent_features[:, 8].sum()))
        print(f"  {tag}: agents done {rec['done_total']:.0f}/{n_agents}, pop events {rec['n_pop_events']}")
        save(tag, **rec)


def gen_nets():
    """Live policy / critic forward (a9-a10): MPNNPolicyNet logits and MPNNValueNetSimple values with saved weights."""
    from src.agents.mpnn_agent import MPNNPolicyNet, MPNNValueNetSimple
    net = synth.torus_network(2, 3)
    R = net.num_roads
    ff = net.x[:, 3 * net.Nmax + 2][net.edge_index[1]]
    torch.manual_seed(13)
    pol = MPNNPolicyNet(net.edge_index, R, ff, device="cpu")
    pol.agent_features = synth.population(40, R, seed=1)
    val = MPNNValueNetSimple(net.edge_index, R, device="cpu")
    x = synth.random_state(net, seed=12, t=30.0, num_agents=40)
    node_features = x[:, 3 * net.Nmax:]
    agent_index = x[:, 0].long().clamp(max=40)
    time = torch.tensor([21600.0])
    logits = pol(node_features, net.edge_attr, agent_index)
    value = val(node_features, net.edge_attr, agent_index, time)
    nb = torch.stack([node_features, node_features.flip(0) * 1.0])
    nb[1, :, 6] = node_features[:, 6]          # keep ROAD_INDEX valid in the second batch row
    tb = torch.tensor([[21600.0], [21700.0]])
    value_b = val(nb, net.edge_attr.expand(2, -1, -1), agent_index.expand(2, -1), tb)
    logits_b = pol(nb, net.edge_attr.expand(2, -1, -1), agent_index.expand(2, -1))
    sd = {("pol." + k): v for k, v in pol.state_dict().items()}
    sd.update({("val." + k): v for k, v in val.state_dict().items()})
    save("nets", edge_index=net.edge_index, edge_attr=net.edge_attr, node_features=node_features,
         agent_index=agent_index, time=time, logits=logits.detach(), value=value.detach(), node_features_b=nb,
         time_b=tb, value_b=value_b.detach(), logits_b=logits_b.detach(),
         **{k.replace(".", "__"): v for k, v in sd.items()})


def gen_edge_mlp():
    """The per-edge MLP head MPNNPolicyNet carries as parameters (src/agents/mpnn_agent.py:35-41), evaluated the way the
    commented lines of its update_edges specify (:227-231) on x = cat(node_features, agent_features[agent_index])
    (:166-178): the reference's own module (its own U(-0.1, 0.1) weights and zero biases), a second copy of it with
    non-zero biases, unbatched and batched inputs, and the parameter gradients of sum(coef * logits) by autograd."""
    from src.agents.mpnn_agent import MPNNPolicyNet
    net = synth.torus_network(3, 3, heterogeneous=True, seed=6)
    R, E = net.num_roads, net.edge_index.size(1)
    ff = net.x[:, 3 * net.Nmax + 2][net.edge_index[1]]
    torch.manual_seed(21)
    pol = MPNNPolicyNet(net.edge_index, R, ff, device="cpu")
    pol.agent_features = synth.population(60, R, seed=3)
    pol.agent_features[1:, 3] = torch.rand(60, generator=torch.Generator().manual_seed(4)) * 50.0   # arrival times
    pol.agent_features[1:, 7] = (torch.rand(60, generator=torch.Generator().manual_seed(5)) < 0.5).float()
    rec = dict(edge_index=net.edge_index, edge_attr=net.edge_attr, agent_features=pol.agent_features.clone())
    g = torch.Generator().manual_seed(8)
    nfs, ais = [], []
    for b in range(3):
        x = synth.random_state(net, seed=30 + b, t=30.0, num_agents=60)
        nfs.append(x[:, 3 * net.Nmax:].clone())
        ais.append(x[:, 0].long().clamp(max=60))
    nf, ai = torch.stack(nfs), torch.stack(ais)                                  # (3, R, 7), (3, R)
    coef = torch.randn((3, E), generator=g)
    for tag, biased in (("ref", False), ("biased", True)):
        mlp = pol.edge_mlp
        if biased:
            with torch.no_grad():
                for lin in (mlp[0], mlp[2], mlp[4]):
                    lin.bias.copy_(torch.randn(lin.bias.shape, generator=g) * 0.1)
        for p_ in mlp.parameters():
            p_.grad = None
        x16 = torch.cat((nf, pol.agent_features[ai]), dim=-1)                    # (:177-178), batched
        ei = net.edge_index
        e_ij = torch.cat([x16[:, ei[0]], x16[:, ei[1]], net.edge_attr.expand(3, -1, -1)], dim=-1)   # (:228-230)
        logits = mlp(e_ij).squeeze(-1)                                           # (:231) -> (3, E)
        (logits * coef).sum().backward()
        rec.update({f"{tag}__logits": logits.detach().clone()})
        for k, v in mlp.state_dict().items():
            rec[f"{tag}__{k.replace('.', '__')}"] = v.clone()
        for k, p_ in mlp.named_parameters():
            rec[f"{tag}__grad__{k.replace('.', '__')}"] = p_.grad.clone()
    rec.update(node_features=nf, agent_index=ai, coef=coef)
    save("edge_mlp", **rec)


def gen_figures():
    """The series behind the reference's analysis figures (src/transportation_simulator.py:387-517, 672-745), read back
    from the matplotlib artists its own plot_* methods create: the binned leg histogram at three timestep sizes, the
    per-road delta-travel-time lines, and the simulated-vs-expected daily counts. Inputs are seeded series of the shape
    run() / _step record them (lists of [departures, arrivals, on the way, clock], (clock, per-edge tensor), (clock, mask))."""
    import contextlib
    import io
    import types
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    g = torch.Generator().manual_seed(77)
    rec = {}
    with contextlib.redirect_stdout(io.StringIO()):
        for dt in (1, 2, 6):
            T = 200 // dt + 7
            dep = torch.randint(0, 9, (T,), generator=g)
            arr = torch.randint(0, 7, (T,), generator=g)
            on = torch.cumsum(dep - arr, 0).clamp(min=0)
            clock = 6 * 3600 - 60 + dt * torch.arange(1, T + 1)
            vals = torch.stack([dep, arr, on, clock], 1)
            sim = TransportationSimulator("cpu")
            sim.timestep = dt
            # rows as the reference records them: tensors for the counts, a python number for the clock
            sim.leg_histogram_values = [[vals[i, 0], vals[i, 1], vals[i, 2], int(vals[i, 3])] for i in range(T)]
            fig = sim.plot_leg_histogram(output_dir=None)
            lines = fig.axes[0].lines
            rec[f"leg{dt}__values"] = vals
            rec[f"leg{dt}__minutes"] = np.asarray(lines[0].get_xdata(), dtype=np.float64)
            for k, name in enumerate(("on", "dep", "arr")):
                rec[f"leg{dt}__{name}"] = np.asarray(lines[k].get_ydata(), dtype=np.float64)
            plt.close(fig)
        # road optimality: 5 roads, 11 route edges, 9 steps
        R, E, T = 5, 11, 9
        src = torch.randint(0, R, (E,), generator=g)
        dtt = torch.rand((T, E), generator=g) * 40.0 - 5.0
        clocks = [6 * 3600 + 30 * i for i in range(T)]
        sim = TransportationSimulator("cpu")
        sim.graph = types.SimpleNamespace(num_roads=R, edge_index_routes=torch.stack([src, torch.zeros_like(src)]))
        sim.road_optimality_values = [(clocks[i], dtt[i]) for i in range(T)]
        fig = sim.plot_road_optimality(output_dir=None)
        lines = fig.axes[0].lines
        rec.update(opt__src=src, opt__dtt=dtt, opt__clocks=np.asarray(clocks),
                   opt__hours=np.asarray(lines[0].get_xdata(), dtype=np.float32),
                   opt__per_road=np.stack([np.asarray(l.get_ydata(), dtype=np.float32) for l in lines], 1))
        plt.close(fig)
        # daily counts: pops then withdrawals over three hours, 6 roads, expected flows for 4 of them
        R = 6
        pops = [(3600 * 7 + 600 * i, torch.rand(R, generator=g) < 0.4) for i in range(14)]
        wds = [(3600 * 7 + 600 * i + 1, torch.rand(R, generator=g) < 0.2) for i in range(14)]
        expected = {4: 3.5, 0: 2.0, 2: 7.25, 5: 0.0}
        sim = TransportationSimulator("cpu")
        sim.model_core = types.SimpleNamespace(response_mpnn=types.SimpleNamespace(update_history=pops))
        sim.agent.withdraw_history = wds
        fig = sim.plot_daily_counts(expected, output_dir=None)
        xy = np.asarray(fig.axes[0].collections[0].get_offsets(), dtype=np.float64)
        rec.update(daily__clocks=np.asarray([t for t, _ in pops + wds]),
                   daily__masks=torch.stack([m for _, m in pops + wds]),
                   daily__expected_keys=np.asarray(list(expected.keys())),
                   daily__expected_vals=np.asarray(list(expected.values())),
                   daily__x=xy[:, 0], daily__y=xy[:, 1])
        plt.close(fig)
    save("figures", **rec)


def gen_value_mpnn():
    """MPNNValueNet (src/agents/mpnn_agent.py:265-402), the message-passing critic the reference defines but never
    instantiates, in eval mode (Dropout = identity): unbatched and batched forward with its own random weights."""
    from src.agents.mpnn_agent import MPNNValueNet
    net = synth.torus_network(3, 2, heterogeneous=True, seed=4)
    R = net.num_roads
    torch.manual_seed(21)
    val = MPNNValueNet(net.edge_index, R, device="cpu")
    val.agent_features = synth.population(60, R, seed=2)
    val.eval()
    x = synth.random_state(net, seed=14, t=30.0, num_agents=60)
    node_features = x[:, 3 * net.Nmax:]
    agent_index = x[:, 0].long().clamp(max=60)
    time = torch.tensor([21600.0])
    with torch.no_grad():
        value = val(node_features, net.edge_attr, agent_index, time)
        nb = torch.stack([node_features, node_features.flip(0) * 1.0, node_features * 0.5])
        ab = torch.stack([agent_index, agent_index.flip(0), agent_index])
        tb = torch.tensor([[21600.0], [21700.0], [30000.0]])
        eb = torch.stack([net.edge_attr, net.edge_attr * 2.0, net.edge_attr])
        value_b = val(nb, eb, ab, tb)
    save("value_mpnn", edge_index=net.edge_index, edge_attr=net.edge_attr, node_features=node_features,
         agent_index=agent_index, agent_features=val.agent_features, time=time, value=value, node_features_b=nb,
         agent_index_b=ab, time_b=tb, edge_attr_b=eb, value_b=value_b,
         **{k.replace(".", "__"): v for k, v in val.state_dict().items()})


from make_golden_fixtures import EQUIL_NETWORK_XML, EQUIL_POPULATION_XML, SIMPLE_NETWORK_XML  # noqa: E402


def _graph_fields(g):
    return dict(x=g.x, edge_index=g.edge_index, edge_attr=g.edge_attr, edge_index_routes=g.edge_index_routes,
                edge_attr_routes=g.edge_attr_routes, num_roads=g.num_roads, adj_matrix=g.adj_matrix, src_adj=g.src_adj,
                critical_number=g.critical_number, congestion_constant=g.congestion_constant)


def gen_builders():
    """TransportationSimulator.config_network / Agents.config_agents_from_xml on the reference's own
    test fixtures and on a synthetic MATSim torus; plus the classical run() loop on the 2-link network with SRC/DEST
    pseudo-nodes (tests/transportation_simulator_test.py:17-25: the agent reaches DONE within 20 steps)."""
    import contextlib
    import io
    import tempfile
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
        for tag, xml in (("simple", SIMPLE_NETWORK_XML), ("equil", EQUIL_NETWORK_XML)):
            d = os.path.join(tmp, tag)
            os.makedirs(d)
            open(os.path.join(d, "network.xml"), "w").write(xml)
        open(os.path.join(tmp, "equil", "population.xml"), "w").write(EQUIL_POPULATION_XML)
        d = os.path.join(tmp, "torus")
        os.makedirs(d)
        synth.write_matsim_network_xml(os.path.join(d, "network.xml"), 3, 4, seed=5)
        synth.write_matsim_population_xml(os.path.join(d, "population.xml"), 3, 4, 60, seed=6)
        rec = {}
        for tag in ("simple", "equil", "torus"):
            sim = TransportationSimulator("cpu")
            sim.config_network(os.path.join(tmp, tag, "network"))
            rec.update({f"{tag}__{k}": v for k, v in _graph_fields(sim.graph).items()})
            rec[f"{tag}__Nmax"] = sim.Nmax
        for tag in ("equil", "torus"):
            ag = Agents("cpu")
            ag.config_agents_from_xml(os.path.join(tmp, tag), verbose=False)
            rec[f"{tag}__agents"] = ag.agent_features
        # classical loop on the simple network (tests/conftest.py:109-120)
        sim = TransportationSimulator("cpu")
        sim.config_network(os.path.join(tmp, "simple", "network"))
        sim.agent.agent_features = torch.zeros((2, 9))
        sim.agent.agent_features[0, sim.agent.DEPARTURE_TIME] = 25 * 3600
        sim.agent.agent_features[1, 0] = 2
        sim.agent.agent_features[1, 1] = 5
        sim.config_parameters(start_time=1)
        sim.agent.set_time(sim.time)
        xs, ags = [], []
        steps = 0
        while sim.agent.agent_features[1, sim.agent.DONE] == 0 and steps < 20:
            torch.manual_seed(700 + steps)
            sim.run()
            steps += 1
            xs.append(sim.graph.x.clone())
            ags.append(sim.agent.agent_features.clone())
        rec.update(run_x=torch.stack(xs), run_agents=torch.stack(ags), run_steps=steps, run_time=sim.time)
    save("builders", **rec)
    print(f"  builders: classical run reached DONE after {steps} steps")


def gen_routing():
    """DijkstraAgents.choice (src/agents/base.py:527-584) driven by the classical run() loop on a MATSim grid with SRC/DEST
    pseudo-nodes (BASELINE config 1's shape: 4 x 6 grid, 76 links, N = 124), on real networkx; one choice on a
    heterogeneous torus with queues; and MPNNPolicyNet.refresh_dijkstra's distance matrix."""
    import contextlib
    import io
    import tempfile
    from src.agents.base import DijkstraAgents
    rec = {}
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
        for tag, het in (("grid", False), ("gridhet", True)):
            d = os.path.join(tmp, tag)
            os.makedirs(d)
            synth.write_matsim_grid_xml(os.path.join(d, "network.xml"), 4, 6, seed=3, heterogeneous=het)
            synth.write_matsim_population_xml(os.path.join(d, "population.xml"), 4, 6, 260, seed=4,
                                              first_departure=21600, spread=60)
            sim = TransportationSimulator("cpu")
            sim.config_network(os.path.join(d, "network"))
            ag = DijkstraAgents("cpu")
            ag.config_agents_from_xml(d, verbose=False)
            ag.agent_features[0, ag.DEPARTURE_TIME] = 48 * 3600
            sim.agent = ag
            sim.config_parameters(timestep_size=1, start_time=21600)
            ag.set_time(21600)
            steps = 70 if not het else 45
            xs, ags, hops = [], [], {}
            for s_ in range(steps):
                torch.manual_seed(900 + s_)
                sim.run()
                xs.append(sim.graph.x.clone())
                ags.append(ag.agent_features.clone())
                if s_ % 10 == 0:
                    hops[s_] = ag.next_hop_tensor.clone()
            rec.update({f"{tag}__x": torch.stack(xs), f"{tag}__agents": torch.stack(ags), f"{tag}__steps": steps,
                        f"{tag}__agents0_n": ag.agent_features.size(0)})
            # TransportationSimulator.compute_node_metrics (src/transportation_simulator.py:563-670) after that run
            nm = sim.compute_node_metrics(output_dir=None)
            # run_msa (src/algorithms/user_equilibrium_msa.py:65-165) on the end state, few iterations and to convergence cap
            from src.algorithms.user_equilibrium_msa import run_msa
            for iters in (1, 3, 25):
                fl = run_msa(sim.graph, ag, max_iter=iters)
                rec[f"{tag}__msa_{iters}"] = torch.tensor([fl[i] for i in range(len(fl))], dtype=torch.float64)
            rec.update({f"{tag}__nm_counts": torch.tensor([nm[n]["hourly_counts"] for n in range(len(nm))]),
                        f"{tag}__nm_avg_vc": torch.tensor([nm[n]["avg_vc"] for n in range(len(nm))]),
                        f"{tag}__nm_std_vc": torch.tensor([nm[n]["std_vc"] for n in range(len(nm))]),
                        f"{tag}__leg_hist": torch.tensor([[float(v) for v in row] for row in sim.leg_histogram_values])})
            for k in (0, 10, 40):
                rec[f"{tag}__next_hop_{k}"] = hops[k].to(torch.int16)
            print(f"  routing {tag}: done {int(ag.agent_features[:, ag.DONE].sum())}/{ag.agent_features.size(0) - 1}",
                  file=sys.stderr)
    # one choice on a heterogeneous torus with queues (pure road graph, every pair reachable, few ties)
    net = synth.torus_network(3, 3, heterogeneous=True, seed=21)
    pop = synth.population(400, net.num_roads, seed=22)
    x = synth.random_state(net, seed=23, fill=0.5, num_agents=400)
    g = Data(x=x.clone(), edge_index=net.edge_index, edge_attr=net.edge_attr, num_roads=net.num_roads,
             congestion_constant=net.congestion_constant, num_nodes=net.num_roads)
    ag = DijkstraAgents("cpu")
    ag.agent_features = pop.clone()
    h = FeatureHelpers(Nmax=net.Nmax)
    with contextlib.redirect_stdout(io.StringIO()):
        out = ag.choice(g, h)
    rec.update(torus__x0=x, torus__agents=pop, torus__x1=out.x, torus__next_hop=ag.next_hop_tensor.to(torch.int16),
               torus__seed=21, torus__Nmax=net.Nmax)
    # refresh_dijkstra (free-flow prior of the policy)
    from src.agents.mpnn_agent import MPNNPolicyNet
    ff = net.x[:, h.FREE_FLOW_TIME_TRAVEL][net.edge_index[1]]
    pol = MPNNPolicyNet(net.edge_index, net.num_roads, ff, device="cpu")
    rec.update(torus__dist_matrix=pol.dist_matrix, torus__ff_edges=ff)
    dest = torch.randint(0, net.num_roads, (net.edge_index.size(1),), generator=torch.Generator().manual_seed(5))
    rec.update(torus__prior_dest=dest, torus__prior_logits=pol.compute_dijkstra_logits(dest, ff))
    save("routing", **rec)


if __name__ == "__main__":
    assert os.path.isdir(REF), f"reference tree not found at {REF} (this script only runs in the build container)"
    gen_core_steps()
    gen_braess()
    gen_agents_tests()
    gen_graphdist()
    gen_env_rollout()
    gen_nets()
    gen_edge_mlp()
    gen_figures()
    gen_value_mpnn()
    gen_builders()
    gen_routing()
