"""GPU parity (through the C ABI) for the traffic-flow step and the environment step: HIP kernels vs the CPU oracle
and vs the golden vectors produced by the reference's own source. Integer state: bit-exact (torch.equal on the whole
fp32 state tensor)."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tarl_hip import ops as _ops
    return _ops


def dev(t):
    return t.cuda()


def make_adj(ei, n):
    adj = torch.zeros((n, n), dtype=torch.bool)
    adj[ei[0], ei[1]] = True
    return adj


@pytest.mark.parametrize("name", ["core_hom", "core_het", "core_noconst"])
def test_core_steps_golden(ops, name):
    g = load_golden(name)
    Nmax, ei = g["Nmax"], g["edge_index"]
    plan = ops.Plan(ei, g["x0"].size(0))
    ec = ops.EdgeConst(g["edge_attr"], "cuda")
    cc = dev(g["congestion_constant"]) if g["with_const"] else None
    x = dev(g["x0"].clone())
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    for s in range(g["steps"]):
        gum = dev(ops.gumbel_from_uniform_cpu(g[f"u{s}"]))
        dtt, _ = ops.direction_step(plan, x, Nmax, ec, g[f"t{s}"], congestion_constant=cc, gumbel=gum)
        assert torch.equal(x.cpu(), g[f"xd{s}"]), f"direction state differs at step {s}"
        assert torch.equal(dtt.cpu().view(-1), g[f"dtt{s}"])
        popped = ops.response_step(plan, x, Nmax, any_flag=flag)
        assert torch.equal(x.cpu(), g[f"xr{s}"]), f"response state differs at step {s}"
        assert torch.equal(popped.cpu().view(-1).bool(), g[f"pop{s}"])
        assert bool(flag.item()) == bool(g[f"pop{s}"].any())


def test_braess_reference_fixture(ops):
    g = load_golden("braess")
    plan = ops.Plan(g["edge_index"], 3)
    ec = ops.EdgeConst(g["edge_attr"], "cuda")
    x = dev(g["x0"].clone())
    gum = dev(ops.gumbel_from_uniform_cpu(g["u0"]))
    dtt, popped = ops.core_step(plan, x, g["Nmax"], ec, 0, gumbel=gum)
    assert torch.equal(x.cpu(), g["x1"]) and torch.equal(dtt.cpu().view(-1), g["dtt"])
    assert int(popped.sum()) == 0
    assert x[:, 3 * g["Nmax"] + 1].tolist() == [1.0, 1.0, 2.0]


@pytest.mark.parametrize("W,H,het,B,length", [(4, 3, True, 5, 100.0), (8, 8, False, 3, 100.0), (5, 5, True, 2, 100.0),
                                               (3, 3, False, 2, 740.0)])
def test_core_batched_vs_oracle(ops, W, H, het, B, length):
    """B environments with different states in ONE launch, strided views, several steps; oracle run per environment.
    The 740 m links give MAX_NUMBER_OF_AGENT = 99, Nmax = 100, F = 307: SURVEY 8d's stress layout with long queues."""
    from oracle import sim
    from tarl_hip import synth
    net = synth.torus_network(W, H, heterogeneous=het, seed=W * 31 + H, length=length)
    assert net.Nmax == (100 if length == 740.0 else net.Nmax)
    R, F, E, Nmax = net.num_roads, net.F, net.edge_index.size(1), net.Nmax
    plan = ops.Plan(net.edge_index, R)
    ec = ops.EdgeConst(net.edge_attr, "cuda")
    cc = dev(net.congestion_constant)
    xs = [synth.random_state(net, seed=100 + b, t=200.0) for b in range(B)]
    big = torch.zeros((B, R + 3, F + 5), device="cuda")        # odd env / row strides on purpose
    x = big[:, :R, :F]
    x.copy_(torch.stack(xs))
    gen = torch.Generator().manual_seed(7)
    total_pops = 0
    for s in range(5):
        t = 200 + s
        u = torch.rand((B, E), generator=gen)
        gum = dev(ops.gumbel_from_uniform_cpu(u))
        dtt, popped = ops.core_step(plan, x, Nmax, ec, t, congestion_constant=cc, gumbel=gum)
        for b in range(B):
            _, dref, pref = sim.core_step(xs[b], net.edge_index, net.edge_attr, t, Nmax, uniform=u[b],
                                          congestion_constant=net.congestion_constant)
            assert torch.equal(x[b].cpu(), xs[b]), f"env {b} step {s}"
            assert torch.equal(dtt[b].cpu(), dref) and torch.equal(popped[b].cpu().bool(), pref)
            total_pops += int(pref.sum())
    assert total_pops > 0
    assert float(big[:, R:, :].abs().sum()) == 0 and float(big[:, :, F:].abs().sum()) == 0  # padding untouched


def test_agents_tiny_reference_facts(ops):
    """tests/agents_test.py:12-73 through the HIP insert / withdraw kernels."""
    g = load_golden("agents_tiny")
    Nmax = g["Nmax"]
    ei = torch.tensor([[1, 0], [0, 0]])
    plan = ops.Plan(ei, 2)
    x, ag = dev(g["x0"].clone()), dev(g["agents0"].clone())
    ops.insert_step(x, Nmax, ag, 0)
    assert torch.equal(x.cpu(), g["x_ins"]) and torch.equal(ag.cpu(), g["agents_ins"])
    ops.withdraw_step(plan, x, Nmax, ag, 0)
    assert torch.equal(x.cpu(), g["x_w0"])
    ops.withdraw_step(plan, x, Nmax, ag, 10)
    assert torch.equal(x.cpu(), g["x_w10"]) and torch.equal(ag.cpu(), g["agents_w10"])
    x2, ag2 = dev(g["x0"].clone()), dev(g["cap_agents0"].clone())
    ops.insert_step(x2, Nmax, ag2, 0)
    assert torch.equal(x2.cpu(), g["cap_x"]) and torch.equal(ag2.cpu(), g["cap_agents"])


def test_agents_torus_golden(ops):
    g = load_golden("agents_torus")
    Nmax, ei = g["Nmax"], g["edge_index"]
    plan = ops.Plan(ei, g["x0"].size(0))
    x, ag, cc = dev(g["x0"].clone()), dev(g["agents0"].clone()), dev(g["congestion_constant"])
    for s in range(g["steps"]):
        t = g[f"t{s}"]
        w = ops.withdraw_step(plan, x, Nmax, ag, t)
        assert torch.equal(x.cpu(), g[f"xw{s}"]) and torch.equal(ag.cpu(), g[f"aw{s}"])
        assert torch.equal(w.cpu().view(-1).bool(), g[f"wmask{s}"])
        ops.insert_step(x, Nmax, ag, t, congestion_constant=cc)
        assert torch.equal(x.cpu(), g[f"xi{s}"]) and torch.equal(ag.cpu(), g[f"ai{s}"])


@pytest.mark.parametrize("name", ["env_hom", "env_het"])
def test_env_rollout_golden(ops, name):
    """The whole env step on device (choice -> core -> withdraw -> insert -> reward), 90 steps, fed with the reference's
    own actions and noise; state, agents and reward bit-exact at every step."""
    g = load_golden(name)
    Nmax, ei = g["Nmax"], g["edge_index"]
    N = g["x_init"].size(0)
    plan = ops.Plan(ei, N)
    ec = ops.EdgeConst(g["edge_attr"], "cuda")
    cc = dev(g["congestion_constant"])
    x, ag = dev(g["x_init"].clone()), dev(g["agents0"].clone())
    ops.reset_state(x, Nmax, ag)
    reward = torch.empty(1, device="cuda")
    counts = torch.empty((1, N), device="cuda")
    t = g["time0"]
    for s in range(g["T"]):
        ops.apply_action(plan, x, Nmax, action_onehot=dev(g["action"][s]))
        gum = dev(ops.gumbel_from_uniform_cpu(g["u_dir"][s]))
        dtt, _ = ops.core_step(plan, x, Nmax, ec, t, congestion_constant=cc, gumbel=gum)
        ops.withdraw_step(plan, x, Nmax, ag, t)
        ops.insert_step(x, Nmax, ag, t, congestion_constant=cc, reward=reward, counts=counts)
        t += 1
        assert torch.equal(x.cpu(), g["x"][s]), f"state differs at step {s}"
        assert torch.equal(ag.cpu(), g["agents"][s]), f"agents differ at step {s}"
        assert torch.equal(reward.cpu(), g["reward"][s]) and t == int(g["time"][s])
        assert torch.equal(dtt.cpu().view(-1), g["dtt"][s])
        assert torch.equal(counts.cpu().view(-1), g["x"][s][:, 3 * Nmax + 1])


def test_env_batched_vs_oracle_with_backlog(ops):
    """Batched env steps incl. an insertion backlog (many agents ready at once, capacity clamps, several per road):
    exercises the ordered-compaction / rank-within-road insert against the oracle."""
    from oracle import sim
    from tarl_hip import synth
    net = synth.torus_network(3, 3, heterogeneous=True, seed=4)
    N, Nmax, E = net.num_roads, net.Nmax, net.edge_index.size(1)
    adj = make_adj(net.edge_index, N)
    B, A = 3, 1500
    plan = ops.Plan(net.edge_index, N)
    ec = ops.EdgeConst(net.edge_attr, "cuda")
    cc = dev(net.congestion_constant)
    pops = [synth.population(A, N, seed=20 + b, t0=100, t1=103) for b in range(B)]   # nearly all ready at once
    xs = [net.x.clone() for _ in range(B)]
    x, ag = dev(torch.stack(xs)), dev(torch.stack(pops))
    reward = torch.empty(B, device="cuda")
    gen = torch.Generator().manual_seed(3)
    for s in range(25):
        t = 100 + s
        choice = torch.randint(0, 4, (B, N), generator=gen, dtype=torch.int32) + 4 * torch.arange(N, dtype=torch.int32)
        u = torch.rand((B, E), generator=gen)
        ops.apply_action(plan, x, Nmax, choice=dev(choice))
        ops.core_step(plan, x, Nmax, ec, t, congestion_constant=cc, gumbel=dev(ops.gumbel_from_uniform_cpu(u)))
        ops.withdraw_step(plan, x, Nmax, ag, t)
        ops.insert_step(x, Nmax, ag, t, congestion_constant=cc, reward=reward)
        for b in range(B):
            onehot = torch.zeros(E, dtype=torch.int64)
            onehot[choice[b].long()] = 1
            out = sim.env_step(xs[b], pops[b], net.edge_index, net.edge_attr, adj, onehot, t, Nmax, uniform=u[b],
                               congestion_constant=net.congestion_constant)
            assert torch.equal(x[b].cpu(), xs[b]), f"env {b} step {s}"
            assert torch.equal(ag[b].cpu(), pops[b]), f"agents env {b} step {s}"
            assert reward[b].item() == out["reward"].item()
    assert sum(float(p[:, sim.DONE].sum()) for p in pops) > 0


def test_full_size_properties(ops):
    """BASELINE config-4 size (10k edges, 16k agents): oracle agreement on one environment for a few steps plus
    size-independent invariants on a batch: agent conservation, FIFO prefix consistency, counters in range."""
    from oracle import sim
    from tarl_hip import synth
    net = synth.torus_network(25, 25)
    N, Nmax, E = net.num_roads, net.Nmax, net.edge_index.size(1)
    assert E == 10000
    B, A = 4, 16384
    plan = ops.Plan(net.edge_index, N)
    ec = ops.EdgeConst(net.edge_attr, "cuda")
    cc = dev(net.congestion_constant)
    pops = [synth.population(A, N, seed=b, t0=1000, t1=1060) for b in range(B)]
    x = dev(net.x.unsqueeze(0).repeat(B, 1, 1))
    ag = dev(torch.stack(pops))
    x0, ag0 = net.x.clone(), pops[0].clone()
    adj = make_adj(net.edge_index, N)
    gen = torch.Generator().manual_seed(1)
    reward = torch.empty(B, device="cuda")
    steps = 70
    for s in range(steps):
        t = 1000 + s
        choice = torch.randint(0, 4, (B, N), generator=gen, dtype=torch.int32) + 4 * torch.arange(N, dtype=torch.int32)
        u = torch.rand((B, E), generator=gen)
        ops.apply_action(plan, x, Nmax, choice=dev(choice))
        ops.core_step(plan, x, Nmax, ec, t, congestion_constant=cc, gumbel=dev(ops.gumbel_from_uniform_cpu(u)))
        ops.withdraw_step(plan, x, Nmax, ag, t)
        ops.insert_step(x, Nmax, ag, t, congestion_constant=cc, reward=reward)
        onehot = torch.zeros(E, dtype=torch.int64)
        onehot[choice[0].long()] = 1
        sim.env_step(x0, ag0, net.edge_index, net.edge_attr, adj, onehot, t, Nmax, uniform=u[0],
                     congestion_constant=net.congestion_constant)
    assert torch.equal(x[0].cpu(), x0) and torch.equal(ag[0].cpu(), ag0)
    xc, agc = x.cpu(), ag.cpu()
    n = xc[:, :, 3 * Nmax + 1]
    assert float(n.min()) >= 0 and bool((n <= xc[:, :, 3 * Nmax]).all())
    for b in range(B):
        on_way = agc[b, :, sim.ON_WAY].sum().item()
        assert n[b].sum().item() == -reward[b].item()
        # every queued agent is ON_WAY; the converse does not hold: the reference's Response round pops an agent out
        # of BOTH roads when an empty road that just received it is linked back by a U-turn edge
        assert on_way >= n[b].sum().item()
        ids = xc[b, :, :Nmax][torch.arange(Nmax).unsqueeze(0) < n[b].unsqueeze(1)]
        assert ids.numel() == torch.unique(ids).numel() and float(ids.min()) >= 1  # no duplicates, no dummy agent
        assert bool((agc[b, ids.long(), sim.ON_WAY] == 1).all())
        done = agc[b, :, sim.DONE] == 1
        assert not bool((done & (agc[b, :, sim.ON_WAY] == 1)).any())
        assert bool((agc[b, done, sim.ARRIVAL_TIME] >= agc[b, done, sim.DEPARTURE_TIME]).all())
    assert float(agc[:, :, sim.DONE].sum()) > 0


def test_device_philox_path(ops):
    """gumbel=NULL draws Philox noise on device: deterministic in (seed, counter), different across counters, and the
    chosen upstream follows the turn probabilities (contested roads with two admissible upstream heads)."""
    from tarl_hip import synth
    net = synth.torus_network(6, 6, heterogeneous=True, seed=9)
    R, Nmax = net.num_roads, net.Nmax
    plan = ops.Plan(net.edge_index, R)
    ec = ops.EdgeConst(net.edge_attr, "cuda")
    x0 = dev(synth.random_state(net, seed=1, t=300.0))
    outs = []
    for seed, counter in [(5, 0), (5, 0), (5, 1), (6, 0)]:
        x = x0.clone()
        ops.direction_step(plan, x, Nmax, ec, 300.0, seed=seed, counter=counter)
        outs.append(x.cpu())
    assert torch.equal(outs[0], outs[1])
    assert not torch.equal(outs[0], outs[2]) or not torch.equal(outs[0], outs[3])
    # statistics: many environments with the same state, count which admissible upstream wins
    B = 4096
    xb = x0.unsqueeze(0).repeat(B, 1, 1).contiguous()
    _, chosen = ops.direction_step(plan, xb, Nmax, ec, 300.0, seed=11, counter=3, want_dtt=False)
    from oracle import sim
    aid, prob, _ = sim.direction_message(x0.cpu(), net.edge_index, net.edge_attr, 300.0, Nmax)
    dst = net.edge_index[1]
    checked = 0
    for i in range(R):
        m = (dst == i) & (prob > 0)
        if int(m.sum()) < 2:
            continue
        ids, p = aid[m], prob[m] / prob[m].sum()
        freq = torch.stack([(chosen[:, i].cpu() == a).float().mean() for a in ids])
        assert abs(float(freq.sum()) - 1.0) < 1e-6
        assert torch.allclose(freq, p, atol=0.04), (i, freq, p)
        checked += 1
    assert checked > 0
