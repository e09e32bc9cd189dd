"""GPU: size-independent properties of the rollout at BASELINE config 4's FULL size (10 000 route edges, 2 500 roads,
16 384 agents, 256 frames) — where the oracle is too slow to replay everything. Checked on both rollout kernels:
  * batch independence: an environment's trajectory does not depend on which other environments share the launch
    (Philox streams are indexed by the environment's seed, not by lanes) — env b of a batch == the same env run alone;
  * determinism: the same seed gives the same bits, run after run;
  * state invariants of the domain after every probe: FIFO counts within [0, MAX], every queued agent id unique and
    marked ON_WAY, nobody DONE is still queued, ON_WAY >= queued (the reference's U-turn double pop can only lose queued
    agents, DESIGN Q24), arrivals stamped within the episode, per-frame reward == -sum of the counts;
  * the two kernels agree with each other at this size."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(B, seeds, seed=11):
    from tarl_hip import synth
    from tarl_hip.engine import SimEngine
    net = synth.torus_network(25, 25)
    N = net.num_roads
    pops = torch.stack([synth.population(16384, N, seed=s) for s in seeds])
    eng = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                    pops.cuda(), congestion_constant=net.congestion_constant, seed=seed)
    eng.reset()
    eng.prepare_policy(torch.randn(N, generator=torch.Generator().manual_seed(1)).cuda())
    return net, eng


def _rollout(eng, T, mode):
    N, B = eng.N, eng.B
    shp = (lambda t: (t, B, N)) if mode == "env" else (lambda t: (t, N, B))
    ch = torch.zeros(shp(T), dtype=torch.uint8, device="cuda")
    ct = torch.zeros(shp(T + 1), dtype=torch.uint8, device="cuda")
    lp, rw = torch.zeros((T, B), device="cuda"), torch.zeros((T, B), device="cuda")
    (eng.rollout_env if mode == "env" else eng.rollout_fused)(T, choice=ch, log_prob=lp, reward=rw, counts=ct)
    ch, ct = eng.decode_rollout(mode != "env", choice=ch, counts=ct)     # (T, B, N) edge ids / fp32 counts
    return ch, lp, rw, ct


def _check_invariants(net, eng, rw, ct):
    Nmax = net.Nmax
    x, ag = eng.x, eng.agents
    n = x[:, :, 3 * Nmax + 1]
    maxn = x[:, :, 3 * Nmax]
    assert bool((n >= 0).all()) and bool((n <= maxn).all())
    assert torch.equal(rw[-1], -n.sum(dim=1)) and torch.equal(ct[-1], n)
    assert torch.equal(rw, -ct[1:].sum(dim=2))                        # every frame, not only the last
    for b in range(eng.B):
        ids = x[b, :, :Nmax]
        slot = torch.arange(Nmax, device="cuda").unsqueeze(0)
        live = ids[slot < n[b].unsqueeze(1)].long()                   # agent ids in the occupied FIFO prefixes
        assert bool((live > 0).all()) and live.numel() == live.unique().numel(), "a queued agent appears twice"
        assert bool((ag[b, live, 7] == 1).all()) and bool((ag[b, live, 8] == 0).all())
        on_way = int(ag[b, :, 7].sum())
        assert on_way >= live.numel()
        done = ag[b, :, 8] == 1
        assert bool((ag[b, done, 7] == 0).all())
        if bool(done.any()):
            assert bool((ag[b, done, 3] >= ag[b, done, 2]).all()) and float(ag[b, done, 3].max()) < eng.time
    assert int(ag[:, :, 7].sum()) > 0


@pytest.mark.parametrize("mode", ["frames", "env"])
def test_full_size_batch_independence_determinism_invariants(mode):
    assert torch.cuda.is_available()
    T = 256
    seeds = [3, 4, 5, 6, 7, 8]
    net, eng = _engine(len(seeds), seeds)
    ch, lp, rw, ct = _rollout(eng, T, mode)
    _check_invariants(net, eng, rw, ct)
    # determinism: a fresh engine with the same seeds reproduces every bit
    _, eng2 = _engine(len(seeds), seeds)
    ch2, lp2, rw2, ct2 = _rollout(eng2, T, mode)
    assert torch.equal(ch, ch2) and torch.equal(lp, lp2) and torch.equal(rw, rw2) and torch.equal(ct, ct2)
    assert torch.equal(eng.x, eng2.x) and torch.equal(eng.agents, eng2.agents)
    # batch independence: environment 0 alone (same engine seed: the Philox streams are indexed by env id 0)
    _, solo = _engine(1, seeds[:1])
    ch1, lp1, rw1, ct1 = _rollout(solo, T, mode)
    assert torch.equal(ch1[:, 0], ch[:, 0]) and torch.equal(lp1[:, 0], lp[:, 0]) and torch.equal(rw1[:, 0], rw[:, 0])
    assert torch.equal(ct1[:, 0], ct[:, 0]) and torch.equal(solo.x[0], eng.x[0]) and torch.equal(solo.agents[0], eng.agents[0])


def test_full_size_both_rollout_kernels_agree():
    T = 256
    seeds = [20, 21, 22]
    net, e1 = _engine(3, seeds)
    _, e2 = _engine(3, seeds)
    a, b = _rollout(e1, T, "frames"), _rollout(e2, T, "env")
    for u, v, what in zip(a, b, ("actions", "log-prob", "reward", "counts")):
        assert torch.equal(u, v), what
    assert torch.equal(e1.x, e2.x) and torch.equal(e1.agents, e2.agents)
    _check_invariants(net, e2, b[2], b[3])


def _gridlocked_engine(fused):
    """Every FIFO full (count == MAX == Nmax - 1) with an overdue head: the gridlock-relief rule (src/direction_mpnn.py:87-89)
    moves heads into full rows, so counts reach Nmax in the first frame — outside the reference's defined domain."""
    from tarl_hip import synth
    from tarl_hip.engine import SimEngine
    net = synth.torus_network(2, 2)
    N, Nmax = net.num_roads, net.Nmax
    x = net.x.clone()
    ids = torch.arange(1, N * (Nmax - 1) + 1, dtype=torch.float32).view(N, Nmax - 1)
    x[:, :Nmax - 1] = ids
    x[:, Nmax:2 * Nmax - 1] = 100.0            # arrivals
    x[:, 2 * Nmax:3 * Nmax - 1] = 200.0        # departures: long overdue at the episode clock
    x[:, 3 * Nmax + 1] = Nmax - 1
    ag = synth.population(N * (Nmax - 1), N, seed=0)
    ag[1:, 7] = 1.0                            # everybody is on the way
    eng = SimEngine(x.cuda().unsqueeze(0).contiguous(), net.edge_index, net.edge_attr, Nmax, ag.cuda().unsqueeze(0),
                    congestion_constant=net.congestion_constant, seed=1, fused=fused)
    if fused:
        eng.prepare_policy(torch.zeros(N, device="cuda"))
    return eng


@pytest.mark.parametrize("mode", ["frames", "env", "frame_api", "unfused"])
def test_count_reaching_nmax_is_flagged_and_raises(mode):
    """The reference raises IndexError when a count reaches Nmax (DESIGN Q25); every kernel family sets the device status
    word and the host raises TarlError when it reads it."""
    from tarl_hip import lib, ops
    eng = _gridlocked_engine(mode != "unfused")
    N, B, T = eng.N, 1, 3
    if mode == "unfused":
        p = ops.graphdist_softmax(eng.plan, torch.zeros((1, eng.E), device="cuda"))
        _, ch = ops.graphdist_sample(eng.plan, p, seed=1, counter=1, want_onehot=False, want_choice=True)
        eng.step(choice=ch)
    elif mode == "frame_api":
        eng.frame_fused()
    else:
        shp = (lambda t: (t, N, B)) if mode == "frames" else (lambda t: (t, B, N))
        ch, ct = torch.zeros(shp(T), dtype=torch.uint8, device="cuda"), torch.zeros(shp(T + 1), dtype=torch.uint8, device="cuda")
        rw = torch.zeros((T, B), device="cuda")
        run = eng.rollout_fused if mode == "frames" else eng.rollout_env
        with pytest.raises(lib.TarlError, match="reached Nmax"):
            run(T, choice=ch, log_prob=None, reward=rw, counts=ct)
        return
    with pytest.raises(lib.TarlError, match="reached Nmax"):
        eng.check_flags()


def test_drop_in_direction_mpnn_reports_the_domain_exit_without_stalling_the_step_path():
    """The drop-in DirectionMPNN polls its device status word asynchronously (a pinned copy behind every forward): the
    forward that drives a count to Nmax returns normally, and check() — or a later forward / set_time once the copy has
    landed — raises IndexError. Also covers a flag raised by the LAST forward of a run (only check() can see that one)."""
    from src.direction_mpnn import DirectionMPNN
    from tarl_hip import synth
    net = synth.torus_network(2, 2)
    N, Nmax = net.num_roads, net.Nmax
    x = net.x.clone()
    x[:, :Nmax - 1] = torch.arange(1, N * (Nmax - 1) + 1, dtype=torch.float32).view(N, Nmax - 1)
    x[:, Nmax:2 * Nmax - 1] = 100.0
    x[:, 2 * Nmax:3 * Nmax - 1] = 200.0
    x[:, 3 * Nmax + 1] = Nmax - 1
    first_out = torch.full((N,), -1, dtype=torch.long)
    for e in range(net.edge_index.size(1) - 1, -1, -1):
        first_out[net.edge_index[0, e]] = net.edge_index[1, e]
    x[:, 3 * Nmax + 5] = first_out.to(torch.float32)          # SELECTED_ROAD: every head wants its first out-neighbour
    xg = x.cuda()
    mp = DirectionMPNN(Nmax=Nmax, time=21540)
    mp.noise_seed = 5
    out = mp(xg, net.edge_index.cuda(), net.edge_attr.cuda(), congestion_constant=net.congestion_constant.cuda())
    assert out is xg and float(xg[:, 3 * Nmax + 1].max()) == Nmax          # the forward itself returned normally
    with pytest.raises(IndexError, match="reached Nmax"):
        mp.check()
    mp.check()                                                             # reported once, then re-armed


@pytest.mark.parametrize("B,T,case", [(1, 1, "nobody"), (3, 5, "nobody"), (1, 7, "one_agent"), (65, 3, "one_agent"),
                                      (257, 2, "everybody_at_once")])
def test_edge_cases_empty_traffic_single_frames_ragged_batches(B, T, case):
    """The corners the reference's own tests poke at (tests/agents_test.py: an agent table where nobody is due, one agent;
    tests/conftest.py: a three-road graph), on all three implementations of the frame at once — per-op kernels, the fused
    frame kernels and the LDS-resident rollout must agree bit for bit on: NOBODY ever due (every row idle in every frame:
    the state stays the reset state, rewards 0, log-probs finite), ONE agent in the whole network, EVERY agent due in the
    first frame (the insert's backlog path: most are refused for lack of room), batch sizes 1 / 65 / 257 (partial waves and
    partial workgroups), one-frame rollouts (no previous count slice)."""
    from tarl_hip import ops, synth
    from tarl_hip.engine import EPISODE_START, SimEngine
    net = synth.torus_network(3, 4, heterogeneous=True, seed=7)
    N = net.num_roads
    A = {"nobody": 40, "one_agent": 1, "everybody_at_once": 600}[case]
    t0 = EPISODE_START + (10_000_000 if case == "nobody" else 0)
    t1 = t0 + (1 if case != "one_agent" else 3)
    pops = torch.stack([synth.population(A, N, seed=90 + b, t0=t0, t1=t1) for b in range(B)]).cuda()
    emb = torch.randn(N, generator=torch.Generator().manual_seed(3)).cuda()

    def engine(fused):
        e = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                      pops.clone(), congestion_constant=net.congestion_constant, seed=5, fused=fused)
        e.reset()
        if fused:
            e.prepare_policy(emb)
        return e
    ef, ee, eu = engine(True), engine(True), engine(False)
    chf, lpf, rwf, ctf = _rollout(ef, T, "frames")
    che, lpe, rwe, cte = _rollout(ee, T, "env")
    assert torch.equal(chf, che) and torch.equal(lpf, lpe) and torch.equal(rwf, rwe) and torch.equal(ctf, cte)
    assert torch.equal(ef.x, ee.x) and torch.equal(ef.agents, ee.agents) and ef.time == ee.time
    # the per-op kernels, frame by frame, with the same draws (the engine's own Philox streams)
    for t in range(T):
        logits = ops.policy_edge_logits(eu.plan, eu.node_features, emb)
        p = ops.graphdist_softmax(eu.plan, logits)
        eu.sample_counter += 1
        _, choice = ops.graphdist_sample(eu.plan, p, seed=eu.seed ^ 0x5DEECE66D, counter=eu.sample_counter, want_onehot=False,
                                         want_choice=True)
        reward, _ = eu.step(choice=choice)
        assert torch.equal(choice, chf[t]) and torch.equal(reward, rwf[t]) and torch.equal(eu.counts, ctf[t + 1]), f"frame {t}"
    assert torch.equal(eu.x, ef.x) and torch.equal(eu.agents, ef.agents)
    ef.check_flags()
    ee.check_flags()
    assert bool(torch.isfinite(lpf).all())
    if case == "nobody":
        assert float(rwf.abs().sum()) == 0.0 and float(ctf.sum()) == 0.0 and float(ef.agents[:, :, 7:].sum()) == 0.0
    elif case == "one_agent":
        assert float(ef.agents[:, :, 7].sum() + ef.agents[:, :, 8].sum()) == (B if T >= 4 else float(ef.agents[:, :, 7].sum()))
    else:
        on_way = ef.agents[:, :, 7].sum(dim=1)
        assert bool((on_way > 0).all()) and bool((on_way < A).all())      # some got in, most found no room
