"""GPU: the fused rollout frame (csrc/fused.hip: 3 launches, env-minor layout) reproduces, frame by frame, the unfused
kernels, the oracle and the reference's golden rollouts — state (after export), agents, actions, rewards, counts and
masks bit-identical; log-probs to fp32 rounding (same terms, different fixed summation order; tolerance 1e-5 rel)."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
LP_RTOL = 1e-5


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tarl_hip import ops as _ops
    return _ops


def dev(t):
    return t.cuda()


def check_records(ops, plan, fs, x, Nmax, ag, cc, ec):
    """The maintained hot records equal a fresh pack of the exported x / agents (tail only where the FIFO is non-empty)."""
    ref = ops.FusedState(plan, fs.B, fs.A, x.device, Nmax)
    ops.fused_pack(plan, ref, x, Nmax, ag, cc, ec=ec)
    # head id, count, head departure — the departure of an empty row that was idle in the last frame (tail word without
    # TLF_AUTH) is derived from the clock by its readers and not stored
    live = ((fs.hdp[..., 0] & 127) > 0) | ((fs.tl & 1) != 0)
    # (bit 7 of the count byte, HD_DIRTY, is bookkeeping state: a row stays dirty once its FIFO has touched its last slot,
    # while a fresh pack derives the flag from the exported x, where the pending garbage triple sits in a dead slot)
    assert torch.equal(fs.hdp[..., 0] & ~0x80, ref.hdp[..., 0] & ~0x80) and torch.equal(fs.hdp[..., 1][live], ref.hdp[..., 1][live])
    assert torch.equal(fs.sel, ref.sel)
    assert torch.equal(fs.sel8 & 0x7F, ref.sel8 & 0x7F) and torch.equal(fs.in_rec, ref.in_rec) and torch.equal(fs.node_rec, ref.node_rec)
    nz = ref.count > 0
    assert torch.equal(fs.tail_id[nz], ref.tail_id[nz])
    assert torch.equal(fs.head_slot_arrival[nz], ref.head_slot_arrival[nz])    # head arrival (read from the head's slot record)
    assert torch.equal(fs.a_status, ref.a_status) and torch.equal(fs.st0, ref.st0)
    # insert cursor: nobody before it is still waiting
    pos = torch.arange(fs.A, device=x.device).unsqueeze(0)
    st_sorted = torch.gather(fs.a_status, 1, fs.a_order.long())
    assert not bool(((pos < fs.cur_lo.unsqueeze(1)) & (st_sorted == 0)).any())
    # the departure-ordered window: "already inserted" flags mirror the status SoA; records = a fresh pack's
    assert torch.equal(fs.a_ins, (st_sorted != 0).to(torch.uint8)) and torch.equal(fs.a_win, ref.a_win)
    assert torch.equal(torch.gather(fs.a_rank, 1, fs.a_order.long()), pos.expand(fs.B, -1).to(torch.int32))
    # some rows carry a pending (lazy, never stored) garbage slot: idle in the last frame, or an event row that received nobody
    assert int((((fs.tl & 1) == 0) | (fs.gc8 > 0)).sum()) > 0


@pytest.mark.parametrize("W,H,het,B,A,frames,with_cc,Nmax,tiny,dt,prune", [
    (3, 3, True, 3, 1500, 60, True, None, False, 1, 0.0),
    (4, 4, False, 2, 600, 80, True, None, False, 1, 0.0),
    (2, 3, True, 2, 300, 40, False, None, False, 1, 0.0),
    (3, 2, False, 70, 200, 30, True, 40, False, 1, 0.0),
    (2, 2, True, 3, 400, 40, True, 100, False, 1, 0.0),
    (3, 3, True, 3, 1500, 40, True, None, True, 15, 0.0),
    (4, 3, True, 5, 900, 60, True, None, False, 1, 0.35),
    (6, 6, True, 4, 1500, 60, True, None, False, 1, 0.01)])
def test_fused_equals_unfused_frame_by_frame(ops, W, H, het, B, A, frames, with_cc, Nmax, tiny, dt, prune):
    """tiny: every third road holds at most 2 or 3 agents (MAX_NUMBER_OF_AGENT <= CONGESTION_FILE) and the clock advances
    15 s per frame: such a road never receives anybody, but EMPTY it passes the Direction gather's second test (its garbage
    head is overdue by more than 10 s) and its id-0 head competes in the Gumbel race — the one reader of an empty row's
    head departure, which the row pass therefore keeps storing for these rows and for no others.
    prune: that fraction of the dual edges removed at random (every road keeps one out-edge): rows no longer group four by
    four by their out-edge targets and in-degrees run from zero to four. With a third removed the row pass falls back to
    consecutive chunks (its chunk table would be mostly padding); with 1 % removed it walks the table, whose groups of one
    to four rows make PARTIAL chunks."""
    from tarl_hip import synth
    net = synth.torus_network(W, H, heterogeneous=het, seed=W + 10 * H, Nmax=Nmax)
    if prune:
        gp = torch.Generator().manual_seed(5)
        Eall = net.edge_index.size(1)
        keep = torch.rand(Eall, generator=gp) > prune
        keep[torch.arange(0, Eall, 4) + torch.randint(0, 4, (Eall // 4,), generator=gp)] = True   # one out-edge per road stays
        net.edge_index, net.edge_attr = net.edge_index[:, keep].contiguous(), net.edge_attr[keep].contiguous()
        out_lists = {}
        for e in range(net.edge_index.size(1)):
            out_lists.setdefault(int(net.edge_index[0, e]), []).append(int(net.edge_index[1, e]))
        sizes = {}
        for lst in out_lists.values():
            sizes[tuple(lst)] = sizes.get(tuple(lst), 0) + 1
        assert len(set(sizes.values())) > 1 and any(v % 4 for v in sizes.values())      # mixed groups, partial chunks
    if tiny:
        nm = net.Nmax
        net.x[::3, 3 * nm + 0] = torch.tensor([2.0, 3.0]).repeat(net.num_roads)[:net.x[::3].size(0)]
        net.congestion_constant = net.x[:, 3 * nm + 2] * (net.x[:, 3 * nm + 0] + 10 - net.critical_number)
    N, Nmax, E = net.num_roads, net.Nmax, net.edge_index.size(1)
    plan = ops.Plan(net.edge_index, N)
    if prune:
        assert plan.row_siblings == (prune < 0.02) and plan.num_row_chunks > N // 4 and not plan.siblings4
    ec = ops.EdgeConst(net.edge_attr, "cuda")
    cc = dev(net.congestion_constant) if with_cc else None
    pops = torch.stack([synth.population(A, N, seed=40 + b, t0=100, t1=130) for b in range(B)])
    x1, a1 = dev(net.x.unsqueeze(0).repeat(B, 1, 1)), dev(pops.clone())
    x2, a2 = x1.clone(), a1.clone()
    fs = ops.FusedState(plan, B, A + 1, "cuda", Nmax)
    ops.fused_pack(plan, fs, x2, Nmax, a2, cc, ec=ec)
    emb = torch.randn(N, generator=torch.Generator().manual_seed(1)).cuda()
    tables = ops.fused_policy_prepare(plan, fs, emb, 0.9)
    gen = torch.Generator().manual_seed(2)
    r1, c1 = torch.empty(B, device="cuda"), torch.empty((B, N), device="cuda")
    r2, c2 = torch.empty(B, device="cuda"), torch.empty((N, B), device="cuda")
    ch2 = torch.empty((N, B), dtype=torch.int32, device="cuda")
    lp2, en2 = torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
    pop2 = torch.empty((B, N), dtype=torch.uint8, device="cuda")
    wd2 = torch.empty((B, N), dtype=torch.uint8, device="cuda")
    dtt2 = torch.empty((B, E), device="cuda")
    events = 0
    for s in range(frames):
        t = 100 + dt * s
        u_s = dev(torch.rand((B, N), generator=gen))
        gum = dev(ops.gumbel_from_uniform_cpu(torch.rand((B, E), generator=gen)))
        # unfused chain
        logits = ops.policy_edge_logits(plan, x1[:, :, 3 * Nmax:], emb)
        p = ops.graphdist_softmax(plan, logits, 0.9)
        _, ch1 = ops.graphdist_sample(plan, p, uniform=u_s, want_onehot=False, want_choice=True)
        lp1, en1 = ops.graphdist_logprob_entropy(plan, p, choice=ch1)
        ops.apply_action(plan, x1, Nmax, choice=ch1)
        dtt1, pop1 = ops.core_step(plan, x1, Nmax, ec, t, congestion_constant=cc, gumbel=gum)
        wd1 = ops.withdraw_step(plan, x1, Nmax, a1, t)
        ops.insert_step(x1, Nmax, a1, t, congestion_constant=cc, reward=r1, counts=c1)
        # fused frame
        ops.fused_frame(plan, fs, tables, a2, ec, t, use_cong=with_cc, prev_time=t - dt, uniform=u_s, gumbel=gum, dtt=dtt2,
                        popped=pop2, withdrawn=wd2, choice=ch2, log_prob=lp2, entropy=en2, reward=r2, counts=c2)
        ops.fused_export(plan, fs, x2, Nmax, t)       # back to the reference's column layout
        assert torch.equal(ch1, ch2.t()), f"actions frame {s}"
        assert torch.allclose(lp1, lp2, rtol=LP_RTOL, atol=1e-5) and torch.equal(en1, en2), f"policy frame {s}"
        assert torch.equal(x1, x2), f"state frame {s}"
        assert torch.equal(a1, a2), f"agents frame {s}"
        assert torch.equal(dtt1, dtt2) and torch.equal(pop1, pop2) and torch.equal(wd1, wd2), f"masks frame {s}"
        assert torch.equal(r1, r2) and torch.equal(c1, c2.t())
        events += int(pop1.sum()) + int(wd1.sum())
        if s % 10 == 0 or s == frames - 1:
            check_records(ops, plan, fs, x2, Nmax, a2, cc, ec)
    assert events > 0 and (tiny or float(a1[:, :, 8].sum()) > 0)


def test_fused_golden_rollout(ops):
    """The reference's own 90-frame rollout (actions sampled from its GraphDistribution with its noise) through the
    fused path: actions, state, agents, reward bit-exact, log-probs <= 1e-4, at every frame."""
    g = load_golden("env_het")
    Nmax, ei = g["Nmax"], g["edge_index"]
    N = g["x_init"].size(0)
    A = g["agents0"].size(0)
    plan = ops.Plan(ei, N)
    ec = ops.EdgeConst(g["edge_attr"], "cuda")
    cc = dev(g["congestion_constant"])
    x, ag = dev(g["x_init"].clone()).unsqueeze(0), dev(g["agents0"].clone()).unsqueeze(0)
    ops.reset_state(x, Nmax, ag)
    fs = ops.FusedState(plan, 1, A, "cuda", Nmax)
    ops.fused_pack(plan, fs, x, Nmax, ag, cc, ec=ec)
    tables = ops.fused_policy_prepare(plan, fs, dev(g["w_emb"]))
    choice = torch.empty((N, 1), dtype=torch.int32, device="cuda")
    lp, reward = torch.empty(1, device="cuda"), torch.empty(1, device="cuda")
    t = g["time0"]
    for s in range(g["T"]):
        ops.fused_frame(plan, fs, tables, ag, ec, t, uniform=dev(g["u_sample"][s]).view(1, -1).contiguous(),
                        gumbel=dev(ops.gumbel_from_uniform_cpu(g["u_dir"][s])).view(1, -1).contiguous(),
                        choice=choice, log_prob=lp, reward=reward)
        onehot = torch.zeros(ei.size(1), dtype=torch.int64)
        onehot[choice.cpu().view(-1).long()] = 1
        assert torch.equal(onehot, g["action"][s]), f"action differs at frame {s}"
        assert abs(lp.item() - g["log_prob"][s].item()) < 1e-4
        ops.fused_export(plan, fs, x, Nmax, t)
        t += 1
        assert torch.equal(x[0].cpu(), g["x"][s]), f"state differs at frame {s}"
        assert torch.equal(ag[0].cpu(), g["agents"][s]) and torch.equal(reward.cpu(), g["reward"][s])


def test_fused_full_size_equals_unfused(ops):
    """BASELINE config-4 size (10k edges, 16k agents), device Philox noise on both paths: identical trajectories.
    B = 130 exercises a partially filled environment tile."""
    from tarl_hip import synth
    from tarl_hip.engine import SimEngine
    net = synth.torus_network(25, 25)
    N = net.num_roads
    B, A = 130, 16384
    pops = torch.stack([synth.population(A, N, seed=b % 7, t0=21540, t1=21600) for b in range(B)])
    mk = lambda fused: SimEngine(dev(net.x.unsqueeze(0).repeat(B, 1, 1)), net.edge_index, net.edge_attr, net.Nmax,
                                 dev(pops.clone()), congestion_constant=net.congestion_constant, seed=5, fused=fused)
    e1, e2 = mk(False), mk(True)
    emb = torch.randn(N, generator=torch.Generator().manual_seed(3)).cuda()
    e1.reset(); e2.reset()
    e2.prepare_policy(emb)
    ch2 = torch.empty((N, B), dtype=torch.int32, device="cuda")
    lp2 = torch.empty(B, device="cuda")
    c2 = torch.empty((N, B), device="cuda")
    for s in range(40):
        logits = ops.policy_edge_logits(e1.plan, e1.node_features, emb)
        p = ops.graphdist_softmax(e1.plan, logits)
        _, ch1 = ops.graphdist_sample(e1.plan, p, seed=e2.seed ^ 0x5DEECE66D, counter=s + 1, want_onehot=False,
                                      want_choice=True)
        lp1, _ = ops.graphdist_logprob_entropy(e1.plan, p, choice=ch1, want_entropy=False)
        e1.step(choice=ch1)
        e2.frame_fused(choice=ch2, log_prob=lp2, counts=c2)
        assert torch.equal(ch1, ch2.t()), f"frame {s}"
        assert torch.allclose(lp1, lp2, rtol=LP_RTOL, atol=1e-4)
        assert torch.equal(e1.agents, e2.agents), f"frame {s}"
        assert torch.equal(e1.reward, e2.reward) and torch.equal(e1.counts, c2.t()) and e1.time == e2.time
        if s % 8 == 0 or s == 39:
            assert torch.equal(e1.x, e2.x), f"frame {s}"
    assert float(e2.agents[:, :, 7].sum()) > 0


def test_critic_slab_mode_matches_row_major(ops):
    """The critic reading the env-minor rollout buffer [frame][node][env] directly == the row-major path."""
    gen = torch.Generator().manual_seed(3)
    S, N, R = 3, 301, 256
    counts = torch.randint(0, 14, (S, N, R), generator=gen).float().cuda()
    times = (torch.arange(S).float() + 21540.0).cuda()
    lin = [torch.nn.Linear(N + 1, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, 1)]
    cw = ops.CriticWeights(*(t.detach().cuda().contiguous() for t in (lin[0].weight, lin[0].bias, lin[1].weight,
                                                                     lin[1].bias, lin[2].weight.reshape(-1), lin[2].bias)))
    v_slab = ops.critic_forward_slabs(cw, counts, times)
    rows = counts.permute(0, 2, 1).contiguous().view(S * R, N)
    v_rows, _, _ = ops.critic_forward(cw, rows, times, rows_per_time=R)
    assert torch.equal(v_slab, v_rows)        # same tiles, same k order, same MFMA chain
    # the rollout's count BYTES: the exact-chain variant is bit-identical too; the default runs the first layer on the bf16
    # matrix cores with W1 split into three exact bf16 pieces — fp32 accuracy, only the order of the fp32 additions differs
    c8 = counts.to(torch.uint8)
    assert torch.equal(ops.critic_forward_slabs(cw, c8, times, exact_chain=True), v_rows)
    v8 = ops.critic_forward_slabs(cw, c8, times)
    ref64 = torch.nn.Sequential(lin[0], torch.nn.ReLU(), lin[1], torch.nn.ReLU(), lin[2]).double()(
        torch.cat([rows.cpu().double(), times.cpu().double().repeat_interleave(R).unsqueeze(1)], dim=1)).view(-1)
    scale = float(ref64.abs().max())
    err8, err32 = float((v8.cpu().double() - ref64).abs().max()), float((v_rows.cpu().double() - ref64).abs().max())
    assert err8 <= 1e-5 * scale and err8 <= 4 * err32 + 1e-6 * scale, (err8, err32, scale)
    # a K that is not a multiple of the 32-node chunk, counts up to 255, one slab of 128 rows
    N2 = 77
    c2 = torch.randint(0, 256, (2, N2, 128), generator=gen).to(torch.uint8).cuda()
    lin2 = torch.nn.Linear(N2 + 1, 64)
    cw2 = ops.CriticWeights(lin2.weight.detach().cuda().contiguous(), lin2.bias.detach().cuda().contiguous(), cw.w2, cw.b2,
                            cw.w3, cw.b3)
    a, b = ops.critic_forward_slabs(cw2, c2, times[:2]), ops.critic_forward_slabs(cw2, c2, times[:2], exact_chain=True)
    assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("ct", [2, 4])
def test_critic_slab_wide_tiles_equal_the_128_environment_tiles(ops, monkeypatch, ct):
    """k_critic_fwd_slab_u8x3<CT>: 256 / 512 environments per workgroup (the weight pieces of a k-tile are fetched once for all
    of them) — every environment's value is the same chain of MFMAs in the same k order as with 128: bit-identical. The
    launcher picks the wide tiles only for large batches; TARL_CRITIC_CT forces them here."""
    gen = torch.Generator().manual_seed(11)
    S, N, R = 3, 333, 1024
    c8 = torch.randint(0, 16, (S, N, R), generator=gen).to(torch.uint8).cuda()
    times = (torch.arange(S).float() + 21540.0).cuda()
    lin = [torch.nn.Linear(N + 1, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, 1)]
    cw = ops.CriticWeights(*(t.detach().cuda().contiguous() for t in (lin[0].weight, lin[0].bias, lin[1].weight,
                                                                     lin[1].bias, lin[2].weight.reshape(-1), lin[2].bias)))
    monkeypatch.setenv("TARL_CRITIC_CT", "1")
    v1 = ops.critic_forward_slabs(cw, c8, times)
    monkeypatch.setenv("TARL_CRITIC_CT", str(ct))
    vw = ops.critic_forward_slabs(cw, c8, times)
    assert torch.equal(v1, vw)
    rows = c8.float().permute(0, 2, 1).contiguous().view(S * R, N)
    v_rows, _, _ = ops.critic_forward(cw, rows, times, rows_per_time=R)
    assert float((vw - v_rows).abs().max()) <= 1e-5 * max(1.0, float(v_rows.abs().max()))


@pytest.mark.parametrize("B,T,merge", [(5, 30, "1"), (5, 30, "0"), (70, 7, "1"), (3, 1, "1"), (130, 2, "1"),
                                       (130, 3, "1"), (300, 5, "1"), (5, 70, "2"), (300, 33, "2"), (3, 1, "2")])
def test_rollout_launcher_equals_frame_loop(ops, monkeypatch, B, T, merge):
    """SimEngine.rollout_fused (tarl_fused_rollout: the whole collector loop in one foreign call; with merge = 2 — the
    default — all actions are drawn in blocks of 32 frames on a side stream; with merge = 1 frame
    t+1's choice shares a launch with frame t's insert, SELECTED_ROAD and the log-prob accumulators double-buffered)
    == T calls of frame_fused; then a second rollout continues from the state the first left (odd and even T end in
    different buffers)."""
    monkeypatch.setenv("TARL_ROLLOUT_MERGE", merge)
    from tarl_hip import synth
    from tarl_hip.engine import SimEngine
    net = synth.torus_network(5, 5, heterogeneous=True, seed=3)
    N = net.num_roads
    A = 700
    pops = torch.stack([synth.population(A, N, seed=b, t0=21540, t1=21560) for b in range(B)])
    mk = lambda: SimEngine(dev(net.x.unsqueeze(0).repeat(B, 1, 1)), net.edge_index, net.edge_attr, net.Nmax,
                           dev(pops.clone()), congestion_constant=net.congestion_constant, seed=9)
    e1, e2 = mk(), mk()
    emb = torch.randn(N, generator=torch.Generator().manual_seed(5)).cuda()
    # frame API: edge ids / fp32 counts; rollout API: one byte each (rank of the chosen out-edge, count)
    ch1, lp1, rw1, ct1 = (torch.zeros((T, N, B), dtype=torch.int32, device="cuda"), torch.zeros((T, B), device="cuda"),
                          torch.zeros((T, B), device="cuda"), torch.zeros((T + 1, N, B), device="cuda"))
    ch2, lp2, rw2, ct2 = (torch.zeros((T, N, B), dtype=torch.uint8, device="cuda"), torch.zeros((T, B), device="cuda"),
                          torch.zeros((T, B), device="cuda"), torch.zeros((T + 1, N, B), dtype=torch.uint8, device="cuda"))
    for e in (e1, e2):
        e.reset()
        e.prepare_policy(emb)
    for t in range(T):
        e1.frame_fused(choice=ch1[t], log_prob=lp1[t], reward=rw1[t], counts=ct1[t + 1])
    times = e2.rollout_fused(T, choice=ch2, log_prob=lp2, reward=rw2, counts=ct2)
    assert len(times) == T + 1 and times[0] == 21540.0 and e1.time == e2.time
    ch2d, ct2d = e2.decode_rollout(True, choice=ch2, counts=ct2)
    assert torch.equal(ch1.permute(0, 2, 1), ch2d) and torch.equal(lp1, lp2) and torch.equal(rw1, rw2)
    assert torch.equal(ct1.permute(0, 2, 1), ct2d)
    assert torch.equal(e1.x, e2.x) and torch.equal(e1.agents, e2.agents)
    # continue: the packed state (incl. the SELECTED_ROAD of the last frame) is where the frame loop left it
    for t in range(T):
        e1.frame_fused(choice=ch1[t], log_prob=lp1[t], reward=rw1[t], counts=ct1[t + 1])
    e2.rollout_fused(T, choice=ch2, log_prob=lp2, reward=rw2, counts=ct2)
    ch2d, ct2d = e2.decode_rollout(True, choice=ch2, counts=ct2)
    assert torch.equal(ch1.permute(0, 2, 1), ch2d) and torch.equal(lp1, lp2) and torch.equal(rw1, rw2)
    assert torch.equal(ct1[1:].permute(0, 2, 1), ct2d[1:])
    assert torch.equal(e1.x, e2.x) and torch.equal(e1.agents, e2.agents) and e1.time == e2.time
    assert float(rw1.abs().sum()) > 0 or T < 5


@pytest.mark.parametrize("knob", ["TARL_INSERT_EPW=1", "TARL_INSERT_EPW=2", "TARL_INSERT_EPW=8", "TARL_INSERT_PAIR=0",
                                  "TARL_CHOICE_QUAD=0", "TARL_DIR_SIBLINGS=0", "TARL_ADDR32=0", "TARL_ROWS_SIBLINGS=0"])
def test_rollout_launcher_developer_knobs(ops, knob):
    """The rollout's kernel variants that a developer knob selects once per process (environments per wave of the insert
    kernel, the one-node-per-step action draw, per-row Direction gathers on a sibling graph, 64-bit addresses in the frame
    kernels — what batches of 2^29 pairs and more run —, consecutive row chunks): each must pass the rollout-vs-frame-loop
    comparison in a process of its own."""
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    k, v = knob.split("=")
    env[k] = v
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.abspath(__file__) + "::test_rollout_launcher_equals_frame_loop", "-k", "300-33-2 or 5-70-2"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def relabelled(net, seed):
    """The same network with its roads renumbered by a seeded permutation (ROAD_INDEX follows the rows)."""
    import dataclasses
    R = net.num_roads
    perm = torch.randperm(R, generator=torch.Generator().manual_seed(seed))      # old id -> new id
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(R)
    x = net.x[inv].clone()
    x[:, 3 * net.Nmax + 6] = torch.arange(R, dtype=torch.float32)
    return dataclasses.replace(net, x=x, edge_index=perm[net.edge_index], critical_number=net.critical_number[inv],
                               congestion_constant=net.congestion_constant[inv])


@pytest.mark.parametrize("B,T", [(5, 40), (130, 9)])
def test_rollout_on_a_relabelled_torus_equals_frame_loop(ops, B, T):
    """A torus keeps its roads in intersection order: the four roads leaving an intersection are consecutive rows and
    share their upstream rows, which the Direction gather of the rollout exploits (one gather per chunk), and its
    all-frames action draw walks four nodes per Philox block. With the roads renumbered at random the first shortcut is
    off (rows of a chunk are unrelated) while the second stays (every node still has out-edges, N % 4 == 0): the rollout
    launcher must still equal the frame loop, bit for bit."""
    from tarl_hip import synth
    from tarl_hip.engine import SimEngine
    net = relabelled(synth.torus_network(5, 5, heterogeneous=True, seed=3), seed=11)
    N = net.num_roads
    A = 700
    pops = torch.stack([synth.population(A, N, seed=b, t0=21540, t1=21560) for b in range(B)])
    mk = lambda: SimEngine(dev(net.x.unsqueeze(0).repeat(B, 1, 1)), net.edge_index, net.edge_attr, net.Nmax,
                           dev(pops.clone()), congestion_constant=net.congestion_constant, seed=9)
    e1, e2 = mk(), mk()
    emb = torch.randn(N, generator=torch.Generator().manual_seed(5)).cuda()
    ch1, lp1, rw1, ct1 = (torch.zeros((T, N, B), dtype=torch.int32, device="cuda"), torch.zeros((T, B), device="cuda"),
                          torch.zeros((T, B), device="cuda"), torch.zeros((T + 1, N, B), device="cuda"))
    ch2, lp2, rw2, ct2 = (torch.zeros((T, N, B), dtype=torch.uint8, device="cuda"), torch.zeros((T, B), device="cuda"),
                          torch.zeros((T, B), device="cuda"), torch.zeros((T + 1, N, B), dtype=torch.uint8, device="cuda"))
    for e in (e1, e2):
        e.reset()
        e.prepare_policy(emb)
    for t in range(T):
        e1.frame_fused(choice=ch1[t], log_prob=lp1[t], reward=rw1[t], counts=ct1[t + 1])
    e2.rollout_fused(T, choice=ch2, log_prob=lp2, reward=rw2, counts=ct2)
    ch2d, ct2d = e2.decode_rollout(True, choice=ch2, counts=ct2)
    assert torch.equal(ch1.permute(0, 2, 1), ch2d) and torch.equal(lp1, lp2) and torch.equal(rw1, rw2)
    assert torch.equal(ct1.permute(0, 2, 1), ct2d)
    assert torch.equal(e1.x, e2.x) and torch.equal(e1.agents, e2.agents) and e1.time == e2.time
    assert float(rw1.abs().sum()) > 0


@pytest.mark.parametrize("W,H,B,T,A", [(5, 5, 5, 30, 700), (3, 3, 70, 7, 300), (8, 8, 3, 40, 1200), (12, 12, 2, 25, 3000)])
def test_env_rollout_equals_frame_loop(ops, W, H, B, T, A):
    """SimEngine.rollout_env (tarl_rollout_env: one workgroup per environment, records in LDS, all T frames in ONE
    launch; 256-thread variant up to 512 roads, 1024-thread variant above) == T calls of frame_fused: actions, log-probs,
    rewards, counts, exported state and agents bit-identical; a second rollout continues from the state it left, and a
    frame-by-frame continuation after it agrees too (the packed records written back are complete)."""
    from tarl_hip import synth
    from tarl_hip.engine import SimEngine
    net = synth.torus_network(W, H, heterogeneous=True, seed=3)
    N = net.num_roads
    pops = torch.stack([synth.population(A, N, seed=b, t0=21540, t1=21560) for b in range(B)])
    mk = lambda: SimEngine(dev(net.x.unsqueeze(0).repeat(B, 1, 1)), net.edge_index, net.edge_attr, net.Nmax,
                           dev(pops.clone()), congestion_constant=net.congestion_constant, seed=9)
    e1, e2 = mk(), mk()
    assert e2.env_rollout_supported
    emb = torch.randn(N, generator=torch.Generator().manual_seed(5)).cuda()
    ch1, lp1, rw1 = (torch.zeros((T, N, B), dtype=torch.int32, device="cuda"), torch.zeros((T, B), device="cuda"),
                     torch.zeros((T, B), device="cuda"))
    ct1 = torch.zeros((T + 1, N, B), device="cuda")
    ch2, lp2, rw2 = (torch.zeros((T, B, N), dtype=torch.uint8, device="cuda"), torch.zeros((T, B), device="cuda"),
                     torch.zeros((T, B), device="cuda"))
    ct2 = torch.zeros((T + 1, B, N), dtype=torch.uint8, device="cuda")
    for e in (e1, e2):
        e.reset()
        e.prepare_policy(emb)
    for rep in range(2):
        for t in range(T):
            e1.frame_fused(choice=ch1[t], log_prob=lp1[t], reward=rw1[t], counts=ct1[t + 1])
        times = e2.rollout_env(T, choice=ch2, log_prob=lp2, reward=rw2, counts=ct2)
        assert len(times) == T + 1 and e1.time == e2.time
        ch2d, ct2d = e2.decode_rollout(False, choice=ch2, counts=ct2)
        assert torch.equal(ch1.permute(0, 2, 1), ch2d), f"actions (rollout {rep})"
        assert torch.equal(lp1, lp2) and torch.equal(rw1, rw2), f"log-prob / reward (rollout {rep})"
        assert torch.equal(ct1[1:].permute(0, 2, 1), ct2d[1:]), f"counts (rollout {rep})"
        assert torch.equal(e1.x, e2.x) and torch.equal(e1.agents, e2.agents), f"state / agents (rollout {rep})"
    assert float(rw1.abs().sum()) > 0
    ca, cb = torch.zeros_like(ch1[0]), torch.zeros_like(ch1[0])
    ra, rb = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda")
    for t in range(5):      # hand the state back to the four-launch path
        e1.frame_fused(choice=ca, reward=ra)
        e2.frame_fused(choice=cb, reward=rb)
        assert torch.equal(ca, cb) and torch.equal(ra, rb)
    assert torch.equal(e1.x, e2.x) and torch.equal(e1.agents, e2.agents)


def test_fused_equals_unfused_on_a_matsim_graph_with_pseudo_nodes(ops, tmp_path):
    """Outside the synthetic pure-road family: the graph config_network builds for a MATSim grid (76 roads + 24 SRC + 24
    DEST pseudo-nodes, SRC -> road and road -> DEST edges with zero turn probability) and a population whose origins /
    destinations are pseudo-nodes. Fused frames, the one-call rollout and the LDS-resident rollout against the per-op
    kernels with the same device noise: identical actions, rewards, state and agents."""
    from src.matsim_io import build_network, build_population
    from tarl_hip import synth
    from tarl_hip.engine import SimEngine
    synth.write_matsim_grid_xml(str(tmp_path / "network.xml"), 4, 6, seed=3, heterogeneous=True)
    synth.write_matsim_population_xml(str(tmp_path / "population.xml"), 4, 6, 260, seed=4, first_departure=21540, spread=40)
    graph, Nmax = build_network(str(tmp_path / "network"))
    agents, _ = build_population(str(tmp_path / "population"), str(tmp_path / "network"))
    agents[0, 2] = 48 * 3600
    N, B, T = graph.x.size(0), 3, 60
    assert N == 124 and int(graph.num_roads) == 76
    mk = lambda fused: SimEngine(dev(graph.x.unsqueeze(0).repeat(B, 1, 1)), graph.edge_index, graph.edge_attr, Nmax,
                                 dev(agents.unsqueeze(0).repeat(B, 1, 1)),
                                 congestion_constant=graph.congestion_constant, seed=5, fused=fused)
    e1, e2, e3 = mk(False), mk(True), mk(True)
    emb = torch.randn(N, generator=torch.Generator().manual_seed(3)).cuda()
    for e in (e1, e2, e3):
        e.reset()
    e2.prepare_policy(emb)
    e3.prepare_policy(emb)
    ch2 = torch.empty((N, B), dtype=torch.int32, device="cuda")
    r2 = torch.empty(B, device="cuda")
    ch3 = torch.zeros((T, B, N), dtype=torch.uint8, device="cuda")
    rw3, ct3 = torch.zeros((T, B), device="cuda"), torch.zeros((T + 1, B, N), dtype=torch.uint8, device="cuda")
    assert e3.env_rollout_supported
    e3.rollout_env(T, choice=ch3, log_prob=None, reward=rw3, counts=ct3)
    ch3 = e3.decode_rollout(False, choice=ch3)[0]
    for s in range(T):
        logits = ops.policy_edge_logits(e1.plan, e1.node_features, emb)
        p = ops.graphdist_softmax(e1.plan, logits)
        _, ch1 = ops.graphdist_sample(e1.plan, p, seed=e2.seed ^ 0x5DEECE66D, counter=s + 1, want_onehot=False,
                                      want_choice=True)
        e1.step(choice=ch1)
        e2.frame_fused(choice=ch2, reward=r2)
        assert torch.equal(ch1, ch2.t()) and torch.equal(ch1, ch3[s]), f"actions frame {s}"
        assert torch.equal(e1.reward, r2) and torch.equal(e1.reward, rw3[s]), f"reward frame {s}"
        assert torch.equal(e1.agents, e2.agents), f"agents frame {s}"
        if s % 10 == 0 or s == T - 1:
            assert torch.equal(e1.x, e2.x), f"state frame {s}"
    assert torch.equal(e1.x, e3.x) and torch.equal(e1.agents, e3.agents)
    assert float(e1.agents[:, :, 8].sum()) > 0 and float(e1.agents[:, :, 7].sum()) > 0      # arrivals and travellers


def test_fused_overflow_paths_equal_unfused(ops):
    """The two LDS lists of the fused frame have in-place fallbacks that the other tests' sizes never reach: the row
    pass's event list (384 of a workgroup's 256 environments x 4 rows) and the insert kernel's candidate list (256 per
    environment). 256 environments on a small torus with every agent due in the first seconds: nearly every (row,
    environment) pair is an event in every frame and thousands of agents are insert candidates at once. Fused frames with
    device Philox noise against the per-op kernels: actions, rewards, counts, state and agents bit-identical."""
    from tarl_hip import synth
    from tarl_hip.engine import SimEngine
    net = synth.torus_network(4, 4, heterogeneous=True, seed=7)
    N, B, A, frames = net.num_roads, 256, 1500, 70
    pops = torch.stack([synth.population(A, N, seed=200 + b, t0=21540, t1=21543) for b in range(B)])
    e1 = SimEngine(dev(net.x.unsqueeze(0).repeat(B, 1, 1)), net.edge_index, net.edge_attr, net.Nmax, dev(pops.clone()),
                   congestion_constant=net.congestion_constant, seed=4, fused=False)
    e2 = SimEngine(dev(net.x.unsqueeze(0).repeat(B, 1, 1)), net.edge_index, net.edge_attr, net.Nmax, dev(pops.clone()),
                   congestion_constant=net.congestion_constant, seed=4, fused=True)
    e1.reset()
    e2.reset()
    emb = torch.randn(N, generator=torch.Generator().manual_seed(8)).cuda()
    e2.prepare_policy(emb)
    ch2 = torch.empty((N, B), dtype=torch.int32, device="cuda")
    rw2, c2 = torch.empty(B, device="cuda"), torch.empty((N, B), device="cuda")
    pop2, wd2 = (torch.zeros((B, N), dtype=torch.uint8, device="cuda") for _ in range(2))
    busiest = 0.0
    for s in range(frames):
        logits = ops.policy_edge_logits(e1.plan, e1.node_features, emb)
        p = ops.graphdist_softmax(e1.plan, logits)
        _, ch1 = ops.graphdist_sample(e1.plan, p, seed=e2.seed ^ 0x5DEECE66D, counter=s + 1, want_onehot=False,
                                      want_choice=True)
        e1.step(choice=ch1)
        e2.frame_fused(choice=ch2, reward=rw2, counts=c2, popped=pop2, withdrawn=wd2)
        assert torch.equal(ch1, ch2.t()), f"actions frame {s}"
        assert torch.equal(e1.reward, rw2) and torch.equal(e1.counts, c2.t()), f"reward / counts frame {s}"
        assert torch.equal(e1.agents, e2.agents), f"agents frame {s}"
        if s % 6 == 5 or s == frames - 1:
            x2 = e2.x
            assert torch.equal(e1.x, x2), f"state frame {s}"
            # event rows of the NEXT frame at least: non-empty rows whose head's departure time has passed
            n_col, dep_col = x2[:, :, 3 * net.Nmax + 1], x2[:, :, 2]
            busiest = max(busiest, float(((n_col > 0) & (dep_col <= e2.time)).float().mean()))
    assert float(e2.agents[:, :, 7].sum() + e2.agents[:, :, 8].sum()) > 300 * B      # > 256 admitted per environment
    assert busiest > 0.45, busiest                     # > 384 of a workgroup's 1024 pairs are event rows in some frame


@pytest.mark.parametrize("W,H,A,window,frames", [(2, 3, 600, 20, 400), (2, 2, 400, 10, 200)])
def test_rows_that_fill_to_their_last_slot_equal_unfused(ops, W, H, A, window, frames):
    """The fused store keeps no physical dead slots for a CLEAN row (count byte bit 7, HD_DIRTY, clear): its pops and
    withdraws skip the last-slot copy and the zero fill, and the export writes zeros there. A row turns dirty for good when
    its FIFO reaches Nmax - 1 agents (the shift then drags the last slot's content in). A small homogeneous torus (Nmax 15)
    with hundreds of agents due within seconds: gridlock relief pushes counts to Nmax - 1 without reaching Nmax, so some
    rows cross that line mid-run while others stay clean. Fused frames with device noise against the per-op kernels:
    actions, rewards, counts, agents every frame and the exported state every few frames, bit-identical; then the
    LDS-resident rollout and the frame loop continue from there and still agree."""
    from tarl_hip import synth
    from tarl_hip.engine import SimEngine
    net = synth.torus_network(W, H, heterogeneous=False, seed=3)
    N, B, Nmax = net.num_roads, 4, net.Nmax
    pops = torch.stack([synth.population(A, N, seed=b, t0=21540, t1=21540 + window) for b in range(B)])
    mk = lambda fused: SimEngine(dev(net.x.unsqueeze(0).repeat(B, 1, 1)), net.edge_index, net.edge_attr, Nmax,
                                 dev(pops.clone()), congestion_constant=net.congestion_constant, seed=5, fused=fused)
    e1, e2 = mk(False), mk(True)
    e1.reset()
    e2.reset()
    emb = torch.randn(N, generator=torch.Generator().manual_seed(1)).cuda()
    e2.prepare_policy(emb)
    assert int((e2.fs.hdp[..., 0] & 0x80).sum()) == 0                     # a fresh network: every row starts clean
    ch2 = torch.empty((N, B), dtype=torch.int32, device="cuda")
    rw2, c2 = torch.empty(B, device="cuda"), torch.empty((N, B), device="cuda")
    fullest = 0
    for s in range(frames):
        logits = ops.policy_edge_logits(e1.plan, e1.node_features, emb)
        p = ops.graphdist_softmax(e1.plan, logits)
        _, ch1 = ops.graphdist_sample(e1.plan, p, seed=e2.seed ^ 0x5DEECE66D, counter=s + 1, want_onehot=False,
                                      want_choice=True)
        e1.step(choice=ch1)
        e2.frame_fused(choice=ch2, reward=rw2, counts=c2)
        assert torch.equal(ch1, ch2.t()), f"actions frame {s}"
        assert torch.equal(e1.reward, rw2) and torch.equal(e1.counts, c2.t()), f"reward / counts frame {s}"
        assert torch.equal(e1.agents, e2.agents), f"agents frame {s}"
        fullest = max(fullest, int(c2.max()))
        if s % 5 == 4 or s == frames - 1:
            assert torch.equal(e1.x, e2.x), f"state frame {s}"
    e2.fs.check_flags()                                                   # nobody reached Nmax (outside the domain)
    dirty = (e2.fs.hdp[..., 0] & 0x80) != 0
    assert fullest == Nmax - 1 and 0 < int(dirty.sum()) < dirty.numel(), (fullest, int(dirty.sum()))
    # hand the mixed clean / dirty state to the LDS-resident rollout and back
    T = 20
    chB, rwB = torch.zeros((T, B, N), dtype=torch.uint8, device="cuda"), torch.zeros((T, B), device="cuda")
    ctB = torch.zeros((T + 1, B, N), dtype=torch.uint8, device="cuda")
    assert e2.env_rollout_supported
    e2.rollout_env(T, choice=chB, log_prob=None, reward=rwB, counts=ctB)
    for t in range(T):
        logits = ops.policy_edge_logits(e1.plan, e1.node_features, emb)
        p = ops.graphdist_softmax(e1.plan, logits)
        _, ch1 = ops.graphdist_sample(e1.plan, p, seed=e2.seed ^ 0x5DEECE66D, counter=frames + t + 1, want_onehot=False,
                                      want_choice=True)
        e1.step(choice=ch1)
        assert torch.equal(e1.reward, rwB[t]), f"rollout reward frame {t}"
    assert torch.equal(e1.x, e2.x) and torch.equal(e1.agents, e2.agents)
    for t in range(10):
        logits = ops.policy_edge_logits(e1.plan, e1.node_features, emb)
        p = ops.graphdist_softmax(e1.plan, logits)
        _, ch1 = ops.graphdist_sample(e1.plan, p, seed=e2.seed ^ 0x5DEECE66D, counter=frames + T + t + 1,
                                      want_onehot=False, want_choice=True)
        e1.step(choice=ch1)
        e2.frame_fused(choice=ch2, reward=rw2)
        assert torch.equal(ch1, ch2.t()) and torch.equal(e1.reward, rw2), f"continuation frame {t}"
    assert torch.equal(e1.x, e2.x) and torch.equal(e1.agents, e2.agents)


def test_raw_selected_road_codes_take_the_exact_direction_pass(ops):
    """A packed state whose SELECTED_ROAD names none of a road's out-edges carries the raw value (code 0x7F); the Direction
    gather's dense pass only notes such a code upstream and the workgroup repeats the pass in its exact form (the raw value
    compared with ROAD_INDEX, as the reference does). Half of the roads of a torus get a valid neighbour, the others values
    that are no neighbour (another road, -1, a value beyond the graph); no action is drawn afterwards (skip_choice), so the
    codes stay. Fused frames against the per-op kernels on the reference layout: state and agents bit-identical."""
    from tarl_hip import synth
    from tarl_hip.engine import SimEngine
    net = synth.torus_network(4, 4, heterogeneous=True, seed=5)
    N, Nmax, B, A = net.num_roads, net.Nmax, 6, 400
    assert ops.Plan(net.edge_index, N).siblings4
    g = torch.Generator().manual_seed(3)
    src_sorted = torch.argsort(net.edge_index[0], stable=True)
    dst_by_rank = net.edge_index[1][src_sorted].view(N, 4)
    sel = dst_by_rank[torch.arange(N), torch.randint(0, 4, (N,), generator=g)].float()
    bogus = torch.tensor([-1.0, float(N + 3), 0.5])[torch.randint(0, 3, (N,), generator=g)]
    other = ((torch.arange(N) + N // 2) % N).float()          # a road on the far side of the torus: not a neighbour
    bogus = torch.where(torch.rand(N, generator=g) < 0.3, other, bogus)
    raw = torch.rand(N, generator=g) < 0.5
    x0 = net.x.unsqueeze(0).repeat(B, 1, 1).contiguous()
    x0[:, :, 3 * Nmax + 5] = torch.where(raw, bogus, sel)
    for b in range(B):                                         # not the same pattern in every environment
        flip = torch.rand(N, generator=g) < 0.2
        x0[b, flip, 3 * Nmax + 5] = sel[flip]
    pops = torch.stack([synth.population(A, N, seed=70 + b, t0=21540, t1=21560) for b in range(B)])
    mk = lambda fused: SimEngine(x0.clone().cuda(), net.edge_index, net.edge_attr, Nmax, pops.clone().cuda(),
                                 congestion_constant=net.congestion_constant, seed=11, fused=fused)
    e1, e2 = mk(False), mk(True)
    assert int((e2.fs.sel8 & 0x7F == 0x7F).sum()) > N * B // 4          # raw codes are there
    rw2 = torch.empty(B, device="cuda")
    moved = 0
    for s in range(60):
        t = float(e1.time)
        e1.noise_counter += 1
        ops.core_step(e1.plan, e1.x, Nmax, e1.ec, t, congestion_constant=e1.cc, seed=e1.seed, counter=e1.noise_counter,
                      chosen=e1.chosen, popped=e1.popped, status=e1.status)
        ops.withdraw_step(e1.plan, e1.x, Nmax, e1.agents, t, want_mask=False)
        ops.insert_step(e1.x, Nmax, e1.agents, t, congestion_constant=e1.cc, scratch=e1.ins_scratch, reward=e1.reward,
                        counts=e1.counts)
        e1.time += e1.timestep
        e2.frame_fused(skip_choice=True, reward=rw2)
        moved += int(e1.popped.sum())
        assert torch.equal(e1.reward, rw2), f"reward frame {s}"
        assert torch.equal(e1.agents, e2.agents), f"agents frame {s}"
        if s % 5 == 4:
            assert torch.equal(e1.x, e2.x), f"state frame {s}"
    assert moved > 0 and float(e1.agents[:, :, 7].sum()) > 0          # agents entered and moved along the fixed choices
    assert int((e2.fs.sel8 & 0x7F == 0x7F).sum()) > N * B // 4          # and the raw codes are still there
