"""GPU: parity / property tests AT THE GEOMETRY THE BENCH LAUNCHES and on config 5's TRAINING leg (round-2 verdict).

* BASELINE config 4 with **B = 16 384** environments through the default ``tarl_fused_rollout`` path — the grids are 64
  environment tiles wide, the insert kernel runs eight environments per wave (auto EPW = 8, ``k_fused_insert2<8>``), the
  action draws are ``k_fused_choice_all<true>`` (four nodes per Philox block), the Direction gather its sibling form.
  Environments {0, 4 095, 8 191, 16 383} of the batch must be bit-identical (actions, log-probs, rewards, counts, final
  state, agents) to the SAME environments simulated alone (``env_base`` = their global id: the noise streams are indexed
  by the environment's global id, not by its lane), and the domain invariants must hold. An index overflow or a
  tile-mapping slip at this size breaks one of the four.
  Reference semantics held: src/agents/base.py:244-331 (insert), src/direction_mpnn.py:103-196.
* two half batches (``env_base`` 0 and B/2) == the whole batch: what a two-rank data-parallel job simulates is exactly
  what one rank simulates with twice the environments.
* BASELINE config 5 (100 000 route edges, 25 000 roads, 262 144 agents): one ``VecPPOTrainer`` collect + update against
  oracle autograd on the same rollout and minibatch frames — the critic with K = 25 001 through
  ``k_critic_fwd_slab_u8x3`` (GAE pass on count bytes) and the split-K minibatch forward, GAE, advantage normalisation,
  clipped loss, backward, Adam — and ``k_edge_mlp_fwd_bf16`` on 100 000 edges against ``oracle/nets.edge_mlp_logits``.
  Reference: src/agents/mpnn_agent.py:35-41,227-231,420-450; src/rl/ppo_trainer.py:129-145."""
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4
BF16_TOL = 2e-2          # tests/test_gpu_edge_mlp.py: bf16 inputs / weights / first hidden activation, fp32 accumulation


def close(a, b, what, tol=TOL):
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} vs scale {scale:.3e}"


def _rollout(eng, T):
    N, B = eng.N, eng.B
    ch = torch.zeros((T, N, B), dtype=torch.uint8, device="cuda")
    ct = torch.zeros((T + 1, N, B), dtype=torch.uint8, device="cuda")
    lp, rw = torch.zeros((T, B), device="cuda"), torch.zeros((T, B), device="cuda")
    leg = torch.zeros((T, B, 2), dtype=torch.int32, device="cuda")
    eng.rollout_fused(T, choice=ch, log_prob=lp, reward=rw, counts=ct, leg=leg)
    return ch, ct, lp, rw, leg


def _invariants(net, x, ag, rw_last, ct_last, t_now):
    """tests/test_gpu_properties.py::_check_invariants for a handful of environments given as (k, N, F) / (k, A, 9)."""
    Nmax = net.Nmax
    n, maxn = x[:, :, 3 * Nmax + 1], x[:, :, 3 * Nmax]
    assert bool((n >= 0).all()) and bool((n <= maxn).all())
    assert torch.equal(rw_last, -n.sum(dim=1)) and torch.equal(ct_last.float(), n)
    slot = torch.arange(Nmax, device=x.device).unsqueeze(0)
    for b in range(x.size(0)):
        live = x[b, :, :Nmax][slot < n[b].unsqueeze(1)].long()
        assert bool((live > 0).all()) and live.numel() == live.unique().numel(), "a queued agent appears twice"
        assert bool((ag[b, live, 7] == 1).all()) and bool((ag[b, live, 8] == 0).all())
        assert int(ag[b, :, 7].sum()) >= live.numel()
        done = ag[b, :, 8] == 1
        assert bool((ag[b, done, 7] == 0).all())
        if bool(done.any()):
            assert bool((ag[b, done, 3] >= ag[b, done, 2]).all()) and float(ag[b, done, 3].max()) < t_now


def test_config4_at_the_bench_batch_size_is_batch_independent():
    from tarl_hip import synth
    from tarl_hip.engine import EPISODE_START, SimEngine
    B, A, T = 16384, 16384, 24
    probe = [0, 4095, 8191, 16383]
    net = synth.torus_network(25, 25)
    N = net.num_roads
    # every agent departs within 800 s: ~20 due per frame and environment — around the packed insert's share of the LDS
    # list (24 entries per environment at EPW = 8), so both its packed path and its one-environment fall-back run; the
    # first agents reach the end of their road after 10 s: pops, enqueues and arrivals from frame 10 on
    pops = synth.population_batch(A, N, B, seed=5, device="cuda", t1=EPISODE_START + 800)
    emb = torch.randn(N, generator=torch.Generator().manual_seed(1)).cuda()
    eng = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                    pops, congestion_constant=net.congestion_constant, seed=11)
    assert eng.plan.handle is not None and eng.B == B
    eng.reset()
    eng.prepare_policy(emb)
    ch, ct, lp, rw, leg = _rollout(eng, T)
    eng.check_flags()
    assert int(leg[..., 0].sum()) > 100 * B and int(leg[..., 1].sum()) > 0        # departures everywhere, some arrivals
    # (a node whose uniform lands beyond its last fp32 threshold draws nothing — ~1e-7 per draw, a hundred of the 1e9 here —
    # and makes that frame's action infeasible: log_prob = -inf, as in the reference, src/reinforcement_learning.py:88-92)
    assert float(rw.abs().sum()) > 0 and float(torch.isfinite(lp).float().mean()) > 0.999
    xb = torch.stack([eng.x[b] for b in probe])           # exports the packed state of the whole batch once
    agb = torch.stack([eng.agents[b] for b in probe])
    pidx = torch.tensor(probe, device="cuda")
    _invariants(net, xb, agb, rw[-1, pidx], ct[-1][:, pidx].t(), eng.time)
    # the same four environments, each simulated ALONE under its global id
    for k, b in enumerate(probe):
        solo = SimEngine(net.x.cuda().unsqueeze(0).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                         pops[b:b + 1].clone(), congestion_constant=net.congestion_constant, seed=11, env_base=b)
        solo.reset()
        solo.prepare_policy(emb)
        ch1, ct1, lp1, rw1, leg1 = _rollout(solo, T)
        solo.check_flags()
        assert torch.equal(ch1[:, :, 0], ch[:, :, b]), f"actions of environment {b}"
        assert torch.equal(lp1[:, 0], lp[:, b]), f"log-probs of environment {b}"
        assert torch.equal(rw1[:, 0], rw[:, b]) and torch.equal(leg1[:, 0], leg[:, b]), f"rewards / leg counts of environment {b}"
        assert torch.equal(ct1[:, :, 0], ct[:, :, b]), f"counts of environment {b}"
        assert torch.equal(solo.x[0], xb[k]) and torch.equal(solo.agents[0], agb[k]), f"final state of environment {b}"
    # every environment's frames are self-consistent at this size: reward == -sum of the count bytes, frame by frame
    assert torch.equal(rw, -ct[1:].sum(dim=1, dtype=torch.float32))


def test_config4_loaded_network_at_the_bench_batch_size_is_batch_independent():
    """The bench's CONGESTED regime at its own geometry: every agent departs within 600 s, a whole rollout of 256 frames at
    B = 16 384 — by its end each environment carries ~7 000 agents, a third of the (road, environment) pairs move something
    in every frame, the row pass's event list overflows into its in-place fall-back in a share of the workgroups, the insert
    kernel runs two environments per wave (due rate ~27 per frame) with blocked candidates queueing in its window, and the
    Direction gather races a sixth of the pairs. Environments {0, 8 191, 16 383} must be bit-identical to the same
    environments simulated ALONE under their global ids; domain invariants and reward = -sum(count bytes) for every
    environment and frame. Reference semantics held: src/direction_mpnn.py:66-196, src/response_mpnn.py:86-127,
    src/agents/base.py:244-403."""
    from tarl_hip import synth
    from tarl_hip.engine import EPISODE_START, SimEngine
    B, A, T = 16384, 16384, 256
    probe = [0, 8191, 16383]
    net = synth.torus_network(25, 25)
    N = net.num_roads
    pops = synth.population_batch(A, N, B, seed=9, device="cuda", t1=EPISODE_START + 600)
    emb = torch.randn(N, generator=torch.Generator().manual_seed(4)).cuda()
    eng = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                    pops, congestion_constant=net.congestion_constant, seed=13)
    eng.reset()
    eng.prepare_policy(emb)
    ch, ct, lp, rw, leg = _rollout(eng, T)
    eng.check_flags()
    # the network is loaded: thousands of agents on the way per environment, most rows hold somebody, and a large share of
    # the pairs changed their count between the last two frames
    on_way = float(-rw[-1].mean())
    moved = float((ct[-1] != ct[-2]).float().mean())
    assert on_way > 5000 and float((ct[-1] > 0).float().mean()) > 0.5 and moved > 0.1, (on_way, moved)
    assert torch.equal(rw, -ct[1:].sum(dim=1, dtype=torch.float32))
    xb = torch.stack([eng.x[b] for b in probe])
    agb = torch.stack([eng.agents[b] for b in probe])
    pidx = torch.tensor(probe, device="cuda")
    _invariants(net, xb, agb, rw[-1, pidx], ct[-1][:, pidx].t(), eng.time)
    for k, b in enumerate(probe):
        solo = SimEngine(net.x.cuda().unsqueeze(0).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                         pops[b:b + 1].clone(), congestion_constant=net.congestion_constant, seed=13, env_base=b)
        solo.reset()
        solo.prepare_policy(emb)
        ch1, ct1, lp1, rw1, leg1 = _rollout(solo, T)
        solo.check_flags()
        assert torch.equal(ch1[:, :, 0], ch[:, :, b]), f"actions of environment {b}"
        assert torch.equal(lp1[:, 0], lp[:, b]), f"log-probs of environment {b}"
        assert torch.equal(rw1[:, 0], rw[:, b]) and torch.equal(leg1[:, 0], leg[:, b]), f"rewards / leg counts of environment {b}"
        assert torch.equal(ct1[:, :, 0], ct[:, :, b]), f"counts of environment {b}"
        assert torch.equal(solo.x[0], xb[k]) and torch.equal(solo.agents[0], agb[k]), f"final state of environment {b}"


def _replay_live_policy_rollout(net, B, A, T, probe, window, pop_seed, eng_seed, emb_seed, min_pops, want_arrivals, max_flips):
    """Roll out T frames of ``tarl_fused_rollout`` with DEVICE noise, then replay the ``probe`` environments with
    ``oracle/sim.env_step`` fed the Gumbel values the kernels consumed (``tarl_noise_export``) and the device's action bytes."""
    from oracle import dist, nets, sim
    from tarl_hip import ops, synth
    from tarl_hip.engine import EPISODE_START, SimEngine
    N, E, Nmax = net.num_roads, net.edge_index.size(1), net.Nmax
    pops = synth.population_batch(A, N, B, seed=pop_seed, device="cuda", t1=EPISODE_START + window)
    emb = torch.randn(N, generator=torch.Generator().manual_seed(emb_seed))
    eng = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, Nmax,
                    pops.clone(), congestion_constant=net.congestion_constant, seed=eng_seed)
    eng.reset()
    eng.prepare_policy(emb.cuda())
    noise0, policy0 = eng.noise_counter + 1, eng.sample_counter + 1       # counters of the rollout's frame 0
    ch, ct, lp, rw, leg = _rollout(eng, T)
    eng.check_flags()
    pidx = torch.tensor(probe, device="cuda")
    ch_p, ct_p = ch[:, :, pidx].cpu(), ct[:, :, pidx].cpu()                 # (T, N, k), (T + 1, N, k)
    lp_p, rw_p, leg_p = lp[:, pidx].cpu(), rw[:, pidx].cpu(), leg[:, pidx].cpu()
    x_fin = torch.stack([eng.x[b] for b in probe]).cpu()
    ag_fin = torch.stack([eng.agents[b] for b in probe]).cpu()
    stats = {"on_way": float(-rw[-1].mean()), "moved": float((ct[-1] != ct[-2]).float().mean())}
    # CSR of the plan on the host: rank r of node i names edge out_eid[out_ptr[i] + r] (stable order of edge_index[0])
    src = net.edge_index[0]
    out_eid = torch.argsort(src, stable=True)
    out_ptr = torch.zeros(N + 1, dtype=torch.long)
    out_ptr[1:] = torch.cumsum(torch.bincount(src, minlength=N), 0)
    adj = net.dense_adjacency()
    c = sim.Cols(Nmax)
    nf0 = net.x[:, 3 * Nmax:]
    gd = dist.GraphDist(nets.policy_logits(nf0, net.edge_index, emb), net.edge_index)
    assert gd.nb_nodes == N
    flips = n_pops = arrivals = 0
    for k, b in enumerate(probe):
        x = net.x.clone()
        x[:, :3 * Nmax] = 0
        x[:, c.N] = 0
        ag = pops[b].cpu().clone()
        ag[:, sim.ON_WAY] = 0
        ag[:, sim.DONE] = 0
        for t in range(T):
            clock = float(EPISODE_START + t)
            g = ops.noise_export(eng.plan, "gumbel", eng.seed, noise0 + t, [b])[0].cpu()
            u = ops.noise_export(eng.plan, "uniform", eng.seed ^ 0x5DEECE66D, policy0 + t, [b])[0].cpu()
            code = ch_p[t, :, k].long()
            drew = (code & 0x80) == 0
            action = torch.zeros(E, dtype=torch.long)
            action[out_eid[out_ptr[:-1][drew] + code[drew]]] = 1
            flips += int((gd.sample(u) != action).sum())      # a node that draws another edge differs in two entries
            lp_o = gd.log_prob(action)
            if bool(drew.all()):
                # (a log-prob is an N-term fp32 sum: 1e-4 of its magnitude, never less than 2 ulp of it)
                tol = max(TOL * abs(float(lp_o)), 2.0 ** -22 * abs(float(lp_o)), TOL)
                assert abs(float(lp_p[t, k]) - float(lp_o)) <= tol, (b, t, float(lp_p[t, k]), float(lp_o))
            else:
                assert float(lp_p[t, k]) == float("-inf") and float(lp_o) == float("-inf")
            before = ((ag[:, sim.ON_WAY] + ag[:, sim.DONE]) > 0).sum(), (ag[:, sim.DONE] > 0).sum()
            out = sim.env_step(x, ag, net.edge_index, net.edge_attr, adj, action, clock, Nmax, gumbel=g,
                               congestion_constant=net.congestion_constant)
            n_pops += int(out["popped"].sum())
            assert torch.equal(x[:, c.N], ct_p[t + 1, :, k].float()), f"counts of environment {b} after frame {t}"
            assert float(out["reward"]) == float(rw_p[t, k]), f"reward of environment {b}, frame {t}"
            after = ((ag[:, sim.ON_WAY] + ag[:, sim.DONE]) > 0).sum(), (ag[:, sim.DONE] > 0).sum()
            assert [int(after[0] - before[0]), int(after[1] - before[1])] == leg_p[t, k].tolist(), f"leg histogram, environment {b}, frame {t}"
        assert torch.equal(x, x_fin[k]), f"final state of environment {b}"
        assert torch.equal(ag, ag_fin[k]), f"agent table of environment {b}"
        arrivals += int(ag[:, sim.DONE].sum())
    # the replay exercised the whole event path: Response pops (agents moving from road to road) and, where asked, arrivals
    assert n_pops > min_pops and (arrivals > 0 or not want_arrivals), (n_pops, arrivals)
    # (measured: 3 nodes of the 1.44 M draws at config 4, 21 of the 1.2 M at config 5, whose running sum is ten times as long —
    # the oracle's thresholds come from CPU exp, the device's from GPU expf, and a
    # running sum of thousands of probabilities carries the one-ulp differences along; the bound only says "the same sampler")
    assert flips <= max_flips, f"{flips} one-hot entries differ between the device draw and GraphDist.sample on the device's uniforms"
    return stats


def test_bench_rollout_replayed_by_the_oracle_with_the_device_noise():
    """The path the bench times — ``tarl_fused_rollout`` with DEVICE noise (Philox Gumbel races, Philox action draws) at
    config 4, B = 16 384, every agent departing within 600 s (the loaded network: event rows, in-place fall-back, the
    two-environments-per-wave insert) — next to the ORACLE: for environments {0, 8 191, 16 383} the Gumbel values the
    kernels consumed are written out by ``tarl_noise_export`` (the same ``philox_uniform`` + ``gumbel_from_u01`` device
    functions) and ``oracle/sim.env_step`` replays all 192 frames from the reset state with those values and the device's
    action bytes: per-node counts, reward and the leg histogram of EVERY frame, the final ``x`` (FIFO slots, clocks,
    SELECTED_ROAD) and the agent table must be bit-exact; the stored log-probs within 1e-4 of
    ``oracle/dist.GraphDist.log_prob``. The actions themselves are the device's (GPU ``expf`` against CPU ``exp`` can move
    a threshold by an ulp: a flip in ~1e-6 of the draws); they are ALSO compared with ``GraphDist.sample`` fed the device's
    uniforms, allowing fifteen differing nodes (thirty one-hot entries) in the 1 440 000 draws. Reference:
    src/reinforcement_learning.py:62-92,222-309, src/direction_mpnn.py:103-146,171-196, src/response_mpnn.py:66-127,
    src/agents/base.py:244-403."""
    from tarl_hip import synth
    st = _replay_live_policy_rollout(synth.torus_network(25, 25), B=16384, A=16384, T=192, probe=[0, 8191, 16383], window=600,
                                     pop_seed=9, eng_seed=13, emb_seed=4, min_pops=1000, want_arrivals=True, max_flips=30)
    assert st["on_way"] > 2000 and st["moved"] > 0.05, st      # loaded, and moving


def test_headline_rollout_replayed_by_the_oracle_with_the_device_noise():
    """The bench's HEADLINE workload itself — departures spread over the whole 61-minute episode (≈4.5 agents due per frame:
    eight environments per wave in the insert kernel), B = 16 384, 96 frames — replayed by the oracle for environments {0, 8 191, 16 383}."""
    from tarl_hip import synth
    st = _replay_live_policy_rollout(synth.torus_network(25, 25), B=16384, A=16384, T=96, probe=[0, 8191, 16383], window=3660,
                                     pop_seed=41, eng_seed=43, emb_seed=8, min_pops=300, want_arrivals=False, max_flips=30)
    assert st["on_way"] > 200, st


def test_default_size_rollout_replayed_by_the_oracle_with_the_device_noise():
    """The same at the bench's DEFAULT size from the end of round 5 on — B = 32 768 environments (twice the largest launch the
    other replays cover: 82 M (road, environment) pairs per frame kernel) —, headline schedule, 64 frames, first / middle /
    last environment."""
    from tarl_hip import synth
    st = _replay_live_policy_rollout(synth.torus_network(25, 25), B=32768, A=16384, T=64, probe=[0, 16383, 32767], window=3660,
                                     pop_seed=47, eng_seed=53, emb_seed=9, min_pops=100, want_arrivals=False, max_flips=30)
    assert st["on_way"] > 100, st


def test_config5_rollout_replayed_by_the_oracle_with_the_device_noise():
    """The same replay on BASELINE config 5's graph and population — 100 000 route edges, 25 000 roads, 262 144 agents, the
    one-wave-per-environment insert — at B = 256, every agent departing within 300 s (≈870 due per frame: more than the insert
    kernel's LDS candidate list holds, so its ordered global-scratch path runs), 24 frames, environments {0, 255}."""
    from tarl_hip import synth
    net = synth.torus_network(25, 250)
    assert (net.num_roads, net.edge_index.size(1)) == (25_000, 100_000)
    st = _replay_live_policy_rollout(net, B=256, A=262_144, T=24, probe=[0, 255], window=300, pop_seed=31, eng_seed=37,
                                     emb_seed=6, min_pops=1000, want_arrivals=False, max_flips=150)
    assert st["on_way"] > 15000, st


@pytest.mark.parametrize("precision,lp_tol", [("x3", 1e-4), ("bf16", 1e-3)])
def test_state_dependent_policy_rollout_replayed_by_the_oracle(precision, lp_tol):
    """The bench's ``state_dependent_policy`` geometry — config 4, B = 2 048, the per-edge MLP head (33 -> 64 -> 32 -> 1, the
    reference's own ``edge_mlp`` initialisation) evaluated on the matrix cores in every frame, GraphDistribution temperature
    2 000, ``tarl_fused_rollout_policy`` — next to the ORACLE: for environments {0, 1 023, 2 047} every frame's observation is
    rebuilt from the oracle's own state (``cat(node_features, agent_features[head])``, src/agents/mpnn_agent.py:166-178), the
    head evaluated with ``oracle/nets.edge_mlp_logits`` (:35-41, :227-231) and ``GraphDist.log_prob`` of the device's action
    compared with the device's stored log-prob (1e-4 of its magnitude for the fp32-accurate kernel, 1e-3 for bf16 logits:
    a log-prob is a 2 500-term sum); then ``oracle/sim.env_step`` advances with the device's action and the Gumbel values the
    kernels consumed: per-node counts, rewards and leg histogram of every frame, final ``x`` and agents bit-exact."""
    from oracle import dist, nets, sim
    from src.agents.mpnn_agent import MPNNPolicyNet
    from tarl_hip import ops, synth
    from tarl_hip.engine import EPISODE_START, SimEngine
    B, A, T, TEMP = 2048, 16384, 64, 2000.0
    probe = [0, 1023, 2047]
    net = synth.torus_network(25, 25)
    N, E, Nmax = net.num_roads, net.edge_index.size(1), net.Nmax
    pops = synth.population_batch(A, N, B, seed=21, device="cuda", t1=EPISODE_START + 300)
    eng = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, Nmax,
                    pops.clone(), congestion_constant=net.congestion_constant, seed=29)
    eng.reset()
    torch.manual_seed(1)
    pol = MPNNPolicyNet(net.edge_index, N, None, device="cuda")
    ws = [pol.edge_mlp[0].weight, pol.edge_mlp[0].bias, pol.edge_mlp[2].weight, pol.edge_mlp[2].bias,
          pol.edge_mlp[4].weight, pol.edge_mlp[4].bias]
    w = ops.EdgeMlpWeights(*(p.data for p in ws))
    ws_cpu = [p.detach().cpu() for p in ws]
    z8 = lambda *shp: torch.zeros(shp, dtype=torch.uint8, device="cuda")
    ch, ct = z8(T, B, N), z8(T + 1, N, B)
    lp, rw = torch.zeros((T, B), device="cuda"), torch.zeros((T, B), device="cuda")
    leg = torch.zeros((T, B, 2), dtype=torch.int32, device="cuda")
    noise0 = eng.noise_counter + 1
    eng.rollout_policy(T, w, precision=precision, temperature=TEMP, policy_seed=77, policy_counter0=5, choice8=ch, log_prob=lp,
                       reward=rw, counts=ct, leg=leg)
    eng.check_flags()
    # (a node whose uniform lands beyond its last fp32 threshold draws nothing, ~1e-7 per draw: a handful of the 3e8 draws here
    # make their frame's action infeasible, log_prob = -inf, as in the reference, src/reinforcement_learning.py:88-92)
    assert float(-rw[-1].mean()) > 500 and float(torch.isfinite(lp).float().mean()) > 0.999
    pidx = torch.tensor(probe, device="cuda")
    ch_p, ct_p = ch[:, pidx].cpu(), ct[:, :, pidx].cpu()                    # (T, 3, N) env-major, (T + 1, N, 3)
    lp_p, rw_p, leg_p = lp[:, pidx].cpu(), rw[:, pidx].cpu(), leg[:, pidx].cpu()
    x_fin = torch.stack([eng.x[b] for b in probe]).cpu()
    ag_fin = torch.stack([eng.agents[b] for b in probe]).cpu()
    src = net.edge_index[0]
    out_eid = torch.argsort(src, stable=True)
    out_ptr = torch.zeros(N + 1, dtype=torch.long)
    out_ptr[1:] = torch.cumsum(torch.bincount(src, minlength=N), 0)
    adj = net.dense_adjacency()
    c = sim.Cols(Nmax)
    n_pops = 0
    for k, b in enumerate(probe):
        x = net.x.clone()
        x[:, :3 * Nmax] = 0
        x[:, c.N] = 0
        ag = pops[b].cpu().clone()
        ag[:, sim.ON_WAY] = 0
        ag[:, sim.DONE] = 0
        for t in range(T):
            clock = float(EPISODE_START + t)
            nf, head = sim.observe(x, Nmax)
            x16 = torch.cat((nf, ag[head.clamp(0, A)]), dim=-1)
            gd = dist.GraphDist(nets.edge_mlp_logits(x16, net.edge_index, net.edge_attr, *ws_cpu), net.edge_index, TEMP)
            code = ch_p[t, k].long()
            drew = (code & 0x80) == 0
            assert bool((code[drew] < 4).all())                        # rank of one of the road's four out-edges
            action = torch.zeros(E, dtype=torch.long)
            action[out_eid[out_ptr[:-1][drew] + code[drew]]] = 1
            lp_o = float(gd.log_prob(action))
            if bool(drew.all()):
                assert abs(float(lp_p[t, k]) - lp_o) <= lp_tol * max(1.0, abs(lp_o)), (b, t, float(lp_p[t, k]), lp_o)
            else:
                assert float(lp_p[t, k]) == float("-inf") and lp_o == float("-inf")
            g = ops.noise_export(eng.plan, "gumbel", eng.seed, noise0 + t, [b])[0].cpu()
            before = ((ag[:, sim.ON_WAY] + ag[:, sim.DONE]) > 0).sum(), (ag[:, sim.DONE] > 0).sum()
            out = sim.env_step(x, ag, net.edge_index, net.edge_attr, adj, action, clock, Nmax, gumbel=g,
                               congestion_constant=net.congestion_constant)
            n_pops += int(out["popped"].sum())
            assert torch.equal(x[:, c.N], ct_p[t + 1, :, k].float()), f"counts of environment {b} after frame {t}"
            assert float(out["reward"]) == float(rw_p[t, k]), f"reward of environment {b}, frame {t}"
            after = ((ag[:, sim.ON_WAY] + ag[:, sim.DONE]) > 0).sum(), (ag[:, sim.DONE] > 0).sum()
            assert [int(after[0] - before[0]), int(after[1] - before[1])] == leg_p[t, k].tolist(), f"leg histogram, environment {b}, frame {t}"
        assert torch.equal(x, x_fin[k]), f"final state of environment {b}"
        assert torch.equal(ag, ag_fin[k]), f"agent table of environment {b}"
    assert n_pops > 300, n_pops


def test_two_half_batches_reproduce_the_whole_batch():
    """What two data-parallel ranks simulate (each its half of the environments, env_base = rank * B / 2, one seed) is
    bit-identical to one rank simulating all of them."""
    from tarl_hip import synth
    from tarl_hip.engine import EPISODE_START, SimEngine
    B, A, T = 512, 2000, 40
    net = synth.torus_network(6, 5, heterogeneous=True, seed=4)
    N = net.num_roads
    pops = synth.population_batch(A, N, B, seed=8, device="cuda", t1=EPISODE_START + 60)
    emb = torch.randn(N, generator=torch.Generator().manual_seed(2)).cuda()

    def run(lo, hi, mode):
        e = SimEngine(net.x.cuda().unsqueeze(0).repeat(hi - lo, 1, 1).contiguous(), net.edge_index, net.edge_attr,
                      net.Nmax, pops[lo:hi].clone(), congestion_constant=net.congestion_constant, seed=21, env_base=lo)
        e.reset()
        e.prepare_policy(emb)
        if mode == "frames":
            ch, ct, lp, rw, _ = _rollout(e, T)
            ch, ct = e.decode_rollout(True, choice=ch, counts=ct)
        else:
            ch = torch.zeros((T, e.B, N), dtype=torch.uint8, device="cuda")
            ct = torch.zeros((T + 1, e.B, N), dtype=torch.uint8, device="cuda")
            lp, rw = torch.zeros((T, e.B), device="cuda"), torch.zeros((T, e.B), device="cuda")
            e.rollout_env(T, choice=ch, log_prob=lp, reward=rw, counts=ct)
            ch, ct = e.decode_rollout(False, choice=ch, counts=ct)
        return ch, ct, lp, rw, e.x.clone(), e.agents.clone()

    for mode in ("frames", "env"):
        whole = run(0, B, mode)
        a, b = run(0, B // 2, mode), run(B // 2, B, mode)
        for w, u, v, what in zip(whole, a, b, ("actions", "counts", "log-probs", "rewards", "state", "agents")):
            dim = 0 if what in ("state", "agents") else 1
            assert torch.equal(w, torch.cat((u, v), dim=dim)), f"{mode}: {what}"
        assert float(whole[3].abs().sum()) > 0


def test_config5_training_leg_matches_oracle_autograd():
    from oracle import dist, nets, ppo
    from src.agents.mpnn_agent import MPNNPolicyNet, MPNNValueNetSimple
    from tarl_hip import synth
    from tarl_hip.engine import EPISODE_START, SimEngine
    from tarl_hip.trainer import VecPPOTrainer
    net = synth.torus_network(25, 250)
    N, E = net.num_roads, net.edge_index.size(1)
    assert (N, E) == (25_000, 100_000)
    B, A, T, M = 128, 262_144, 10, 16
    # 262 144 agents departing within 40 s: the network is loaded inside the 10 frames (counts matter to the critic)
    pops = synth.population_batch(A, N, B, seed=3, device="cuda", t1=EPISODE_START + 40)
    eng = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                    pops, congestion_constant=net.congestion_constant, seed=3)
    torch.manual_seed(0)
    pol = MPNNPolicyNet(net.edge_index, N, None, device="cuda")
    val = MPNNValueNetSimple(net.edge_index, N, device="cuda")
    l = val.final_mlp
    crit = [l[0].weight, l[0].bias, l[2].weight, l[2].bias, l[4].weight, l[4].bias]
    assert crit[0].shape == (64, N + 1)                                   # K = 25 001
    tr = VecPPOTrainer(eng, pol.nodes_embedding.weight, crit, rollout_steps=T, num_epochs=1, sub_batch_size=M,
                       extra_params=[p for n, p in pol.named_parameters() if not n.startswith("nodes_embedding")])
    assert tr.rollout == "frames" and tr.env_minor and B % 128 == 0       # -> critic_forward_slabs (k_critic_fwd_slab_u8x3)
    tr.keep_grad = True
    tr.collect()
    tr.check_flags()
    emb0 = pol.nodes_embedding.weight.detach().cpu().clone()
    crit0 = [p.detach().cpu().clone() for p in crit]
    choice, counts = (t_.cpu() for t_ in eng.decode_rollout(True, choice=tr.choice, counts=tr.counts))
    reward, times = tr.reward.cpu(), tr.times.cpu()
    assert float(counts[-1].sum()) > 1000 * B and float(reward.abs().sum()) > 0
    idx = torch.randperm(T * B, generator=torch.Generator().manual_seed(4))[:M]
    adv_g, tgt_g = tr.advantages()
    out = tr.minibatch_step(adv_g, tgt_g, idx=idx)
    # ---- oracle (torch autograd on the CPU) ----
    emb = emb0.clone().requires_grad_(True)
    cw = [p.clone().requires_grad_(True) for p in crit0]
    with torch.no_grad():
        # MPNNValueNetSimple reads the count column and the clock (src/agents/mpnn_agent.py:428-450): evaluated frame by
        # frame to keep the (T+1, B, N, 7) observation tensor out of host memory
        v_all = torch.stack([nets.critic_value(_obs(counts[t], N), times[t].view(1, 1).expand(B, 1), *cw).squeeze(-1)
                             for t in range(T + 1)])
        nodone = torch.zeros((T, B), dtype=torch.bool)
        adv, tgt = ppo.gae(reward, v_all[:T], v_all[1:], nodone, nodone, average_gae=True)
    close(tr.values.cpu(), v_all, "critic values over all frames (GAE pass on count bytes)")
    close(adv_g.cpu(), adv, "advantage")
    close(tgt_g.cpu(), tgt, "value_target")
    t_idx, b_idx = idx // B, idx % B
    onehot = torch.zeros((M, E), dtype=torch.int64)
    onehot.scatter_(1, choice[t_idx, b_idx].long(), 1)
    nf_mb = _obs(counts[t_idx, b_idx], N)
    with torch.no_grad():
        lp_old = dist.GraphDist(nets.policy_logits(nf_mb, net.edge_index, emb0), net.edge_index).log_prob(onehot)
    # An action's log-prob here is a sum of 25 000 terms of ~ -1.1: magnitude 2^14..2^15, where ONE fp32 ulp is
    # Q = 2^-9 = 2e-3. The importance ratio exp(lp_new - lp_old) of the reference itself therefore carries a quantum of
    # ~0.2 % at this size (its torch.sum is fp32 too), and two correct fp32 evaluations of the same sum differ by a few
    # ulp. Tolerances on the actor side are stated in that quantum: log-probs within 4 ulp, the surrogate loss within
    # 8 ulp x max|A| (two log-probs per ratio), the embedding gradient within 8 ulp of its scale. Everything that does
    # not pass through a 25 000-term fp32 sum (critic values, GAE, advantages, critic loss and gradients, entropy) keeps
    # the 1e-4 bar.
    import math
    assert bool(torch.isfinite(lp_old).all()) and float(lp_old.abs().max()) > 2.0 ** 14
    Q = 2.0 ** (math.floor(math.log2(float(lp_old.abs().max()))) - 23)      # one fp32 ulp at the log-probs' magnitude
    err_lp = float((tr.logp.view(-1).cpu()[idx] - lp_old).abs().max())
    assert err_lp <= 4 * Q, f"sample_log_prob: {err_lp:.3e} (> 4 ulp = {4 * Q:.3e})"
    lp_old = tr.logp.view(-1).cpu()[idx]                    # the update uses the stored one; so does the oracle from here
    d = dist.GraphDist(nets.policy_logits(nf_mb, net.edge_index, emb), net.edge_index)
    lp_new, ent = d.log_prob(onehot), d.entropy()
    value = nets.critic_value(nf_mb, times[t_idx].view(M, 1), *cw).squeeze(-1)
    adv_mb = adv.view(-1)[idx]
    losses = ppo.clip_ppo_loss(lp_new, lp_old, adv_mb, value, tgt.view(-1)[idx], ent)
    (losses["loss_objective"] + losses["loss_critic"] + losses["loss_entropy"]).backward()
    o = out.cpu()
    tol_obj = 8 * Q * float(adv_mb.abs().max())
    assert abs(o[0].item() - losses["loss_objective"].item()) <= tol_obj, \
        f"loss_objective: {o[0].item()} vs {losses['loss_objective'].item()} (tolerance {tol_obj:.3e})"
    for i, k in ((1, "loss_critic"), (2, "loss_entropy")):
        assert abs(o[i].item() - losses[k].item()) <= TOL * max(1.0, abs(losses[k].item())), \
            f"{k}: {o[i].item()} vs {losses[k].item()}"
    g = tr.last_grad.cpu()
    n_emb = emb.grad.numel()
    close(g[:n_emb], emb.grad.reshape(-1), "grad emb", 8 * Q)
    assert float(emb.grad.abs().max()) > 0
    # ---- fp64 referee for the actor side (round-3 verdict) ------------------------------------------------------------------
    # The ulp-sized bounds above compare two fp32 evaluations with each other. Here both are compared with the SAME quantities
    # evaluated once in float64 on the CPU (the reference's formulas, src/reinforcement_learning.py:82-92, on double inputs):
    # the HIP result must be no farther from the exact value than twice the fp32 oracle's own distance, plus 1e-4 of the
    # quantity's scale — i.e. the ulp-sized slack is the fp32 format's, shared by the reference, not the kernels'.
    emb64 = emb0.double().clone().requires_grad_(True)
    d64 = dist.GraphDist(nets.policy_logits(nf_mb.double(), net.edge_index, emb64), net.edge_index)
    lp64 = d64.log_prob(onehot)
    assert lp64.dtype == torch.float64

    def referee(hip, orc, exact, what):
        scale = max(1.0, float(exact.abs().max()))
        e_hip, e_orc = float((hip.double() - exact).abs().max()), float((orc.double() - exact).abs().max())
        assert e_hip <= 2.0 * e_orc + 1e-4 * scale, f"{what}: |hip - fp64| = {e_hip:.3e}, |oracle fp32 - fp64| = {e_orc:.3e}, scale {scale:.3e}"
        return e_hip, e_orc

    with torch.no_grad():
        lp_or32 = dist.GraphDist(nets.policy_logits(nf_mb, net.edge_index, emb0), net.edge_index).log_prob(onehot)
    e_hip, e_orc = referee(lp_old, lp_or32, lp64.detach(), "sample_log_prob")
    # the rollout sums its log-prob terms in 2^-32 fixed point: the stored value is the correctly rounded sum of its fp32
    # terms (half an ulp + the terms' own rounding), at least as close to the exact value as the fp32 tree sum
    assert e_hip <= 2 * Q
    # surrogate loss and embedding gradient with the stored behaviour log-prob, exact arithmetic elsewhere
    adv64, lpo64 = adv_mb.double(), lp_old.double()
    lw = lp64 - lpo64
    gain = torch.min(lw.exp() * adv64, lw.clamp(math.log1p(-0.2), math.log1p(0.2)).exp() * adv64)
    obj64 = -gain.mean()
    (obj64 + (-0.01) * d64.entropy().mean()).backward()
    referee(o[0].double().reshape(1), losses["loss_objective"].detach().reshape(1), obj64.detach().reshape(1), "loss_objective")
    referee(g[:n_emb], emb.grad.reshape(-1), emb64.grad.reshape(-1), "grad emb")
    off = n_emb
    for name, ref in [(f"critic{i}", c.grad) for i, c in enumerate(cw)]:
        n = ref.numel()
        close(g[off:off + n], ref.reshape(-1), f"grad {name}")
        off += n
    assert float(g[off:].abs().sum()) == 0.0                       # dormant heads receive no gradient
    # (the Adam kernel is pinned against torch.optim.Adam in tests/test_gpu_ppo_parity.py; here: the step was applied)
    assert not torch.equal(pol.nodes_embedding.weight.detach().cpu(), emb0)
    assert not torch.equal(crit[0].detach().cpu(), crit0[0])


def _obs(counts, N):
    """(rows, N) counts -> (rows, N, 7) node features with the columns the nets read (count, ROAD_INDEX)."""
    nf = torch.zeros(counts.shape + (7,))
    nf[..., 1] = counts
    nf[..., 6] = torch.arange(N, dtype=torch.float32)
    return nf


def test_config5_edge_mlp_bf16_on_100k_edges():
    """k_edge_mlp_fwd_bf16 (and the fp32 MFMA forward) on config 5's 100 000 edges, observations taken from a loaded
    packed state, against the oracle's restatement of the reference's nn.Sequential (oracle/nets.edge_mlp_logits)."""
    from oracle import nets
    from src.agents.mpnn_agent import MPNNPolicyNet
    from tarl_hip import ops, synth
    from tarl_hip.engine import EPISODE_START, SimEngine
    net = synth.torus_network(25, 250)
    N, E = net.num_roads, net.edge_index.size(1)
    B, A = 3, 262_144
    pops = synth.population_batch(A, N, B, seed=6, device="cuda", t1=EPISODE_START + 30)
    eng = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                    pops, congestion_constant=net.congestion_constant, seed=2)
    eng.reset()
    eng.prepare_policy(torch.randn(N, generator=torch.Generator().manual_seed(3)).cuda())
    for _ in range(20):
        eng.frame_fused()
    eng.check_flags()
    obs = ops.fused_obs16(eng.plan, eng.fs, eng._x, eng.Nmax, eng.agents)            # (B, N, 16) fp32
    obs_b = ops.fused_obs16_bf16(eng.plan, eng.fs, eng._x, eng.Nmax, eng.agents)
    assert float(obs[:, :, 1].sum()) > 1000 and torch.equal(obs_b, obs.to(torch.bfloat16))
    torch.manual_seed(5)
    pol = MPNNPolicyNet(net.edge_index, N, None, device="cuda")
    mm = pol.edge_mlp
    ws = [mm[0].weight, mm[0].bias, mm[2].weight, mm[2].bias, mm[4].weight, mm[4].bias]
    w = ops.EdgeMlpWeights(*(p.data for p in ws))
    ref = nets.edge_mlp_logits(obs.cpu(), net.edge_index, net.edge_attr.expand(B, -1, -1), *[p.detach().cpu() for p in ws])
    assert ref.shape == (B, E)
    close(ops.policy_edge_mlp(eng.plan, obs, eng.ec, w).cpu(), ref, "logits (fp32 MFMA, 100k edges)")
    close(ops.policy_edge_mlp(eng.plan, obs, eng.ec, w, precision="x3").cpu(), ref, "logits (bf16x3 MFMA, 100k edges)")
    lb = ops.policy_edge_mlp(eng.plan, obs, eng.ec, w, bf16=True)
    close(lb.cpu(), ref, "logits (bf16 MFMA, 100k edges)", BF16_TOL)
    assert torch.equal(ops.policy_edge_mlp(eng.plan, obs_b, eng.ec, w), lb)          # bf16 rows in == fp32 rows rounded inside
