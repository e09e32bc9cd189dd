"""GPU: the per-edge MLP policy head (csrc/edge_mlp.hip; reference src/agents/mpnn_agent.py:35-41,227-231) against the
golden fixture generated from the reference's own module (tests/golden/edge_mlp.npz): forward on fp32 MFMA <= 1e-4,
forward on bf16 MFMA within the stated bf16 tolerance, parameter gradients <= 1e-4 (relative to the tensor's scale); the
observation builders; and a PPO update with the head as the (state-dependent) policy against oracle autograd."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4
# bf16 keeps 8 significant bits; inputs (clock times ~2e4, ids), weights and the first hidden activation are each rounded
# once (relative 2^-9), products accumulate in fp32: logits agree to a few 1e-3 of their scale
BF16_TOL = 2e-2


def close(a, b, what, tol=TOL):
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} vs scale {scale:.3e}"


def _weights(g, tag, ops):
    k = lambda n: g[f"{tag}__{n}"].cuda()
    return ops.EdgeMlpWeights(k("0__weight"), k("0__bias"), k("2__weight"), k("2__bias"), k("4__weight"), k("4__bias"))


@pytest.mark.parametrize("tag", ["ref", "biased"])
def test_edge_mlp_forward_and_gradients_golden(tag):
    from tarl_hip import ops
    g = load_golden("edge_mlp")
    ei = g["edge_index"]
    N, E = g["node_features"].size(1), ei.size(1)
    plan = ops.Plan(ei, N)
    ec = ops.EdgeConst(g["edge_attr"], "cuda")
    w = _weights(g, tag, ops)
    obs = ops.policy_obs16(g["node_features"].cuda(), g["agent_index"].cuda(), g["agent_features"].cuda())
    ref_x = torch.cat((g["node_features"], g["agent_features"][g["agent_index"]]), dim=-1)
    assert torch.equal(obs.cpu(), ref_x)
    logits = ops.policy_edge_mlp(plan, obs, ec, w)
    close(logits.cpu(), g[f"{tag}__logits"], "logits (fp32 MFMA)")
    # unbatched call == row 0
    l0 = ops.policy_edge_mlp(plan, obs[:1].contiguous(), ec, w)
    assert torch.equal(l0[0], logits[0])
    lb = ops.policy_edge_mlp(plan, obs, ec, w, bf16=True)
    close(lb.cpu(), g[f"{tag}__logits"], "logits (bf16 MFMA)", BF16_TOL)
    assert not torch.equal(lb, logits)
    # fp32 accuracy on the bf16 pipe (operands split into three exact bf16 pieces): the same 1e-4 contract against the
    # reference's module, and as close to the fp64 evaluation of that module as the fp32 MFMA chain is (referee: the
    # oracle's restatement of the head in float64)
    l3 = ops.policy_edge_mlp(plan, obs, ec, w, precision="x3")
    close(l3.cpu(), g[f"{tag}__logits"], "logits (bf16x3 MFMA)")
    from oracle import nets
    ws64 = [t.double() for t in (w.w1.cpu(), w.b1.cpu(), w.w2.cpu(), w.b2.cpu(), w.w3.cpu().view(1, -1), w.b3.cpu())]
    ref64 = nets.edge_mlp_logits(ref_x.double(), ei, g["edge_attr"].double().reshape(1, E, 1).expand(ref_x.size(0), E, 1), *ws64)
    scale = max(1.0, float(ref64.abs().max()))
    e3, e32 = float((l3.cpu().double() - ref64).abs().max()), float((logits.cpu().double() - ref64).abs().max())
    assert e3 <= 4 * e32 + 1e-6 * scale, (e3, e32, scale)
    grads = [torch.zeros_like(t) for t in (w.w1, w.b1, w.w2, w.b2, w.w3, w.b3)]
    ops.policy_edge_mlp_bwd(plan, obs, ec, w, g["coef"].cuda(), grads)
    for gr, name in zip(grads, ("0__weight", "0__bias", "2__weight", "2__bias", "4__weight", "4__bias")):
        close(gr.cpu().reshape(-1), g[f"{tag}__grad__{name}"].reshape(-1), f"grad {name}")
    # deterministic: a second evaluation gives the same bits
    grads2 = [torch.zeros_like(t) for t in grads]
    ops.policy_edge_mlp_bwd(plan, obs, ec, w, g["coef"].cuda(), grads2)
    assert all(torch.equal(a, b) for a, b in zip(grads, grads2))


def test_edge_mlp_ragged_tile_and_large_batch():
    """E not a multiple of the 128-edge tile, many samples: against the oracle restatement."""
    from oracle import nets
    from tarl_hip import ops, synth
    net = synth.torus_network(3, 4, heterogeneous=True, seed=2)            # 48 roads, 192 edges
    N = net.num_roads
    ei = net.edge_index[:, :-7]                                            # 185 edges: a ragged last tile
    ea = net.edge_attr[:-7]
    plan = ops.Plan(ei, N)
    ec = ops.EdgeConst(ea, "cuda")
    gen = torch.Generator().manual_seed(3)
    ws = [torch.randn(s, generator=gen) * 0.2 for s in ((64, 33), (64,), (32, 64), (32,), (1, 32), (1,))]
    w = ops.EdgeMlpWeights(*[t.cuda() for t in ws])
    x16 = torch.randn((70, N, 16), generator=gen) * 3.0
    ref = nets.edge_mlp_logits(x16, ei, ea.expand(70, -1, -1), *ws)
    out = ops.policy_edge_mlp(plan, x16.cuda(), ec, w)
    close(out.cpu(), ref, "logits")


@pytest.mark.parametrize("drop,M", [(0, 1), (7, 5), (160, 3), (191, 2), (37, 4097)])
def test_edge_mlp_chunk_walk_edge_cases_all_precisions(drop, M):
    """The forwards' walk over their 32-edge chunks (``emr_walk``: a wave's range cut into per-sample segments, ids one
    chunk ahead of the rows, the last chunk re-requested past a segment's end) on the shapes that stress it: one sample,
    a ragged last chunk, a single (ragged) chunk per sample, ONE edge, and more samples than waves so that every wave's
    range straddles sample boundaries — fp32, bf16 (fp32 and bf16 rows) and x3 against the fp64 evaluation of the head."""
    from oracle import nets
    from tarl_hip import ops, synth
    net = synth.torus_network(3, 4, heterogeneous=True, seed=5)            # 48 roads, 192 edges
    N = net.num_roads
    E = net.edge_index.size(1) - drop
    ei, ea = net.edge_index[:, :E], net.edge_attr[:E]
    plan = ops.Plan(ei, N)
    ec = ops.EdgeConst(ea, "cuda")
    gen = torch.Generator().manual_seed(100 + drop)
    ws = [torch.randn(s, generator=gen) * 0.2 for s in ((64, 33), (64,), (32, 64), (32,), (1, 32), (1,))]
    w = ops.EdgeMlpWeights(*[t.cuda() for t in ws])
    x16 = torch.randn((M, N, 16), generator=gen) * 2.0
    ref = nets.edge_mlp_logits(x16.double(), ei, ea.double().expand(M, -1, -1), *[t.double() for t in ws])
    scale = float(ref.abs().max())
    xg = x16.cuda()
    for prec, obs, tol in (("fp32", xg, 1e-5), ("x3", xg, 1e-4), ("bf16", xg, 2e-2), (None, xg.to(torch.bfloat16), 2e-2)):
        out = torch.full((M, E), float("nan"), device="cuda")
        ops.policy_edge_mlp(plan, obs, ec, w, precision=prec, out=out)
        err = float((out.cpu().double() - ref).abs().max())
        assert err <= tol * scale, (prec, obs.dtype, drop, M, err, scale)      # (a NaN left in `out` = a chunk nobody wrote)


def test_fused_obs16_matches_the_exported_state():
    from tarl_hip import ops, synth
    from tarl_hip.engine import SimEngine
    net = synth.torus_network(4, 4, heterogeneous=True, seed=1)
    N, B, A = net.num_roads, 5, 500
    pops = torch.stack([synth.population(A, N, seed=b, t0=21540, t1=21560) for b in range(B)])
    eng = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                    pops.cuda(), congestion_constant=net.congestion_constant, seed=4)
    eng.reset()
    eng.prepare_policy(torch.randn(N, generator=torch.Generator().manual_seed(1)).cuda())
    for _ in range(25):
        eng.frame_fused()
    obs = ops.fused_obs16(eng.plan, eng.fs, eng._x, eng.Nmax, eng.agents)
    x = eng.x                                                              # exports the packed state
    nf = x[:, :, 3 * eng.Nmax:]
    head = x[:, :, 0].long()
    ref = torch.cat((nf, torch.gather(eng.agents, 1, head.unsqueeze(-1).expand(B, N, 9))), dim=-1)
    assert torch.equal(obs, ref)
    assert float(obs[:, :, 1].sum()) > 0 and float(obs[:, :, 14].sum()) > 0   # counts and ON_WAY flags are live
    # the bf16 observation (round to nearest even) and the fp32 rows of a few environments
    ob = ops.fused_obs16_bf16(eng.plan, eng.fs, eng._x, eng.Nmax, eng.agents)
    assert ob.dtype == torch.bfloat16 and torch.equal(ob, ref.to(torch.bfloat16))
    env = torch.tensor([3, 0, 3], dtype=torch.int32, device="cuda")
    slot = torch.tensor([1, 2, 0], dtype=torch.int32, device="cuda")
    rows = ops.fused_obs16_rows(eng.plan, eng.fs, eng._x, eng.Nmax, eng.agents, env, slot,
                                torch.zeros((3, N, 16), device="cuda"))
    assert torch.equal(rows[1], ref[3]) and torch.equal(rows[2], ref[0]) and torch.equal(rows[0], ref[3])
    # the bf16 MLP gives the same logits from either observation
    from src.agents.mpnn_agent import MPNNPolicyNet
    torch.manual_seed(2)
    pol = MPNNPolicyNet(net.edge_index, N, None, device="cuda")
    w = ops.EdgeMlpWeights(*(p.data for p in (pol.edge_mlp[0].weight, pol.edge_mlp[0].bias, pol.edge_mlp[2].weight,
                                              pol.edge_mlp[2].bias, pol.edge_mlp[4].weight, pol.edge_mlp[4].bias)))
    assert torch.equal(ops.policy_edge_mlp(eng.plan, obs, eng.ec, w, bf16=True), ops.policy_edge_mlp(eng.plan, ob, eng.ec, w))


@pytest.mark.parametrize("bf16_rollout", [False, True])
def test_ppo_update_with_edge_mlp_policy_matches_oracle_autograd(bf16_rollout):
    """One PPO update with the state-dependent head (per-frame observation -> MLP -> GraphDistribution in the rollout; MLP
    forward / backward in the minibatch step) against the oracle with torch autograd on the SAME rollout and frames."""
    from oracle import dist, nets, ppo
    from src.agents.mpnn_agent import MPNNPolicyNet, MPNNValueNetSimple
    from tarl_hip import synth
    from tarl_hip.engine import SimEngine
    from tarl_hip.trainer import VecPPOTrainer
    net = synth.torus_network(4, 4, heterogeneous=True, seed=2)
    N, E = net.num_roads, net.edge_index.size(1)
    B, A, T, M = 128, 300, 12, 16
    TEMP = 500.0            # the head sees raw features (clock times ~2e4): a temperature keeps the softmax from saturating
    pops = torch.stack([synth.population(A, N, seed=b, t0=21540, t1=21550) for b in range(B)])
    eng = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                    pops.cuda(), congestion_constant=net.congestion_constant, seed=3)
    torch.manual_seed(0)
    pol = MPNNPolicyNet(net.edge_index, N, None, device="cuda")
    val = MPNNValueNetSimple(net.edge_index, N, device="cuda")
    l = val.final_mlp
    crit = [l[0].weight, l[0].bias, l[2].weight, l[2].bias, l[4].weight, l[4].bias]
    mlp = [pol.edge_mlp[0].weight, pol.edge_mlp[0].bias, pol.edge_mlp[2].weight, pol.edge_mlp[2].bias,
           pol.edge_mlp[4].weight, pol.edge_mlp[4].bias]
    extra = [p for n, p in pol.named_parameters() if not n.startswith("nodes_embedding")]
    tr = VecPPOTrainer(eng, pol.nodes_embedding.weight, crit, rollout_steps=T, num_epochs=1, sub_batch_size=M,
                       extra_params=extra, policy="edge_mlp", edge_mlp_params=mlp, policy_bf16=bf16_rollout,
                       temperature=TEMP)
    assert tr.rollout == "frames+policy"
    tr.keep_grad = True
    idx = torch.randperm(T * B, generator=torch.Generator().manual_seed(4))[:M]
    tr.obs_idx = idx
    tr.collect()
    mlp0 = [p.detach().cpu().clone() for p in mlp]
    crit0 = [p.detach().cpu().clone() for p in crit]
    assert tr.counts.dtype == torch.uint8 and tr.choice.dtype == torch.uint8     # byte rollout buffers
    counts = tr.counts.permute(0, 2, 1).float().cpu()    # (T+1, B, N)
    choice = eng.decode_rollout(False, choice=tr.choice)[0].cpu()               # rank bytes -> (T, B, N) edge ids
    reward, times = tr.reward.cpu(), tr.times.cpu()
    x16 = tr.obs_mb.cpu()                                 # observations of the minibatch frames (tested on their own above)
    assert float(reward.abs().sum()) > 0 and float(x16[:, :, 1].sum()) > 0
    adv_g, tgt_g = tr.advantages()
    out = tr.minibatch_step(adv_g, tgt_g)
    # ---- oracle ----
    wm = [p.clone().requires_grad_(True) for p in mlp0]
    cw = [p.clone().requires_grad_(True) for p in crit0]
    nf_all = torch.zeros((T + 1, B, N, 7))
    nf_all[..., 1] = counts
    with torch.no_grad():
        v_all = nets.critic_value(nf_all, times.view(T + 1, 1, 1).expand(T + 1, B, 1), *cw).squeeze(-1)
        nodone = torch.zeros((T, B), dtype=torch.bool)
        adv, tgt = ppo.gae(reward, v_all[:T], v_all[1:], nodone, nodone, average_gae=True)
    close(adv_g.cpu(), adv, "advantage")
    t_idx, b_idx = idx // B, idx % B
    onehot = torch.zeros((M, E), dtype=torch.int64)
    onehot.scatter_(1, choice[t_idx, b_idx].long(), 1)
    ea = net.edge_attr.expand(M, -1, -1)
    with torch.no_grad():
        lp_old = dist.GraphDist(nets.edge_mlp_logits(x16, net.edge_index, ea, *mlp0), net.edge_index, TEMP).log_prob(onehot)
    if not bf16_rollout:     # the rollout's stored behaviour log-prob == the exact one (fp32 rollout)
        close(tr.logp.view(-1).cpu()[idx], lp_old, "sample_log_prob")
    lp_old = tr.logp.view(-1).cpu()[idx]
    d = dist.GraphDist(nets.edge_mlp_logits(x16, net.edge_index, ea, *wm), net.edge_index, TEMP)
    lp_new, ent = d.log_prob(onehot), d.entropy()
    value = nets.critic_value(nf_all[t_idx, b_idx], times[t_idx].view(M, 1), *cw).squeeze(-1)
    losses = ppo.clip_ppo_loss(lp_new, lp_old, adv.view(-1)[idx], value, tgt.view(-1)[idx], ent)
    (losses["loss_objective"] + losses["loss_critic"] + losses["loss_entropy"]).backward()
    o = out.cpu()
    for i, k in enumerate(["loss_objective", "loss_critic", "loss_entropy"]):
        assert abs(o[i].item() - losses[k].item()) <= TOL * max(1.0, abs(losses[k].item())), k
    g = tr.last_grad.cpu()
    assert float(g[:N].abs().sum()) == 0.0                                  # the embedding is not on this policy's path
    off = N + sum(c.numel() for c in crit0) + sum(p.numel() for p in extra[:4])       # edge_mlp_test comes first
    for i, ref in enumerate(wm):
        n = ref.numel()
        close(g[off:off + n], ref.grad.reshape(-1), f"grad edge_mlp[{i}]")
        off += n
    assert sum(float(r.grad.abs().sum()) for r in wm) > 0
    for p_gpu, p0, gr, name in [(mlp[i], mlp0[i], wm[i].grad, f"edge_mlp{i}") for i in range(6)]:
        q = p0.clone()
        ppo.adam_step(q, gr, torch.zeros_like(q), torch.zeros_like(q), 1)
        close(p_gpu.detach().cpu(), q, f"param {name}")


@pytest.mark.parametrize("precision", ["fp32", "bf16", "x3"])
def test_rollout_policy_call_equals_its_launches_one_by_one(precision):
    """tarl_fused_rollout_policy (T frames of observation -> MLP -> sample + log-prob -> simulation frame queued by one
    foreign call) against the same entry points called one by one from Python on a twin engine: identical action bytes,
    log-probs, rewards, counts, kept observations and final state; the per-step logs add up."""
    from src.agents.mpnn_agent import MPNNPolicyNet
    from tarl_hip import ops, synth
    from tarl_hip.engine import SimEngine
    net = synth.torus_network(5, 4, heterogeneous=True, seed=3)
    N, E = net.num_roads, net.edge_index.size(1)
    B, A, T, TEMP = 96, 200, 14, 400.0
    pops = torch.stack([synth.population(A, N, seed=40 + b, t0=21540, t1=21552) for b in range(B)])

    def engine():
        e = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                      pops.cuda(), congestion_constant=net.congestion_constant, seed=9)
        e.reset()
        return e
    torch.manual_seed(1)
    pol = MPNNPolicyNet(net.edge_index, N, None, device="cuda")
    w = ops.EdgeMlpWeights(*(p.data for p in (pol.edge_mlp[0].weight, pol.edge_mlp[0].bias, pol.edge_mlp[2].weight,
                                              pol.edge_mlp[2].bias, pol.edge_mlp[4].weight, pol.edge_mlp[4].bias)))
    e1, e2 = engine(), engine()
    gen = torch.Generator().manual_seed(6)
    flat = torch.randperm(T * B, generator=gen)[:20]
    order = torch.argsort(flat, stable=True)
    t_sorted = (flat[order] // B).tolist()
    kenv, kslot = (flat[order] % B).to(torch.int32).cuda(), order.to(torch.int32).cuda()
    ptr = [sum(1 for v in t_sorted if v < t) for t in range(T + 1)]
    z8 = lambda *shp: torch.zeros(shp, dtype=torch.uint8, device="cuda")
    zf = lambda *shp: torch.zeros(shp, dtype=torch.float32, device="cuda")
    m = 2
    ch1, ct1, lp1, rw1, keep1 = z8(T, B, N), z8(T + 1, N, B), zf(T, B), zf(T, B), zf(20, N, 16)
    leg1, dtt1, ev1 = torch.zeros((T, B, 2), dtype=torch.int32, device="cuda"), zf(T, N, m), z8(T, N, m)
    e1.rollout_policy(T, w, precision=precision, temperature=TEMP, policy_seed=77, policy_counter0=5, choice8=ch1, log_prob=lp1,
                      reward=rw1, counts=ct1, keep=(ptr, kenv, kslot), obs_keep=keep1, metrics_envs=m, dtt_node=dtt1,
                      events=ev1, leg=leg1)
    # twin: one entry point at a time
    ch2, lp2, rw2, keep2 = z8(T, B, N), zf(T, B), zf(T, B), zf(20, N, 16)
    cf = zf(N, B)
    for t in range(T):
        obs = ops.fused_obs16(e2.plan, e2.fs, e2._x, e2.Nmax, e2.agents)
        if ptr[t + 1] > ptr[t]:
            j = slice(ptr[t], ptr[t + 1])
            keep2.index_copy_(0, kslot[j].long(), obs.index_select(0, kenv[j].long()))
        logits = ops.policy_edge_mlp(e2.plan, obs, e2.ec, w, precision=precision)
        ops.graphdist_rollout(e2.plan, logits, TEMP, seed=77, counter=5 + t, choice8=ch2[t], sel8=e2.fs.sel8,
                              log_prob=lp2[t])
        e2.frame_fused(skip_choice=True, reward=rw2[t], counts=cf)
        assert torch.equal(ct1[t + 1].float(), cf), f"counts frame {t}"
    assert torch.equal(ch1, ch2) and torch.equal(lp1, lp2) and torch.equal(rw1, rw2) and torch.equal(keep1, keep2)
    assert e1.time == e2.time and torch.equal(e1.x, e2.x) and torch.equal(e1.agents, e2.agents)
    assert float(rw1.abs().sum()) > 0 and int(leg1[..., 0].sum()) > 0 and bool(torch.isfinite(lp1).all())
    assert int(leg1[..., 0].sum()) == int((e1.agents[:, :, 7] + e1.agents[:, :, 8] > 0).sum())     # departed == on way + done
    assert bool(((ch1 & 0x7F) < 4).all())                     # rank bytes of a torus: four out-edges per road


def test_policy_collect_across_an_episode_end_equals_a_manual_loop():
    """VecPPOTrainer(policy="edge_mlp").collect() with 300 s frames: the episode ends inside the batch (clock past 7 h), the
    collector resets and goes on. The trainer queues one foreign call per episode segment; a twin engine driven entry
    point by entry point, with the reset done by hand, must give the same action bytes, log-probs, rewards, counts, clocks,
    kept observations and done marks."""
    from src.agents.mpnn_agent import MPNNPolicyNet, MPNNValueNetSimple
    from tarl_hip import ops, synth
    from tarl_hip.engine import EPISODE_END, SimEngine
    from tarl_hip.trainer import VecPPOTrainer
    net = synth.torus_network(4, 4, heterogeneous=True, seed=5)
    N = net.num_roads
    B, A, T, M, TEMP = 128, 120, 20, 16, 300.0
    pops = torch.stack([synth.population(A, N, seed=60 + b, t0=21540, t1=23000) for b in range(B)])

    def engine():
        return SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                         pops.clone().cuda(), congestion_constant=net.congestion_constant, seed=3, timestep=300)
    torch.manual_seed(0)
    pol = MPNNPolicyNet(net.edge_index, N, None, device="cuda")
    val = MPNNValueNetSimple(net.edge_index, N, device="cuda")
    l = val.final_mlp
    crit = [l[0].weight, l[0].bias, l[2].weight, l[2].bias, l[4].weight, l[4].bias]
    mlp = [pol.edge_mlp[0].weight, pol.edge_mlp[0].bias, pol.edge_mlp[2].weight, pol.edge_mlp[2].bias,
           pol.edge_mlp[4].weight, pol.edge_mlp[4].bias]
    extra = [p for n, p in pol.named_parameters() if not n.startswith("nodes_embedding")]
    e1 = engine()
    tr = VecPPOTrainer(e1, pol.nodes_embedding.weight, crit, rollout_steps=T, num_epochs=1, sub_batch_size=M,
                       extra_params=extra, policy="edge_mlp", edge_mlp_params=mlp, policy_bf16=True, temperature=TEMP)
    idx = torch.randperm(T * B, generator=torch.Generator().manual_seed(9))[:M]
    tr.obs_idx = idx
    tr.collect()
    done = tr.done_frames.tolist()
    assert sum(done) == 1 and done.index(True) == 12          # (25200 - 21540) / 300 = 12.2: frame 12 passes 7 h
    # twin, by hand
    e2 = engine()
    e2.reset()
    w = ops.EdgeMlpWeights(*(p.data for p in mlp))
    order = torch.argsort(idx, stable=True)
    ch2 = torch.zeros((T, B, N), dtype=torch.uint8, device="cuda")
    ct2 = torch.zeros((T + 1, N, B), device="cuda")
    lp2, rw2 = torch.zeros((T, B), device="cuda"), torch.zeros((T, B), device="cuda")
    keep2 = torch.zeros((M, N, 16), device="cuda")
    clocks = []
    pseed = tr.seed ^ 0x5DEECE66D
    for t in range(T):
        clocks.append(float(e2.time))
        obs = ops.fused_obs16(e2.plan, e2.fs, e2._x, e2.Nmax, e2.agents)
        for j in order.tolist():
            if int(idx[j]) // B == t:
                keep2[j] = obs[int(idx[j]) % B]
        logits = ops.policy_edge_mlp(e2.plan, obs, e2.ec, w, bf16=True)
        ops.graphdist_rollout(e2.plan, logits, TEMP, seed=pseed, counter=1 + t, choice8=ch2[t], sel8=e2.fs.sel8,
                              log_prob=lp2[t])
        is_done = e2.frame_fused(skip_choice=True, reward=rw2[t], counts=ct2[t + 1])
        assert is_done == (e2.time > EPISODE_END) == done[t]
        if is_done and t + 1 < T:
            e2.reset()
            ct2[t + 1].zero_()
    clocks.append(float(e2.time))
    assert torch.equal(tr.choice, ch2) and torch.equal(tr.logp, lp2) and torch.equal(tr.reward, rw2)
    assert torch.equal(tr.counts.float(), ct2) and tr.times.tolist() == clocks
    assert torch.equal(tr.obs_mb, keep2)
    assert torch.equal(e1.x, e2.x) and torch.equal(e1.agents, e2.agents)
    assert float(rw2.abs().sum()) > 0 and float(rw2[13:].abs().sum()) > 0          # agents are on the road again after the reset
    adv, tgt = tr.advantages()                                  # GAE with the done mask runs on the byte buffers
    assert bool(torch.isfinite(adv).all()) and bool(torch.isfinite(tgt).all())


def test_mirror_policy_net_with_edge_mlp_head_golden_and_cli(tmp_path, monkeypatch, capsys):
    """MPNNPolicyNet(policy_head="edge_mlp").forward == the reference module's evaluation of its edge_mlp (golden), with
    autograd through the HIP kernels; and `main.py --algo mpnn+ppo --policy-head edge_mlp` trains that head end to end."""
    import importlib
    import sys
    from conftest import PKG
    sys.path.insert(0, PKG)
    from src.agents.mpnn_agent import MPNNPolicyNet
    g = load_golden("edge_mlp")
    ei = g["edge_index"]
    N = g["node_features"].size(1)
    pol = MPNNPolicyNet(ei, N, None, device="cuda")
    pol.policy_head = "edge_mlp"
    pol.agent_features = g["agent_features"].cuda()
    with torch.no_grad():
        for k in ("0", "2", "4"):
            getattr(pol.edge_mlp, k).weight.copy_(g[f"biased__{k}__weight"])
            getattr(pol.edge_mlp, k).bias.copy_(g[f"biased__{k}__bias"])
    ea = g["edge_attr"].cuda()
    logits = pol(g["node_features"].cuda(), ea.expand(3, -1, -1), g["agent_index"].cuda())
    close(logits.detach().cpu(), g["biased__logits"], "mirror logits")
    (logits * g["coef"].cuda()).sum().backward()
    for k in ("0", "2", "4"):
        close(getattr(pol.edge_mlp, k).weight.grad.cpu().reshape(-1), g[f"biased__grad__{k}__weight"].reshape(-1), f"grad {k}.weight")
        close(getattr(pol.edge_mlp, k).bias.grad.cpu().reshape(-1), g[f"biased__grad__{k}__bias"].reshape(-1), f"grad {k}.bias")
    l1 = pol(g["node_features"][0].cuda(), ea, g["agent_index"][0].cuda())          # unbatched, like the env's frames
    assert l1.shape == (ei.size(1),) and torch.allclose(l1, logits[0].detach(), atol=1e-4)
    # end to end through the CLI
    monkeypatch.chdir(tmp_path)
    main = importlib.import_module("main").main
    from src.runner import Runner
    created = []
    orig_setup = Runner.setup

    def spy_setup(self):
        orig_setup(self)
        created.append(self)
    monkeypatch.setattr(Runner, "setup", spy_setup)
    main(["--algo", "mpnn+ppo", "--mode", "train", "--scenario", "synthetic-1024-1024", "--rollout-steps", "24",
          "--epochs", "2", "--steps", "6", "--num-envs", "4", "--policy-head", "edge_mlp", "--output-dir",
          str(tmp_path / "run"), "--seed", "1"])
    assert "Simulation Summary" in capsys.readouterr().out
    r = created[-1]
    torch.manual_seed(1)
    fresh = MPNNPolicyNet(r.policy_net.edge_index, r.policy_net.num_nodes, None, device="cuda")
    assert not torch.equal(fresh.edge_mlp[4].weight, r.policy_net.edge_mlp[4].weight)      # the head trained
    assert torch.equal(fresh.nodes_embedding.weight, r.policy_net.nodes_embedding.weight)   # the embedding is off the path


@pytest.mark.parametrize("W,H,B", [(5, 4, 96), (3, 3, 1), (7, 5, 130), (8, 8, 64)])
def test_set_actions_turns_env_major_bytes_into_the_selected_road_column(W, H, B):
    """tarl_fused_set_actions: choice8 (B, N) -> sel8 (N, B), 64 x 64 byte tiles (ragged on both sides here); a byte with
    bit 7 set ("drew nothing") keeps the road's previous rank and comes back completed. Afterwards the exported
    SELECTED_ROAD column names the chosen out-edge's target."""
    from tarl_hip import ops, synth
    from tarl_hip.engine import SimEngine
    net = synth.torus_network(W, H, heterogeneous=True, seed=2)
    N = net.num_roads
    pops = torch.stack([synth.population(50, N, seed=b) for b in range(B)])
    e = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                  pops.cuda(), congestion_constant=net.congestion_constant, seed=1)
    e.reset()
    g = torch.Generator().manual_seed(B + N)
    first = torch.randint(0, 4, (B, N), generator=g, dtype=torch.uint8).cuda()
    ops.fused_set_actions(e.plan, e.fs, first)
    assert torch.equal(e.fs.sel8, first.t())
    second = torch.randint(0, 4, (B, N), generator=g, dtype=torch.uint8)
    carried = torch.rand((B, N), generator=g) < 0.2
    second[carried] = 0x80
    second = second.cuda()
    want = torch.where(carried.cuda(), first | 0x80, second)
    ops.fused_set_actions(e.plan, e.fs, second)
    assert torch.equal(second, want) and torch.equal(e.fs.sel8, want.t())
    assert int(carried.sum()) > 0 or B * N < 20
    # the reference's view of it: SELECTED_ROAD = target of the out-edge with that rank
    e._x_stale = True
    sel = e.x[:, :, 3 * net.Nmax + 5]
    src_sorted = torch.argsort(net.edge_index[0], stable=True)
    dst_by_rank = net.edge_index[1][src_sorted].view(N, 4).cuda()           # a torus: four out-edges per road
    assert torch.equal(sel, torch.gather(dst_by_rank.unsqueeze(0).expand(B, N, 4), 2,
                                         (want & 0x7F).long().unsqueeze(2)).squeeze(2).float())


def test_bf16x3_forward_on_cancellation_heavy_inputs():
    """k_edge_mlp_fwd_x3 takes its bf16 pieces by TRUNCATION and drops the mid*lo, lo*mid and lo*lo products: per multiply-add
    up to ~2^-20 |w| |x| of one-signed error (csrc/edge_mlp.hip), i.e. the distance from the exact head is bounded by the
    magnitude that PROPAGATES through the three layers, S(e) = |w3| . (|W2| . (|W1| |x(e)| + |b1|) + |b2|) + |b3|, not by the
    logit. Two regimes, both against an fp64 evaluation of the head:
      * the reference's own shape of inputs — U(-0.1, 0.1) weights of mixed sign on raw features with clock-time columns
        around 2e4 (src/agents/mpnn_agent.py:35-41, :166-178): the north star's contract, 1e-4 of the logits' scale;
      * weights built to cancel — +w on the source's clock column, -w on the target's, so that terms of ~2e3 sum to ~1 —:
        4e-6 of S(e) (= 4 * 2^-20), and no more than that; the fp32 MFMA kernel (exact products, fp32 accumulation) is held
        to 2^-19 of S(e) on the same inputs for comparison."""
    from oracle import nets
    from tarl_hip import ops, synth
    net = synth.torus_network(6, 6)
    N, E = net.num_roads, net.edge_index.size(1)
    plan = ops.Plan(net.edge_index, N)
    ec = ops.EdgeConst(net.edge_attr, "cuda")
    g = torch.Generator().manual_seed(17)
    M = 9
    x = torch.zeros((M, N, 16))
    x[..., 0] = 14.0
    x[..., 1] = torch.randint(0, 14, (M, N), generator=g).float()
    x[..., 2] = 10.0
    x[..., 3] = 100.0
    x[..., 5] = torch.randint(0, N, (M, N), generator=g).float()
    x[..., 6] = torch.arange(N).float()
    x[..., 7] = torch.randint(0, N, (M, N), generator=g).float()          # agent origin / destination
    x[..., 8] = torch.randint(0, N, (M, N), generator=g).float()
    x[..., 9] = 21540.0 + torch.randint(0, 3600, (M, N), generator=g).float()      # departure / arrival clock times
    x[..., 10] = x[..., 9] + torch.rand((M, N), generator=g) * 200.0

    def run(ws):
        w = ops.EdgeMlpWeights(*(t.cuda().contiguous() for t in ws))
        ws64 = [t.double() for t in ws]
        ea = net.edge_attr.double().reshape(1, E, 1).expand(M, E, 1)
        ref = nets.edge_mlp_logits(x.double(), net.edge_index, ea, *ws64)
        a = [t.abs() for t in ws64]
        S = nets.edge_mlp_logits(x.double().abs(), net.edge_index, ea.abs(), *a)        # all terms positive: ReLU is the identity
        l3 = ops.policy_edge_mlp(plan, x.cuda(), ec, w, precision="x3").cpu().double()
        l32 = ops.policy_edge_mlp(plan, x.cuda(), ec, w).cpu().double()
        return ref, S, (l3 - ref).abs(), (l32 - ref).abs()

    u = lambda *s: (torch.rand(s, generator=g) - 0.5) * 0.2
    ref, S, e3, e32 = run([u(64, 33), u(64), u(32, 64), u(32), u(1, 32), u(1)])
    scale = float(ref.abs().max())
    assert scale > 1.0 and float(e3.max()) <= TOL * scale, (float(e3.max()), scale)
    assert float((e3 / S).max()) <= 4e-6 and float((e32 / S).max()) <= 2.0 ** -19
    w1 = u(64, 33)
    w1[:, 16 + 9] = -w1[:, 9]                  # the target row's departure clock against the source row's
    w1[:, 16 + 10] = -w1[:, 10]
    ref, S, e3, e32 = run([w1, u(64), u(32, 64), u(32), u(1, 32), u(1)])
    assert float(S.max()) > 20 * float(ref.abs().max())               # the terms do cancel
    assert float((e3 / S).max()) <= 4e-6, float((e3 / S).max())
    assert float((e32 / S).max()) <= 2.0 ** -19, float((e32 / S).max())
