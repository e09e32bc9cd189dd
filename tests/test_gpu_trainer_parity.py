"""GPU: one full PPO update of VecPPOTrainer (critic over all frames, GAE, global advantage normalisation, minibatch,
clipped loss, backward through GraphDistribution / policy / critic, Adam) against the oracle evaluated with torch
autograd on the CPU over the SAME rollout and the SAME minibatch frames. Losses, gradients and updated parameters
within 1e-4 (relative to the tensor's scale) — the north-star tolerance for "MPNN logits and PPO losses"."""
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4


def close(a, b, what):
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= TOL * scale, f"{what}: max abs err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("fused,lazy,rollout,timestep", [(True, False, "frames", 1), (True, True, "frames", 1),
                                                         (True, False, "env", 1), (True, True, "env", 1),
                                                         (False, False, None, 1),
                                                         # 300-s steps: the episode ends at frame 13 of 24 -> reset inside
                                                         # the batch, done / terminated masks in GAE
                                                         (True, False, "frames", 300), (True, False, "env", 300),
                                                         (False, False, None, 300)])
def test_ppo_update_matches_oracle_autograd(fused, lazy, rollout, timestep):
    assert torch.cuda.is_available()
    from oracle import dist, nets, ppo
    from src.agents.mpnn_agent import MPNNPolicyNet, MPNNValueNetSimple
    from tarl_hip import synth
    from tarl_hip.engine import SimEngine
    from tarl_hip.trainer import VecPPOTrainer

    net = synth.torus_network(4, 4, heterogeneous=True, seed=2)
    N, E = net.num_roads, net.edge_index.size(1)
    B, A, T, M = 128 if fused else 5, 300, 24, 16
    pops = torch.stack([synth.population(A, N, seed=b, t0=21540, t1=21555) for b in range(B)])
    eng = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                    pops.cuda(), congestion_constant=net.congestion_constant, seed=3, fused=fused, timestep=timestep)
    torch.manual_seed(0)
    pol = MPNNPolicyNet(net.edge_index, N, None, device="cuda")
    val = MPNNValueNetSimple(net.edge_index, N, device="cuda")
    l = val.final_mlp
    crit = [l[0].weight, l[0].bias, l[2].weight, l[2].bias, l[4].weight, l[4].bias]
    tr = VecPPOTrainer(eng, pol.nodes_embedding.weight, crit, rollout_steps=T, num_epochs=1, sub_batch_size=M,
                       extra_params=[p for n, p in pol.named_parameters() if not n.startswith("nodes_embedding")],
                       lazy_log_prob=lazy, rollout=rollout)
    assert tr.rollout == (rollout or "unfused")
    tr.keep_grad = True
    tr.collect()
    done_t = tr.done_frames
    assert done_t.tolist() == [timestep == 300 and t == 12 for t in range(T)]
    if timestep == 300:     # the frame after the episode end observes the reset state at the reset clock
        assert float(tr.times[13]) == 21540.0 and float(tr.times[12]) == 21540.0 + 12 * 300
        assert int(tr.counts[13].sum()) == 0
    # ---- snapshot everything the update reads (CPU copies, reference (frame, env, node) order) ----
    emb0 = pol.nodes_embedding.weight.detach().cpu().clone()
    crit0 = [p.detach().cpu().clone() for p in crit]
    if fused:   # the rollout's byte buffers (count, rank of the chosen out-edge) in the reference's terms
        choice, counts = (t_.cpu() for t_ in eng.decode_rollout(tr.env_minor, choice=tr.choice, counts=tr.counts))
    else:
        counts, choice = tr.counts.cpu(), tr.choice.cpu()                                    # (T+1, B, N), (T, B, N) edge ids
    reward, times = tr.reward.cpu(), tr.times.cpu()
    assert float(reward.abs().sum()) > 0
    idx = torch.randperm(T * B, generator=torch.Generator().manual_seed(4))[:M]
    # ---- GPU update ----
    adv_g, tgt_g = tr.advantages()
    out = tr.minibatch_step(adv_g, tgt_g, idx=idx)
    # ---- oracle update (torch autograd on the CPU) ----
    emb = emb0.clone().requires_grad_(True)
    cw = [p.clone().requires_grad_(True) for p in crit0]
    nf_all = torch.zeros((T + 1, B, N, 7))
    nf_all[..., 1] = counts
    nf_all[..., 6] = torch.arange(N, dtype=torch.float32)
    with torch.no_grad():
        v_all = nets.critic_value(nf_all, times.view(T + 1, 1, 1).expand(T + 1, B, 1), *cw).squeeze(-1)   # (T+1, B)
        dmask = done_t.view(T, 1).expand(T, B)
        adv, tgt = ppo.gae(reward, v_all[:T], v_all[1:], dmask, dmask, average_gae=True)
    close(adv_g.cpu(), adv, "advantage")
    close(tgt_g.cpu(), tgt, "value_target")
    t_idx, b_idx = idx // B, idx % B
    onehot = torch.zeros((M, E), dtype=torch.int64)
    onehot.scatter_(1, choice[t_idx, b_idx].long(), 1)
    nf_mb = nf_all[t_idx, b_idx]                                                             # (M, N, 7)
    with torch.no_grad():
        lp_old = dist.GraphDist(nets.policy_logits(nf_mb, net.edge_index, emb0), net.edge_index).log_prob(onehot)
    if not lazy and fused:   # the rollout's stored behaviour log-prob (fixed-point accumulation) == the exact one
        close(tr.logp.view(-1).cpu()[idx], lp_old, "sample_log_prob")
    d = dist.GraphDist(nets.policy_logits(nf_mb, net.edge_index, emb), net.edge_index)
    lp_new, ent = d.log_prob(onehot), d.entropy()
    value = nets.critic_value(nf_mb, times[t_idx].view(M, 1), *cw).squeeze(-1)
    losses = ppo.clip_ppo_loss(lp_new, lp_old, adv.view(-1)[idx], value, tgt.view(-1)[idx], ent)
    (losses["loss_objective"] + losses["loss_critic"] + losses["loss_entropy"]).backward()
    o = out.cpu()
    for i, k in enumerate(["loss_objective", "loss_critic", "loss_entropy"]):
        assert abs(o[i].item() - losses[k].item()) <= TOL * max(1.0, abs(losses[k].item())), k
    # gradients (flat buffer order: embedding, critic, dormant heads)
    g = tr.last_grad.cpu()
    off = 0
    for name, ref in [("emb", emb.grad)] + [(f"critic{i}", c.grad) for i, c in enumerate(cw)]:
        n = ref.numel()
        close(g[off:off + n], ref.reshape(-1), f"grad {name}")
        off += n
    assert float(g[off:].abs().sum()) == 0.0                       # dormant heads receive no gradient
    # one Adam step
    for p_gpu, p0, gr, name in [(pol.nodes_embedding.weight, emb0, emb.grad, "emb")] + \
            [(crit[i], crit0[i], cw[i].grad, f"critic{i}") for i in range(6)]:
        q = p0.clone()
        ppo.adam_step(q, gr, torch.zeros_like(q), torch.zeros_like(q), 1)
        close(p_gpu.detach().cpu(), q, f"param {name}")
    assert not torch.equal(pol.nodes_embedding.weight.detach().cpu(), emb0)


def test_consecutive_collects_with_an_episode_end_keep_every_log_prob_exact():
    """An episode end splits a collector batch into rollouts of different lengths (here 13 + 11 frames) that share one
    choice scratch buffer. The log-prob accumulators of the long call must not be overlaid by the record tables of the
    short one (round-2 advisor finding: offsets inside the scratch depended on T). Two consecutive collect() calls; the
    stored sample_log_prob of EVERY frame of BOTH batches must equal the exact log-prob of the stored action."""
    from oracle import dist, nets
    from src.agents.mpnn_agent import MPNNPolicyNet, MPNNValueNetSimple
    from tarl_hip import synth
    from tarl_hip.engine import SimEngine
    from tarl_hip.trainer import VecPPOTrainer

    net = synth.torus_network(4, 4, heterogeneous=True, seed=2)
    N, E = net.num_roads, net.edge_index.size(1)
    B, A, T = 128, 300, 24
    pops = torch.stack([synth.population(A, N, seed=b, t0=21540, t1=21555) for b in range(B)])
    eng = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                    pops.cuda(), congestion_constant=net.congestion_constant, seed=3, fused=True, timestep=300)
    torch.manual_seed(0)
    pol = MPNNPolicyNet(net.edge_index, N, None, device="cuda")
    val = MPNNValueNetSimple(net.edge_index, N, device="cuda")
    l = val.final_mlp
    tr = VecPPOTrainer(eng, pol.nodes_embedding.weight, [l[0].weight, l[0].bias, l[2].weight, l[2].bias, l[4].weight, l[4].bias],
                       rollout_steps=T, num_epochs=1, sub_batch_size=16,
                       extra_params=[p for n, p in pol.named_parameters() if not n.startswith("nodes_embedding")],
                       rollout="frames")
    emb0 = pol.nodes_embedding.weight.detach().cpu().clone()
    nf = torch.zeros((1, N, 7))
    nf[..., 6] = torch.arange(N, dtype=torch.float32)
    d = dist.GraphDist(nets.policy_logits(nf, net.edge_index, emb0), net.edge_index)
    for it in range(3):
        tr.collect()
        torch.cuda.synchronize()
        assert tr.done_frames.tolist() == [t == 12 for t in range(T)]
        choice = eng.decode_rollout(True, choice=tr.choice)[0].cpu()          # (T, B, N) edge ids
        onehot = torch.zeros((T * B, E), dtype=torch.int64)
        onehot.scatter_(1, choice.view(T * B, N).long(), 1)
        exact = d.log_prob(onehot).view(T, B)
        got = tr.logp.cpu()
        err = (got - exact).abs().max(dim=1).values
        assert float(err.max()) <= TOL * max(1.0, float(exact.abs().max())), \
            f"collect {it}: sample_log_prob off by up to {float(err.max()):.3e} (frames {err.nonzero().flatten().tolist()[:8]})"
    tr.check_flags()
