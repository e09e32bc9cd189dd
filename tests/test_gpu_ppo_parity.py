"""GPU parity (through the C ABI) for the critic MLP (MFMA), GAE, the clipped PPO loss and Adam: HIP kernels vs the CPU
oracle / torch autograd of the oracle expressions. Floating point: tolerance 1e-4 (north_star), stated per assert."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tarl_hip import ops as _ops
    return _ops


def dev(t):
    return t.cuda()


def test_critic_golden(ops):
    """MPNNValueNetSimple forward with the reference module's own weights and inputs (tests/golden/nets.npz)."""
    g = load_golden("nets")
    w = [g[f"val__final_mlp__{i}__{p}"] for i in (0, 2, 4) for p in ("weight", "bias")]
    cw = ops.CriticWeights(*(dev(t.contiguous()) for t in (w[0], w[1], w[2], w[3], w[4].reshape(-1), w[5])))
    counts = dev(g["node_features"][:, 1].reshape(1, -1).contiguous())
    v, _, _ = ops.critic_forward(cw, counts, dev(g["time"]))
    ref = g["value"].item()
    assert abs(v.item() - ref) <= TOL * max(1.0, abs(ref))
    cb = dev(g["node_features_b"][:, :, 1].contiguous())
    vb, _, _ = ops.critic_forward(cw, cb, dev(g["time_b"].reshape(-1).contiguous()))
    assert torch.allclose(vb.cpu(), g["value_b"].view(-1), rtol=TOL, atol=TOL)


# (the last two: the chunked backward for many rows, csrc/critic.hip CB_MANY_ROWS = 512 — whole and ragged chunks)
@pytest.mark.parametrize("M,N,rpt", [(1, 24, 1), (37, 256, 1), (300, 2500, 4), (129, 31, 1), (1024, 333, 8), (777, 70, 1)])
def test_critic_fwd_bwd_vs_autograd(ops, M, N, rpt):
    from oracle import nets
    gen = torch.Generator().manual_seed(M * 7 + N)
    counts = torch.randint(0, 14, (M, N), generator=gen).float()
    times = torch.rand((M + rpt - 1) // rpt, generator=gen) * 10 + 21540.0 / 1000.0
    lin = [torch.nn.Linear(N + 1, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, 1)]
    params = [p for l in lin for p in (l.weight, l.bias)]
    trow = times.repeat_interleave(rpt)[:M].unsqueeze(1)
    nf = torch.zeros(M, N, 7)
    nf[:, :, 1] = counts
    ref = nets.critic_value(nf, trow, *params).view(-1)
    gv = torch.randn(M, generator=gen)
    (ref * gv).sum().backward()
    cw = ops.CriticWeights(*(dev(p.detach().reshape(-1) if i == 4 else p.detach()).contiguous() for i, p in enumerate(params)))
    big = torch.zeros((M, N + 12), device="cuda")          # row stride != N
    cview = big[:, :N]
    cview.copy_(counts)
    v, h1, h2 = ops.critic_forward(cw, cview, dev(times), rpt, keep_hidden=True)
    scale = max(1.0, float(ref.abs().max()))
    assert float((v.cpu() - ref.detach()).abs().max()) <= TOL * scale
    # the split-K form of the same forward (few rows: the optimiser minibatch): same value and hidden activations up to
    # the order of the fp32 additions, and deterministic
    vk, h1k, h2k = ops.critic_forward(cw, cview, dev(times), rpt, keep_hidden=True, split_k=True)
    assert float((vk.cpu() - ref.detach()).abs().max()) <= TOL * scale
    assert torch.allclose(h1k, h1, rtol=1e-5, atol=1e-4 * scale) and torch.allclose(h2k, h2, rtol=1e-5, atol=1e-4 * scale)
    vk2, _, _ = ops.critic_forward(cw, cview, dev(times), rpt, split_k=True)
    assert torch.equal(vk, vk2)
    grads = [torch.zeros_like(dev(p.detach().reshape(-1) if i == 4 else p.detach())) for i, p in enumerate(params)]
    ops.critic_backward(cw, cview, dev(times), rpt, h1, h2, dev(gv), grads)
    for i, (gp, p) in enumerate(zip(grads, params)):
        r = p.grad.reshape(gp.shape)
        tol = TOL * max(1.0, float(r.abs().max()))
        assert float((gp.cpu() - r).abs().max()) <= tol, f"grad {i}"


def test_gae_and_normalisation_vs_oracle(ops):
    from oracle import ppo
    gen = torch.Generator().manual_seed(5)
    T, B = 64, 37
    r = torch.randn((T, B), generator=gen) * 5 - 20
    v = torch.randn((T + 1, B), generator=gen) * 3
    done = torch.rand((T, B), generator=gen) < 0.05
    term = done & (torch.rand((T, B), generator=gen) < 0.5)
    adv, tgt = ops.gae(dev(r), dev(v[:-1].contiguous()), dev(v[1:].contiguous()),
                       done=dev(done.to(torch.uint8)), terminated=dev(term.to(torch.uint8)))
    # oracle per environment (its normalisation is over the whole batch, applied afterwards)
    a_ref, t_ref = ppo.gae(r, v[:-1], v[1:], done, term, average_gae=False)
    assert torch.allclose(adv.cpu(), a_ref, rtol=1e-5, atol=1e-4) and torch.allclose(tgt.cpu(), t_ref, rtol=1e-5, atol=1e-4)
    stats = ops.advantage_stats(adv)
    ops.advantage_normalize_(adv, stats)
    an, _ = ppo.gae(r, v[:-1], v[1:], done, term, average_gae=True)
    assert torch.allclose(adv.cpu(), an, rtol=1e-4, atol=1e-4)
    assert abs(adv.mean().item()) < 1e-5 and abs(adv.std().item() - 1) < 1e-4


def test_ppo_loss_values_and_grads_vs_autograd(ops):
    from oracle import ppo
    gen = torch.Generator().manual_seed(9)
    M = 32
    lp_old = -torch.rand(M, generator=gen) * 50
    lp_new = (lp_old + torch.randn(M, generator=gen) * 0.3).requires_grad_(True)   # ratios on both sides of the clip
    adv = torch.randn(M, generator=gen)
    value = (torch.randn(M, generator=gen) * 2).requires_grad_(True)
    target = torch.randn(M, generator=gen) * 2
    ent = (torch.rand(M, generator=gen) * 100).requires_grad_(True)
    ref = ppo.clip_ppo_loss(lp_new, lp_old, adv, value, target, ent)
    (ref["loss_objective"] + ref["loss_critic"] + ref["loss_entropy"]).backward()
    out, g_lp, g_ent, g_val = ops.ppo_loss(dev(lp_new.detach()), dev(lp_old), dev(adv), dev(value.detach()),
                                           dev(target), dev(ent.detach()))
    o = out.cpu()
    for i, k in enumerate(["loss_objective", "loss_critic", "loss_entropy"]):
        assert abs(o[i].item() - ref[k].item()) <= TOL * max(1.0, abs(ref[k].item())), k
    assert torch.allclose(g_lp.cpu(), lp_new.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(g_val.cpu(), value.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(g_ent.cpu(), ent.grad, rtol=1e-4, atol=1e-7)
    lw = (lp_new.detach() - lp_old)
    import math
    outside = ((lw < math.log1p(-0.2)) | (lw > math.log1p(0.2))).float().mean().item()
    assert abs(o[3].item() - outside) < 1e-6 and 0.0 < outside < 1.0


def test_adam_matches_torch_optim(ops):
    torch.manual_seed(0)
    p = torch.randn(100003, requires_grad=True)
    q = dev(p.detach().clone())
    m, v = torch.zeros_like(q), torch.zeros_like(q)
    opt = torch.optim.Adam([p], lr=1e-3)
    for step in range(1, 8):
        grad = torch.randn(100003) * (10.0 if step % 2 else 0.01)
        p.grad = grad.clone()
        opt.step()
        ops.adam_step_(q, dev(grad), m, v, step)
        assert torch.allclose(q.cpu(), p.detach(), rtol=1e-6, atol=1e-7), f"step {step}"


@pytest.mark.parametrize("B", [8, 300, 1000])
def test_policy_logits_backward_many_rows_vs_autograd(ops, B):
    """tarl_policy_edge_logits_bwd: the reference's live head (logit = embedding of the target road,
    src/agents/mpnn_agent.py:215-217) differentiated over B batch rows that share ONE observation (the optimiser minibatch:
    a broadcast view, stride 0) — from 256 rows on through the row-chunked kernels — against torch autograd; twice the same bits."""
    from tarl_hip import synth
    net = synth.torus_network(5, 4, heterogeneous=True, seed=2)
    N, E = net.num_roads, net.edge_index.size(1)
    plan = ops.Plan(net.edge_index, N)
    gen = torch.Generator().manual_seed(B)
    emb = torch.randn(N, generator=gen, requires_grad=True)
    g = torch.randn((B, E), generator=gen)
    nf = net.x[:, 3 * net.Nmax:].cuda()
    nfb = nf.unsqueeze(0).expand(B, N, 7)
    assert nfb.stride(0) == 0
    logits = ops.policy_edge_logits(plan, nfb, emb.detach().cuda())
    ref = emb[net.x[:, 3 * net.Nmax + 6].long()][net.edge_index[1]].unsqueeze(0).expand(B, E)
    assert torch.equal(logits.cpu(), ref.detach())
    (ref * g).sum().backward()
    ge = ops.policy_edge_logits_bwd(plan, nfb, g.cuda(), N)
    scale = max(1.0, float(emb.grad.abs().max()))
    assert float((ge.cpu() - emb.grad).abs().max()) <= 1e-5 * scale
    assert torch.equal(ge, ops.policy_edge_logits_bwd(plan, nfb, g.cuda(), N))
    # the same rows as separate observations (batch stride != 0): the per-row kernel, same gradient
    gs = ops.policy_edge_logits_bwd(plan, nfb.contiguous(), g.cuda(), N)
    assert float((gs - ge).abs().max()) <= 1e-5 * scale
