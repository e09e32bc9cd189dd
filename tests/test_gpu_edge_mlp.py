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
