"""GPU: MPNNValueNet (the reference's message-passing critic, src/agents/mpnn_agent.py:265-402; evaluation-mode dropout)
on the HIP kernels tarl_value_mpnn_{fwd,bwd}: forward against the golden produced by the reference's own class with
the same weights (unbatched and batched), forward + parameter gradients against a plain-torch fp32 restatement on the
CPU. Floating-point kernel: tolerance 1e-5 relative (summation order differs from a BLAS GEMV)."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def _torch_reference(sd, edge_index, agent_features, nf, ef, ai, tm):
    from oracle import nets
    return nets.value_mpnn(sd, edge_index, agent_features, nf, ef, ai, tm)


def _module(g):
    from src.agents.mpnn_agent import MPNNValueNet
    N = g["node_features"].size(0)
    net = MPNNValueNet(g["edge_index"].cuda(), N, device="cuda")
    sd = {k.replace("__", "."): v for k, v in g.items() if "__" in k and k.split("__")[0] in
          ("message_mlp", "node_mlp", "final_mlp", "time_net")}
    assert sorted(sd) == sorted(net.state_dict())                                          # same module tree / keys
    net.load_state_dict({k: v.cuda() for k, v in sd.items()})
    net.agent_features = g["agent_features"].cuda()
    return net, sd


def test_forward_matches_reference_golden():
    assert torch.cuda.is_available()
    g = load_golden("value_mpnn")
    net, _ = _module(g)
    with pytest.raises(RuntimeError):
        net(g["node_features"].cuda(), g["edge_attr"].cuda(), g["agent_index"].cuda(), g["time"].cuda())   # train mode
    net.eval()
    v = net(g["node_features"].cuda(), g["edge_attr"].cuda(), g["agent_index"].cuda(), g["time"].cuda())
    assert v.shape == g["value"].shape
    assert torch.allclose(v.cpu(), g["value"], rtol=RTOL, atol=1e-6)
    vb = net(g["node_features_b"].cuda(), g["edge_attr_b"].cuda(), g["agent_index_b"].cuda(), g["time_b"].cuda())
    assert vb.shape == g["value_b"].shape
    assert torch.allclose(vb.cpu(), g["value_b"], rtol=RTOL, atol=1e-6)


def test_forward_and_gradients_match_torch():
    g = load_golden("value_mpnn")
    net, sd = _module(g)
    net.eval()
    # the simulator's raw features (times ~2e4, road indices) saturate both tanh layers (the golden test above covers
    # them); for informative gradients use unit-scale inputs
    gen = torch.Generator().manual_seed(8)
    M, N, E = 3, g["node_features"].size(0), g["edge_index"].size(1)
    nf = torch.randn((M, N, 7), generator=gen) * 0.5
    ef = torch.rand((M, E), generator=gen)
    ai = g["agent_index_b"]
    agent_features = torch.randn(g["agent_features"].shape, generator=gen) * 0.5
    net.agent_features = agent_features.cuda()
    tm = torch.tensor([0.3, -1.2, 2.0])
    w = torch.tensor([0.7, -1.3, 0.4])
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    v_ref = _torch_reference(ref_sd, g["edge_index"], agent_features, nf, ef, ai, tm)
    (v_ref * w).sum().backward()
    v = net(nf.cuda(), ef.unsqueeze(-1).cuda(), ai.cuda(), tm.view(-1, 1).cuda()).view(-1)
    assert torch.allclose(v.cpu(), v_ref.detach(), rtol=RTOL, atol=1e-6)
    (v * w.cuda()).sum().backward()
    for name, p in net.named_parameters():
        gr, want = p.grad.cpu(), ref_sd[name].grad
        scale = max(1e-6, float(want.abs().max()))
        assert float((gr - want).abs().max()) <= 2e-5 * scale, name
        assert float(want.abs().max()) > 0, name
