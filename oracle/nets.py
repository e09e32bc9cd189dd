"""CPU restatement of the live policy / critic forward (TEST INFRASTRUCTURE — see oracle/__init__.py)."""
from __future__ import annotations

import torch
import torch.nn.functional as Fn


def policy_logits(node_features, edge_index, emb_weight):
    """MPNNPolicyNet.forward live path (src/agents/mpnn_agent.py:175-178,215-217):
    ``logits[e] = W_emb[int(node_features[dst(e), ROAD_INDEX])]``; batched input (B,N,7) -> (B,E).
    Everything at ``:181-190`` is computed and discarded by the reference."""
    road = node_features[..., 6].to(torch.long)          # ObservationFeatureHelpers.ROAD_INDEX = 6
    emb = emb_weight.reshape(-1)[road]                     # (..., N)
    return emb[..., edge_index[1]]


def edge_mlp_logits(x16, edge_index, edge_attr, w0, b0, w2, b2, w4, b4):
    """Dormant ``edge_mlp`` 33->64->32->1 on cat(x[src], x[dst], edge_attr) (src/agents/mpnn_agent.py:35-41,227-231)."""
    e = torch.cat([x16[..., edge_index[0], :], x16[..., edge_index[1], :], edge_attr], dim=-1)
    hdn = Fn.relu(Fn.linear(e, w0, b0))
    hdn = Fn.relu(Fn.linear(hdn, w2, b2))
    return Fn.linear(hdn, w4, b4).squeeze(-1)


def critic_value(node_features, time, w0, b0, w2, b2, w4, b4):
    """MPNNValueNetSimple.forward (src/agents/mpnn_agent.py:428-450): MLP(cat(NUMBER_OF_AGENT per node, time))."""
    xin = torch.cat((node_features[..., 1], time), dim=-1)
    hdn = Fn.relu(Fn.linear(xin, w0, b0))
    hdn = Fn.relu(Fn.linear(hdn, w2, b2))
    return Fn.linear(hdn, w4, b4)


def value_mpnn(sd, edge_index, agent_features, nf, ef, ai, tm):
    """MPNNValueNet.forward in evaluation mode (src/agents/mpnn_agent.py:300-402): ``sd`` = its state dict;
    nf (M,N,7), ef (M,E), ai (M,N) agent ids, tm (M,) -> (M,). Message of edge (u -> v) from x_v (flow
    target_to_source), mean at u over its out-edges in edge order, node tanh, time MLP, final Linear(N+1, 1)."""
    M, N = nf.shape[:2]
    x = torch.cat([nf, agent_features[ai.long()]], dim=-1)
    src, dst = edge_index
    msg_in = torch.cat([x[:, dst], ef.unsqueeze(-1)], dim=-1)
    m = torch.tanh(msg_in @ sd["message_mlp.1.weight"].t() + sd["message_mlp.1.bias"]).squeeze(-1)
    summ = torch.zeros((M, N)).index_add_(1, src, m)
    deg = torch.zeros(N).index_add_(0, src, torch.ones(src.numel()))
    a = torch.where(deg > 0, summ / deg.clamp(min=1), torch.zeros_like(summ))
    nd = torch.tanh(a * sd["node_mlp.0.weight"].view(()) + sd["node_mlp.0.bias"].view(()))
    h = torch.relu(tm.view(M, 1) @ sd["time_net.0.weight"].t() + sd["time_net.0.bias"])
    h = torch.relu(h @ sd["time_net.3.weight"].t() + sd["time_net.3.bias"])
    te = h @ sd["time_net.6.weight"].t() + sd["time_net.6.bias"]
    return (torch.cat([nd, te], dim=1) @ sd["final_mlp.0.weight"].t() + sd["final_mlp.0.bias"]).view(M)
