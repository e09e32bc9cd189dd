"""MPNNPolicyNet / MPNNValueNetSimple — the learned actor and critic (reference: src/agents/mpnn_agent.py).

Same constructors, ``forward`` signatures and state-dict keys (``nodes_embedding.weight``, ``edge_mlp.{0,2,4}.*``,
``edge_mlp_test.{0,2}.*``, ``final_mlp.{0,2,4}.*``). Live actor: logits[e] = W_emb[ROAD_INDEX(dst(e))]
(``tarl_policy_edge_logits_fwd/bwd``); live critic: MLP(cat(NUMBER_OF_AGENT per node, time)) on fp32 MFMA
(``tarl_critic_mlp_fwd/bwd``). What the reference computes and discards after the logits (Dijkstra prior, travel
time, norm — :181-190) is not evaluated; the all-pairs Dijkstra matrix of its constructor is built lazily on request.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .._compat import MessagePassingBase, cached_plan, require_cuda
from ..feature_helpers import ObservationFeatureHelpers
from .base import Agents


class _EdgeLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb_weight, node_features, plan):
        from tarl_hip import ops
        ctx.plan, ctx.nf, ctx.shape = plan, node_features, emb_weight.shape
        return ops.policy_edge_logits(plan, node_features, emb_weight.detach().reshape(-1).contiguous())

    @staticmethod
    def backward(ctx, grad_logits):
        from tarl_hip import ops
        g = ops.policy_edge_logits_bwd(ctx.plan, ctx.nf, grad_logits.contiguous(), int(torch.Size(ctx.shape).numel()))
        return g.view(ctx.shape), None, None


class _EdgeMlp(torch.autograd.Function):
    """The per-edge MLP head on the HIP kernels (tarl_policy_edge_mlp_fwd / _bwd); gradients for its six parameters."""

    @staticmethod
    def forward(ctx, obs16, plan, ec, bf16, w1, b1, w2, b2, w3, b3):
        from tarl_hip import ops
        w = ops.EdgeMlpWeights(w1, b1, w2, b2, w3, b3)
        ctx.saved = (obs16, plan, ec, w, [t.shape for t in (w1, b1, w2, b2, w3, b3)])
        return ops.policy_edge_mlp(plan, obs16, ec, w, bf16=bf16)

    @staticmethod
    def backward(ctx, grad_logits):
        from tarl_hip import ops
        obs16, plan, ec, w, shapes = ctx.saved
        grads = [torch.zeros_like(t) for t in (w.w1, w.b1, w.w2, w.b2, w.w3, w.b3)]
        ops.policy_edge_mlp_bwd(plan, obs16, ec, w, grad_logits.contiguous(), grads)
        return (None, None, None, None) + tuple(g.view(sh) for g, sh in zip(grads, shapes))


class MPNNPolicyNet(MessagePassingBase, Agents):
    h = ObservationFeatureHelpers()
    # Which head produces the logits. "embedding": the reference's live forward (logit = nodes_embedding of the target
    # road, src/agents/mpnn_agent.py:215-217). "edge_mlp" / "edge_mlp_fp32" / "edge_mlp_bf16": the per-edge MLP the reference
    # keeps as parameters and spells out in its commented lines (:227-231), on fp32 / bf16 MFMA — a state-dependent policy
    # (this module's forward runs the fp32 MFMA kernel for the first two; the names differ in the ROLLOUT kernel the trainer
    # picks: fp32 accuracy on the bf16 pipe / exact fp32 products / bf16).
    policy_head = "embedding"

    def __init__(self, edge_index, num_nodes, free_flow_time_travel, device):
        Agents.__init__(self, device=device)
        MessagePassingBase.__init__(self, aggr="mean", flow="target_to_source")
        self.edge_index = edge_index
        self.num_nodes = num_nodes
        self.num_edges = edge_index.size(1)
        self.dim_node_features = 16
        self.dim_edge_features = 1
        self._free_flow = free_flow_time_travel
        self._dist_matrix = None
        self.nodes_embedding = nn.Embedding(num_nodes, 1)
        self.edge_mlp_test = nn.Sequential(nn.Linear(2 * self.dim_node_features, 16), nn.ReLU(), nn.Linear(16, 1))
        self.edge_mlp = nn.Sequential(nn.Linear(2 * self.dim_node_features + self.dim_edge_features, 64), nn.ReLU(),
                                      nn.Linear(64, 32), nn.ReLU(), nn.Linear(32, 1))
        for seq in (self.edge_mlp, self.edge_mlp_test):       # dormant heads: U(-0.1, 0.1) weights, zero bias
            for m in seq:
                if isinstance(m, nn.Linear):
                    nn.init.uniform_(m.weight, -0.1, 0.1)
                    nn.init.constant_(m.bias, 0)
        self.to(device)

    @property
    def dist_matrix(self):
        """All-pairs free-flow shortest-path matrix, built on first access (the reference builds it eagerly with
        networkx in the constructor although the live forward never reads it)."""
        if self._dist_matrix is None:
            self.refresh_dijkstra(self.edge_index, self._free_flow)
        return self._dist_matrix

    def refresh_dijkstra(self, edge_index: torch.Tensor, free_flow_travel: torch.Tensor):
        """All-pairs free-flow distances (src/agents/mpnn_agent.py:53-79) by ``tarl_apsp`` on the device."""
        from tarl_hip import ops
        assert free_flow_travel.size(0) == edge_index.size(1) and edge_index.size(0) == 2
        plan = cached_plan(edge_index, self.num_nodes)
        w = free_flow_travel.detach().to(self.device, torch.float32)
        require_cuda(w, "free_flow_travel")
        self._dist_matrix = ops.all_pairs_shortest_paths(plan, w, want_next_hop=False, want_dist=True)[1][0]

    def compute_dijkstra_logits(self, agent_destination: torch.Tensor, time_travel: torch.Tensor) -> torch.Tensor:
        """Shortest-path prior (src/agents/mpnn_agent.py:81-113; dormant in the live forward):
        logits[e] = -dist[dst(e), destination[e]] - time_travel[e]; a destination vector of k * E entries is batched."""
        E = self.edge_index.size(1)
        rep = agent_destination.size(0) // E
        head = self.edge_index[1].to(self.dist_matrix.device).repeat(rep)
        logits = -self.dist_matrix[head, agent_destination.to(head.device)] - time_travel
        return logits.view(rep, -1) if rep > 1 else logits.view(-1)

    def forward(self, node_features: torch.Tensor, edge_features: torch.Tensor, agent_index: torch.Tensor):
        """node_features (N,7) or (B,N,7) -> logits (E,) or (B,E)."""
        require_cuda(node_features, "node_features")
        plan = cached_plan(self.edge_index, self.num_nodes)
        if self.policy_head != "embedding":
            from tarl_hip import ops
            from .._compat import cached_edge_const
            ea = edge_features if edge_features.dim() <= 2 else edge_features[0]     # static: the same for every sample
            ec = cached_edge_const(ea.reshape(-1, 1), node_features.device)
            obs16 = ops.policy_obs16(node_features, agent_index, self.agent_features.to(node_features.device))
            m = self.edge_mlp
            logits = _EdgeMlp.apply(obs16, plan, ec, self.policy_head == "edge_mlp_bf16", m[0].weight, m[0].bias,
                                    m[2].weight, m[2].bias, m[4].weight, m[4].bias)
            return logits if node_features.dim() == 3 else logits.view(-1)
        return _EdgeLogits.apply(self.nodes_embedding.weight, node_features, plan)

    def update_edges(self, x, edge_index, edge_attr=None):
        plan = cached_plan(edge_index, x.size(0))
        return _EdgeLogits.apply(self.nodes_embedding.weight, x[:, :7], plan).view(-1, 1)


class _CriticMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, counts, time_rows, w1, b1, w2, b2, w3, b3):
        from tarl_hip import ops
        cw = ops.CriticWeights(w1.detach().contiguous(), b1.detach().contiguous(), w2.detach().contiguous(),
                               b2.detach().contiguous(), w3.detach().reshape(-1).contiguous(), b3.detach().contiguous())
        value, h1, h2 = ops.critic_forward(cw, counts, time_rows, 1, keep_hidden=True)
        ctx.cw, ctx.saved = cw, (counts, time_rows, h1, h2)
        ctx.w3_shape = w3.shape
        return value

    @staticmethod
    def backward(ctx, grad_value):
        from tarl_hip import ops
        counts, time_rows, h1, h2 = ctx.saved
        cw = ctx.cw
        grads = [torch.zeros_like(t) for t in (cw.w1, cw.b1, cw.w2, cw.b2, cw.w3, cw.b3)]
        ops.critic_backward(cw, counts, time_rows, 1, h1, h2, grad_value.contiguous(), grads)
        return None, None, grads[0], grads[1], grads[2], grads[3], grads[4].view(ctx.w3_shape), grads[5]


class _ValueMPNN(torch.autograd.Function):
    """MPNNValueNet forward / backward on the HIP kernels (tarl_value_mpnn_{fwd,bwd}); gradients for the 12 parameters."""

    @staticmethod
    def forward(ctx, plan, nf, ar, ef, tm, *params):
        from tarl_hip import ops
        value, act, agg = ops.value_mpnn_forward(plan, nf, ar, ef, tm, params, keep=True)
        ctx.plan, ctx.saved, ctx.params = plan, (nf, ar, ef, tm, act, agg), params
        return value

    @staticmethod
    def backward(ctx, grad_value):
        from tarl_hip import ops
        nf, ar, ef, tm, act, agg = ctx.saved
        grads = ops.value_mpnn_backward(ctx.plan, nf, ar, ef, tm, ctx.params, grad_value.contiguous(), act, agg)
        return (None, None, None, None, None) + tuple(g.view_as(p) for g, p in zip(grads, ctx.params))


class MPNNValueNet(MessagePassingBase, Agents):
    """The message-passing critic of the reference (src/agents/mpnn_agent.py:265-402; never instantiated by its runner):
    per-edge message tanh(Linear(17,1)), mean over a road's out-edges, tanh(Linear(1,1)), a 1-32-32-1 time MLP and a
    final Linear(N+1, 1). Same module tree / state-dict keys. The Dropout(0.05) layers are kept in the tree but the
    kernels implement evaluation-mode semantics (identity); ``forward`` refuses training-mode dropout explicitly."""

    def __init__(self, edge_index, num_nodes, device):
        Agents.__init__(self, device=device)
        MessagePassingBase.__init__(self, aggr="mean", flow="target_to_source")
        self.edge_index = edge_index
        self.num_nodes = num_nodes
        self.num_edges = edge_index.size(1)
        self.dim_nodes_features = 16
        self.dim_edges_features = 1
        self.message_mlp = nn.Sequential(nn.Dropout(0.05), nn.Linear(17, 1), nn.Tanh())
        self.node_mlp = nn.Sequential(nn.Linear(1, 1), nn.Tanh())
        self.final_mlp = nn.Sequential(nn.Linear(num_nodes + 1, 1))
        self.time_net = nn.Sequential(nn.Linear(1, 32), nn.Dropout(0.05), nn.ReLU(), nn.Linear(32, 32), nn.Dropout(0.05),
                                      nn.ReLU(), nn.Linear(32, 1))
        self.to(device)

    def _params(self):
        m, n, f, t = self.message_mlp[1], self.node_mlp[0], self.final_mlp[0], self.time_net
        return (m.weight, m.bias, n.weight, n.bias, f.weight, f.bias, t[0].weight, t[0].bias, t[3].weight, t[3].bias,
                t[6].weight, t[6].bias)

    def forward(self, node_features, edge_features, agent_index, time):
        """node_features (N,7) or (B,N,7); edge_features (E,1) or (B,E,1); agent_index (N,) or (B,N); time (1,) or
        (B,1) -> (1,) or (B,1)."""
        require_cuda(node_features, "node_features")
        if self.training:
            raise RuntimeError("MPNNValueNet runs with evaluation-mode dropout only: call .eval() (the kernels implement "
                               "Dropout as the identity)")
        batched = node_features.dim() == 3
        nf = node_features.reshape(-1, self.num_nodes, node_features.size(-1)).to(torch.float32).contiguous()
        M = nf.size(0)
        ar = None
        if self.agent_features is not None:
            ar = self.agent_features[agent_index.reshape(M, self.num_nodes).long()].to(torch.float32).contiguous()
        ef = edge_features.reshape(-1, self.num_edges).to(torch.float32).contiguous()
        tm = time.reshape(-1).to(torch.float32).contiguous()
        plan = cached_plan(self.edge_index, self.num_nodes)
        v = _ValueMPNN.apply(plan, nf, ar, ef, tm, *self._params())
        return v.view(M, 1) if batched else v.view(1)


class MPNNValueNetSimple(MessagePassingBase, Agents):
    """Critic actually used by the runner: ``final_mlp`` = Linear(N+1,64)-ReLU-Linear(64,64)-ReLU-Linear(64,1)."""

    def __init__(self, edge_index, num_nodes, device):
        Agents.__init__(self, device=device)
        MessagePassingBase.__init__(self, aggr="mean", flow="target_to_source")
        self.edge_index = edge_index
        self.num_nodes = num_nodes
        self.num_edges = edge_index.size(1)
        self.dim_nodes_features = 16
        self.dim_edges_features = 1
        self.final_mlp = nn.Sequential(nn.Linear(num_nodes + 1, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU(),
                                       nn.Linear(64, 1))
        self.to(device)

    def forward(self, node_features, edge_features, agent_index, time):
        """node_features (...,N,7), time (...,1) -> (...,1)."""
        require_cuda(node_features, "node_features")
        lead = node_features.shape[:-2]
        counts = node_features[..., 1].reshape(-1, self.num_nodes).contiguous()
        time_rows = time.reshape(-1).to(torch.float32).contiguous()
        l0, l2, l4 = self.final_mlp[0], self.final_mlp[2], self.final_mlp[4]
        v = _CriticMLP.apply(counts, time_rows, l0.weight, l0.bias, l2.weight, l2.bias, l4.weight, l4.bias)
        return v.view(*lead, 1)
