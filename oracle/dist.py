"""CPU restatement of GraphDistribution (TEST INFRASTRUCTURE — see oracle/__init__.py).

Follows ``src/reinforcement_learning.py:15-96``: one categorical per source node over its out-edges.
Defined on the parity domain: every source id in ``0..nb_nodes-1`` has >= 1 out-edge.
"""
from __future__ import annotations

import torch


def segment_softmax(logits: torch.Tensor, index: torch.Tensor, n: int) -> torch.Tensor:
    """torch-scatter 2.1.2 ``scatter_softmax`` over the last dim: exp(l - max[g]) / sum[g], no epsilon; the group sum
    is a sequential fp32 accumulation in edge order."""
    idx = index.view((1,) * (logits.dim() - 1) + (-1,)).expand_as(logits)
    mx = logits.new_full(logits.shape[:-1] + (n,), float("-inf")).scatter_reduce_(-1, idx, logits, reduce="amax")
    ex = (logits - mx.gather(-1, idx)).exp()
    sm = logits.new_zeros(logits.shape[:-1] + (n,)).scatter_add_(-1, idx, ex)
    return ex / sm.gather(-1, idx)


class GraphDist:
    """``GraphDistribution.__init__`` (``:17-55``)."""

    def __init__(self, logits: torch.Tensor, edge_index: torch.Tensor, temperature: float = 1.0, proba=None):
        """``proba`` (test hook): use these probabilities instead of the softmax of ``logits`` so that the integer
        part (cumsum / sample) can be checked on bit-identical fp inputs."""
        src = edge_index[0]
        self.edge_index = edge_index
        self.groups, self.index = torch.sort(src, stable=True)  # reference: unstable call, stable on CPU in practice
        self.inv_index = torch.argsort(self.index)
        self.nodes = torch.unique(self.groups)
        self.nb_nodes = self.nodes.numel()
        n_all = int(src.max()) + 1
        self.proba = segment_softmax(logits / temperature, src, n_all) if proba is None else proba
        self.proba_sort = self.proba[..., self.index]
        self.log_proba_sort = torch.log(self.proba_sort + 1e-8)
        g = self.groups
        self.last = torch.ones_like(g, dtype=torch.bool)
        self.last[:-1] = g[1:] != g[:-1]
        # global prefix sum (CPU cumsum accumulates fp32 inputs in double, rounds each output to fp32),
        # then rebased per group in fp32 (``:38-42``)
        cs = torch.cumsum(self.proba_sort, dim=-1)
        bsum = torch.zeros(logits.shape[:-1] + (self.nb_nodes,))
        bsum[..., 1:] = cs[..., self.last][..., :-1]
        self.cumsum = cs - bsum[..., g]

    @property
    def mode(self):
        """``:45-55`` unbatched: one-hot of the per-node argmax, first maximum wins."""
        assert self.proba.dim() == 1
        from .sim import segment_argmax_first
        arg = segment_argmax_first(self.proba, self.edge_index[0], int(self.edge_index[0].max()) + 1)
        out = torch.zeros_like(self.proba)
        out[arg] = 1
        return out

    def sample(self, uniform: torch.Tensor | None = None) -> torch.Tensor:
        """``:62-80`` (unbatched). ``uniform`` (nb_nodes,) or drawn from the global generator like the reference."""
        if uniform is None:
            uniform = torch.rand(self.nb_nodes)
        s = uniform[..., self.groups]
        r = torch.where(s < self.cumsum, 1, 0)
        r = torch.cumsum(r, dim=-1)
        rb = torch.zeros_like(self.nodes)
        rb[1:] = r[..., self.last][:-1]
        r = r - rb[self.groups]
        hot = torch.where(r == 1, 1, 0)
        return hot[..., self.inv_index]

    def log_prob(self, action: torch.Tensor) -> torch.Tensor:
        """``:82-93``; works batched (B,E)->(B,)."""
        a = action[..., self.index]
        cs = torch.cumsum(a, dim=-1)
        possible = torch.all(cs[..., self.last] == torch.arange(1, self.nb_nodes + 1), dim=-1)
        lp = torch.sum(a * self.log_proba_sort, dim=-1)
        lp = torch.where(possible, lp, torch.full_like(lp, float("-inf")))
        return lp

    def entropy(self) -> torch.Tensor:
        """``:95-96``."""
        return -torch.sum(self.proba_sort * self.log_proba_sort, dim=-1).flatten()
