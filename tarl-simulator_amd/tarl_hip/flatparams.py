"""FlatParams — all trainable parameters of the actor and the critic in ONE contiguous fp32 buffer (plus one gradient
buffer and the two Adam moment buffers), with the nn.Module parameters re-pointed at views of it. One fused Adam launch
and one gradient all-reduce per optimiser step (SURVEY §5 'distributed communication backend')."""
from __future__ import annotations

import torch

from . import dist_utils, ops


class FlatParams:
    def __init__(self, params, device=None):
        params = [p for p in params]
        assert params, "no parameters"
        device = device or params[0].device
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.flat = torch.empty(self.numel, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step = 0
        self.offsets = {}
        off = 0
        for p in params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
            self.offsets[id(p)] = (off, n)
            off += n

    def grad_view(self, p):
        off, n = self.offsets[id(p)]
        return self.grad[off:off + n].view(p.shape)

    def zero_grad(self):
        self.grad.zero_()

    def allreduce_grads(self):
        """Sum over ranks; the 1/world_size factor is applied where the gradient seeds are formed."""
        dist_utils.allreduce_sum_(self.grad)

    def adam_step(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
        self.step += 1
        ops.adam_step_(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.step, lr=lr, beta1=beta1, beta2=beta2,
                       eps=eps, grad_scale=grad_scale)
