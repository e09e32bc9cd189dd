"""Minimal actor / critic wrappers with the call conventions of the torchrl / tensordict classes the reference's runner
uses (``TensorDictModule``, ``ProbabilisticActor``, ``ValueOperator``; src/runner.py:83-105). torchrl is not required."""
from __future__ import annotations

import torch
import torch.nn as nn


class TensorDictModule(nn.Module):
    def __init__(self, module, in_keys, out_keys):
        super().__init__()
        self.module, self.in_keys, self.out_keys = module, list(in_keys), list(out_keys)

    def forward(self, td):
        out = self.module(*[td[k] for k in self.in_keys])
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for k, v in zip(self.out_keys, outs):
            td[k] = v
        return td


class ProbabilisticActor(nn.Module):
    """``module`` writes the distribution parameters; an action is drawn from ``distribution_class`` (``mode`` when
    ``deterministic`` is set — torchrl's ExplorationType.MODE) and, optionally, its log-probability is recorded."""

    def __init__(self, module, spec=None, distribution_class=None, in_keys=("logits",), distribution_kwargs=None,
                 return_log_prob=False):
        super().__init__()
        self.module, self.spec, self.distribution_class = module, spec, distribution_class
        self.in_keys, self.distribution_kwargs = list(in_keys), dict(distribution_kwargs or {})
        self.return_log_prob = return_log_prob
        self.deterministic = False

    def get_dist(self, td):
        td = self.module(td)
        return self.distribution_class(*[td[k] for k in self.in_keys], **self.distribution_kwargs)

    def forward(self, td):
        dist = self.get_dist(td)
        action = dist.mode.to(torch.int64) if self.deterministic else dist.sample()
        td["action"] = action
        if self.return_log_prob:
            td["sample_log_prob"] = dist.log_prob(action)
        return td


class ValueOperator(TensorDictModule):
    def __init__(self, module, in_keys, out_keys=("state_value",)):
        inner = module.module if isinstance(module, TensorDictModule) else module
        super().__init__(inner, in_keys, out_keys)


def unwrap(module, cls_name):
    """Find the first sub-module whose class is called ``cls_name`` (works for these wrappers and for torchrl's)."""
    for m in module.modules():
        if type(m).__name__ == cls_name:
            return m
    raise TypeError(f"no {cls_name} inside {type(module).__name__}")
