"""ResponseMPNN — second round: every road whose offered agent now sits at the tail of a downstream road pops its head
(reference: src/response_mpnn.py). Same class surface; the arithmetic is ``tarl_response_step``. ``update_history`` is
kept on the device and only filtered (one host sync) when somebody reads it."""
from __future__ import annotations

import torch

from ._compat import MessagePassingBase, cached_plan, require_cuda
from .feature_helpers import FeatureHelpers


class ResponseMPNN(MessagePassingBase):
    def __init__(self, Nmax: int = 100, time: int = 0):
        MessagePassingBase.__init__(self, aggr="max", flow="target_to_source")
        FeatureHelpers.__init__(self, Nmax=Nmax)   # same quirk as the reference: constants without inheriting (Q1)
        self.time = time
        self._pending = []     # (time, mask uint8 (R,), flag int32[1]) not yet filtered
        self._history = []

    def set_time(self, time: int):
        self.time = time

    @property
    def update_history(self):
        """list[(time, BoolTensor(R))] — appended only for steps with >= 1 pop (src/response_mpnn.py:106,125)."""
        if self._pending:
            flags = torch.stack([f for _, _, f in self._pending]).view(-1).cpu()
            for (t, m, _), keep in zip(self._pending, flags.tolist()):
                if keep:
                    self._history.append((t, m.bool()))
            self._pending = []
        return self._history

    @update_history.setter
    def update_history(self, value):
        self._pending = []
        self._history = list(value)

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor, edge_attr: torch.Tensor = None) -> torch.Tensor:
        from tarl_hip import ops
        require_cuda(x, "x")
        plan = cached_plan(edge_index, x.size(0))
        flag = torch.empty(1, dtype=torch.int32, device=x.device)
        popped = ops.response_step(plan, x, self.Nmax, any_flag=flag)
        self._pending.append((self.time, popped.view(-1), flag))
        return x
