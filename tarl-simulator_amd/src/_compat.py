"""Optional third-party bases. torch-geometric / torchrl / tensordict are NOT needed (and absent on the GPU box); when
they are importable the mirror classes inherit from them so they still slot into the reference's runner and tests
(e.g. ``issubclass(DirectionMPNN, MessagePassing)``, tests/direction_mpnn_test.py:7-8)."""
from __future__ import annotations

import copy

import torch

try:  # pragma: no cover - depends on the environment
    from torch_geometric.nn import MessagePassing as _MP
    HAVE_PYG = True
except Exception:  # noqa: BLE001
    _MP = None
    HAVE_PYG = False

if HAVE_PYG:
    class MessagePassingBase(_MP):
        def __init__(self, **kw):
            super().__init__(**kw)
else:
    class MessagePassingBase(torch.nn.Module):
        def __init__(self, aggr="add", flow="source_to_target", **kw):
            super().__init__()
            self.aggr, self.flow = aggr, flow

try:  # pragma: no cover
    from torch_geometric.data import Data
except Exception:  # noqa: BLE001
    class Data:
        """Attribute bag with ``.to()`` / ``.clone()`` (what the path uses of torch_geometric.data.Data)."""

        def __init__(self, **kw):
            self.__dict__.update(kw)

        def keys(self):
            return [k for k in self.__dict__ if not k.startswith("_")]

        def to(self, device):
            for k in self.keys():
                v = getattr(self, k)
                if torch.is_tensor(v):
                    setattr(self, k, v.to(device))
            return self

        def clone(self):
            return Data(**{k: (v.clone() if torch.is_tensor(v) else copy.deepcopy(v)) for k, v in self.__dict__.items()})

try:  # pragma: no cover
    from tensordict import TensorDict, TensorDictBase
except Exception:  # noqa: BLE001
    class TensorDictBase(dict):
        pass

    class TensorDict(TensorDictBase):
        """Minimal stand-in: a dict with ``batch_size`` and nested ("next", key) access."""

        def __init__(self, source=None, batch_size=None, **kw):
            super().__init__(source or {})
            self.batch_size = list(batch_size) if batch_size is not None else []

        def __getitem__(self, key):
            if isinstance(key, tuple):
                cur = self
                for k in key:
                    cur = dict.__getitem__(cur, k)
                return cur
            return dict.__getitem__(self, key)

        def set(self, key, value):
            self[key] = value
            return self

        def to(self, device):
            return TensorDict({k: (v.to(device) if hasattr(v, "to") else v) for k, v in self.items()}, self.batch_size)

try:  # pragma: no cover
    from torchrl.envs import EnvBase
except Exception:  # noqa: BLE001
    class EnvBase:
        def __init__(self, device="cpu", **kw):
            self.device = torch.device(device)

        def to(self, device):
            self.device = torch.device(device)
            return self


class Spec:
    """Shape/dtype record standing in for torchrl's tensor specs (only ``shape`` / ``dtype`` are consulted)."""

    def __init__(self, shape, dtype, device=None, low=None, high=None):
        self.shape, self.dtype, self.device, self.low, self.high = torch.Size(shape), dtype, device, low, high


_PLAN_CACHE: dict = {}


def cached_plan(edge_index: torch.Tensor, num_nodes: int):
    """One static plan per (edge_index storage, version, node count): built once instead of per step
    (the reference re-sorts the static topology every step, src/reinforcement_learning.py:21-35)."""
    from tarl_hip import ops
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), str(edge_index.device),
           -1 if num_nodes is None else int(num_nodes))
    hit = _PLAN_CACHE.get(key)
    if hit is None:
        if len(_PLAN_CACHE) > 64:
            _PLAN_CACHE.clear()
        if num_nodes is None:                                  # only on a cache miss: one host read per topology
            num_nodes = int(edge_index.max()) + 1 if edge_index.numel() else 0
        hit = (ops.Plan(edge_index, num_nodes), edge_index)   # keep the tensor alive so data_ptr stays unique
        _PLAN_CACHE[key] = hit
    return hit[0]


_ROUTING_CACHE: dict = {}


def cached_routing_plan(edge_index: torch.Tensor, num_nodes: int, num_roads: int):
    """Plan over the edges whose destination is a ROAD (road -> road and SRC -> road): the candidates of the classical
    random agent (src/agents/base.py:447-495 samples inside ``adj[:, :num_roads]``; road -> DEST edges never compete)."""
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), str(edge_index.device), int(num_nodes),
           int(num_roads))
    hit = _ROUTING_CACHE.get(key)
    if hit is None:
        if len(_ROUTING_CACHE) > 64:
            _ROUTING_CACHE.clear()
        sub = edge_index[:, edge_index[1] < num_roads].contiguous()
        hit = (cached_plan(sub, num_nodes), sub, edge_index)
        _ROUTING_CACHE[key] = hit
    return hit[0]


_EC_CACHE: dict = {}


def cached_edge_const(edge_attr: torch.Tensor, device):
    from tarl_hip import ops
    key = (edge_attr.data_ptr(), edge_attr._version, tuple(edge_attr.shape), str(edge_attr.device), str(device))
    hit = _EC_CACHE.get(key)
    if hit is None:
        if len(_EC_CACHE) > 64:
            _EC_CACHE.clear()
        hit = (ops.EdgeConst(edge_attr, device), edge_attr)
        _EC_CACHE[key] = hit
    return hit[0]


def require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        from tarl_hip.lib import TarlError
        raise TarlError(f"{what} lives on {t.device}: this build runs the path on the GPU only "
                        "(hand-written HIP kernels, no CPU fallback); move the graph to 'cuda'")
