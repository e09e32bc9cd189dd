"""Stand-in for torch_geometric.nn: only MessagePassing.propagate as the reference uses it.

Restated semantics (torch-geometric 2.5.0, not verifiable offline):
* flow='source_to_target': (i, j) = (1, 0); 'target_to_source': (i, j) = (0, 1).
* for each parameter of ``message`` named ``<k>_i`` / ``<k>_j``: ``kwargs[k].index_select(node_dim, edge_index[i|j])``;
  other parameters are passed through from kwargs when present.
* ``aggregate(inputs, index=edge_index[i], ptr=None, dim_size=N)``; the default implementation is the
  ``aggr`` reduction ('add' | 'mean' | 'max') into zeros, groups without inputs stay 0.
* ``update(aggr_out, **kwargs named in its signature)``.
"""
import inspect
import torch


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2, **kwargs):
        super().__init__()
        self.aggr = aggr
        self.flow = flow
        self.node_dim = node_dim
        assert flow in ("source_to_target", "target_to_source")

    @staticmethod
    def _named(fn, skip):
        return [p for p in inspect.signature(fn).parameters if p not in skip]

    def propagate(self, edge_index, size=None, **kwargs):
        i, j = (1, 0) if self.flow == "source_to_target" else (0, 1)
        num_nodes = None
        for v in kwargs.values():
            if torch.is_tensor(v) and v.dim() >= 2:
                num_nodes = v.size(self.node_dim)
                break
        if "x" in kwargs and torch.is_tensor(kwargs["x"]):
            num_nodes = kwargs["x"].size(self.node_dim)
        msg_kwargs = {}
        for name in self._named(self.message, ()):
            if name.endswith("_i") or name.endswith("_j"):
                data = kwargs.get(name[:-2])
                if torch.is_tensor(data):
                    dim = i if name.endswith("_i") else j
                    data = data.index_select(self.node_dim, edge_index[dim])
                msg_kwargs[name] = data
            elif name in kwargs:
                msg_kwargs[name] = kwargs[name]
        out = self.message(**msg_kwargs)
        aggr_all = {"index": edge_index[i], "ptr": None, "dim_size": num_nodes}
        aggr_all.update(kwargs)
        aggr_kwargs = {k: aggr_all[k] for k in self._named(self.aggregate, ("inputs",)) if k in aggr_all}
        out = self.aggregate(out, **aggr_kwargs)
        upd_kwargs = {k: kwargs[k] for k in self._named(self.update, ("aggr_out", "inputs")) if k in kwargs}
        return self.update(out, **upd_kwargs)

    def message(self, x_j):
        return x_j

    def aggregate(self, inputs, index, ptr=None, dim_size=None):
        shape = list(inputs.shape)
        shape[0] = dim_size
        out = inputs.new_zeros(shape)
        idx = index.view(-1, *([1] * (inputs.dim() - 1))).expand_as(inputs)
        if self.aggr in ("add", "sum"):
            return out.scatter_add_(0, idx, inputs)
        if self.aggr == "mean":
            return out.scatter_reduce_(0, idx, inputs, reduce="mean", include_self=False)
        if self.aggr == "max":
            return out.scatter_reduce_(0, idx, inputs, reduce="amax", include_self=False)
        raise NotImplementedError(self.aggr)

    def update(self, aggr_out):
        return aggr_out
