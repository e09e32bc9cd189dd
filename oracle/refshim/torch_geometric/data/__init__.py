"""Stand-in for torch_geometric.data.Data: an attribute bag with .to() and .clone()."""
import copy
import torch


class Data:
    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def __getattr__(self, name):
        # torch_geometric infers num_nodes from x (or the largest edge index) when it was not given explicitly
        if name == "num_nodes":
            x = self.__dict__.get("x")
            if x is not None:
                return x.size(0)
            ei = self.__dict__.get("edge_index")
            return int(ei.max()) + 1 if ei is not None and ei.numel() else 0
        raise AttributeError(name)

    def keys(self):
        return [k for k in self.__dict__ if not k.startswith("_")]

    def to(self, device):
        for k in self.keys():
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self

    def clone(self):
        out = Data()
        for k in self.keys():
            v = getattr(self, k)
            setattr(out, k, v.clone() if torch.is_tensor(v) else copy.deepcopy(v))
        return out
