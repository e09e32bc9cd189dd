"""Stand-in for torch_geometric.data.Data: an attribute bag with .to() and .clone()."""
import copy
import torch


class Data:
    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

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
