"""Stand-in for torch-scatter==2.1.2 on CPU (see oracle/refshim/README.md). Test infrastructure only.

Restated semantics (1-D `src`, or reduction over the last dimension with a 1-D `index`):
* scatter_add : sequential fp32 accumulation in element order into zeros.
* scatter_max : running maximum with a strict ``>`` (first maximum wins); groups that receive nothing
  return value 0 and ``arg = src.size(dim)``.
* scatter_softmax : ``exp(src - max[index]) / sum[index]`` per group, no epsilon.
"""
import torch


def _dim_size(index, dim_size):
    if dim_size is not None:
        return int(dim_size)
    return int(index.max()) + 1 if index.numel() else 0


def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    n = _dim_size(index, dim_size)
    res = src.new_zeros(src.shape[:-1] + (n,))
    idx = index.expand_as(src) if index.dim() == src.dim() else index.view((1,) * (src.dim() - 1) + (-1,)).expand_as(src)
    return res.scatter_add_(-1, idx, src)


scatter_sum = scatter_add


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    n = _dim_size(index, dim_size)
    length = src.size(-1)
    flat = src.reshape(-1, length)
    vals = flat.new_full((flat.size(0), n), float("-inf"))
    args = torch.full((flat.size(0), n), length, dtype=torch.long)
    idx = index.tolist()
    for r in range(flat.size(0)):
        row = flat[r].tolist()
        best = [None] * n
        barg = [length] * n
        for e, (g, v) in enumerate(zip(idx, row)):
            if best[g] is None or v > best[g]:
                best[g] = v
                barg[g] = e
        # re-read the winning values from the tensor so fp32 bits are preserved
        for g in range(n):
            if barg[g] < length:
                vals[r, g] = flat[r, barg[g]]
            else:
                vals[r, g] = 0.0
            args[r, g] = barg[g]
    shape = src.shape[:-1] + (n,)
    return vals.reshape(shape), args.reshape(shape)


def scatter_softmax(src, index, dim=-1, dim_size=None):
    n = _dim_size(index, dim_size)
    idx = index.view((1,) * (src.dim() - 1) + (-1,)).expand_as(src)
    mx = src.new_full(src.shape[:-1] + (n,), float("-inf")).scatter_reduce_(-1, idx, src, reduce="amax", include_self=True)
    rec = src - mx.gather(-1, idx)
    ex = rec.exp()
    sm = scatter_add(ex, index, dim_size=n)
    return ex.div(sm.gather(-1, idx))
