"""Stand-in for the torch_geometric.utils helpers the reference imports (graph conversion only)."""
import torch


def to_networkx(data, edge_attrs=None, to_undirected=False):
    import networkx as nx
    g = nx.Graph() if to_undirected else nx.DiGraph()
    g.add_nodes_from(range(int(data.num_nodes)))
    ei = data.edge_index.cpu()
    for e in range(ei.size(1)):
        attrs = {}
        for name in (edge_attrs or []):
            v = getattr(data, name)[e]
            attrs[name] = v.item() if v.numel() == 1 else v.tolist()
        g.add_edge(int(ei[0, e]), int(ei[1, e]), **attrs)
    return g


def degree(index, num_nodes=None, dtype=None):
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros(n, dtype=dtype or torch.float32, device=index.device)
    return out.scatter_add_(0, index, torch.ones_like(index, dtype=out.dtype))


def to_scipy_sparse_matrix(edge_index, edge_attr=None, num_nodes=None):
    import scipy.sparse as sp
    ei = edge_index.cpu().numpy()
    vals = edge_attr.cpu().numpy().reshape(-1) if edge_attr is not None else [1.0] * ei.shape[1]
    return sp.coo_matrix((vals, (ei[0], ei[1])), shape=(num_nodes, num_nodes))
