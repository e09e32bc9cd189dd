"""Thin, validated wrappers over the C ABI (include/tarl_hip.h). Tensors in, tensors out; no compute happens in Python
and there is no fallback: every function enqueues hand-written HIP kernels on torch's current stream.

Batched state convention: ``x`` is fp32 ``(R, F)`` or ``(B, R, F)`` (last dim contiguous, arbitrary row / env
strides), mutated in place like the reference does.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import torch

from . import lib as _lib

EPS_GUMBEL = 1e-12   # src/direction_mpnn.py:136
# revision of the fused path's packed HBM layout / kernel set: a PMC traffic record (profiles/*_pmc_traffic.json) only
# applies to the revision it was measured on
FUSED_LAYOUT = "v11"


def frame_kernel_source_hash() -> str:
    """sha256 (first 16 hex digits) of the frame kernels' sources, csrc/fused.hip + csrc/fused_common.h: a PMC traffic record
    (tools/pmc_bench.py -> profiles/*.json) is only paired with kernel times measured on the very code it was taken on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
    for name in ("fused.hip", "fused_common.h"):
        with open(os.path.join(csrc, name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _check_dev(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise _lib.TarlError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")


def _contig(t: torch.Tensor, dtype, name: str):
    _check_dev(t, dtype, name)
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def _state(x: torch.Tensor, Nmax: int):
    """-> (B, rows, x_bstride, ldx)."""
    _check_dev(x, torch.float32, "x")
    if x.dim() not in (2, 3) or x.stride(-1) != 1 or x.size(-1) < 3 * Nmax + 7:
        raise ValueError(f"x must be (R,F) or (B,R,F) with F >= 3*Nmax+7 and a contiguous last dim, got {tuple(x.shape)}")
    if x.dim() == 2:
        return 1, x.size(0), x.size(0) * x.stride(0), x.stride(0)
    return x.size(0), x.size(1), x.stride(0), x.stride(1)


def _agents(a: torch.Tensor, B: int):
    """-> (A, a_bstride)."""
    _check_dev(a, torch.float32, "agent_features")
    if a.size(-1) != 9 or a.stride(-1) != 1 or a.stride(-2) != 9:
        raise ValueError("agent_features must be (A,9) or (B,A,9) with contiguous rows")
    if a.dim() == 2:
        if B != 1:
            raise ValueError("batched state needs batched agent_features (B,A,9)")
        return a.size(0), a.size(0) * 9
    if a.size(0) != B:
        raise ValueError("agent_features batch does not match x")
    return a.size(1), a.stride(0)


class Plan:
    """Static per-graph plan (``tarl_plan_create``): int32 CSC/CSR built once on the host and kept on the device."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, src_order: torch.Tensor | None = None):
        L = _lib.load()
        ei = edge_index.detach().to("cpu", torch.int64).contiguous()
        if ei.dim() != 2 or ei.size(0) != 2:
            raise ValueError("edge_index must be (2, E)")
        so = None if src_order is None else src_order.detach().to("cpu", torch.int64).contiguous()
        handle = C.c_void_p()
        _lib.check(L.tarl_plan_create(ei.data_ptr(), ei.size(1), int(num_nodes), _lib.ptr(so), C.byref(handle)))
        self._h = handle
        self._finalizer = weakref.finalize(self, L.tarl_plan_destroy, handle)
        info = (C.c_int64 * 6)()
        _lib.check(L.tarl_plan_info(handle, info))
        self.num_nodes, self.num_edges, self.num_groups, self.max_in, self.max_out, src_sorted = [int(v) for v in info]
        self.src_sorted = bool(src_sorted)
        geo = (C.c_int64 * 3)()
        _lib.check(L.tarl_plan_geometry(handle, geo))
        self.siblings4, self.row_siblings, self.num_row_chunks = bool(geo[0]), bool(geo[1]), int(geo[2])
        self.device = torch.device("cuda", torch.cuda.current_device())

    @property
    def handle(self):
        return self._h


_LOG_EPS = None


def log_eps() -> float:
    """fp32 ``log(0 + 1e-12)`` evaluated by torch on the CPU, as the reference does (src/direction_mpnn.py:138)."""
    global _LOG_EPS
    if _LOG_EPS is None:
        _LOG_EPS = float(torch.log(torch.zeros(1, dtype=torch.float32) + EPS_GUMBEL)[0])
    return _LOG_EPS


class EdgeConst:
    """Per-graph edge constants on the device: ``edge_attr`` (E,) and ``log(edge_attr + 1e-12)`` evaluated on the CPU
    with torch so that the Gumbel-max scores are bit-identical to the reference's."""

    def __init__(self, edge_attr: torch.Tensor, device):
        ea = edge_attr.detach().to("cpu", torch.float32).reshape(-1).contiguous()
        self.edge_attr = ea.to(device)
        self.log_edge_attr = torch.log(ea + EPS_GUMBEL).to(device)
        self.log_eps = log_eps()


def gumbel_from_uniform_cpu(u: torch.Tensor) -> torch.Tensor:
    """``-log(-log(u))`` evaluated on the CPU (parity runs feed the reference's own noise, src/direction_mpnn.py:137)."""
    u = u.detach().to("cpu", torch.float32)
    return -torch.log(-torch.log(u))


def direction_step(plan: Plan, x, Nmax, ec: EdgeConst, t, *, congestion_constant=None, gumbel=None, seed=0, counter=0,
                   want_dtt=True, chosen=None, status=None):
    """DirectionMPNN.forward on B environments. Returns (delta_travel_time (B,E) or None, chosen (B,R))."""
    L = _lib.load()
    B, R, bs, ldx = _state(x, Nmax)
    E = plan.num_edges
    if chosen is None:
        chosen = torch.empty((B, R), dtype=torch.float32, device=x.device)
    dtt = torch.empty((B, E), dtype=torch.float32, device=x.device) if want_dtt else None
    if gumbel is not None:
        _contig(gumbel, torch.float32, "gumbel")
        if gumbel.numel() != B * E:
            raise ValueError("gumbel must hold B*E values")
    if congestion_constant is not None:
        _contig(congestion_constant, torch.float32, "congestion_constant")
        if congestion_constant.numel() < R:
            raise ValueError("congestion_constant shorter than num_roads")
    _lib.check(L.tarl_direction_step(plan.handle, x.data_ptr(), B, bs, ldx, Nmax, R, ec.edge_attr.data_ptr(),
                                     ec.log_edge_attr.data_ptr(), ec.log_eps, _lib.ptr(congestion_constant), float(t),
                                     _lib.ptr(gumbel), int(seed), int(counter), _lib.ptr(dtt), chosen.data_ptr(),
                                     _lib.ptr(status), _lib.current_stream()))
    return dtt, chosen


def response_step(plan: Plan, x, Nmax, *, popped=None, any_flag=None):
    """ResponseMPNN.forward on B environments. Returns popped (B,R) uint8; ``any_flag`` int32[1] set on device."""
    L = _lib.load()
    B, R, bs, ldx = _state(x, Nmax)
    if popped is None:
        popped = torch.empty((B, R), dtype=torch.uint8, device=x.device)
    _lib.check(L.tarl_response_step(plan.handle, x.data_ptr(), B, bs, ldx, Nmax, R, popped.data_ptr(),
                                    _lib.ptr(any_flag), _lib.current_stream()))
    return popped


def core_step(plan: Plan, x, Nmax, ec: EdgeConst, t, *, congestion_constant=None, gumbel=None, seed=0, counter=0,
              want_dtt=True, chosen=None, popped=None, any_flag=None, status=None):
    """SimulationCoreModel.forward (both rounds). Returns (dtt or None, popped)."""
    L = _lib.load()
    B, R, bs, ldx = _state(x, Nmax)
    E = plan.num_edges
    if chosen is None:
        chosen = torch.empty((B, R), dtype=torch.float32, device=x.device)
    if popped is None:
        popped = torch.empty((B, R), dtype=torch.uint8, device=x.device)
    dtt = torch.empty((B, E), dtype=torch.float32, device=x.device) if want_dtt else None
    if gumbel is not None:
        _contig(gumbel, torch.float32, "gumbel")
    _lib.check(L.tarl_core_step(plan.handle, x.data_ptr(), B, bs, ldx, Nmax, R, ec.edge_attr.data_ptr(),
                                ec.log_edge_attr.data_ptr(), ec.log_eps, _lib.ptr(congestion_constant), float(t),
                                _lib.ptr(gumbel), int(seed), int(counter), _lib.ptr(dtt), chosen.data_ptr(),
                                popped.data_ptr(), _lib.ptr(any_flag), _lib.ptr(status), _lib.current_stream()))
    return dtt, popped


def apply_action(plan: Plan, x, Nmax, *, action_onehot=None, choice=None):
    L = _lib.load()
    B, N, bs, ldx = _state(x, Nmax)
    if N != plan.num_nodes:
        raise ValueError("x rows must equal the plan's node count")
    if action_onehot is not None:
        _contig(action_onehot, torch.int64, "action")
    if choice is not None:
        _contig(choice, torch.int32, "choice")
    _lib.check(L.tarl_apply_action(plan.handle, x.data_ptr(), B, bs, ldx, Nmax, _lib.ptr(action_onehot),
                                   _lib.ptr(choice), _lib.current_stream()))


def withdraw_step(plan: Plan, x, Nmax, agent_features, t, *, withdrawn=None, want_mask=True):
    L = _lib.load()
    B, N, bs, ldx = _state(x, Nmax)
    A, abs_ = _agents(agent_features, B)
    if withdrawn is None and want_mask:
        withdrawn = torch.empty((B, N), dtype=torch.uint8, device=x.device)
    _lib.check(L.tarl_withdraw_step(plan.handle, x.data_ptr(), B, bs, ldx, Nmax, N, agent_features.data_ptr(), A, abs_,
                                    float(t), _lib.ptr(withdrawn), _lib.current_stream()))
    return withdrawn


def insert_step(x, Nmax, agent_features, t, *, congestion_constant=None, scratch=None, reward=None, counts=None):
    L = _lib.load()
    B, N, bs, ldx = _state(x, Nmax)
    A, abs_ = _agents(agent_features, B)
    if scratch is None:
        scratch = torch.empty((B, 2 * A), dtype=torch.int32, device=x.device)
    _lib.check(L.tarl_insert_step(x.data_ptr(), B, bs, ldx, Nmax, N, agent_features.data_ptr(), A, abs_,
                                  _lib.ptr(congestion_constant), float(t), scratch.data_ptr(), _lib.ptr(reward),
                                  _lib.ptr(counts), _lib.current_stream()))


# ---- shortest-path routing -------------------------------------------------------------------------------------------------
def edge_travel_time(plan: Plan, x, Nmax, congestion_constant):
    """(B, E) current travel time of every edge, original edge order (src/agents/base.py:541-550)."""
    L = _lib.load()
    B, N, bs, ldx = _state(x, Nmax)
    out = torch.empty((B, plan.num_edges), dtype=torch.float32, device=x.device)
    cc = congestion_constant.to(torch.float32).contiguous()
    _lib.check(L.tarl_edge_travel_time(plan.handle, x.data_ptr(), B, bs, ldx, Nmax, cc.data_ptr(), out.data_ptr(),
                                       _lib.current_stream()))
    return out


def all_pairs_shortest_paths(plan: Plan, weights, *, want_next_hop=True, want_dist=False):
    """``weights`` (E,) or (B, E) fp32 in original edge order -> (next_hop int64 (B, N, N) | None, dist fp32 (B, N, N) |
    None), networkx-compatible tie order (tarl_apsp)."""
    L = _lib.load()
    f64 = weights.dtype == torch.float64          # double weights stay double (tarl_apsp_f64)
    w = weights.contiguous() if f64 else weights.to(torch.float32).contiguous()
    w = w.view(1, -1) if w.dim() == 1 else w
    B, N = w.size(0), plan.num_nodes
    assert w.size(1) == plan.num_edges, "one weight per edge"
    nh = torch.empty((B, N, N), dtype=torch.int64, device=w.device) if want_next_hop else None
    d = torch.empty((B, N, N), dtype=torch.float32, device=w.device) if want_dist else None
    need = int(L.tarl_apsp_scratch_bytes(plan.handle, B))
    scratch = torch.empty(need, dtype=torch.uint8, device=w.device) if need > 0 else None
    fn = L.tarl_apsp_f64 if f64 else L.tarl_apsp
    _lib.check(fn(plan.handle, w.data_ptr(), B, plan.num_edges, _lib.ptr(scratch), need, _lib.ptr(nh), _lib.ptr(d),
                  _lib.current_stream()))
    return nh, d


def msa_assign(next_hop, od_origin, od_dest, od_volume, is_road, aux_flow):
    """All-or-nothing assignment: aux_flow (N,) float64 += volume of every OD pair on the road nodes of its path."""
    L = _lib.load()
    N = next_hop.size(-1)
    for t, dt, nm in ((next_hop, torch.int64, "next_hop"), (od_origin, torch.int64, "od_origin"),
                      (od_dest, torch.int64, "od_dest"), (od_volume, torch.float64, "od_volume"),
                      (is_road, torch.uint8, "is_road"), (aux_flow, torch.float64, "aux_flow")):
        _contig(t, dt, nm)
    _lib.check(L.tarl_msa_assign(next_hop.data_ptr(), N, od_origin.data_ptr(), od_dest.data_ptr(), od_volume.data_ptr(),
                                 od_origin.numel(), is_road.data_ptr(), aux_flow.data_ptr(), _lib.current_stream()))


def select_next_hop(x, Nmax, agent_features, next_hop):
    """x[b, i, SELECTED_ROAD] = next_hop[b, i, DESTINATION[head agent of i]] (src/agents/base.py:572-580)."""
    L = _lib.load()
    B, N, bs, ldx = _state(x, Nmax)
    A, abs_ = _agents(agent_features, B)
    nh = next_hop.view(-1, N, N)
    assert nh.dtype == torch.int64 and nh.is_contiguous() and nh.size(0) in (1, B)
    _lib.check(L.tarl_select_next_hop(x.data_ptr(), B, bs, ldx, Nmax, N, agent_features.data_ptr(), A, abs_,
                                      nh.data_ptr(), 0 if nh.size(0) == 1 else N * N, _lib.current_stream()))


def reset_state(x, Nmax, agent_features=None):
    L = _lib.load()
    B, N, bs, ldx = _state(x, Nmax)
    A, abs_ = (0, 0) if agent_features is None else _agents(agent_features, B)
    _lib.check(L.tarl_reset_state(x.data_ptr(), B, bs, ldx, Nmax, N, _lib.ptr(agent_features), A, abs_,
                                  _lib.current_stream()))


# ---- GraphDistribution ------------------------------------------------------------------------------------------------
def _rows(t: torch.Tensor, E: int, name: str):
    if t.size(-1) != E:
        raise ValueError(f"{name} last dim must be E={E}")
    return t.numel() // E


def graphdist_softmax(plan: Plan, logits, temperature=1.0):
    L = _lib.load()
    _contig(logits, torch.float32, "logits")
    B = _rows(logits, plan.num_edges, "logits")
    proba = torch.empty_like(logits)
    _lib.check(L.tarl_graphdist_softmax(plan.handle, logits.data_ptr(), B, float(temperature), proba.data_ptr(),
                                        _lib.current_stream()))
    return proba


def graphdist_sample(plan: Plan, proba, *, uniform=None, seed=0, counter=0, want_onehot=True, want_choice=False):
    L = _lib.load()
    _contig(proba, torch.float32, "proba")
    E, G, N = plan.num_edges, plan.num_groups, plan.num_nodes
    B = _rows(proba, E, "proba")
    if uniform is not None:
        _contig(uniform, torch.float32, "uniform")
        if uniform.numel() != B * G:
            raise ValueError("uniform must hold B*num_groups values")
    sums = torch.empty((B, G + 1), dtype=torch.float64, device=proba.device)
    onehot = torch.empty(proba.shape, dtype=torch.int64, device=proba.device) if want_onehot else None
    choice = torch.empty(proba.shape[:-1] + (N,), dtype=torch.int32, device=proba.device) if want_choice else None
    _lib.check(L.tarl_graphdist_sample(plan.handle, proba.data_ptr(), B, _lib.ptr(uniform), int(seed), int(counter),
                                       sums.data_ptr(), _lib.ptr(onehot), _lib.ptr(choice), _lib.current_stream()))
    return onehot, choice


def graphdist_rollout(plan: Plan, logits, temperature=1.0, *, uniform=None, seed=0, counter=0, choice=None, choice8=None,
                      sel8=None, log_prob=None, scratch=None):
    """sample() + log_prob() of GraphDistribution(logits / temperature) in one launch (bit-identical to softmax -> sample
    -> logprob). logits (B, E); outputs written in place where given: choice int32 (B, N), choice8 uint8 (B, N) rank
    bytes, sel8 uint8 (N, B) = the packed state's SELECTED_ROAD bytes; returns log_prob (B,)."""
    L = _lib.load()
    _contig(logits, torch.float32, "logits")
    E, G, N = plan.num_edges, plan.num_groups, plan.num_nodes
    B = _rows(logits, E, "logits")
    if uniform is not None:
        _contig(uniform, torch.float32, "uniform")
        if uniform.numel() != B * G:
            raise ValueError("uniform must hold B*num_groups values")
    for name, t, dt, n in (("choice", choice, torch.int32, B * N), ("choice8", choice8, torch.uint8, B * N),
                           ("sel8", sel8, torch.uint8, B * N)):
        if t is not None:
            _contig(t, dt, name)
            if t.numel() != n:
                raise ValueError(f"{name} must hold B * num_nodes values")
    need = int(L.tarl_graphdist_rollout_scratch_bytes(plan.handle, B))
    if scratch is None or scratch.numel() * scratch.element_size() < need:
        scratch = torch.empty((need + 7) // 8, dtype=torch.float64, device=logits.device)
    if log_prob is None:
        log_prob = torch.empty(B, dtype=torch.float32, device=logits.device)
    else:
        _contig(log_prob, torch.float32, "log_prob")
    _lib.check(L.tarl_graphdist_rollout(plan.handle, logits.data_ptr(), B, float(temperature), _lib.ptr(uniform), int(seed),
                                        int(counter), scratch.data_ptr(), _lib.ptr(choice), _lib.ptr(choice8),
                                        _lib.ptr(sel8), log_prob.data_ptr(), _lib.current_stream()))
    return log_prob


def graphdist_mode(plan: Plan, proba, *, want_choice=False):
    L = _lib.load()
    _contig(proba, torch.float32, "proba")
    B = _rows(proba, plan.num_edges, "proba")
    onehot = torch.zeros_like(proba)
    choice = (torch.empty(proba.shape[:-1] + (plan.num_nodes,), dtype=torch.int32, device=proba.device)
              if want_choice else None)
    _lib.check(L.tarl_graphdist_mode(plan.handle, proba.data_ptr(), B, onehot.data_ptr(), _lib.ptr(choice),
                                     _lib.current_stream()))
    return onehot, choice


def graphdist_logprob_entropy(plan: Plan, proba, *, action_onehot=None, choice=None, want_logprob=True,
                              want_entropy=True):
    L = _lib.load()
    _contig(proba, torch.float32, "proba")
    B = _rows(proba, plan.num_edges, "proba")
    shape = proba.shape[:-1]
    if action_onehot is not None:
        _contig(action_onehot, torch.int64, "action")
    if choice is not None:
        _contig(choice, torch.int32, "choice")
    lp = torch.empty(shape, dtype=torch.float32, device=proba.device) if want_logprob else None
    ent = torch.empty(shape, dtype=torch.float32, device=proba.device) if want_entropy else None
    _lib.check(L.tarl_graphdist_logprob_entropy_fwd(plan.handle, proba.data_ptr(), B, _lib.ptr(action_onehot),
                                                    _lib.ptr(choice), _lib.ptr(lp), _lib.ptr(ent),
                                                    _lib.current_stream()))
    return lp, ent


def graphdist_logprob_entropy_bwd(plan: Plan, proba, temperature, *, action_onehot=None, choice=None,
                                  grad_log_prob=None, grad_entropy=None, log_prob_fwd=None):
    L = _lib.load()
    _contig(proba, torch.float32, "proba")
    B = _rows(proba, plan.num_edges, "proba")
    for name, t in (("grad_log_prob", grad_log_prob), ("grad_entropy", grad_entropy), ("log_prob_fwd", log_prob_fwd)):
        if t is not None:
            _contig(t, torch.float32, name)
    grad = torch.zeros_like(proba)
    _lib.check(L.tarl_graphdist_logprob_entropy_bwd(plan.handle, proba.data_ptr(), B, float(temperature),
                                                    _lib.ptr(action_onehot), _lib.ptr(choice),
                                                    _lib.ptr(grad_log_prob), _lib.ptr(grad_entropy),
                                                    _lib.ptr(log_prob_fwd), grad.data_ptr(), _lib.current_stream()))
    return grad


# ---- policy -----------------------------------------------------------------------------------------------------------
def _road_index_view(node_features: torch.Tensor, plan: Plan):
    """node_features (..., N, C>=7): the ROAD_INDEX observation column (ObservationFeatureHelpers.ROAD_INDEX = 6)."""
    _check_dev(node_features, torch.float32, "node_features")
    if node_features.dim() == 2:
        nf = node_features.unsqueeze(0)
    else:
        nf = node_features.reshape(-1, node_features.size(-2), node_features.size(-1))
    if nf.size(1) != plan.num_nodes or nf.size(2) < 7:
        raise ValueError("node_features must be (..., N, >=7)")
    col = nf[:, :, 6]
    return col, nf.size(0), col.stride(0), col.stride(1)


def policy_edge_logits(plan: Plan, node_features, emb):
    L = _lib.load()
    _contig(emb, torch.float32, "emb")
    col, B, bs, ns = _road_index_view(node_features, plan)
    shape = (plan.num_edges,) if node_features.dim() == 2 else tuple(node_features.shape[:-2]) + (plan.num_edges,)
    logits = torch.empty(shape, dtype=torch.float32, device=emb.device)
    _lib.check(L.tarl_policy_edge_logits_fwd(plan.handle, col.data_ptr(), bs, ns, B, emb.data_ptr(), emb.numel(),
                                             logits.data_ptr(), _lib.current_stream()))
    return logits


def policy_edge_logits_bwd(plan: Plan, node_features, grad_logits, num_embeddings):
    L = _lib.load()
    _contig(grad_logits, torch.float32, "grad_logits")
    col, B, bs, ns = _road_index_view(node_features, plan)
    grad_emb = torch.zeros(num_embeddings, dtype=torch.float32, device=grad_logits.device)
    scratch = None
    if bs == 0 and B >= 256:      # one observation broadcast over many rows: the row sum in parallel chunks
        scratch = torch.empty(int(L.tarl_policy_edge_logits_bwd_scratch_floats(plan.handle, B)), dtype=torch.float32,
                              device=grad_logits.device)
    _lib.check(L.tarl_policy_edge_logits_bwd(plan.handle, col.data_ptr(), bs, ns, B, grad_logits.data_ptr(),
                                             grad_emb.data_ptr(), num_embeddings, _lib.ptr(scratch), _lib.current_stream()))
    return grad_emb


# ---- per-edge MLP policy head ----------------------------------------------------------------------------------------------
class EdgeMlpWeights:
    """Flat views of the reference's ``edge_mlp.{0,2,4}.{weight,bias}`` tensors (device, fp32, contiguous)."""

    def __init__(self, w1, b1, w2, b2, w3, b3):
        self.w1, self.b1, self.w2, self.b2, self.w3, self.b3 = (_contig(t.detach(), torch.float32, n) for t, n in
                                                               ((w1, "w1"), (b1, "b1"), (w2, "w2"), (b2, "b2"),
                                                                (w3.reshape(-1), "w3"), (b3, "b3")))
        if self.w1.shape != (64, 33) or self.w2.shape != (32, 64) or self.w3.numel() != 32:
            raise ValueError("edge_mlp must be 33 -> 64 -> 32 -> 1")

    def ptrs(self):
        return [t.data_ptr() for t in (self.w1, self.b1, self.w2, self.b2, self.w3, self.b3)]


def policy_obs16(node_features, agent_index, agent_features):
    """x = cat(node_features[..., :7], agent_features[agent_index]) -> (M, N, 16) (src/agents/mpnn_agent.py:166-178).
    ``node_features`` (N, >=7) or (M, N, >=7) (last dim contiguous), ``agent_index`` int64 matching, ``agent_features``
    (A, 9) shared or (M, A, 9)."""
    L = _lib.load()
    _check_dev(node_features, torch.float32, "node_features")
    nf = node_features.unsqueeze(0) if node_features.dim() == 2 else node_features
    M, N = nf.shape[:2]
    if nf.stride(-1) != 1 or nf.stride(0) != N * nf.stride(1):     # rows may be strided (a view of x), samples may not
        nf = nf.contiguous()
    ai = _contig(agent_index.reshape(M, N).to(torch.int64), torch.int64, "agent_index")
    ag = _contig(agent_features, torch.float32, "agent_features")
    A = ag.size(-2)
    obs = torch.empty((M, N, 16), dtype=torch.float32, device=nf.device)
    _lib.check(L.tarl_policy_obs16(nf.data_ptr(), nf.stride(1), ai.data_ptr(), ag.data_ptr(), A,
                                   A * 9 if ag.dim() == 3 else 0, M, N, obs.data_ptr(), _lib.current_stream()))
    return obs


def fused_obs16(plan: Plan, fs, x, Nmax, agent_features, out=None):
    """The same observation from the packed state of the fused engine: (B, N, 16)."""
    L = _lib.load()
    B, N, bs, ldx = _state(x, Nmax)
    A, abs_ = _agents(agent_features, B)
    obs = out if out is not None else torch.empty((B, N, 16), dtype=torch.float32, device=x.device)
    _lib.check(L.tarl_fused_obs16(plan.handle, fs.ref, x.data_ptr(), B, bs, ldx, Nmax, agent_features.data_ptr(), A, abs_,
                                  obs.data_ptr(), _lib.current_stream()))
    return obs


def fused_obs16_bf16(plan: Plan, fs, x, Nmax, agent_features, out=None):
    """The packed state's observation rounded to bf16 (RNE): (B, N, 16) torch.bfloat16."""
    L = _lib.load()
    B, N, bs, ldx = _state(x, Nmax)
    A, abs_ = _agents(agent_features, B)
    obs = out if out is not None else torch.empty((B, N, 16), dtype=torch.bfloat16, device=x.device)
    _lib.check(L.tarl_fused_obs16_bf16(plan.handle, fs.ref, x.data_ptr(), B, bs, ldx, Nmax, agent_features.data_ptr(), A,
                                       abs_, obs.data_ptr(), _lib.current_stream()))
    return obs


def fused_obs16_rows(plan: Plan, fs, x, Nmax, agent_features, env, slot, out):
    """fp32 observation rows of a few environments: ``out[slot[j]] = obs[env[j]]`` (``out`` (K, N, 16), int32 lists)."""
    L = _lib.load()
    B, N, bs, ldx = _state(x, Nmax)
    A, abs_ = _agents(agent_features, B)
    _contig(env, torch.int32, "env")
    _contig(slot, torch.int32, "slot")
    _contig(out, torch.float32, "out")
    _lib.check(L.tarl_fused_obs16_rows(plan.handle, fs.ref, x.data_ptr(), B, bs, ldx, Nmax, agent_features.data_ptr(), A,
                                       abs_, env.data_ptr(), slot.data_ptr(), env.numel(), out.data_ptr(),
                                       _lib.current_stream()))
    return out


def fused_set_actions(plan: Plan, fs, choice8):
    """Env-major action bytes ``choice8`` (B, N) uint8 -> the SELECTED_ROAD column of the packed state (env-minor). A byte
    with bit 7 set keeps the road's previous value; the completed code is written back into ``choice8``."""
    L = _lib.load()
    _contig(choice8, torch.uint8, "choice8")
    if choice8.dim() != 2 or choice8.size(0) != fs.B or choice8.size(1) != plan.num_nodes:
        raise ValueError(f"choice8 must be (B, N) = ({fs.B}, {plan.num_nodes})")
    _lib.check(L.tarl_fused_set_actions(plan.handle, fs.ref, fs.B, choice8.data_ptr(), _lib.current_stream()))
    return choice8


EDGE_MLP_PRECISIONS = ("fp32", "bf16", "x3")


def _edge_mlp_precision(bf16, precision):
    """``precision``: "fp32" = fp32 MFMA (exact fp32 products), "bf16" = bf16 MFMA, "x3" = fp32 accuracy on the bf16 pipe
    (operands split into three exact bf16 pieces). ``bf16=True`` is the older spelling of precision="bf16"."""
    if precision is None:
        precision = "bf16" if bf16 else "fp32"
    if precision not in EDGE_MLP_PRECISIONS:
        raise ValueError(f"precision must be one of {EDGE_MLP_PRECISIONS}")
    return precision


def policy_edge_mlp(plan: Plan, obs16, ec: EdgeConst, w: EdgeMlpWeights, *, bf16=False, precision=None, out=None):
    """obs16 (M, N, 16) -> logits (M, E) of the per-edge MLP head: fp32 MFMA (default), bf16 MFMA (``bf16=True`` /
    ``precision="bf16"``) or fp32-accurate on the bf16 pipe (``precision="x3"``); ``obs16`` in torch.bfloat16
    (fused_obs16_bf16) selects the bf16 MFMA kernel that reads bf16 observations."""
    L = _lib.load()
    if obs16.dtype == torch.bfloat16:
        _contig(obs16, torch.bfloat16, "obs16")
        if obs16.shape[1:] != (plan.num_nodes, 16):
            raise ValueError("obs16 must be (M, num_nodes, 16)")
        M = obs16.size(0)
        logits = out if out is not None else torch.empty((M, plan.num_edges), dtype=torch.float32, device=obs16.device)
        _lib.check(L.tarl_policy_edge_mlp_fwd(plan.handle, obs16.data_ptr(), M, ec.edge_attr.data_ptr(), *w.ptrs(), 2,
                                              logits.data_ptr(), _lib.current_stream()))
        return logits
    _contig(obs16, torch.float32, "obs16")
    M = obs16.size(0)
    if obs16.shape[1:] != (plan.num_nodes, 16):
        raise ValueError("obs16 must be (M, num_nodes, 16)")
    logits = out if out is not None else torch.empty((M, plan.num_edges), dtype=torch.float32, device=obs16.device)
    code = {"fp32": 0, "bf16": 1, "x3": 3}[_edge_mlp_precision(bf16, precision)]
    _lib.check(L.tarl_policy_edge_mlp_fwd(plan.handle, obs16.data_ptr(), M, ec.edge_attr.data_ptr(), *w.ptrs(),
                                          code, logits.data_ptr(), _lib.current_stream()))
    return logits


def policy_edge_mlp_bwd(plan: Plan, obs16, ec: EdgeConst, w: EdgeMlpWeights, grad_logits, grads):
    """Accumulates into ``grads`` = (gw1, gb1, gw2, gb2, gw3, gb3), tensors shaped like the weights (fp32, contiguous)."""
    L = _lib.load()
    _contig(obs16, torch.float32, "obs16")
    gl = _contig(grad_logits, torch.float32, "grad_logits")
    M = obs16.size(0)
    if gl.numel() != M * plan.num_edges:
        raise ValueError("grad_logits must be (M, E)")
    scratch = torch.empty(int(L.tarl_policy_edge_mlp_bwd_scratch_floats(plan.handle, M)), dtype=torch.float32,
                          device=obs16.device)
    gs = [_contig(g, torch.float32, "grad") for g in grads]
    _lib.check(L.tarl_policy_edge_mlp_bwd(plan.handle, obs16.data_ptr(), M, ec.edge_attr.data_ptr(), *w.ptrs(),
                                          gl.data_ptr(), scratch.data_ptr(), *(g.data_ptr() for g in gs),
                                          _lib.current_stream()))


# ---- critic -----------------------------------------------------------------------------------------------------------
class CriticWeights:
    """Flat views of the reference's ``final_mlp.{0,2,4}.{weight,bias}`` tensors (device, fp32, contiguous)."""

    def __init__(self, w1, b1, w2, b2, w3, b3):
        self.w1, self.b1, self.w2, self.b2, self.w3, self.b3 = (_contig(t, torch.float32, n) for t, n in
                                                               ((w1, "w1"), (b1, "b1"), (w2, "w2"), (b2, "b2"),
                                                                (w3, "w3"), (b3, "b3")))
        if self.w1.size(0) != 64 or self.w2.shape != (64, 64) or self.w3.numel() != 64:
            raise ValueError("critic must be (N+1)->64->64->1")
        self.N = self.w1.size(1) - 1


def critic_forward(cw: CriticWeights, counts, time_rows, rows_per_time=1, *, keep_hidden=False, split_k=False):
    """counts (M, N) fp32 with contiguous last dim (row stride free); time_rows (ceil(M / rows_per_time),).
    split_k (fp32 rows only): spread the first layer of FEW rows over the input columns (tarl_critic_mlp_fwd_splitk)."""
    L = _lib.load()
    u8 = counts.dtype == torch.uint8           # the rollout buffers' count bytes (widened inside the kernel)
    _check_dev(counts, torch.uint8 if u8 else torch.float32, "counts")
    if counts.dim() != 2 or counts.stride(1) != 1 or counts.size(1) != cw.N:
        raise ValueError(f"counts must be (M, {cw.N}) with a contiguous last dim")
    _contig(time_rows, torch.float32, "time_rows")
    M = counts.size(0)
    if time_rows.numel() * rows_per_time < M:
        raise ValueError("time_rows too short")
    value = torch.empty(M, dtype=torch.float32, device=counts.device)
    h1 = torch.empty((M, 64), dtype=torch.float32, device=counts.device) if keep_hidden else None
    h2 = torch.empty((M, 64), dtype=torch.float32, device=counts.device) if keep_hidden else None
    if split_k and not u8:
        scratch = torch.empty(int(L.tarl_critic_splitk_scratch_floats(M, cw.N)), dtype=torch.float32,
                              device=counts.device)
        _lib.check(L.tarl_critic_mlp_fwd_splitk(counts.data_ptr(), counts.stride(0), M, cw.N, time_rows.data_ptr(),
                                                rows_per_time, cw.w1.data_ptr(), cw.b1.data_ptr(), cw.w2.data_ptr(),
                                                cw.b2.data_ptr(), cw.w3.data_ptr(), cw.b3.data_ptr(), scratch.data_ptr(),
                                                value.data_ptr(), _lib.ptr(h1), _lib.ptr(h2), _lib.current_stream()))
        return value, h1, h2
    fn = L.tarl_critic_mlp_fwd_u8 if u8 else L.tarl_critic_mlp_fwd
    _lib.check(fn(counts.data_ptr(), counts.stride(0), M, cw.N, time_rows.data_ptr(), rows_per_time,
                  cw.w1.data_ptr(), cw.b1.data_ptr(), cw.w2.data_ptr(), cw.b2.data_ptr(),
                  cw.w3.data_ptr(), cw.b3.data_ptr(), value.data_ptr(), _lib.ptr(h1), _lib.ptr(h2),
                  _lib.current_stream()))
    return value, h1, h2


def critic_backward(cw: CriticWeights, counts, time_rows, rows_per_time, h1, h2, grad_value, grads):
    """Accumulates into ``grads`` = (gw1, gb1, gw2, gb2, gw3, gb3), tensors shaped like the weights."""
    L = _lib.load()
    M = counts.size(0)
    _contig(grad_value, torch.float32, "grad_value")
    scratch = torch.empty(int(L.tarl_critic_mlp_bwd_scratch_floats(M, cw.N)), dtype=torch.float32, device=counts.device)
    gw1, gb1, gw2, gb2, gw3, gb3 = (_contig(g, torch.float32, "grad") for g in grads)
    _lib.check(L.tarl_critic_mlp_bwd(counts.data_ptr(), counts.stride(0), M, cw.N, time_rows.data_ptr(), rows_per_time,
                                     cw.w1.data_ptr(), cw.w2.data_ptr(), cw.w3.data_ptr(), h1.data_ptr(), h2.data_ptr(),
                                     grad_value.data_ptr(), scratch.data_ptr(), gw1.data_ptr(), gb1.data_ptr(),
                                     gw2.data_ptr(), gb2.data_ptr(), gw3.data_ptr(), gb3.data_ptr(),
                                     _lib.current_stream()))


# ---- PPO --------------------------------------------------------------------------------------------------------------
def gae(reward, value, next_value, *, done=None, terminated=None, gamma=0.99, lmbda=0.95):
    """Time-major (T, B) fp32 tensors -> (advantage, value_target), un-normalised."""
    L = _lib.load()
    for n, t in (("reward", reward), ("value", value), ("next_value", next_value)):
        _contig(t, torch.float32, n)
    T, B = reward.shape
    adv, tgt = torch.empty_like(reward), torch.empty_like(reward)
    _lib.check(L.tarl_gae(reward.data_ptr(), value.data_ptr(), next_value.data_ptr(), _lib.ptr(done),
                          _lib.ptr(terminated), T, B, float(gamma), float(lmbda), adv.data_ptr(), tgt.data_ptr(),
                          _lib.current_stream()))
    return adv, tgt


def advantage_stats(adv):
    L = _lib.load()
    _contig(adv, torch.float32, "advantage")
    partial = torch.empty(512, dtype=torch.float64, device=adv.device)
    stats = torch.empty(3, dtype=torch.float64, device=adv.device)
    _lib.check(L.tarl_advantage_stats(adv.data_ptr(), adv.numel(), partial.data_ptr(), stats.data_ptr(),
                                      _lib.current_stream()))
    return stats


def advantage_normalize_(adv, stats):
    L = _lib.load()
    _lib.check(L.tarl_advantage_normalize(adv.data_ptr(), adv.numel(), stats.data_ptr(), _lib.current_stream()))
    return adv


def ppo_loss(lp_new, lp_old, adv, value, target, entropy, *, clip_epsilon=0.2, entropy_coef=0.01, critic_coef=1.0,
             grad_scale=1.0, want_grads=True):
    """-> (out6, g_lp, g_ent, g_val); out6 = loss_objective, loss_critic, loss_entropy, clip_fraction, kl_approx, ESS."""
    L = _lib.load()
    ts = [_contig(t.reshape(-1), torch.float32, "ppo input") for t in (lp_new, lp_old, adv, value, target, entropy)]
    M = ts[0].numel()
    if any(t.numel() != M for t in ts):
        raise ValueError("ppo_loss inputs must have the same number of elements")
    out = torch.empty(6, dtype=torch.float32, device=ts[0].device)
    gs = [torch.empty(M, dtype=torch.float32, device=ts[0].device) if want_grads else None for _ in range(3)]
    _lib.check(L.tarl_ppo_loss(*(t.data_ptr() for t in ts), M, float(clip_epsilon), float(entropy_coef),
                               float(critic_coef), float(grad_scale), out.data_ptr(), *(_lib.ptr(g) for g in gs),
                               _lib.current_stream()))
    return out, gs[0], gs[1], gs[2]


def adam_step_(param, grad, exp_avg, exp_avg_sq, step, *, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    L = _lib.load()
    for n, t in (("param", param), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        _contig(t, torch.float32, n)
    _lib.check(L.tarl_adam_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                param.numel(), int(step), float(lr), float(beta1), float(beta2), float(eps),
                                float(grad_scale), _lib.current_stream()))
    return param


# ---- fused rollout frame ----------------------------------------------------------------------------------------------
class FusedState:
    """Side buffers of the fused path (``tarl_fused`` in include/tarl_hip.h), ENV-MINOR ([node][env]): the packed dense
    words (hdp, tl, post, sel8), the event-only byte gc8, static node records, the slot-interleaved FIFO store and the
    agent SoA. They hold the state between :func:`fused_pack` and :func:`fused_export`."""

    def __init__(self, plan: Plan, B: int, A: int, device, Nmax: int = 15, env_base: int = 0):
        """``env_base``: global id of environment 0 of this batch — the device noise streams are indexed by
        ``env_base + b`` (include/tarl_hip.h: tarl_fused.env_base), so a shard of a larger batch reproduces the larger
        batch's trajectories."""
        L = _lib.load()
        N, E = plan.num_nodes, plan.num_edges
        if Nmax > 127 or plan.max_out > 126:
            raise _lib.TarlError(f"the fused path needs Nmax <= 127 and out-degree <= 126 (got Nmax={Nmax}, max out-degree="
                                 f"{plan.max_out}); construct SimEngine(..., fused=False) for this graph")
        f32 = dict(dtype=torch.float32, device=device)
        i32 = dict(dtype=torch.int32, device=device)
        # Never zeroed after this: a CLEAN row's dead slots are zero by rule, whatever the store holds (csrc/fused_common.h)
        self.ld_slots = int(L.tarl_fused_slot_floats(Nmax))
        self.slots = torch.zeros((N, B, self.ld_slots), **f32)
        self.hdp = torch.zeros((N, B, 2), **i32)
        self.tl = torch.zeros((N, B), **i32)
        self.gc8 = torch.zeros((N, B), dtype=torch.uint8, device=device)     # pending-garbage code (event-only byte)
        self.post = torch.zeros((N, B), **i32)
        self.st0 = torch.zeros((N, 4), **f32)
        self.sel8 = torch.zeros((N, B), dtype=torch.uint8, device=device)
        self.sel = torch.zeros((N, B), **f32)
        self.node_rec = torch.zeros((N, 36), **i32)          # static records (fused_common.h: NodeRec / InRec)
        self.in_rec = torch.zeros((E + 4, 5), **i32)
        self.out_pad = torch.zeros(E + 4, **i32)
        self.acc_slots = 32      # accumulator banks (spread the per-environment atomics of the N/chunk workgroups)
        self.acc_lp = torch.zeros((self.acc_slots, B), dtype=torch.int64, device=device)
        self.acc_n = torch.zeros((self.acc_slots, B), **f32)
        self.acc_w = torch.zeros((self.acc_slots, B), **f32)
        self.a_origin = torch.zeros((B, A), **i32)
        self.a_dest = torch.zeros((B, A), **i32)
        self.a_dep = torch.zeros((B, A), **f32)
        self.a_status = torch.zeros((B, A), dtype=torch.uint8, device=device)
        self.a_order = torch.zeros((B, A), **i32)
        self.a_dep_sorted = torch.zeros((B, A), **f32)
        self.cur_lo = torch.zeros(B, **i32)
        self.a_win = torch.zeros((B, A, 4), **i32)
        self.a_ins = torch.zeros((B, A), dtype=torch.uint8, device=device)
        self.a_rank = torch.zeros((B, A), **i32)
        self.flags = torch.zeros(1, **i32)
        # the library's device-resident copy of the pointer table below (include/tarl_hip.h: tarl_fused.bufs_dev)
        self.bufs_dev = torch.zeros(int(L.tarl_fused_bufs_bytes()), dtype=torch.uint8, device=device)
        self.order_valid = False
        self.struct = _lib.FusedStruct(self.hdp.data_ptr(), self.tl.data_ptr(), self.gc8.data_ptr(),
                                       self.post.data_ptr(), self.st0.data_ptr(), self.slots.data_ptr(), self.ld_slots,
                                       self.sel8.data_ptr(), self.sel.data_ptr(), self.node_rec.data_ptr(),
                                       self.in_rec.data_ptr(), self.out_pad.data_ptr(),
                                       self.acc_lp.data_ptr(), self.acc_n.data_ptr(), self.acc_w.data_ptr(),
                                       self.a_origin.data_ptr(), self.a_dest.data_ptr(), self.a_dep.data_ptr(),
                                       self.a_status.data_ptr(), None, self.cur_lo.data_ptr(), None, None, None, None,
                                       self.acc_slots, self.flags.data_ptr(), int(env_base), 0.0, 0, self.bufs_dev.data_ptr())
        self.B, self.N, self.A, self.Nmax, self.env_base = B, N, A, Nmax, int(env_base)

    # -- unpacked views of the dense words (tests / debugging; torch plumbing, never on a hot path) ----------------------
    @property
    def count(self):
        return (self.hdp[..., 0] & 127).to(torch.float32)      # (bit 7 of the count byte: HD_DIRTY, csrc/fused_common.h)

    @property
    def head_id(self):
        return ((self.hdp[..., 0] >> 8) & 0xFFFFFF).to(torch.float32)

    @property
    def head_dep(self):
        """Stored head departure; not maintained for an empty row that idled in the last frame (count 0, tl bit 0 clear)."""
        return self.hdp[..., 1].contiguous().view(torch.float32)

    @property
    def tail_id(self):
        return ((self.tl >> 8) & 0xFFFFFF).to(torch.float32)

    @property
    def head_slot_arrival(self):
        """Arrival field of the slot record at every row's ring offset: the head's arrival time where the row holds
        somebody (csrc/fused_common.h: head_arrival)."""
        hoff = ((self.tl >> 1) & 127).long().unsqueeze(-1)
        w = 8 if int(_lib.load().tarl_fused_slot_floats(1)) == 8 else 3    # (developer build -DTARL_SLW=8: 32-byte records)
        arr = self.slots[..., :w * self.Nmax].reshape(self.N, self.B, self.Nmax, w)[..., 1]
        return torch.gather(arr, 2, hoff).squeeze(-1)

    def sort_agents(self, agent_features):
        """Departure-time order of every environment's population (static while DEPARTURE_TIME is not edited): lets the
        insert kernel scan a small window per frame. Plain torch sort — set-up plumbing, not on the per-frame path."""
        dep = agent_features.reshape(self.B, self.A, 9)[:, :, 2]
        order = torch.argsort(dep, dim=1, stable=True)
        self.a_order.copy_(order.to(torch.int32))
        self.a_dep_sorted.copy_(torch.gather(dep, 1, order))
        self.a_rank.scatter_(1, order, torch.arange(self.A, dtype=torch.int32, device=order.device).expand(self.B, -1))
        self.struct.a_order = self.a_order.data_ptr()
        self.struct.a_dep_sorted = self.a_dep_sorted.data_ptr()
        self.struct.a_win = self.a_win.data_ptr()       # filled by tarl_fused_pack
        self.struct.a_ins = self.a_ins.data_ptr()
        self.struct.a_rank = self.a_rank.data_ptr()
        # hint for the insert kernel's geometry: departures per second and environment at the busiest second of the
        # schedule (the never-departing dummy, 48 h, is left out). One histogram over all environments: set-up plumbing.
        real = self.a_dep_sorted[self.a_dep_sorted < 86400.0 * 1.5]
        if real.numel() > 0:
            lo, hi = float(real.min()), float(real.max())
            bins = max(1, min(1 << 20, int(hi - lo) + 1))
            self.struct.due_rate = float(torch.histc(real, bins=bins, min=lo, max=lo + bins).max()) / self.B
        self.order_valid = True

    def check_flags(self):
        """Read the device status word (one host synchronisation) and raise on a domain exit."""
        raise_on_flags(int(self.flags.item()))

    @property
    def ref(self):
        return C.byref(self.struct)


def fused_path_supported(edge_index: torch.Tensor, Nmax: int) -> bool:
    """Can the packed path (FusedState) represent this graph? Nmax <= 127 (count byte + 7-bit ring offset), out-degree
    <= 126 (7-bit rank of the chosen out-edge) and no parallel dual edges (two out-edges of one node to the same target
    have no unique rank: FLAG_AMBIGUOUS_EDGES). Decided on the host from the topology alone, so every rank of a
    data-parallel job takes the same branch; graphs outside it run on the unfused entry points."""
    ei = edge_index.detach().to("cpu", torch.int64)
    if ei.numel() == 0:
        return Nmax <= 127
    n = int(ei.max()) + 1
    key = ei[0] * n + ei[1]
    return bool(Nmax <= 127 and int(torch.bincount(ei[0]).max()) <= 126 and key.unique().numel() == key.numel())


def raise_on_flags(v: int):
    """Turn the bits of a device status word (include/tarl_hip.h: TARL_FLAG_*) into a :class:`TarlError`."""
    if v & _lib.FLAG_COUNT_AT_NMAX:
        raise _lib.TarlError("a FIFO count reached Nmax: the state left the reference's defined domain (its "
                             "DirectionMPNN.update silently overwrites the neighbouring FIFO blocks there and raises "
                             "IndexError only a few steps later, src/direction_mpnn.py:172-191)")
    if v & _lib.FLAG_AMBIGUOUS_EDGES:
        raise _lib.TarlError("two out-edges of one node lead to the same ROAD_INDEX: SELECTED_ROAD has no unique rank "
                             "on this graph; construct SimEngine(..., fused=False)")
    if v & _lib.FLAG_PACK_RANGE:
        raise _lib.TarlError("pack: a FIFO count above 255 or an agent id at / above 2^24 does not fit the packed words")
    if v & _lib.FLAG_CHOICE_OVERFLOW:
        raise _lib.TarlError("more than 65536 nodes drew no action in one block of frames: degenerate policy tables")


def fused_pack(plan: Plan, fs: FusedState, x, Nmax, agent_features, congestion_constant=None, sort_agents=None, *,
               ec: EdgeConst):
    """``ec``: the graph's edge constants (the turn probabilities go into the static in-edge records);
    ``sort_agents``: True = (re)build the departure-time order, None = build it once, False = never."""
    L = _lib.load()
    B, N, bs, ldx = _state(x, Nmax)
    A, abs_ = _agents(agent_features, B)
    if sort_agents or (sort_agents is None and not fs.order_valid):
        fs.sort_agents(agent_features)
    _lib.check(L.tarl_fused_pack(plan.handle, fs.ref, x.data_ptr(), B, bs, ldx, Nmax, _lib.ptr(congestion_constant),
                                 ec.edge_attr.data_ptr(), agent_features.data_ptr(), A, abs_, _lib.current_stream()))


def fused_reset(plan: Plan, fs: FusedState, agent_features):
    """SimulatorEnv._reset on the packed state (no round trip through x)."""
    L = _lib.load()
    A, abs_ = _agents(agent_features, fs.B)
    _lib.check(L.tarl_fused_reset(plan.handle, fs.ref, fs.B, fs.Nmax, agent_features.data_ptr(), A, abs_,
                                  _lib.current_stream()))


def fused_export(plan: Plan, fs: FusedState, x, Nmax, last_step_time):
    """Write the packed state (FIFO columns, NUMBER_OF_AGENT, SELECTED_ROAD) back into ``x`` (reference layout).
    ``last_step_time``: the clock passed to the most recent :func:`fused_frame`."""
    L = _lib.load()
    B, N, bs, ldx = _state(x, Nmax)
    _lib.check(L.tarl_fused_export(plan.handle, fs.ref, x.data_ptr(), B, bs, ldx, Nmax, float(last_step_time),
                                   _lib.current_stream()))


class PolicyTables:
    """Per-edge tables of the live policy's GraphDistribution (plan order), valid until the embedding changes."""

    def __init__(self, plan: Plan, device):
        self.thresholds = torch.empty(plan.num_edges, dtype=torch.float32, device=device)
        self.log_probs = torch.empty(plan.num_edges, dtype=torch.int64, device=device)   # 2^-32 fixed point
        self.entropy = torch.empty(1, dtype=torch.float32, device=device)
        self.base = torch.empty(plan.num_groups + 1, dtype=torch.float64, device=device)


def fused_policy_prepare(plan: Plan, fs: FusedState, emb, temperature=1.0, tables: PolicyTables | None = None):
    L = _lib.load()
    _contig(emb, torch.float32, "emb")
    if tables is None:
        tables = PolicyTables(plan, emb.device)
    _lib.check(L.tarl_fused_policy_prepare(plan.handle, fs.ref, emb.data_ptr(), emb.numel(), float(temperature),
                                           tables.base.data_ptr(), tables.thresholds.data_ptr(),
                                           tables.log_probs.data_ptr(), tables.entropy.data_ptr(),
                                           _lib.current_stream()))
    return tables


def fused_apply_choice(plan: Plan, fs: FusedState, choice):
    """SELECTED_ROAD of the packed state <- an externally sampled action: ``choice`` (B, N) int32 edge ids (-1: none)."""
    L = _lib.load()
    _contig(choice, torch.int32, "choice")
    if tuple(choice.shape) != (fs.B, fs.N):
        raise ValueError("choice must be (B, N)")
    _lib.check(L.tarl_fused_apply_choice(plan.handle, fs.ref, fs.B, choice.data_ptr(), _lib.current_stream()))


def fused_frame(plan: Plan, fs: FusedState, tables: PolicyTables | None, agent_features, ec: EdgeConst, t, *,
                use_cong=True, prev_time=None, uniform=None, policy_seed=0, policy_counter=0, gumbel=None, seed=0,
                counter=0, dtt=None, popped=None, withdrawn=None, scratch=None, choice=None, log_prob=None, entropy=None,
                reward=None, counts=None):
    """One collector frame for all B environments: sample + log_prob + choice phase, core step, withdraw, insert, reward.
    ``tables=None`` skips the choice phase (the action was written by :func:`fused_apply_choice`).
    ``choice`` (N, B) int32 and ``counts`` (N, B) fp32 are ENV-MINOR; ``dtt`` (B, E), ``popped`` / ``withdrawn`` (B, N)
    uint8, ``log_prob`` / ``entropy`` / ``reward`` (B,). Outputs are written into the tensors passed in.
    ``prev_time``: the previous frame's clock (default ``t - 1``; only ``dtt`` reads it)."""
    L = _lib.load()
    B, Nmax = fs.B, fs.Nmax
    A, abs_ = _agents(agent_features, B)
    if scratch is None:
        scratch = torch.empty((B, 2 * A), dtype=torch.int32, device=agent_features.device)
    for n, tt, dt in (("gumbel", gumbel, torch.float32), ("uniform", uniform, torch.float32),
                      ("dtt", dtt, torch.float32), ("reward", reward, torch.float32), ("counts", counts, torch.float32),
                      ("popped", popped, torch.uint8), ("withdrawn", withdrawn, torch.uint8),
                      ("choice", choice, torch.int32), ("log_prob", log_prob, torch.float32),
                      ("entropy", entropy, torch.float32)):
        if tt is not None:
            _contig(tt, dt, n)
    th, lg, en = ((tables.thresholds.data_ptr(), tables.log_probs.data_ptr(), tables.entropy.data_ptr())
                  if tables is not None else (None, None, None))
    _lib.check(L.tarl_fused_frame(plan.handle, fs.ref, B, Nmax, th, lg, en, _lib.ptr(uniform),
                                  int(policy_seed), int(policy_counter), agent_features.data_ptr(), A, abs_,
                                  ec.edge_attr.data_ptr(), ec.log_edge_attr.data_ptr(), ec.log_eps,
                                  1 if use_cong else 0, float(t), float(t - 1 if prev_time is None else prev_time),
                                  _lib.ptr(gumbel), int(seed), int(counter),
                                  _lib.ptr(dtt), _lib.ptr(popped), _lib.ptr(withdrawn), scratch.data_ptr(),
                                  _lib.ptr(choice), _lib.ptr(log_prob), _lib.ptr(entropy), _lib.ptr(reward),
                                  _lib.ptr(counts), _lib.current_stream()))


def _check_rollout_outputs(T, B, N, env_minor, m_env, choice, counts, log_prob, entropy, reward, dtt_node, events, leg):
    nb = (lambda t_, k: (t_, N, k)) if env_minor else (lambda t_, k: (t_, k, N))
    for name, tns, dt, shp in (("choice", choice, torch.uint8, nb(T, B)), ("counts", counts, torch.uint8, nb(T, B)),
                               ("log_prob", log_prob, torch.float32, (T, B)), ("entropy", entropy, torch.float32, (T, B)),
                               ("reward", reward, torch.float32, (T, B)),
                               ("dtt_node", dtt_node, torch.float32, nb(T, m_env)),
                               ("events", events, torch.uint8, nb(T, m_env)), ("leg", leg, torch.int32, (T, B, 2))):
        if tns is not None and (tns.dtype != dt or tuple(tns.shape) != shp or not tns.is_contiguous() or not tns.is_cuda):
            raise ValueError(f"{name} must be a contiguous cuda {dt} tensor of shape {shp}")


def fused_rollout(plan: Plan, fs: FusedState, tables: PolicyTables, agent_features, ec: EdgeConst, times, *, use_cong,
                  policy_seed, policy_counter0, seed, counter0, scratch, prev_time=None, choice=None, log_prob=None,
                  entropy=None, reward=None, counts=None, metrics_envs=0, dtt_node=None, events=None, leg=None):
    """``T = len(times)`` frames in one foreign call (tarl_fused_rollout). ``choice`` (T,N,B) uint8 (rank of the chosen
    out-edge, bit 7: none), ``counts`` (T,N,B) uint8 (counts[t] = per-node counts after frame t), ``log_prob`` / ``entropy``
    / ``reward`` (T,B) fp32, ``leg`` (T,B,2) int32, ``dtt_node`` (T,N,metrics_envs) fp32, ``events`` (T,N,metrics_envs)
    uint8 — all optional, contiguous device tensors. Frame t uses policy counter ``policy_counter0 + t`` and noise counter
    ``counter0 + t``."""
    L = _lib.load()
    T, B, N = len(times), fs.B, fs.N
    A, abs_ = _agents(agent_features, B)
    _contig(scratch, torch.int32, "scratch")
    _check_rollout_outputs(T, B, N, True, metrics_envs, choice, counts, log_prob, entropy, reward, dtt_node, events, leg)
    if getattr(fs, "acc_scratch", None) is None:     # double buffers of the merged insert + choice launch
        fs.acc_scratch = torch.zeros_like(fs.acc_lp)
    if choice is None and getattr(fs, "sel_scratch", None) is None:
        fs.sel_scratch = torch.empty_like(fs.sel8)
    need = int(L.tarl_fused_rollout_scratch_ints(plan.handle, T, B))
    if getattr(fs, "choice_scratch", None) is None or fs.choice_scratch.numel() < need:
        # unresolved-draw list + packed policy records + per-(frame, env) log-prob accumulators of the side stream
        fs.choice_scratch = torch.zeros(need, dtype=torch.int32, device=fs.sel8.device)
    tarr = (C.c_float * T)(*[float(t) for t in times])
    _lib.check(L.tarl_fused_rollout(plan.handle, fs.ref, B, fs.Nmax, T, tarr,
                                    float(times[0] - 1 if prev_time is None else prev_time), tables.thresholds.data_ptr(),
                                    tables.log_probs.data_ptr(), tables.entropy.data_ptr(), int(policy_seed),
                                    int(policy_counter0), agent_features.data_ptr(), A, abs_, ec.edge_attr.data_ptr(),
                                    ec.log_edge_attr.data_ptr(), ec.log_eps, 1 if use_cong else 0, int(seed),
                                    int(counter0), scratch.data_ptr(),
                                    _lib.ptr(getattr(fs, "sel_scratch", None)) if choice is None else None,
                                    fs.acc_scratch.data_ptr(), fs.choice_scratch.data_ptr(), _lib.ptr(choice),
                                    _lib.ptr(log_prob), _lib.ptr(entropy),
                                    _lib.ptr(reward), _lib.ptr(counts), int(metrics_envs), _lib.ptr(dtt_node),
                                    _lib.ptr(events), _lib.ptr(leg), _lib.current_stream()))


def fused_rollout_policy(plan: Plan, fs: FusedState, x, agent_features, ec: EdgeConst, w: EdgeMlpWeights, times, *,
                         use_cong, bf16=False, temperature, policy_seed, policy_counter0, seed, counter0, scratch,
                         prev_time=None, keep=None, obs_keep=None, choice8=None, log_prob=None, reward=None, counts=None,
                         metrics_envs=0, dtt_node=None, events=None, leg=None, precision=None):
    """``T = len(times)`` frames under the per-edge MLP policy in one foreign call (tarl_fused_rollout_policy).
    ``keep`` = (ptr, env, slot): ``ptr`` a Python list of T + 1 offsets, ``env`` / ``slot`` int32 device tensors — the
    observations (frame t, environment env[j]) for ptr[t] <= j < ptr[t + 1] are copied to ``obs_keep[slot[j]]``
    ((K, N, 16) fp32). ``choice8`` (T, B, N) uint8 ENV-MAJOR rank bytes; ``counts`` (T, N, B) uint8 env-minor; the
    other outputs as :func:`fused_rollout`."""
    L = _lib.load()
    T, B, N = len(times), fs.B, fs.N
    A, abs_ = _agents(agent_features, B)
    _, _, bs, ldx = _state(x, fs.Nmax)
    _contig(scratch, torch.int32, "scratch")
    _check_rollout_outputs(T, B, N, True, metrics_envs, None, counts, log_prob, None, reward, dtt_node, events, leg)
    _check_rollout_outputs(T, B, N, False, metrics_envs, choice8, None, None, None, None, None, None, None)
    dev = fs.sel8.device
    if getattr(fs, "obs_scratch", None) is None:
        fs.obs_scratch = torch.empty((B, N, 16), dtype=torch.float32, device=dev)
        fs.logits_scratch = torch.empty((B, plan.num_edges), dtype=torch.float32, device=dev)
        fs.dist_scratch = torch.empty((int(L.tarl_graphdist_rollout_scratch_bytes(plan.handle, B)) + 7) // 8,
                                      dtype=torch.float64, device=dev)
    kptr = kenv = kslot = None
    if keep is not None:
        ptr, kenv, kslot = keep
        if len(ptr) != T + 1 or ptr[0] != 0 or any(b < a for a, b in zip(ptr, ptr[1:])):
            raise ValueError("keep pointer list must be T + 1 non-decreasing offsets starting at 0")
        _contig(kenv, torch.int32, "keep env")
        _contig(kslot, torch.int32, "keep slot")
        _contig(obs_keep, torch.float32, "obs_keep")
        if kenv.numel() < ptr[-1] or kslot.numel() < ptr[-1] or obs_keep.shape[1:] != (N, 16):
            raise ValueError("keep arrays shorter than the pointer list, or obs_keep not (K, N, 16)")
        kptr = (C.c_int64 * (T + 1))(*[int(v) for v in ptr])
    tarr = (C.c_float * T)(*[float(t) for t in times])
    _lib.check(L.tarl_fused_rollout_policy(
        plan.handle, fs.ref, B, fs.Nmax, T, tarr, float(times[0] - 1 if prev_time is None else prev_time), x.data_ptr(),
        bs, ldx, agent_features.data_ptr(), A, abs_, ec.edge_attr.data_ptr(), ec.log_edge_attr.data_ptr(), ec.log_eps,
        1 if use_cong else 0, *w.ptrs(), {"fp32": 0, "bf16": 1, "x3": 2}[_edge_mlp_precision(bf16, precision)],
        float(temperature), int(policy_seed), int(policy_counter0),
        int(seed), int(counter0), kptr, _lib.ptr(kenv), _lib.ptr(kslot), _lib.ptr(obs_keep), fs.obs_scratch.data_ptr(),
        fs.logits_scratch.data_ptr(), fs.dist_scratch.data_ptr(), scratch.data_ptr(), _lib.ptr(choice8),
        _lib.ptr(log_prob), _lib.ptr(reward), _lib.ptr(counts), int(metrics_envs), _lib.ptr(dtt_node), _lib.ptr(events),
        _lib.ptr(leg), _lib.current_stream()))


def rollout_gather(plan: Plan, T, B, env_minor, idx=None, *, choice=None, counts=None):
    """Rollout bytes -> (choice_eid int32 (rows, N) | None, counts_f fp32 (rows, N) | None) for the (frame, env) pairs
    ``idx`` (int64 flat indices t * B + b; None = all ``T * B`` in order). ``choice`` / ``counts``: the uint8 buffers
    ((T,N,B) when ``env_minor`` else (T,B,N)); ``T`` is the buffers' leading extent."""
    L = _lib.load()
    N = plan.num_nodes
    some = choice if choice is not None else counts
    rows = T * B if idx is None else idx.numel()
    for nm, t_ in (("choice", choice), ("counts", counts)):
        if t_ is not None:
            _contig(t_, torch.uint8, nm)
            if t_.numel() != T * B * N:
                raise ValueError(f"{nm} must hold T*B*N bytes")
    if idx is not None:
        _contig(idx, torch.int64, "idx")
    ce = torch.empty((rows, N), dtype=torch.int32, device=some.device) if choice is not None else None
    cf = torch.empty((rows, N), dtype=torch.float32, device=some.device) if counts is not None else None
    _lib.check(L.tarl_rollout_gather(plan.handle, _lib.ptr(choice), _lib.ptr(counts), T, B, 1 if env_minor else 0,
                                     _lib.ptr(idx), rows, _lib.ptr(ce), _lib.ptr(cf), _lib.current_stream()))
    return ce, cf


# ---- MPNNValueNet (dormant message-passing critic) ---------------------------------------------------------------------------
def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def value_mpnn_forward(plan: Plan, node_features, agent_rows, edge_features, time, params, *, keep=False):
    """``node_features`` (M, N, 7), ``agent_rows`` (M, N, 9) or None, ``edge_features`` (M, E) or (E,), ``time`` (M,),
    ``params`` = the 12 parameter tensors in tarl_value_mpnn_fwd's order -> value (M,) [+ node_act, agg (M, N)]."""
    L = _lib.load()
    M, N = node_features.size(0), node_features.size(1)
    if N != plan.num_nodes or node_features.size(2) != 7:
        raise ValueError("node_features must be (M, num_nodes, 7)")
    nf = _contig(node_features, torch.float32, "node_features")
    ar = None if agent_rows is None else _contig(agent_rows, torch.float32, "agent_rows")
    ef = _contig(edge_features, torch.float32, "edge_features")
    ef_stride = 0 if ef.dim() == 1 or ef.size(0) == 1 else plan.num_edges
    tm = _contig(time, torch.float32, "time")
    ps = [_contig(p.detach(), torch.float32, "param") for p in params]
    value = torch.empty(M, dtype=torch.float32, device=nf.device)
    act = torch.empty((M, N), dtype=torch.float32, device=nf.device) if keep else None
    agg = torch.empty((M, N), dtype=torch.float32, device=nf.device) if keep else None
    _lib.check(L.tarl_value_mpnn_fwd(plan.handle, nf.data_ptr(), M, _lib.ptr(ar), ef.data_ptr(), ef_stride, tm.data_ptr(),
                                     _ptr_array(ps), value.data_ptr(), _lib.ptr(act), _lib.ptr(agg),
                                     _lib.current_stream()))
    return value, act, agg


def value_mpnn_backward(plan: Plan, node_features, agent_rows, edge_features, time, params, grad_value, act, agg):
    """Parameter gradients (list of 12 tensors shaped like ``params``) of sum(grad_value * value)."""
    L = _lib.load()
    M = node_features.size(0)
    nf = _contig(node_features, torch.float32, "node_features")
    ar = None if agent_rows is None else _contig(agent_rows, torch.float32, "agent_rows")
    ef = _contig(edge_features, torch.float32, "edge_features")
    ef_stride = 0 if ef.dim() == 1 or ef.size(0) == 1 else plan.num_edges
    ps = [_contig(p.detach(), torch.float32, "param") for p in params]
    grads = [torch.zeros_like(p) for p in ps]
    gv = _contig(grad_value, torch.float32, "grad_value")
    _lib.check(L.tarl_value_mpnn_bwd(plan.handle, nf.data_ptr(), M, _lib.ptr(ar), ef.data_ptr(), ef_stride,
                                     _contig(time, torch.float32, "time").data_ptr(), _ptr_array(ps), gv.data_ptr(),
                                     act.data_ptr(), agg.data_ptr(), _ptr_array(grads), _lib.current_stream()))
    return grads


def noise_export(plan: Plan, kind: str, seed: int, counter: int, env_ids):
    """The device noise of the Philox path for the listed GLOBAL environment ids (tarl_noise_export; test hook):
    ``kind="gumbel"`` -> (n, E) fp32 Gumbel values of DirectionMPNN.aggregate's race in ORIGINAL edge order (seed = the
    engine's ``seed``, counter = the frame's noise counter); ``kind="uniform"`` -> (n, G) uniforms of the action draw."""
    L = _lib.load()
    env = torch.as_tensor(env_ids, dtype=torch.int64).to(plan.device).contiguous()
    n = env.numel()
    width = plan.num_edges if kind == "gumbel" else plan.num_groups
    out = torch.empty((n, width), dtype=torch.float32, device=plan.device)
    _lib.check(L.tarl_noise_export(plan.handle, {"gumbel": 0, "uniform": 1}[kind], int(seed), int(counter), env.data_ptr(),
                                   n, out.data_ptr(), _lib.current_stream()))
    return out


def rollout_env_supported(plan: Plan) -> bool:
    return bool(_lib.load().tarl_rollout_env_supported(plan.handle))


def rollout_env(plan: Plan, fs: FusedState, tables: PolicyTables, agent_features, ec: EdgeConst, times, *, use_cong,
                policy_seed, policy_counter0, seed, counter0, scratch, prev_time=None, choice=None, log_prob=None,
                entropy=None, reward=None, counts=None, metrics_envs=0, dtt_node=None, events=None, leg=None):
    """Same contract as :func:`fused_rollout` through ``tarl_rollout_env`` (one workgroup per environment, LDS-resident
    records, one launch for all frames); the per-node buffers are ENV-MAJOR: ``choice`` / ``counts`` (T, B, N),
    ``dtt_node`` / ``events`` (T, metrics_envs, N)."""
    L = _lib.load()
    T, B, N = len(times), fs.B, fs.N
    A, abs_ = _agents(agent_features, B)
    _contig(scratch, torch.int32, "scratch")
    _check_rollout_outputs(T, B, N, False, metrics_envs, choice, counts, log_prob, entropy, reward, dtt_node, events, leg)
    tdev = torch.tensor([float(t) for t in times], dtype=torch.float32).to(fs.sel.device, non_blocking=True)
    if getattr(fs, "env_scratch", None) is None:
        fs.env_scratch = torch.empty(int(L.tarl_rollout_env_scratch_bytes(plan.handle)), dtype=torch.uint8,
                                     device=fs.sel.device)
    _lib.check(L.tarl_rollout_env(plan.handle, fs.ref, B, fs.Nmax, T, tdev.data_ptr(),
                                  float(times[0] - 1 if prev_time is None else prev_time), tables.thresholds.data_ptr(),
                                  tables.log_probs.data_ptr(), tables.entropy.data_ptr(), int(policy_seed),
                                  int(policy_counter0), agent_features.data_ptr(), A, abs_, ec.edge_attr.data_ptr(),
                                  ec.log_edge_attr.data_ptr(), ec.log_eps, 1 if use_cong else 0, int(seed),
                                  int(counter0), scratch.data_ptr(), fs.env_scratch.data_ptr(), _lib.ptr(choice),
                                  _lib.ptr(log_prob), _lib.ptr(entropy), _lib.ptr(reward), _lib.ptr(counts),
                                  int(metrics_envs), _lib.ptr(dtt_node), _lib.ptr(events), _lib.ptr(leg),
                                  _lib.current_stream()))
    return tdev


_SPLIT_SCRATCH = {}


def critic_forward_slabs(cw: CriticWeights, counts, time_rows, *, exact_chain=False):
    """counts (S, N, R) fp32 or uint8 contiguous = [frame][node][env] with R % 128 == 0 -> value (S*R,) in (frame, env)
    order; ``time_rows`` (S,) is each frame's clock."""
    L = _lib.load()
    u8 = counts.dtype == torch.uint8
    _contig(counts, torch.uint8 if u8 else torch.float32, "counts")
    _contig(time_rows, torch.float32, "time_rows")
    S, N, R = counts.shape
    if N != cw.N or R % 128 or time_rows.numel() < S:
        raise ValueError("counts must be (S, N, R) with R a multiple of 128 and one time per slab")
    value = torch.empty(S * R, dtype=torch.float32, device=counts.device)
    args = (counts.data_ptr(), R, S * R, N, time_rows.data_ptr(), R, cw.w1.data_ptr(), cw.b1.data_ptr(), cw.w2.data_ptr(),
            cw.b2.data_ptr(), cw.w3.data_ptr(), cw.b3.data_ptr())
    if u8:
        # count bytes: first layer on the bf16 matrix cores at fp32 accuracy (W1 as three exact bf16 pieces);
        # exact_chain=True keeps the k-ordered fp32 MFMA chain (bit-identical to the row-major kernel)
        scratch = None
        if not exact_chain:
            key = (str(counts.device), N)
            scratch = _SPLIT_SCRATCH.get(key)
            if scratch is None:
                scratch = torch.empty(int(L.tarl_critic_split_scratch_bytes(N)), dtype=torch.uint8, device=counts.device)
                _SPLIT_SCRATCH[key] = scratch
        _lib.check(L.tarl_critic_mlp_fwd_slabs_u8(*args, _lib.ptr(scratch), value.data_ptr(), _lib.current_stream()))
    else:
        _lib.check(L.tarl_critic_mlp_fwd_slabs(*args, value.data_ptr(), _lib.current_stream()))
    return value
