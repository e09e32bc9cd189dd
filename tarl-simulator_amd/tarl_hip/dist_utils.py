"""Data-parallel helpers over torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The hot path shards by *rollouts*: every rank owns its own environments and minibatch; the only exchange per optimiser
step is ONE all-reduce of the flat fp32 gradient buffer, plus a 3-double all-reduce of the advantage statistics so that
``average_gae`` normalises with the global mean / std. Parameters stay replicated (identical Adam on
every rank)."""
from __future__ import annotations

import os

# dmabuf IPC: what RCCL (and device-tensor sharing across processes) needs on hosts whose driver has no legacy IPC; must be in
# the environment before the HIP runtime starts, i.e. before torch is imported. A launcher's own setting wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend=None):
    """Initialise from RANK / WORLD_SIZE / MASTER_* (torchrun). Returns (rank, world_size, local_rank)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:   # TARL_DIST_BACKEND=gloo lets several ranks share one GPU for rehearsals (RCCL refuses that)
            backend = os.environ.get("TARL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        kw = {}
        if backend == "nccl":    # bind the communicator to this rank's GPU up front (no "device under current context")
            kw["device_id"] = torch.device("cuda", local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=ws, **kw)
    return rank, ws, local


def allreduce_sum_(t: torch.Tensor):
    """In-place sum over ranks (no-op for a single process)."""
    if world()[1] > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_max_float(v: float, device) -> float:
    if world()[1] == 1:
        return v
    t = torch.tensor([v], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(vals, device):
    """Every rank's list of floats -> [rank][k] on every rank (all_gather of one small fp64 tensor)."""
    t = torch.tensor([float(v) for v in vals], dtype=torch.float64, device=device)
    if world()[1] == 1:
        return [t.tolist()]
    out = [torch.empty_like(t) for _ in range(world()[1])]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]


def replica_max_abs_diff(t: torch.Tensor) -> float:
    """max over ranks and elements of |t - rank 0's t|: 0.0 exactly when every replica holds rank 0's bits (one broadcast
    of a copy + one MAX all-reduce; 0.0 for a single process). The data-parallel contract: replicated
    parameters, one averaged gradient, the same Adam step on every rank."""
    if world()[1] == 1:
        return 0.0
    ref = t.detach().clone()
    dist.broadcast(ref, src=0)
    d = (t.detach() - ref).abs().max().to(torch.float64).reshape(1)
    dist.all_reduce(d, op=dist.ReduceOp.MAX)
    return float(d.item())


def backend_name() -> str:
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else "none"


def broadcast_(t: torch.Tensor, src=0):
    if world()[1] > 1:
        dist.broadcast(t, src=src)
    return t


def barrier():
    if world()[1] > 1:
        dist.barrier()
