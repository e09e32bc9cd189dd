"""CPU, world_size 2, gloo: the data-parallel exchange of the path (one all-reduce of the flat gradient buffer + the
3-double advantage-statistics all-reduce) reproduces the single-process result."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from tarl_hip import dist_utils
    r, w, _ = dist_utils.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist_utils.world() == (rank, world)
    g = torch.Generator().manual_seed(100 + rank)
    # (1) gradient averaging: each rank's seeds are pre-scaled by 1/world, the all-reduce sums
    grad = torch.randn(1000, generator=g, dtype=torch.float32) / world
    dist_utils.allreduce_sum_(grad)
    # (2) global advantage statistics from per-rank {sum, sumsq, count}
    adv = torch.randn(257, generator=g, dtype=torch.float32) * (rank + 1) + rank
    stats = torch.tensor([adv.double().sum(), (adv.double() ** 2).sum(), adv.numel()], dtype=torch.float64)
    dist_utils.allreduce_sum_(stats)
    # (3) replicas start from rank 0's weights
    wts = torch.full((5,), float(rank))
    dist_utils.broadcast_(wts, src=0)
    mx = dist_utils.allreduce_max_float(1.0 + rank, "cpu")
    # (4) the replica check bench.py reports (replica_param_max_abs_diff): 0.0 exactly for identical replicas, the largest
    # deviation from rank 0 otherwise — the same number on every rank
    assert dist_utils.replica_max_abs_diff(wts) == 0.0
    assert dist_utils.replica_max_abs_diff(torch.full((7,), float(rank))) == float(world - 1)
    dist_utils.barrier()
    # numpy arrays pickle by value (torch tensors travel as file descriptors that die with this process)
    q.put((rank, grad.numpy(), stats.numpy(), adv.numpy(), wts.numpy(), mx))
    dist.destroy_process_group()


def test_eight_rank_exchange():
    """The 8-GPU job's exchange pattern (one rank per GPU), rehearsed on the CPU: gradient all-reduce of pre-scaled seeds =
    the mean over 8 ranks, identical on every rank; global advantage statistics; rank 0's weights everywhere; max-reduce."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = [(r, *(torch.from_numpy(a) for a in arrs), m) for r, *arrs, m in res]
    ref = torch.stack([torch.randn(1000, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]).mean(0)
    allv = torch.cat([r[3] for r in res]).double()
    for _, g, s_, _, w, m in res:
        assert torch.equal(g, res[0][1]) and torch.allclose(g, ref, atol=1e-6)
        assert torch.equal(s_, res[0][2]) and torch.equal(w, torch.zeros(5)) and m == float(world)
    s0 = res[0][2]
    cnt, mean = s0[2], s0[0] / s0[2]
    std = torch.sqrt((s0[1] - cnt * mean * mean) / (cnt - 1))
    assert int(cnt) == 257 * world and abs(mean - allv.mean()) < 1e-9 and abs(std - allv.std()) < 1e-9


def test_two_rank_exchange_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res = [(r, *(torch.from_numpy(a) for a in arrs), m) for r, *arrs, m in res]
    (_, g0, s0, a0, w0, m0), (_, g1, s1, a1, w1, m1) = res
    # identical on both ranks, equal to the mean of the per-rank gradients
    ref = torch.stack([torch.randn(1000, generator=torch.Generator().manual_seed(100 + r)) for r in range(2)]).mean(0)
    assert torch.equal(g0, g1) and torch.allclose(g0, ref, atol=1e-7)
    assert torch.equal(s0, s1) and torch.equal(w0, torch.zeros(5)) and torch.equal(w1, torch.zeros(5))
    assert m0 == m1 == 2.0
    # normalising with the combined statistics == normalising the concatenated advantages (torchrl average_gae)
    allv = torch.cat([a0, a1]).double()
    cnt, mean = s0[2], s0[0] / s0[2]
    std = torch.sqrt((s0[1] - cnt * mean * mean) / (cnt - 1))
    assert abs(mean - allv.mean()) < 1e-9 and abs(std - allv.std()) < 1e-9
