"""GPU, 2 ranks sharing the one card over gloo (RCCL refuses two ranks per device; the collectives' arithmetic is the
same): SURVEY §8e's correctness statement — replicas fed IDENTICAL data reproduce the single-process gradient and
parameter update (to 1e-6), and with per-rank data the applied gradient is the mean of the ranks' gradients."""
import os

import pytest
import torch
import torch.multiprocessing as mp

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu


def _build(rank_offset, seed_shift=0):
    from src.agents.mpnn_agent import MPNNPolicyNet, MPNNValueNetSimple
    from tarl_hip import synth
    from tarl_hip.engine import SimEngine
    from tarl_hip.trainer import VecPPOTrainer
    net = synth.torus_network(4, 4, heterogeneous=True, seed=2)
    N = net.num_roads
    B, A, T, M = 128, 300, 16, 16
    pops = torch.stack([synth.population(A, N, seed=b + 1000 * seed_shift, t0=21540, t1=21550) for b in range(B)])
    eng = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                    pops.cuda(), congestion_constant=net.congestion_constant, seed=3 + seed_shift)
    torch.manual_seed(0)
    pol = MPNNPolicyNet(net.edge_index, N, None, device="cuda")
    val = MPNNValueNetSimple(net.edge_index, N, device="cuda")
    l = val.final_mlp
    tr = VecPPOTrainer(eng, pol.nodes_embedding.weight, [l[0].weight, l[0].bias, l[2].weight, l[2].bias, l[4].weight, l[4].bias],
                       rollout_steps=T, num_epochs=1, sub_batch_size=M, seed=5, rank_offset=rank_offset)
    tr.keep_grad = True
    return tr


def _one_iteration(tr):
    tr.collect()
    adv, tgt = tr.advantages()
    idx = torch.randperm(tr.T * tr.eng.B, generator=torch.Generator().manual_seed(4))[:tr.M]
    tr.minibatch_step(adv, tgt, idx=idx)
    return tr.last_grad.cpu(), tr.flat.flat.detach().cpu().clone(), adv.cpu()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _worker(rank, world, port, identical, q):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", TARL_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from tarl_hip import dist_utils
    dist_utils.init_from_env()
    tr = _build(rank_offset=not identical, seed_shift=0 if identical else rank)
    g, w, adv = _one_iteration(tr)
    q.put((rank, g.numpy(), w.numpy(), adv.numpy()))   # by value: the worker may exit before the parent unpickles
    dist_utils.barrier()
    dist.destroy_process_group()


def _run_two(identical):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, identical, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return [(r, torch.from_numpy(g), torch.from_numpy(w), torch.from_numpy(a)) for r, g, w, a in res]


def test_identical_replicas_reproduce_single_process_update():
    assert torch.cuda.is_available()
    (_, g0, w0, a0), (_, g1, w1, a1) = _run_two(identical=True)
    g, w, a = _one_iteration(_build(rank_offset=False))
    assert torch.equal(g0, g1) and torch.equal(w0, w1) and torch.equal(a0, a1)
    # The only legitimate difference: torchrl's average_gae divides by the UNBIASED std, and duplicating the n frames
    # turns SS/(n-1) into 2SS/(2n-1): the normalised advantages grow by sqrt((2n-1)/(2n-2)) (1.2e-4 at n = 2048).
    n = a.numel()
    f = ((2 * n - 1) / (2 * n - 2)) ** 0.5
    assert float((a0 - a * f).abs().max()) <= 1e-5
    # critic part of the flat gradient (value targets are not normalised): the single-process gradient to 1e-6;
    # policy part (linear in the advantages while nothing is clipped): to that factor
    N = 64
    gc, gc0 = g[N:], g0[N:]
    assert float(gc.abs().max()) > 0 and float((gc0 - gc).abs().max()) <= 1e-6 * max(1.0, float(gc.abs().max()))
    assert float((g0[:N] - g[:N]).abs().max()) <= 5e-4 * max(1e-3, float(g[:N].abs().max()))
    assert float((w0 - w).abs().max()) <= 1e-5


def test_ranks_with_their_own_rollouts_average_gradients():
    (_, g0, w0, a0), (_, g1, w1, a1) = _run_two(identical=False)
    assert torch.equal(g0, g1) and torch.equal(w0, w1)          # one all-reduced gradient, identical Adam step
    assert not torch.equal(a0, a1)                               # different rollouts, normalised with GLOBAL statistics
    both = torch.cat([a0.view(-1), a1.view(-1)]).double()
    assert abs(float(both.mean())) < 1e-4 and abs(float(both.std()) - 1.0) < 1e-3


def _main_worker(rank, world, port, out_dir, q):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", TARL_DIST_BACKEND="gloo")
    os.chdir(out_dir)
    import importlib
    main = importlib.import_module("main").main
    from src.rl.ppo_trainer import ppo_train
    main(["--algo", "mpnn+ppo", "--mode", "train", "--scenario", "synthetic-1024-1024", "--rollout-steps", "32",
          "--epochs", "2", "--steps", "5", "--num-envs", "4", "--output-dir", os.path.join(out_dir, "run"), "--seed", "2"])
    tr = ppo_train.last_trainer
    import torch.distributed as dist
    # numpy arrays pickle by value (torch tensors travel as file descriptors that die with this process)
    q.put((rank, tr.world, tr.flat.flat.detach().cpu().numpy(), tr.reward.cpu().numpy(), tr.choice.cpu().numpy(),
           dist.is_initialized()))


def test_main_under_torchrun_trains_data_parallel(tmp_path):
    """`torchrun --nproc-per-node 2 main.py --algo mpnn+ppo --mode train` (2 ranks, here sharing the card over gloo): the
    ranks join the process group, roll out DIFFERENT trajectories, apply the same averaged update (identical weights), only
    rank 0 writes the checkpoint / logs, and the group is torn down at exit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_main_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, w0, f0, r0, c0, init0), (_, w1, f1, r1, c1, init1) = res
    assert w0 == w1 == 2 and not init0 and not init1            # joined, and left again in main()'s finally
    assert (f0 == f1).all()                                        # replicas stay identical
    assert (c0 != c1).any()                                        # but every rank rolled out its own trajectories
    import json
    logs = [json.loads(l) for l in open(tmp_path / "run" / "train_log.jsonl")]
    assert len(logs) == 1                                          # one writer
    assert (tmp_path / "run" / "policy.pt").exists()
