#!/usr/bin/env python3
"""bench.py — PPO env-steps/sec (+ MPNN message-pass edges/sec) of the MPNN+PPO routing hot path on N MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

A *step* is one PPO iteration of the reference's ``ppo_train`` (src/rl/ppo_trainer.py:129-145) for every environment
of a rank: a rollout of T frames (policy logits -> GraphDistribution sample/log_prob -> env step: choice,
DirectionMPNN, ResponseMPNN, withdraw, insert, reward) followed by ``epochs`` x (critic over all frames, GAE,
advantage normalisation, minibatch, clipped PPO loss, backward, gradient all-reduce, Adam). Nothing is skipped or cached.
Workload: BASELINE.json config 4 — 10k-edge synthetic torus dual graph, 16k agents, rollout-steps 256 — with ``--envs``
vectorised environments per GPU (weak scaling: per-GPU work is fixed). Inputs are resident in HBM before the timed
region. The timed region is bracketed by barrier + synchronize on both sides; the slowest rank's time is used.

Also reported on the same JSON line:
  roofline      — the Direction message+aggregate kernel (the scatter kernel named by the north star): algorithmic bytes
                  per launch / average launch duration measured live with HIP events on the launch stream (the T frames
                  of the first timed iteration), vs 8 TB/s; roofline_row_pass: the same for the row pass (Direction
                  update + Response + withdraw), the kernel with the largest share of a frame.
  cpu_baseline  — the oracle (CPU restatement of the reference path, torch CPU) timed on this host's cores on a bounded
                  sample of the same workload (rank 0, N=1 only). A reported baseline, not a target.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tarl-simulator_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy kernel achieves
# Direction message+aggregate, algorithmic bytes per edge (DESIGN.md "Kernels"): int32 src,eid (8) + edge_attr (4) +
# upstream id/arr/dep/max/n/ff/sel (28) + downstream max/n/road_index (12) = 52; + 4 when Gumbel noise is read from
# HBM (parity mode) + 4 when delta_travel_time is materialised. The bench draws noise in-kernel and skips dtt -> 52.
DIR_BYTES_PER_EDGE = 52.0
DIR_BYTES_PER_NODE = 4.0  # chosen[] out
# Row pass = DirectionMPNN.update (32 B/node: max, n, ff, cong in; id, arr, dep, n out) + ResponseMPNN message+aggregate
# (24 B/edge: indices 8 + upstream n, head 8 + downstream n, tail 8); the per-pop FIFO movement (344 B/pop in the
# reference's layout) is not counted (SURVEY 8d's per-unit figures).
ROWS_BYTES_PER_EDGE = 24.0
ROWS_BYTES_PER_NODE = 32.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--edges", type=int, default=10000)
    ap.add_argument("--agents", type=int, default=16384)
    ap.add_argument("--envs", type=int, default=2048, help="vectorised environments per GPU")
    ap.add_argument("--rollout-steps", type=int, default=256)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--sub-batch", type=int, default=32)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="do not bracket the Direction kernel with HIP events (lets small batches use the graph replay)")
    return ap.parse_args()


def build_trainer(args, rank, device):
    from tarl_hip import synth
    from tarl_hip.engine import SimEngine
    from tarl_hip.trainer import VecPPOTrainer
    from src.agents.mpnn_agent import MPNNPolicyNet, MPNNValueNetSimple

    W, H = synth.torus_for_edges(args.edges)
    net = synth.torus_network(W, H)
    N = net.num_roads
    # every environment of every rank gets its own population (seed + rank, env index)
    pops = torch.stack([synth.population(args.agents, N, seed=args.seed + 1000 * rank + b) for b in range(args.envs)])
    engine = SimEngine(net.x.to(device).unsqueeze(0).repeat(args.envs, 1, 1).contiguous(), net.edge_index,
                       net.edge_attr, net.Nmax, pops.to(device), congestion_constant=net.congestion_constant,
                       device=device, seed=args.seed + rank)
    torch.manual_seed(args.seed)       # identical initial weights on every rank (also broadcast by the trainer)
    ff = net.x[:, 3 * net.Nmax + 2][net.edge_index[1]]
    pol = MPNNPolicyNet(net.edge_index, N, ff, device=str(device))
    val = MPNNValueNetSimple(net.edge_index, N, device=str(device))
    l = val.final_mlp
    dormant = [p for n, p in pol.named_parameters() if not n.startswith("nodes_embedding")]
    trainer = VecPPOTrainer(engine, pol.nodes_embedding.weight,
                            [l[0].weight, l[0].bias, l[2].weight, l[2].bias, l[4].weight, l[4].bias],
                            rollout_steps=args.rollout_steps, num_epochs=args.epochs, sub_batch_size=args.sub_batch,
                            extra_params=dormant, seed=args.seed)
    return net, engine, trainer


def cpu_baseline(args, net):
    """Oracle rollout (policy logits -> GraphDistribution -> sample -> env step) of ONE environment at the bench's
    config size on the host CPU; bounded by --cpu-seconds."""
    from oracle import sim, dist, nets
    from tarl_hip import synth
    N, Nmax, E = net.num_roads, net.Nmax, net.edge_index.size(1)
    adj = net.dense_adjacency()
    w = torch.randn(N)

    def rollout(x, ag, t, budget_s, max_steps):
        steps, t0 = 0, time.perf_counter()
        while True:
            d = dist.GraphDist(nets.policy_logits(sim.observe(x, Nmax)[0], net.edge_index, w), net.edge_index)
            a = d.sample()
            d.log_prob(a)
            sim.env_step(x, ag, net.edge_index, net.edge_attr, adj, a, t, Nmax,
                         congestion_constant=net.congestion_constant)
            t += 1
            steps += 1
            el = time.perf_counter() - t0
            if el >= budget_s or steps >= max_steps:
                return steps, el

    # the oracle's ops are small: more threads can be slower. Probe a few thread counts, keep the fastest.
    ncpu = os.cpu_count() or 1
    best_nt, best_rate = 1, 0.0
    for nt in sorted({1, min(8, ncpu), min(16, ncpu)}):
        torch.set_num_threads(nt)
        rollout(net.x.clone(), synth.population(args.agents, N, seed=args.seed), 21540, 0.3, 2)       # warm up
        st, el = rollout(net.x.clone(), synth.population(args.agents, N, seed=args.seed), 21540, 1.0, 64)
        if st / el > best_rate:
            best_nt, best_rate = nt, st / el
    torch.set_num_threads(best_nt)
    steps, el = rollout(net.x.clone(), synth.population(args.agents, N, seed=args.seed), 21540, args.cpu_seconds, 8192)
    return {"value": steps / el, "unit": "env-steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} env steps of 1 environment (oracle rollout: policy logits, GraphDistribution sample + "
                      f"log_prob, env step) on the {E}-edge / {args.agents}-agent workload, {el:.1f} s, torch CPU"}


def main():
    args = parse()
    from tarl_hip import dist_utils, lib
    rank, world, local = dist_utils.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run for N>1)"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (the product path has no CPU fallback)"
    local = local % torch.cuda.device_count()      # (rehearsal: several ranks may share one GPU under gloo)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:     # N ranks share the host: keep each rank's CPU-side set-up (population generation) in its share
        torch.set_num_threads(max(1, (os.cpu_count() or 1) // world))
    L = lib.load()

    net, engine, trainer = build_trainer(args, rank, device)
    E, B, T = engine.E, engine.B, args.rollout_steps

    for _ in range(args.warmup):
        trainer.train_iteration()
    if not args.no_kernel_timing:
        L.tarl_prof_enable(T)      # HIP events around the two message-passing kernels of the first timed iteration's frames
    dist_utils.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames = 0
    for _ in range(args.steps):
        frames += trainer.train_iteration()
    torch.cuda.synchronize()
    dist_utils.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = dist_utils.allreduce_max_float(elapsed, device)

    k_ms, r_ms, k_n = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_int64(0)
    lib.check(L.tarl_prof_collect2(ctypes.byref(k_ms), ctypes.byref(r_ms), ctypes.byref(k_n)))
    L.tarl_prof_enable(0)

    if rank == 0:
        total_frames = frames * world
        value = total_frames / elapsed
        avg_s = (k_ms.value / max(1, k_n.value)) * 1e-3
        alg_bytes = DIR_BYTES_PER_EDGE * B * E + DIR_BYTES_PER_NODE * B * engine.N
        achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        # HBM traffic per launch from the PMC counters (2*FETCH_SIZE + WRITE_SIZE, collected in separate rocprofv3 --pmc
        # passes of the same kernels at the same sizes and committed under profiles/); null when no matching record.
        traffic, traffic_rows, traffic_src = None, None, None
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "r01_v5_pmc_traffic.json")))
            if rec["config"] == {"edges": E, "agents": args.agents, "envs": B}:
                traffic = rec["kernels"]["k_fused_direction"]["hbm_bytes_per_launch"]
                traffic_rows = rec["kernels"]["k_fused_rows"]["hbm_bytes_per_launch"]
                traffic_src = "profiles/r01_v5_pmc_traffic.json (rocprofv3 --pmc, 2*FETCH_SIZE + WRITE_SIZE)"
        except (OSError, KeyError, ValueError):
            pass
        rows_s = (r_ms.value / max(1, k_n.value)) * 1e-3
        rows_bytes = ROWS_BYTES_PER_EDGE * B * E + ROWS_BYTES_PER_NODE * B * engine.N
        rows_achieved = rows_bytes / rows_s / 1e9 if rows_s > 0 else 0.0
        out = {
            "metric": "ppo_env_steps_per_sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"mpnn+ppo train, {E}-edge synthetic torus dual graph ({engine.N} roads), "
                                   f"{args.agents} agents per environment, rollout-steps {T}, epochs {args.epochs}, "
                                   f"sub-batch {args.sub_batch}" +
                                   (" (BASELINE config 4)" if (E, args.agents) == (10000, 16384) else ""),
                       "envs_per_gpu": B, "env_steps_per_step": B * T, "parallelism": f"dp{world} (rollouts sharded, "
                       "one gradient all-reduce per optimiser step)"},
            "msgpass_edges_per_sec": value * E,
            # Direction + Response pair alone (SURVEY 8d's second metric): B*E edges per frame / the two kernels' live time
            "msgpass_pair_edges_per_sec": (B * E) / (avg_s + rows_s) if (avg_s + rows_s) > 0 else None,
            "roofline": {"bound": "hbm", "kernel": "k_fused_direction (DirectionMPNN message+aggregate on the packed hot records)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_us": avg_s * 1e6, "launches_timed": k_n.value},
            # the kernel with the largest share of the frame (35 %): Direction update + Response + withdraw
            "roofline_row_pass": {"bound": "hbm", "kernel": "k_fused_rows (DirectionMPNN.update + ResponseMPNN + withdraw)",
                                  "achieved": rows_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": rows_achieved / HBM_PEAK_GBS, "traffic": traffic_rows,
                                  "traffic_source": traffic_src, "algorithmic_bytes_per_launch": rows_bytes,
                                  "avg_launch_us": rows_s * 1e6, "launches_timed": k_n.value},
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args, net)
        print(json.dumps(out), flush=True)
    dist_utils.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
