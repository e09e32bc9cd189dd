#!/usr/bin/env python3
"""bench.py — PPO env-steps/sec (+ MPNN message-pass edges/sec) of the MPNN+PPO routing hot path on N MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

A *step* is one PPO iteration of the reference's ``ppo_train`` (src/rl/ppo_trainer.py:129-145) for every environment
of a rank: a rollout of T frames (policy logits -> GraphDistribution sample/log_prob -> env step: choice,
DirectionMPNN, ResponseMPNN, withdraw, insert, reward) followed by ``epochs`` x (critic over all frames, GAE,
advantage normalisation, minibatch, clipped PPO loss, backward, gradient all-reduce, Adam).
What is hoisted: the live policy's logits are state-independent (embedding of the target road), so its segment softmax,
inverse-CDF thresholds, log-probabilities and entropy are evaluated once per parameter update, not per frame (bit-identical
to evaluating them per frame; ``--policy edge_mlp`` runs the state-dependent head with per-frame logits instead).
Workload: BASELINE.json config 4 — 10k-edge synthetic torus dual graph, 16k agents, rollout-steps 256 — with ``--envs``
vectorised environments per GPU (weak scaling: per-GPU work is fixed). Inputs are resident in HBM before the timed
region. The timed region is bracketed by barrier + synchronize on both sides; the slowest rank's time is used.

Also reported on the same JSON line:
  roofline            — the DOMINANT kernel of a frame, k_fused_rows (DirectionMPNN.update + ResponseMPNN message /
                        aggregate / update + withdraw): HBM bytes per launch from the PMC counters (2*FETCH_SIZE +
                        WRITE_SIZE, rocprofv3 --pmc passes of THIS script reduced by tools/pmc_bench.py over frames >= 200
                        of an iteration and committed under profiles/) / the kernel's average launch duration over the
                        same frames, measured live with HIP events on the launch stream, vs the 8 TB/s HBM peak.
                        ``compulsory_*``: the same with the bytes the packed layout must move instead of
                        the counter bytes; ``survey_8d_*``: SURVEY §8d's per-edge figure for the reference's AoS layout,
                        kept for continuity only (it charges every record once per out-edge: not a fraction of peak).
  roofline_direction  — the same for k_fused_direction (DirectionMPNN.message + aggregate, the scatter kernel the
                        north star names); roofline_insert — the insert launch (latency-bound).
  state_dependent_policy — the same iteration with the per-edge MLP head (nothing hoisted): "fp32" = rollout logits at fp32
                        accuracy on the bf16 matrix pipe (operands split into exact bf16 pieces), "fp32_mfma" = on the fp32
                        matrix pipe (exact fp32 products), "bf16" = bf16 logits on bf16 observations; each with an MFMA
                        roofline object for the MLP launch (live HIP-event time).
  cpu_baseline        — the oracle (CPU restatement of the reference path, torch CPU) timed on this host's cores on a
                        bounded sample of the same workload (rank 0, N=1 only). A reported baseline, not a target.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

# before torch is imported (the HIP runtime reads it once): dmabuf IPC, which RCCL needs on hosts without legacy IPC — also
# when the ranks come from an external `torchrun bench.py` (the driver's launch line) whose environment lacks it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tarl-simulator_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # dense MFMA peaks (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy kernel achieves
LATE_FRAME = 200          # the roofline window: frames >= 200 of an iteration (traffic and live time alike)
# written by tools/pmc_bench.py from --pmc passes of this script (one record per workload: the default line and the
# congested regime), with the per-kernel counter corrections measured by tools/pmc_cal.hip (profiles/r03_pmc_calibration.txt)
PMC_RECORDS = (os.path.join("profiles", "r04_pmc_traffic.json"), os.path.join("profiles", "r04_pmc_traffic_congested.json"))
# Compulsory HBM bytes per (road, environment) and launch of the packed env-minor layout: what each kernel
# must read and write once, neighbour gathers served by the XCD's L2, statics / topology / policy tables through the
# scalar cache (shared by all environments, not counted).
COMPULSORY = {
    # head words 8 + tail word 4 + SELECTED_ROAD byte 1 in; post word 4 out
    "k_fused_direction": {"per_node_env": 13.0 + 4.0},
    # post word 4 + head words 8 + tail word 4 in; count byte 1 out (an idle row's words already hold what a refresh would
    # store: head / tail words, event word, slot store and agent rows are written only where something moves)
    "k_fused_rows": {"per_node_env": 16.0 + 1.0},
    # insert(t): departure window, a few words per admitted agent, the accumulator banks (the actions of all frames are
    # drawn on a side stream: k_fused_choice_all, 1 byte per (frame, road, environment))
    "k_fused_insert": {"per_node_env": 0.0},
}
# SURVEY §8d's per-unit figures for the reference's AoS layout (kept as ``survey_8d_*`` keys only): Direction message +
# aggregate 60 B/edge (8 indices + 4 edge_attr + 4 noise + 28 x_j + 12 x_i + 4 delta_tt out); row pass = Direction update
# 32 B/node + Response message/aggregate 24 B/edge (its 344 B per popped road not counted).
SURVEY_8D = {"k_fused_direction": (60.0, 0.0), "k_fused_rows": (24.0, 32.0)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--edges", type=int, default=10000)
    ap.add_argument("--agents", type=int, default=16384)
    ap.add_argument("--envs", type=int, default=16384, help="vectorised environments per GPU")
    ap.add_argument("--rollout-steps", type=int, default=256)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--sub-batch", type=int, default=32)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--departure-window", type=int, default=0,
                    help="seconds after the episode start within which ALL agents depart (0 = the whole 61-minute episode, "
                         "BASELINE's workload; e.g. 600 = the congested regime: the network fills up inside the rollout)")
    ap.add_argument("--congested-window", type=int, default=600,
                    help="also time --congested-steps iterations with every agent departing within this many seconds "
                         "(0 = skip); reported under congested_regime")
    ap.add_argument("--congested-steps", type=int, default=2)
    ap.add_argument("--policy-envs", type=int, default=2048,
                    help="environments per GPU of the state-dependent-policy line (policy_head=edge_mlp; 0 = skip)")
    ap.add_argument("--policy-steps", type=int, default=2)
    ap.add_argument("--policy-temperature", type=float, default=2000.0,
                    help="GraphDistribution temperature of the state-dependent-policy line: the head reads raw features "
                         "(clock times ~2e4), so an untrained head at temperature 1 is near-deterministic, drives every "
                         "agent down the same turn and gridlocks the network out of the reference's domain")
    ap.add_argument("--metrics-envs", type=int, default=1,
                    help="environments that keep the per-node logs of SimulatorEnv._step (delta_travel_time, pop / withdraw "
                         "masks); the per-frame leg histogram is kept for all of them")
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
    # every environment of every rank gets its own population, drawn on the device (seed + rank)
    pops = synth.population_batch(args.agents, N, args.envs, seed=args.seed + 1000 * rank, device=device,
                                  t1=args.departure_window + synth.EPISODE_START if args.departure_window else synth.EPISODE_END)
    engine = SimEngine(net.x.to(device).unsqueeze(0).repeat(args.envs, 1, 1).contiguous(), net.edge_index,
                       net.edge_attr, net.Nmax, pops, congestion_constant=net.congestion_constant,
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
                            extra_params=dormant, seed=args.seed, metrics_envs=args.metrics_envs)
    return net, engine, trainer


def physical_cores_one_socket():
    """Physical cores of socket 0 (unique core ids under physical id 0 in /proc/cpuinfo); the logical count if unknown."""
    try:
        cores, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = int(line.split(":")[1])
            elif line.startswith("core id"):
                core = int(line.split(":")[1])
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        first = min(p for p, _ in cores)
        n = len([1 for p, _ in cores if p == first])
        if n > 0:
            return min(n, os.cpu_count() or n)
    except (OSError, ValueError):
        pass
    return os.cpu_count() or 1


def cpu_baseline(args, net):
    """The oracle (CPU restatement of the reference path, torch CPU) on ONE environment at the bench's config size, on the
    host's cores, bounded by --cpu-seconds: a rollout (policy logits -> GraphDistribution sample + log_prob -> env step)
    that keeps what the collector keeps, then ONE PPO update on those frames as the reference's loop runs it
    (src/rl/ppo_trainer.py:129-145: critic over all frames, GAE, a minibatch of --sub-batch frames, ClipPPOLoss, backward,
    Adam) — ``value`` = frames / (rollout + update), like the GPU ``value``; ``value_rollout_only`` beside it."""
    from oracle import sim, dist, nets, ppo
    from tarl_hip import synth
    N, Nmax, E = net.num_roads, net.Nmax, net.edge_index.size(1)
    adj = net.dense_adjacency()
    gen = torch.Generator().manual_seed(args.seed)
    w = torch.randn(N, generator=gen)

    def rollout(x, ag, t, budget_s, max_steps, keep=None):
        steps, t0 = 0, time.perf_counter()
        while True:
            nf = sim.observe(x, Nmax)[0]
            d = dist.GraphDist(nets.policy_logits(nf, net.edge_index, w), net.edge_index)
            a = d.sample()
            lp = d.log_prob(a)
            if keep is not None:
                keep["counts"].append(nf[..., 1].reshape(-1).clone())
                keep["action"].append(a.reshape(-1).clone())
                keep["logp"].append(lp.reshape(-1)[0].clone())
                keep["time"].append(float(t))
            out = sim.env_step(x, ag, net.edge_index, net.edge_attr, adj, a, t, Nmax,
                               congestion_constant=net.congestion_constant)
            if keep is not None:
                keep["reward"].append(-float(x[:, 3 * Nmax + 1].sum()))     # reward = -(agents in the network), as _step's
            del out
            t += 1
            steps += 1
            el = time.perf_counter() - t0
            if el >= budget_s or steps >= max_steps:
                if keep is not None:
                    keep["counts"].append(sim.observe(x, Nmax)[0][..., 1].reshape(-1).clone())
                    keep["time"].append(float(t))
                return steps, el

    def update(keep):
        """One PPO update on the kept frames: oracle/ppo.py + autograd, the reference's optimiser step."""
        t0 = time.perf_counter()
        T = len(keep["reward"])
        counts = torch.stack(keep["counts"])                               # (T + 1, N)
        times = torch.tensor(keep["time"], dtype=torch.float32)
        nf_all = torch.zeros((T + 1, 1, N, 7))
        nf_all[:, 0, :, 1] = counts
        nf_all[:, 0, :, 6] = torch.arange(N, dtype=torch.float32)
        g2 = torch.Generator().manual_seed(args.seed + 1)
        cw = [torch.randn(s_, generator=g2) * 0.05 for s_ in ((64, N + 1), (64,), (64, 64), (64,), (1, 64), (1,))]
        emb = w.clone().requires_grad_(True)
        cw = [c.requires_grad_(True) for c in cw]
        with torch.no_grad():
            v_all = nets.critic_value(nf_all, times.view(T + 1, 1, 1), *cw).reshape(T + 1, 1)
            reward = torch.tensor(keep["reward"], dtype=torch.float32).view(T, 1)
            z = torch.zeros((T, 1))
            adv, tgt = ppo.gae(reward, v_all[:T], v_all[1:], z, z, average_gae=T > 1)
        M = min(args.sub_batch, T)
        idx = torch.randperm(T, generator=g2)[:M]
        onehot = torch.stack(keep["action"])[idx]
        nf_mb = nf_all[idx, 0]
        d = dist.GraphDist(nets.policy_logits(nf_mb, net.edge_index, emb), net.edge_index)
        lp_new, ent = d.log_prob(onehot), d.entropy()
        value = nets.critic_value(nf_mb, times[idx].view(M, 1), *cw).reshape(-1)
        losses = ppo.clip_ppo_loss(lp_new, torch.stack(keep["logp"])[idx], adv.view(-1)[idx], value, tgt.view(-1)[idx], ent)
        (losses["loss_objective"] + losses["loss_critic"] + losses["loss_entropy"]).backward()
        for p in [emb] + cw:
            ppo.adam_step(p.data, p.grad, torch.zeros_like(p), torch.zeros_like(p), 1)
        return time.perf_counter() - t0

    # the oracle's ops are small: more threads can be slower. Probe a few thread counts — up to the PHYSICAL cores of one
    # socket (beyond that the probe measures oversubscription of a few-thousand-element ops, not the machine) — report each,
    # time the long sample with the fastest.
    ncpu, nsock = os.cpu_count() or 1, physical_cores_one_socket()
    best_nt, best_rate, by_threads = 1, 0.0, {}
    for nt in sorted({1, min(8, nsock), min(16, nsock), nsock}):
        torch.set_num_threads(nt)
        many = nt > 16
        rollout(net.x.clone(), synth.population(args.agents, N, seed=args.seed), 21540, 0.3, 1 if many else 2)       # warm up
        st, el = rollout(net.x.clone(), synth.population(args.agents, N, seed=args.seed), 21540, 1.0, 2 if many else 64)
        by_threads[str(nt)] = st / el
        if st / el > best_rate:
            best_nt, best_rate = nt, st / el
    torch.set_num_threads(best_nt)
    # PPO iterations as the reference runs them (reset, collect rollout_steps frames, one update) until the budget is spent
    steps, el, up, iters, t_all = 0, 0.0, 0.0, 0, time.perf_counter()
    while True:
        keep = {"counts": [], "action": [], "logp": [], "time": [], "reward": []}
        left = max(0.05, args.cpu_seconds - (time.perf_counter() - t_all))
        st_i, el_i = rollout(net.x.clone(), synth.population(args.agents, N, seed=args.seed + iters), 21540, left,
                             args.rollout_steps, keep)
        up += update(keep)
        steps, el, iters = steps + st_i, el + el_i, iters + 1
        if time.perf_counter() - t_all >= args.cpu_seconds:
            break
    return {"value": steps / (el + up), "unit": "env-steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "value_rollout_only": steps / el, "update_seconds": up, "rollout_seconds": el,
            "all_cores": {"cores": nsock, "value": by_threads[str(nsock)], "unit": "env-steps/s",
                          "logical_cpus_of_the_host": ncpu,
                          "sample": "1 s probe of the same rollout with torch.set_num_threads(physical cores of one socket)"},
            "by_threads_1s_probe": by_threads,
            "sample": f"{iters} PPO iteration(s) of 1 environment: {steps} env steps (oracle rollout: policy logits, "
                      f"GraphDistribution sample + log_prob, env step; at most {args.rollout_steps} per iteration) + one update "
                      f"per iteration (critic over all frames, GAE, minibatch of {args.sub_batch}, ClipPPOLoss, autograd "
                      f"backward, Adam) on the {E}-edge / {args.agents}-agent workload, {el:.1f} s + {up:.2f} s, torch CPU"}


def spawn_ranks(args):
    """``python bench.py --gpus N`` without a launcher: start the N ranks ourselves. This parent never touches the GPU (no
    HIP call, no torch.cuda query): it starts ``python -m torch.distributed.run --nproc-per-node N bench.py <same flags>``
    as a CHILD process (no exec of a process that has initialised the GPU anywhere), relays rank 0's JSON line to stdout
    and everything else to stderr, and exits with the children's return code (non-zero if any rank failed)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this pool
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env)
    lines = 0
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            lines += 1
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {lines}", file=sys.stderr)
        rc = 1
    sys.exit(rc)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)              # does not return
    from tarl_hip import dist_utils, lib
    rank, world, local = dist_utils.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (the product path has no CPU fallback)"
    local = local % torch.cuda.device_count()      # (rehearsal: several ranks may share one GPU under gloo)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:     # N ranks share the host: keep each rank's CPU-side set-up (population generation) in its share
        torch.set_num_threads(max(1, (os.cpu_count() or 1) // world))
    L = lib.load()

    t_setup = time.perf_counter()
    net, engine, trainer = build_trainer(args, rank, device)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_setup
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
    elapsed_rank = time.perf_counter() - t0
    elapsed = dist_utils.allreduce_max_float(elapsed_rank, device)
    per_rank = dist_utils.gather_floats([setup_s, elapsed_rank], device)     # [rank][setup_seconds, timed_seconds]
    # data-parallel evidence (outside the timed region): after `steps` averaged-gradient Adam steps every replica must still
    # hold rank 0's parameter bits
    replica_diff = dist_utils.replica_max_abs_diff(trainer.flat.flat)

    ms_all, ms_late, nfr = (ctypes.c_double * 3)(), (ctypes.c_double * 3)(), (ctypes.c_int64 * 2)()
    late0 = LATE_FRAME if T > LATE_FRAME else 0
    lib.check(L.tarl_prof_collect(late0, ms_all, ms_late, nfr))
    L.tarl_prof_enable(0)
    trainer.check_flags()

    # rollout-only figure (BASELINE.md §3 / SURVEY §8d: "rollout + update, and rollout-only"): the collector loop alone, timed
    # the same way right behind the headline region
    dist_utils.barrier()
    torch.cuda.synchronize()
    t_ro = time.perf_counter()
    ro_frames = 0
    for _ in range(min(args.steps, 3)):
        ro_frames += trainer.collect()
    torch.cuda.synchronize()
    dist_utils.barrier()
    ro_elapsed = dist_utils.allreduce_max_float(time.perf_counter() - t_ro, device)
    trainer.check_flags()

    # ---- second line of evidence: the congested regime -------------------------------------------------------------------
    # The headline workload spreads the departures over the 61-minute episode (BASELINE config), so the 256 timed frames see
    # a filling network. Here every agent departs within --congested-window seconds: the FIFOs fill up, most rows pop /
    # withdraw / enqueue in every frame. Same engine, same kernels, populations re-drawn and re-packed.
    congested = None
    if args.congested_window > 0 and not args.departure_window:
        from tarl_hip import synth
        engine.agents.copy_(synth.population_batch(args.agents, engine.N, B, seed=args.seed + 1000 * rank + 17,
                                                   device=device, t1=synth.EPISODE_START + args.congested_window))
        engine.fs.order_valid = False
        engine._packed_stale = True
        trainer.train_iteration()                       # warm-up (re-pack, re-sort)
        if not args.no_kernel_timing:
            L.tarl_prof_enable(T)
        dist_utils.barrier()
        torch.cuda.synchronize()
        t1_ = time.perf_counter()
        cf = 0
        for _ in range(args.congested_steps):
            cf += trainer.train_iteration()
        torch.cuda.synchronize()
        dist_utils.barrier()
        cel = dist_utils.allreduce_max_float(time.perf_counter() - t1_, device)
        c_all, c_late, c_nfr = (ctypes.c_double * 3)(), (ctypes.c_double * 3)(), (ctypes.c_int64 * 2)()
        lib.check(L.tarl_prof_collect(late0, c_all, c_late, c_nfr))
        L.tarl_prof_enable(0)
        trainer.check_flags()
        on_way = float(engine.agents[:, :, 7].sum()) / B
        arrived = float(engine.agents[:, :, 8].sum()) / B
        congested = {"value": cf * world / cel, "unit": "env-steps/s", "steps": args.congested_steps,
                     "ms_per_step": cel / args.congested_steps * 1e3, "departure_window_s": args.congested_window,
                     "agents_on_the_way_at_the_end_per_env": on_way, "agents_arrived_per_env": arrived,
                     "note": "all agents depart within the window: the network is loaded for most of the rollout"}

    # ---- third line of evidence: the state-dependent policy ---------------------------------------------------------------
    # The per-edge MLP head (33 -> 64 -> 32 -> 1 on cat(x[src], x[dst], edge_attr); the reference keeps it as parameters,
    # src/agents/mpnn_agent.py:35-41,227-231) reads the dynamic state, so NOTHING is hoisted: every frame builds the
    # observation from the packed state, runs the MLP on the matrix cores (fp32 and bf16 variants), the segment softmax,
    # the sample and the log-prob, then the four-launch simulation frame; the update runs the MLP forward / backward.
    policy_lines = None
    layout_tag, rollout_mode, n_roads = trainer.layout_tag, trainer.rollout, engine.N
    if args.policy_envs > 0:
        from tarl_hip import synth
        from tarl_hip.engine import SimEngine
        from tarl_hip.trainer import VecPPOTrainer
        from src.agents.mpnn_agent import MPNNPolicyNet, MPNNValueNetSimple
        Bp = args.policy_envs
        del trainer, engine
        torch.cuda.empty_cache()
        pops = synth.population_batch(args.agents, net.num_roads, Bp, seed=args.seed + 1000 * rank + 31, device=device)
        eng_p = SimEngine(net.x.to(device).unsqueeze(0).repeat(Bp, 1, 1).contiguous(), net.edge_index, net.edge_attr,
                          net.Nmax, pops, congestion_constant=net.congestion_constant, device=device,
                          seed=args.seed + rank)
        policy_lines = {}
        # "fp32": rollout logits at fp32 accuracy (the north star's 1e-4 contract) on the bf16 pipe — operands split into
        # exact bf16 pieces, k_edge_mlp_fwd_x3; "fp32_mfma": the same contract on the fp32 matrix pipe (exact fp32 products,
        # round 3's "fp32" line); "bf16": bf16 logits on bf16 observations (BASELINE config 5's "bf16 MPNN features")
        for tag, prec in (("fp32", "x3"), ("fp32_mfma", "fp32"), ("bf16", "bf16")):
            bf = prec == "bf16"
            torch.manual_seed(args.seed)
            pol = MPNNPolicyNet(net.edge_index, net.num_roads, None, device=str(device))
            val = MPNNValueNetSimple(net.edge_index, net.num_roads, device=str(device))
            l, mm = val.final_mlp, pol.edge_mlp
            tr_p = VecPPOTrainer(eng_p, pol.nodes_embedding.weight,
                                 [l[0].weight, l[0].bias, l[2].weight, l[2].bias, l[4].weight, l[4].bias],
                                 rollout_steps=T, num_epochs=args.epochs, sub_batch_size=args.sub_batch,
                                 extra_params=[p for n, p in pol.named_parameters() if not n.startswith("nodes_embedding")],
                                 seed=args.seed, policy="edge_mlp", policy_precision=prec, temperature=args.policy_temperature,
                                 edge_mlp_params=[mm[0].weight, mm[0].bias, mm[2].weight, mm[2].bias, mm[4].weight,
                                                  mm[4].bias])
            tr_p.train_iteration()
            if not args.no_kernel_timing:
                L.tarl_prof_enable(T)      # HIP events around the per-edge MLP launch of the first timed iteration's frames
            dist_utils.barrier()
            torch.cuda.synchronize()
            t2_ = time.perf_counter()
            pf = 0
            for _ in range(args.policy_steps):
                pf += tr_p.train_iteration()
            torch.cuda.synchronize()
            dist_utils.barrier()
            pel = dist_utils.allreduce_max_float(time.perf_counter() - t2_, device)
            p_all, p_late, p_nfr = (ctypes.c_double * 3)(), (ctypes.c_double * 3)(), (ctypes.c_int64 * 2)()
            lib.check(L.tarl_prof_collect(0, p_all, p_late, p_nfr))
            L.tarl_prof_enable(0)
            tr_p.check_flags()
            mlp_s = p_all[0] / max(1, p_nfr[0]) * 1e-3
            # MPNNPolicyNet.edge_mlp per edge: 2 * (33*64 + 64*32 + 32) flop (src/agents/mpnn_agent.py:35-41); the matrix
            # cores are the bound of this kernel: dense MFMA peaks from MI355X_MICROARCH.md (bf16 2.5 PFLOP/s, fp32 157.3 TF)
            flops = 2.0 * (33 * 64 + 64 * 32 + 32) * Bp * E
            # x3: the kernel ISSUES six bf16 piece products per multiply-add of the head (hi hi, hi mid, mid hi, hi lo, mid mid,
            # lo hi; the edge_attr / bias k-step once): achieved / frac are the HEAD's flops over the bf16 peak,
            # issued_* the piece products the matrix cores actually execute
            peak = MFMA_PEAK_TFLOPS["f32" if prec == "fp32" else "bf16"]
            kname = {"x3": "k_edge_mlp_fwd_x3", "fp32": "k_edge_mlp_fwd_f32", "bf16": "k_edge_mlp_fwd_bf16"}[prec]
            mlp_roof = ({"bound": "mfma", "kernel": kname + " (per-edge MLP 33->64->32->1)",
                         "achieved": flops / mlp_s / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": flops / mlp_s / 1e12 / peak,
                         "traffic": None, "avg_launch_us": mlp_s * 1e6, "launches_timed": int(p_nfr[0]),
                         "flop_per_launch": flops} if mlp_s > 0 else None)
            if mlp_roof and prec == "x3":
                issued = 2.0 * 32 * 32 * 16 * 50 * (Bp * E / 32.0)      # 50 MFMAs of 32x32x16 per 32 edges
                mlp_roof["issued_tflops"] = issued / mlp_s / 1e12
                mlp_roof["issued_frac"] = issued / mlp_s / 1e12 / peak
            policy_lines[tag] = {"value": pf * world / pel, "unit": "env-steps/s", "envs_per_gpu": Bp, "roofline": mlp_roof,
                                 "steps": args.policy_steps, "ms_per_step": pel / args.policy_steps * 1e3,
                                 "edge_mlp_edges_per_sec": pf * world / pel * E,
                                 "rollout_logits": {"bf16": "bf16 MFMA (v_mfma_f32_32x32x16_bf16) on bf16 observations",
                                                    "fp32": "fp32 MFMA (v_mfma_f32_32x32x2_f32), exact fp32 products",
                                                    "x3": "fp32-accurate on the bf16 pipe: operands in three exact bf16 "
                                                          "pieces, six products per multiply-add (v_mfma_f32_32x32x16_bf16)"}[prec]}
            del tr_p
        policy_lines["note"] = ("policy_head=edge_mlp: per-frame observation + 33->64->32->1 MLP per edge + GraphDistribution "
                                "softmax / sample / log_prob (no table hoist), then the simulation frame; "
                                f"GraphDistribution temperature {args.policy_temperature:g}")

    if rank == 0:
        total_frames = frames * world
        value = total_frames / elapsed
        NB = B * n_roads

        def pmc_record(window):
            """The committed PMC record of this workload, or None: the config must match AND the record must have been taken
            on the very frame-kernel sources this run executes (sha256 of csrc/fused.hip + fused_common.h) — a kernel edit
            without a fresh PMC pass falls back to the compulsory bytes instead of pairing new times with old bytes."""
            from tarl_hip.ops import frame_kernel_source_hash
            sha = frame_kernel_source_hash()
            want = {"edges": E, "agents": args.agents, "envs": B, "rollout_steps": T}
            if window:
                want["departure_window"] = window
            for path in PMC_RECORDS:
                try:
                    rec = json.load(open(os.path.join(ROOT, path)))
                    if rec["config"] == want and rec.get("source_sha16") == sha:
                        rec["path"] = path
                        return rec
                except (OSError, KeyError, ValueError):
                    pass
            return None

        def roofline(pmc, ms_late_, ms_all_, nfr_, slot, kernel, what):
            """achieved = HBM bytes per launch (PMC counters of this script's own rollout, frames >= LATE_FRAME, corrected
            per kernel with the factors tools/pmc_cal.hip measured for its access shapes) / the live average launch
            duration over the same frames; without a matching PMC record: the compulsory bytes."""
            n_late, n_all = max(1, nfr_[1]), max(1, nfr_[0])
            late_s, all_s = ms_late_[slot] / n_late * 1e-3, ms_all_[slot] / n_all * 1e-3
            comp = COMPULSORY[kernel]["per_node_env"] * NB
            k = pmc["kernels"].get(kernel) if pmc else None
            traffic = k["hbm_bytes_per_launch"] if k else None
            byts = traffic if traffic is not None else comp
            achieved = byts / late_s / 1e9 if late_s > 0 else 0.0
            out = {"bound": "hbm", "kernel": f"{kernel} ({what})", "achieved": achieved, "peak": HBM_PEAK_GBS,
                   "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                   "traffic_source": (f"{pmc['path']}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, "
                                      f"mean over frames >= {LATE_FRAME}; bytes = {k.get('formula', '2*FETCH_SIZE + WRITE_SIZE')}"
                                      ) if k else None,
                   "bytes_basis": "pmc_counters" if traffic is not None else "compulsory",
                   "avg_launch_us": late_s * 1e6, "frames_timed": int(nfr_[1]), "first_frame": late0,
                   "avg_launch_us_all_frames": all_s * 1e6,
                   "compulsory_bytes_per_launch": comp,
                   "compulsory_frac": (comp / late_s / 1e9 / HBM_PEAK_GBS) if late_s > 0 else 0.0,
                   "traffic_over_compulsory": (traffic / comp) if (traffic is not None and comp > 0) else None}
            if traffic is not None:
                # frac = traffic / time / peak moves with BOTH terms: round 3 cut the row pass's traffic (1 289 -> 889 MB) by
                # more than its time (245 -> 205 us), so the fraction fell while the kernel got faster; compulsory_frac and
                # traffic_over_compulsory separate the two
                out["note"] = ("frac falls when re-reads are removed faster than time: compare avg_launch_us and "
                               "traffic_over_compulsory across rounds, not frac alone")
            if k and "raw_frac_bracket" in k:      # the same fraction with no correction / the guide's blanket 2x on FETCH_SIZE
                out["frac_uncorrected_to_blanket_2x"] = [b_ / late_s / 1e9 / HBM_PEAK_GBS for b_ in k["raw_frac_bracket"]]
            if kernel in SURVEY_8D:
                pe, pn = SURVEY_8D[kernel]
                out["survey_8d_bytes_per_launch"] = pe * B * E + pn * NB
            return out, late_s

        pmc = pmc_record(args.departure_window)
        rf_rows, rows_s = roofline(pmc, ms_late, ms_all, nfr, 1, "k_fused_rows",
                                   "DirectionMPNN.update + ResponseMPNN + withdraw; the dominant kernel")
        rf_dir, dir_s = roofline(pmc, ms_late, ms_all, nfr, 0, "k_fused_direction",
                                 "DirectionMPNN message + aggregate on the packed hot records")
        rf_ic, _ = roofline(pmc, ms_late, ms_all, nfr, 2, "k_fused_insert",
                            "insert_agent_into_network + reward; a latency chain, several environments per wave")
        if congested is not None and not args.no_kernel_timing:
            cp = pmc_record(args.congested_window)
            congested["roofline"], c_rows_s = roofline(cp, c_late, c_all, c_nfr, 1, "k_fused_rows",
                                                       "the dominant kernel of the loaded network: most rows move something")
            congested["roofline_direction"], c_dir_s = roofline(cp, c_late, c_all, c_nfr, 0, "k_fused_direction",
                                                                "DirectionMPNN message + aggregate")
            congested["roofline_insert"], _ = roofline(cp, c_late, c_all, c_nfr, 2, "k_fused_insert", "insert + reward")
            congested["msgpass_pair_edges_per_sec"] = (B * E) / (c_dir_s + c_rows_s) if (c_dir_s + c_rows_s) > 0 else None
        out = {
            "metric": "ppo_env_steps_per_sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "value_rollout_only": ro_frames * world / ro_elapsed if ro_elapsed > 0 else None,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"mpnn+ppo train, {E}-edge synthetic torus dual graph ({n_roads} roads), "
                                   f"{args.agents} agents per environment, rollout-steps {T}, epochs {args.epochs}, "
                                   f"sub-batch {args.sub_batch}" +
                                   (" (BASELINE config 4)" if (E, args.agents) == (10000, 16384) else ""),
                       "envs_per_gpu": B, "env_steps_per_step": B * T, "parallelism": f"dp{world} (rollouts sharded, "
                       "one gradient all-reduce per optimiser step)", "rollout_kernels": rollout_mode},
            "msgpass_edges_per_sec": value * E,
            # Direction + Response pair alone (SURVEY 8d's second metric): B*E edges per frame / the two kernels' live time
            "msgpass_pair_edges_per_sec": (B * E) / (dir_s + rows_s) if (dir_s + rows_s) > 0 else None,
            "roofline": rf_rows, "roofline_direction": rf_dir, "roofline_insert": rf_ic,
            "setup_seconds": setup_s, "timed_seconds": elapsed,
            "per_rank": {"setup_seconds": [r_[0] for r_ in per_rank], "timed_seconds": [r_[1] for r_ in per_rank]},
            "world_size_seen_by_backend": dist_utils.world()[1], "dist_backend": dist_utils.backend_name(),
            "replica_param_max_abs_diff": replica_diff,
            "congested_regime": congested,
            "state_dependent_policy": policy_lines,
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args, net)
        print(json.dumps(out), flush=True)
    dist_utils.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
