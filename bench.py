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
to evaluating them per frame; the ``state_dependent_policy`` lines run the per-edge MLP head with per-frame logits instead).
Workload: BASELINE.json config 4 — 10k-edge synthetic torus dual graph, 16k agents, rollout-steps 256 — with ``--envs``
vectorised environments per GPU (weak scaling: per-GPU work is fixed). Inputs are resident in HBM before the timed
region. The timed region is bracketed by barrier + synchronize on both sides; the slowest rank's time is used.

ONE JSON line on stdout, numbers only and below 4 KB (the driver keeps the line's tail); every descriptive field — what a
kernel does, where a traffic figure comes from, the CPU sample in words — goes to the sidecar
``gpurun_out/bench_details.json`` (and its path to stderr). Objects on the line:
  roofline            — the DOMINANT kernel of a frame, k_fused_rows (DirectionMPNN.update + ResponseMPNN message /
                        aggregate / update + withdraw): ``traffic`` = HBM bytes per launch from the PMC counters
                        (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc passes of THIS script reduced by tools/pmc_bench.py
                        over frames >= 200 of an iteration, committed under profiles/ and matched by the sha256 of the
                        frame-kernel sources) / the kernel's average launch duration over the same frames, measured live
                        with HIP events on the launch stream, vs the 8 TB/s HBM peak. ``compulsory_frac``: the same with
                        the bytes the packed layout must move; ``rd_req`` / ``wr_req``: memory-side
                        read / write requests per launch (TCC_EA0_RDREQ / WRREQ).
  roofline_direction  — the same for k_fused_direction (DirectionMPNN.message + aggregate, the scatter kernel the
                        north star names); roofline_insert — the insert launch (latency-bound).
  config5             — BASELINE config 5 (100k route edges, 262 144 agents) at ``--config5-envs`` environments: one or two
                        PPO iterations, its own three roofline objects, and the bf16 per-edge MLP line at that size.
  update_path         — the same iteration with ``--update-epochs`` x ``--update-sub-batch`` (the update at a size where
                        its kernels exist): HIP-event time and algorithmic bytes per update stage.
  congested_regime    — every agent departs within ``--congested-window`` seconds (the loaded network).
  state_dependent_policy — the per-edge MLP head (nothing hoisted): "fp32" = rollout logits on the fp32 matrix pipe (exact
                        fp32 products), "fp32_x3" = fp32 accuracy on the bf16 pipe (operands in exact bf16 pieces),
                        "bf16" = bf16 logits on bf16 observations; each with an MFMA roofline object for the MLP launch.
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
# written by tools/pmc_bench.py from --pmc passes of this script (one record per workload), matched by config + source hash
PMC_RECORDS = tuple(os.path.join("profiles", n) for n in
                    ("r05_pmc_traffic.json", "r05_pmc_traffic_congested.json", "r05_pmc_traffic_c5.json"))
# Compulsory HBM bytes per (road, environment) and launch of the packed env-minor layout: what each kernel
# must read and write once, neighbour gathers served by the XCD's L2, statics / topology / policy tables through the
# scalar cache (shared by all environments, not counted).
COMPULSORY = {
    "k_fused_direction": 13.0 + 4.0,   # head words 8 + tail word 4 + SELECTED_ROAD byte 1 in; post word 4 out
    "k_fused_rows": 16.0 + 1.0,        # post word 4 + head words 8 + tail word 4 in; count byte 1 out
    "k_fused_insert": 0.0,             # departure window, a few words per admitted agent, the accumulator banks
}
KERNEL_WHAT = {
    "k_fused_rows": "DirectionMPNN.update + ResponseMPNN message / aggregate / update + withdraw; the dominant kernel",
    "k_fused_direction": "DirectionMPNN message + aggregate on the packed dense words (the scatter kernel)",
    "k_fused_insert": "insert_agent_into_network + reward; a latency chain, several environments per wave",
}


def sig(x, n=4):
    return float(f"{x:.{n}g}") if isinstance(x, float) else x


def compact(o, key=None):
    """Numbers at four significant digits, five for the throughput figures (the line must stay below 4 KB)."""
    if isinstance(o, dict):
        return {k: compact(v, k) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [compact(v, key) for v in o]
    return sig(o, 5 if key in ("value", "value_rollout_only", "ms_per_step") else 4)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--edges", type=int, default=10000)
    ap.add_argument("--agents", type=int, default=16384)
    ap.add_argument("--envs", type=int, default=32768,
                    help="vectorised environments per GPU (32 768: 150 GB of the 288; 16 384 until round 5: -6 % env-steps/s)")
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
    ap.add_argument("--update-epochs", type=int, default=8,
                    help="update_path object: epochs per iteration (0 = skip); with --update-sub-batch frames per minibatch")
    ap.add_argument("--update-sub-batch", type=int, default=4096)
    ap.add_argument("--update-steps", type=int, default=2)
    ap.add_argument("--config5-envs", type=int, default=2048,
                    help="environments per GPU of the config5 object (BASELINE config 5: 100k edges, 262 144 agents; 0 = skip)")
    ap.add_argument("--config5-steps", type=int, default=2)
    ap.add_argument("--policy-envs", type=int, default=4096,
                    help="environments per GPU of the state-dependent-policy lines (policy_head=edge_mlp; 0 = skip)")
    ap.add_argument("--policy-steps", type=int, default=2)
    ap.add_argument("--policy-temperature", type=float, default=2000.0,
                    help="GraphDistribution temperature of the state-dependent-policy lines: the head reads raw features "
                         "(clock times ~2e4), so an untrained head at temperature 1 is near-deterministic, drives every "
                         "agent down the same turn and gridlocks the network out of the reference's domain")
    ap.add_argument("--metrics-envs", type=int, default=1,
                    help="environments that keep the per-node logs of SimulatorEnv._step (delta_travel_time, pop / withdraw "
                         "masks); the per-frame leg histogram is kept for all of them")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="do not bracket the frame kernels with HIP events (PMC / kernel-trace passes)")
    ap.add_argument("--details", type=str, default=os.path.join("gpurun_out", "bench_details.json"),
                    help="sidecar file with every descriptive field (relative to the repo root; '' = stderr only)")
    return ap.parse_args()


def make_network(edges):
    from tarl_hip import synth
    W, H = synth.torus_for_edges(edges)
    return synth.torus_network(W, H)


def build_trainer(args, rank, device, net=None, *, agents=None, envs=None, window=None, seed_off=0, **trainer_kw):
    """Engine + trainer of one workload: every environment of every rank gets its own population, drawn on the device."""
    from tarl_hip import synth
    from tarl_hip.engine import SimEngine
    from tarl_hip.trainer import VecPPOTrainer
    from src.agents.mpnn_agent import MPNNPolicyNet, MPNNValueNetSimple

    net = net if net is not None else make_network(args.edges)
    agents = args.agents if agents is None else agents
    envs = args.envs if envs is None else envs
    window = args.departure_window if window is None else window
    N = net.num_roads
    pops = synth.population_batch(agents, N, envs, seed=args.seed + 1000 * rank + seed_off, device=device,
                                  t1=window + synth.EPISODE_START if window else synth.EPISODE_END)
    engine = SimEngine(net.x.to(device).unsqueeze(0).repeat(envs, 1, 1).contiguous(), net.edge_index,
                       net.edge_attr, net.Nmax, pops, congestion_constant=net.congestion_constant,
                       device=device, seed=args.seed + rank)
    torch.manual_seed(args.seed)       # identical initial weights on every rank (also broadcast by the trainer)
    head = trainer_kw.get("policy") == "edge_mlp"
    ff = None if head else net.x[:, 3 * net.Nmax + 2][net.edge_index[1]]
    pol = MPNNPolicyNet(net.edge_index, N, ff, device=str(device))
    val = MPNNValueNetSimple(net.edge_index, N, device=str(device))
    l, mm = val.final_mlp, pol.edge_mlp
    if head:
        trainer_kw["edge_mlp_params"] = [mm[0].weight, mm[0].bias, mm[2].weight, mm[2].bias, mm[4].weight, mm[4].bias]
    trainer_kw.setdefault("num_epochs", args.epochs)
    trainer_kw.setdefault("sub_batch_size", args.sub_batch)
    trainer_kw.setdefault("metrics_envs", args.metrics_envs)
    trainer = VecPPOTrainer(engine, pol.nodes_embedding.weight,
                            [l[0].weight, l[0].bias, l[2].weight, l[2].bias, l[4].weight, l[4].bias],
                            rollout_steps=args.rollout_steps,
                            extra_params=[p for n, p in pol.named_parameters() if not n.startswith("nodes_embedding")],
                            seed=args.seed, **trainer_kw)
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
            "sample": f"{iters} PPO iter x 1 env: {steps} oracle env steps + 1 update each, {el:.1f}+{up:.2f} s",
            "value_rollout_only": steps / el, "update_seconds": up, "rollout_seconds": el,
            "all_cores": {"cores": nsock, "value": by_threads[str(nsock)], "logical_cpus_of_the_host": ncpu},
            "by_threads_1s_probe": by_threads}


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


def time_iterations(trainer, steps, L, T, prof, device, run=None):
    """``steps`` iterations bracketed by barrier + synchronize; HIP events around the frame kernels of the first one.
    -> (frames, slowest rank's seconds, this rank's seconds, (ms_all[3], ms_late[3], frames[2]))"""
    from tarl_hip import dist_utils, lib
    if prof:
        L.tarl_prof_enable(T)
    dist_utils.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames = 0
    for _ in range(steps):
        frames += (run or trainer.train_iteration)()
    torch.cuda.synchronize()
    dist_utils.barrier()
    mine = time.perf_counter() - t0
    elapsed = dist_utils.allreduce_max_float(mine, device)
    ms_all, ms_late, nfr = (ctypes.c_double * 3)(), (ctypes.c_double * 3)(), (ctypes.c_int64 * 2)()
    lib.check(L.tarl_prof_collect(LATE_FRAME if T > LATE_FRAME else 0, ms_all, ms_late, nfr))
    L.tarl_prof_enable(0)
    trainer.check_flags()
    return frames, elapsed, mine, (list(ms_all), list(ms_late), list(nfr))


def pmc_record(cfg):
    """The committed PMC record of a workload, or None: the config must match AND the record must have been taken on the
    very frame-kernel sources this run executes (sha256 of csrc/fused.hip + fused_common.h) — a kernel edit without a fresh
    PMC pass falls back to the compulsory bytes instead of pairing new times with old bytes."""
    from tarl_hip.ops import frame_kernel_source_hash
    sha = frame_kernel_source_hash()
    for path in PMC_RECORDS:
        try:
            rec = json.load(open(os.path.join(ROOT, path)))
            if rec["config"] == cfg and rec.get("source_sha16") == sha:
                rec["path"] = path
                return rec
        except (OSError, KeyError, ValueError):
            pass
    return None


def rooflines(cfg, NB, prof, T, full_first=True, want_req=True):
    """The three frame kernels' roofline objects of one workload: (compact objects for the line, verbose ones for the
    sidecar, {kernel: late seconds}). achieved = HBM bytes per launch (PMC counters of this script's own rollout, frames >=
    LATE_FRAME; without a matching record: the compulsory bytes, ``traffic`` null) / the live average launch duration."""
    ms_all, ms_late, nfr = prof
    pmc = pmc_record(cfg)
    late0 = LATE_FRAME if T > LATE_FRAME else 0
    line, detail, secs = {}, {}, {}
    for key, slot, kernel in (("roofline", 1, "k_fused_rows"), ("roofline_direction", 0, "k_fused_direction"),
                              ("roofline_insert", 2, "k_fused_insert")):
        n_late, n_all = max(1, nfr[1]), max(1, nfr[0])
        late_s, all_s = ms_late[slot] / n_late * 1e-3, ms_all[slot] / n_all * 1e-3
        comp = COMPULSORY[kernel] * NB
        k = pmc["kernels"].get(kernel) if pmc else None
        traffic = k["hbm_bytes_per_launch"] if k else None
        byts = traffic if traffic is not None else comp
        achieved = byts / late_s / 1e9 if late_s > 0 else 0.0
        o = {"kernel": kernel, "achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
             "avg_launch_us": late_s * 1e6, "avg_launch_us_all_frames": all_s * 1e6,
             "bytes_basis": "pmc" if traffic is not None else "compulsory",
             "compulsory_frac": (comp / late_s / 1e9 / HBM_PEAK_GBS) if late_s > 0 else 0.0,
             "traffic_over_compulsory": (traffic / comp) if (traffic is not None and comp > 0) else None}
        if k and "TCC_EA0_RDREQ_sum" in k:
            o["rd_req"], o["wr_req"] = k["TCC_EA0_RDREQ_sum"], k.get("TCC_EA0_WRREQ_sum")
        full = {"bound": "hbm", "kernel": kernel, "achieved": o["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": o["frac"], "traffic": traffic, **{a: b for a, b in o.items() if a not in ("kernel", "achieved", "frac", "traffic")}}
        if key == "roofline" and full_first:      # the contract's object: bound / achieved / peak / unit / frac / traffic
            line[key] = full
        else:                                     # the other kernels / workloads: the same numbers under fewer keys (the 4 KB line)
            line[key] = {"frac": o["frac"], "achieved": o["achieved"], "traffic": traffic, "avg_launch_us": o["avg_launch_us"],
                         "us_all": o["avg_launch_us_all_frames"], "t_over_c": o["traffic_over_compulsory"]}
            if want_req and "rd_req" in o:
                line[key]["req"] = [o["rd_req"], o["wr_req"]]
        detail[key] = dict(full, what=KERNEL_WHAT[kernel], frames_timed=int(nfr[1]),
                           first_frame=late0, compulsory_bytes_per_launch=comp,
                           traffic_source=(f"{pmc['path']}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, mean over "
                                           f"frames >= {LATE_FRAME}; bytes = {k.get('formula', '2*FETCH_SIZE + WRITE_SIZE')}") if k else None,
                           note="frac falls when re-reads are removed faster than time: compare avg_launch_us and "
                                "traffic_over_compulsory across rounds, not frac alone")
        secs[kernel] = late_s
    return line, detail, secs


def update_path(args, trainer, engine, world, L, T, device):
    """The update at a size where its kernels exist (the reference's loop shape, src/rl/ppo_trainer.py:129-145, with more
    epochs and a larger sub-batch): env-steps/s of the iteration, and per update stage the HIP-event time per call plus its
    algorithmic bytes (what the stage must read + write once)."""
    from tarl_hip.trainer import NoStageTimer, StageTimer
    B, N, E = engine.B, engine.N, engine.E
    M = min(args.update_sub_batch, T * B)
    saved = trainer.num_epochs, trainer.M
    trainer.num_epochs, trainer.M = args.update_epochs, M
    trainer.train_iteration()                                   # warm-up at this shape (scratch allocations)
    trainer.stage = st = StageTimer()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    up_s, frames = 0.0, 0
    for _ in range(args.update_steps):
        frames += trainer.collect()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        trainer.update()
        torch.cuda.synchronize()
        up_s += time.perf_counter() - t1
    el = time.perf_counter() - t0
    ms = st.ms()
    trainer.stage = NoStageTimer()
    trainer.num_epochs, trainer.M = saved
    trainer.check_flags()
    P = trainer.flat.numel
    K = N + 1
    # algorithmic bytes per call of each stage (fp32 unless noted): inputs read once + outputs written once
    byts = {"critic_all_frames": (T + 1) * B * N + 3 * 64 * K * 2 + (T + 1) * B * 4,     # count bytes + W1 pieces + values
            "gae": T * B * 4 * 5,
            "minibatch_gather": M * N * (1 + 1 + 4 + 4),                                  # action / count bytes in, ids / fp32 rows out
            "actor_logits_fwd": M * E * 4 + N * 4,
            "graphdist_fwd": M * E * 4 * 3 + M * N * 4,
            "critic_fwd": M * K * 4 + 64 * K * 4 + M * 64 * 4 * 2,
            "ppo_loss": M * 4 * 9,
            "graphdist_bwd": M * E * 4 * 2 + M * N * 4,
            "actor_logits_bwd": M * E * 4 + N * 4,
            "critic_bwd": M * K * 4 + 2 * 64 * K * 4 + M * 64 * 4 * 2,
            "grad_allreduce": P * 4 * 2, "adam": P * 4 * 7}
    stages = {k: {"us": v[0] / v[1] * 1e3, "GBs": byts.get(k, 0) / (v[0] / v[1] * 1e-3) / 1e9 if v[0] > 0 else None}
              for k, v in ms.items()}
    n_up = args.update_steps * args.update_epochs
    slow = max((k for k in stages if k != "critic_all_frames"), key=lambda k: stages[k]["us"])
    line = {"value": frames * world / el, "epochs": args.update_epochs, "sub_batch": M, "ms_per_step": el / args.update_steps * 1e3,
            "update_frac": up_s / el, "ms_per_minibatch_step": (up_s / n_up) * 1e3,
            "stage_us": {k: v["us"] for k, v in stages.items()}, "slowest_minibatch_stage": slow,
            "slowest_GBs": stages[slow]["GBs"]}
    detail = dict(line, stages={k: dict(v, bytes_per_call=byts.get(k)) for k, v in stages.items()}, update_seconds=up_s,
                  note="HIP events on the launch stream around each stage of VecPPOTrainer.advantages / minibatch_step; "
                       "bytes = algorithmic (inputs once + outputs once)")
    return line, detail


def policy_lines(args, net, rank, world, device, L, T, envs, precisions, steps, seed_off=31, agents=None):
    """The state-dependent policy (per-edge MLP head 33 -> 64 -> 32 -> 1 on cat(x[src], x[dst], edge_attr); the reference
    keeps it as parameters, src/agents/mpnn_agent.py:35-41,227-231): NOTHING is hoisted — every frame builds the
    observation from the packed state, runs the MLP on the matrix cores, the segment softmax, the sample and the log-prob,
    then the simulation frame; the update runs the MLP forward / backward."""
    from tarl_hip import dist_utils, lib
    E = net.edge_index.size(1)
    out, detail, eng_p = {}, {}, None
    for tag, prec in precisions:
        if eng_p is None:
            _, eng_p, tr_p = build_trainer(args, rank, device, net, agents=agents, envs=envs, window=0, seed_off=seed_off,
                                           policy="edge_mlp", policy_precision=prec, temperature=args.policy_temperature,
                                           metrics_envs=1)
        else:
            tr_p = rebuild_policy_trainer(args, eng_p, net, device, prec)
        tr_p.train_iteration()
        if not args.no_kernel_timing:
            L.tarl_prof_enable(T)      # HIP events around the per-edge MLP launch of the first timed iteration's frames
        dist_utils.barrier()
        torch.cuda.synchronize()
        t2_ = time.perf_counter()
        pf = 0
        for _ in range(steps):
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
        flops = 2.0 * (33 * 64 + 64 * 32 + 32) * envs * E
        peak = MFMA_PEAK_TFLOPS["f32" if prec == "fp32" else "bf16"]
        kname = {"x3": "k_edge_mlp_fwd_x3", "fp32": "k_edge_mlp_fwd_f32", "bf16": "k_edge_mlp_fwd_bf16"}[prec]
        roof = None
        if mlp_s > 0:
            roof = {"bound": "mfma", "kernel": kname, "achieved": flops / mlp_s / 1e12, "peak": peak, "unit": "TFLOP/s",
                    "frac": flops / mlp_s / 1e12 / peak, "traffic": None, "avg_launch_us": mlp_s * 1e6}
            if prec == "x3":
                # the kernel ISSUES six bf16 piece products per multiply-add of the head: issued_frac = the piece products the
                # matrix cores actually execute (50 MFMAs of 32x32x16 per 32 edges) over the bf16 peak
                roof["issued_frac"] = 2.0 * 32 * 32 * 16 * 50 * (envs * E / 32.0) / mlp_s / 1e12 / peak
        out[tag] = {"value": pf * world / pel, "ms_per_step": pel / steps * 1e3,
                    "roofline": {k: roof[k] for k in ("bound", "achieved", "frac", "avg_launch_us", "issued_frac") if k in roof} if roof else None}
        detail[tag] = dict(out[tag], roofline=roof, unit="env-steps/s", envs_per_gpu=envs, steps=steps, edge_mlp_edges_per_sec=pf * world / pel * E,
                           launches_timed=int(p_nfr[0]), flop_per_launch=flops,
                           rollout_logits={"bf16": "bf16 MFMA (v_mfma_f32_32x32x16_bf16) on bf16 observations",
                                           "fp32": "fp32 MFMA (v_mfma_f32_32x32x2_f32), exact fp32 products",
                                           "x3": "fp32-accurate on the bf16 pipe: operands in three exact bf16 pieces, six "
                                                 "products per multiply-add (v_mfma_f32_32x32x16_bf16)"}[prec])
        del tr_p
    out["envs_per_gpu"], out["temperature"] = envs, args.policy_temperature
    detail["note"] = ("policy_head=edge_mlp: per-frame observation + 33->64->32->1 MLP per edge + GraphDistribution softmax / "
                      f"sample / log_prob (no table hoist), then the simulation frame; GraphDistribution temperature {args.policy_temperature:g}")
    del eng_p
    torch.cuda.empty_cache()
    return out, detail


def rebuild_policy_trainer(args, eng_p, net, device, prec):
    from tarl_hip.trainer import VecPPOTrainer
    from src.agents.mpnn_agent import MPNNPolicyNet, MPNNValueNetSimple
    torch.manual_seed(args.seed)
    pol = MPNNPolicyNet(net.edge_index, net.num_roads, None, device=str(device))
    val = MPNNValueNetSimple(net.edge_index, net.num_roads, device=str(device))
    l, mm = val.final_mlp, pol.edge_mlp
    return VecPPOTrainer(eng_p, pol.nodes_embedding.weight,
                         [l[0].weight, l[0].bias, l[2].weight, l[2].bias, l[4].weight, l[4].bias],
                         rollout_steps=args.rollout_steps, num_epochs=args.epochs, sub_batch_size=args.sub_batch,
                         extra_params=[p for n, p in pol.named_parameters() if not n.startswith("nodes_embedding")],
                         seed=args.seed, policy="edge_mlp", policy_precision=prec, temperature=args.policy_temperature,
                         edge_mlp_params=[mm[0].weight, mm[0].bias, mm[2].weight, mm[2].bias, mm[4].weight, mm[4].bias])


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
    prof = not args.no_kernel_timing

    t_setup = time.perf_counter()
    net, engine, trainer = build_trainer(args, rank, device)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_setup
    E, B, T, n_roads = engine.E, engine.B, args.rollout_steps, engine.N
    NB = B * n_roads
    layout_tag, rollout_mode = trainer.layout_tag, trainer.rollout

    # ---- the headline: `steps` PPO iterations of BASELINE's workload ----------------------------------------------------------
    for _ in range(args.warmup):
        trainer.train_iteration()
    frames, elapsed, elapsed_rank, prof_main = time_iterations(trainer, args.steps, L, T, prof, device)
    per_rank = dist_utils.gather_floats([setup_s, elapsed_rank], device)     # [rank][setup_seconds, timed_seconds]
    # data-parallel evidence (outside the timed region): after `steps` averaged-gradient Adam steps every replica must still
    # hold rank 0's parameter bits
    replica_diff = dist_utils.replica_max_abs_diff(trainer.flat.flat)
    # rollout-only figure (BASELINE.md §3 / SURVEY §8d: "rollout + update, and rollout-only"): the collector loop alone
    ro_frames, ro_elapsed, _, _ = time_iterations(trainer, min(args.steps, 3), L, T, False, device, run=trainer.collect)

    # ---- the update at a size where its kernels exist -----------------------------------------------------------------------------
    upd = upd_detail = None
    if args.update_epochs > 0 and not args.departure_window and world == 1:      # (N = 1 only: a multi-rank run times the headline)
        upd, upd_detail = update_path(args, trainer, engine, world, L, T, device)

    # ---- the congested regime: every agent departs within --congested-window seconds -----------------------------------------------
    # The headline workload spreads the departures over the 61-minute episode (BASELINE config), so the 256 timed frames see
    # a filling network. Here the FIFOs fill up, most rows pop / withdraw / enqueue in every frame. Same engine, same kernels,
    # populations re-drawn and re-packed.
    congested = cong_detail = None
    if args.congested_window > 0 and not args.departure_window:
        from tarl_hip import synth
        engine.agents.copy_(synth.population_batch(args.agents, engine.N, B, seed=args.seed + 1000 * rank + 17,
                                                   device=device, t1=synth.EPISODE_START + args.congested_window))
        engine.fs.order_valid = False
        engine._packed_stale = True
        trainer.train_iteration()                       # warm-up (re-pack, re-sort)
        cf, cel, _, prof_c = time_iterations(trainer, args.congested_steps, L, T, prof, device)
        congested = {"value": cf * world / cel, "ms_per_step": cel / args.congested_steps * 1e3,
                     "departure_window_s": args.congested_window,
                     "agents_on_the_way_at_the_end_per_env": float(engine.agents[:, :, 7].sum()) / B}
        cong_detail = dict(congested, unit="env-steps/s", steps=args.congested_steps,
                           agents_arrived_per_env=float(engine.agents[:, :, 8].sum()) / B,
                           note="all agents depart within the window: the network is loaded for most of the rollout")
        if prof:
            cfg_c = {"edges": E, "agents": args.agents, "envs": B, "rollout_steps": T, "departure_window": args.congested_window}
            rl, rd, secs = rooflines(cfg_c, NB, prof_c, T, full_first=False)
            congested.update(rl)
            cong_detail.update(rd)
            pair = secs["k_fused_direction"] + secs["k_fused_rows"]
            congested["msgpass_pair_edges_per_sec"] = (B * E) / pair if pair > 0 else None
    del trainer, engine
    torch.cuda.empty_cache()

    # ---- BASELINE config 5: 100k route edges, 262 144 agents ("HBM-bound scatter stress") --------------------------------------------
    c5 = c5_detail = None
    if args.config5_envs > 0 and world == 1 and not args.departure_window and (E, args.agents) == (10000, 16384):
        from tarl_hip import synth
        B5, A5 = args.config5_envs, 262144
        net5 = synth.torus_network(25, 250)
        E5, N5 = net5.edge_index.size(1), net5.num_roads
        t5 = time.perf_counter()
        _, eng5, tr5 = build_trainer(args, rank, device, net5, agents=A5, envs=B5, window=0, seed_off=53)
        tr5.train_iteration()
        torch.cuda.synchronize()
        setup5 = time.perf_counter() - t5
        f5, el5, _, prof5 = time_iterations(tr5, args.config5_steps, L, T, prof, device)
        c5 = {"value": f5 * world / el5, "ms_per_step": el5 / args.config5_steps * 1e3, "envs_per_gpu": B5,
              "edges": E5, "agents": A5, "msgpass_edges_per_sec": f5 * world / el5 * E5}
        c5_detail = dict(c5, unit="env-steps/s", roads=N5, steps=args.config5_steps, setup_seconds=setup5,
                         rollout_kernels=tr5.rollout,
                         workload="BASELINE config 5: mpnn+ppo train, 100 000-edge torus dual graph, 262 144 agents per environment")
        if prof:
            rl, rd, secs = rooflines({"edges": E5, "agents": A5, "envs": B5, "rollout_steps": T}, B5 * N5, prof5, T, full_first=False, want_req=False)
            c5.update(rl)
            c5_detail.update(rd)
            pair = secs["k_fused_direction"] + secs["k_fused_rows"]
            c5["msgpass_pair_edges_per_sec"] = (B5 * E5) / pair if pair > 0 else None
            # picoseconds per (road, environment) pair of the Direction + row pass pair: config 4's figure beside it
            c5["ps_per_pair"] = pair / (B5 * N5) * 1e12
        del tr5, eng5
        torch.cuda.empty_cache()
        # the bf16 per-edge MLP head at this size ("bf16 MPNN features")
        pl, pd = policy_lines(args, net5, rank, world, device, L, T, B5, (("bf16", "bf16"),), 1, seed_off=59, agents=A5)
        c5["edge_mlp_bf16"] = pl["bf16"]
        c5_detail["edge_mlp_bf16"] = pd["bf16"]

    # ---- the state-dependent policy at config 4 -------------------------------------------------------------------------------------
    pol = pol_detail = None
    if args.policy_envs > 0:
        # "fp32": rollout logits on the fp32 matrix pipe (exact fp32 products); "fp32_x3": the same 1e-4 contract on the bf16
        # pipe — operands split into exact bf16 pieces, k_edge_mlp_fwd_x3 (the trainer's default for an fp32 policy);
        # "bf16": bf16 logits on bf16 observations (BASELINE config 5's "bf16 MPNN features")
        pol, pol_detail = policy_lines(args, net, rank, world, device, L, T, args.policy_envs,
                                       (("fp32", "fp32"), ("fp32_x3", "x3"), ("bf16", "bf16")), args.policy_steps)

    if rank == 0:
        value = frames * world / elapsed
        rl, rd, secs = rooflines({"edges": E, "agents": args.agents, "envs": B, "rollout_steps": T,
                                  **({"departure_window": args.departure_window} if args.departure_window else {})},
                                 NB, prof_main, T)
        pair = secs["k_fused_direction"] + secs["k_fused_rows"]
        head = {
            "metric": "ppo_env_steps_per_sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"mpnn+ppo train, {E}-edge torus dual graph, {args.agents} agents, T={T}" +
                                   (" (BASELINE config 4)" if (E, args.agents) == (10000, 16384) else ""),
                       "envs_per_gpu": B, "parallelism": f"dp{world}", "rollout_kernels": rollout_mode},
            # Direction + Response pair alone (SURVEY 8d's second metric): B*E edges per frame / the two kernels' live time
            "msgpass_pair_edges_per_sec": (B * E) / pair if pair > 0 else None,
            "timed_seconds": elapsed,
            "per_rank": {"setup_seconds": [r_[0] for r_ in per_rank], "timed_seconds": [r_[1] for r_ in per_rank]},
            "world_size_seen_by_backend": dist_utils.world()[1], "dist_backend": dist_utils.backend_name(),
            "replica_param_max_abs_diff": replica_diff,
        }
        out = dict(head, **rl)
        out["config5"], out["update_path"], out["congested_regime"], out["state_dependent_policy"] = c5, upd, congested, pol
        out["value_rollout_only"] = ro_frames * world / ro_elapsed if ro_elapsed > 0 else None
        details = dict(head, **rd)
        details.update(config5=c5_detail, update_path=upd_detail, congested_regime=cong_detail, state_dependent_policy=pol_detail,
                       value_rollout_only=out["value_rollout_only"], layout=layout_tag,
                       env_steps_per_step=B * T, msgpass_edges_per_sec=value * E, epochs=args.epochs, sub_batch=args.sub_batch,
                       roads=n_roads, setup_seconds=setup_s,
                       parallelism=f"dp{world}: rollouts sharded, one gradient all-reduce per optimiser step")
        if world == 1 and args.cpu_seconds > 0:
            cb = cpu_baseline(args, net)
            details["cpu_baseline"] = cb
            out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "value_rollout_only",
                                                      "update_seconds", "all_cores")}
        line = json.dumps(compact(out), separators=(",", ":"))
        if args.details:
            try:
                path = os.path.join(ROOT, args.details)
                os.makedirs(os.path.dirname(path), exist_ok=True)
                with open(path, "w") as f:
                    json.dump(details, f, indent=1)
                print(f"bench.py: descriptive fields in {args.details}", file=sys.stderr)
            except OSError as exc:
                print(f"bench.py: could not write {args.details}: {exc}", file=sys.stderr)
        else:
            print(json.dumps(details), file=sys.stderr)
        print(line, flush=True)
    dist_utils.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
