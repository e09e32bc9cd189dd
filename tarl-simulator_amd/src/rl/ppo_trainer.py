"""ppo_train — same signature as the reference's (src/rl/ppo_trainer.py:12-14); the loop itself is
``tarl_hip.trainer.VecPPOTrainer`` (collector + GAE + clipped PPO loss + Adam, all HIP kernels). The replay-buffer round
trip through the CPU (:133) and per-frame TensorDict bookkeeping are not reproduced. The reference's scalar tags
(:60-88: ``PPO/avg_episode_return``, ``loss/*``, ``approx_kl``, ``clip_fraction``, ``grad_global_norm``,
``transport/*``) go to ``<log_dir>/train_log.jsonl`` — and to a TensorBoard ``SummaryWriter`` as well when
``torch.utils.tensorboard`` is importable. The periodic evaluation pass (:89-127,147-151) runs on ``eval_env`` through the drop-in classes (MODE = GraphDistribution.mode,
optionally a sampled rollout); its scalars ``eval/avg_return``, ``eval/episode_len``, ``eval/computation_time_ms`` and the
arrays behind the reference's histograms (``eval/nodes_metrics/avg_vc|std_vc``) and leg-histogram figure go into the same
record; matplotlib figures are not drawn.

Extension: ``num_envs`` (default 1) vectorises the rollout over B environments per GPU; under ``torchrun`` every rank
trains on its own environments and gradients are averaged with one RCCL all-reduce per optimiser step.
"""
from __future__ import annotations

import json
import os
import time

import torch

from .modules import unwrap


def _episode_returns(reward_tb, done_t):
    """The reference's logged return (src/rl/ppo_trainer.py:45-58): mean return of the episodes that END inside the batch,
    or — when none does — the sum of the rewards collected so far. ``reward_tb`` (T, B) on the host, ``done_t`` (T,) bool;
    averaged over the B environments."""
    returns, cum = [], torch.zeros(reward_tb.size(1))
    for t in range(reward_tb.size(0)):
        cum = cum + reward_tb[t]
        if bool(done_t[t]):
            returns.append(cum.mean().item())
            cum = torch.zeros_like(cum)
    return sum(returns) / len(returns) if returns else cum.mean().item()


def _evaluate(prefix, deterministic, eval_env, policy_module, frames_per_batch):
    """The reference's ``_evaluate`` (src/rl/ppo_trainer.py:89-127): one rollout of the policy on ``eval_env`` — its MODE
    (``deterministic``: GraphDistribution.mode, torchrl's ExplorationType.MODE) or a sampled action per frame — until the
    episode ends or ``frames_per_batch`` frames; scalars + the arrays behind the reference's histograms / figures."""
    start = time.perf_counter()
    actor = policy_module
    was = getattr(actor, "deterministic", False)
    actor.deterministic = bool(deterministic)
    try:
        with torch.no_grad():
            frames = eval_env.rollout(frames_per_batch, policy_module, break_when_any_done=True)
    finally:
        actor.deterministic = was
    comp_ms = (time.perf_counter() - start) * 1000.0
    rewards = torch.stack([f["next"]["reward"].reshape(-1) for f in frames]).view(-1)
    rec = {f"{prefix}/avg_return": float(rewards.sum()), f"{prefix}/episode_len": int(rewards.numel()),
           f"{prefix}/computation_time_ms": comp_ms}
    sim = getattr(eval_env, "simulator", None)
    if sim is not None:
        try:      # the data of writer.add_histogram(nodes_metrics/*) and of the leg-histogram figure, as arrays
            nm = sim.compute_node_metrics(output_dir=None)
            if nm:
                rec[f"{prefix}/nodes_metrics/avg_vc"] = [float(m["avg_vc"]) for m in nm.values()]
                rec[f"{prefix}/nodes_metrics/std_vc"] = [float(m["std_vc"]) for m in nm.values()]
            leg = sim.leg_histogram()           # (T, 4): departures, arrivals, on the way, time
            if leg.numel():
                rec[f"{prefix}/leg_histogram"] = leg.tolist()
        except Exception:  # noqa: BLE001 - the reference swallows metric errors here too (:125-126)
            pass
    return rec, frames


def ppo_train(env, policy_module, value_module, *, total_frames=128, frames_per_batch=32, num_epochs=1,
              sub_batch_size=32, device=torch.device("cpu"), checkpoint_path=None, log_dir=None, eval_env=None,
              eval_interval=0, log_interval=1, stochastic_eval=False, num_envs=1, seed=0):
    from tarl_hip import dist_utils, ops
    from tarl_hip.engine import SimEngine
    from tarl_hip.trainer import VecPPOTrainer

    rank, world = dist_utils.world()
    policy_net = unwrap(policy_module, "MPNNPolicyNet")
    value_net = unwrap(value_module, "MPNNValueNetSimple")
    sim = env.simulator
    g = sim.graph
    agents = sim.agent.agent_features
    # every rank (one process per GPU) rolls out its own environments: the device noise streams are keyed by the engine
    # seed, so seed + rank gives different trajectories per rank; parameters stay replicated (rank 0's are broadcast)
    engine = SimEngine(g.x, g.edge_index, g.edge_attr, sim.Nmax, agents,
                       congestion_constant=getattr(g, "congestion_constant", None), num_envs=num_envs,
                       device=g.x.device, timestep=sim.timestep, seed=seed + 7919 * rank,
                       fused=ops.fused_path_supported(g.edge_index, sim.Nmax))   # same decision on every rank
    l = value_net.final_mlp
    dormant = [p for n, p in policy_net.named_parameters() if not n.startswith("nodes_embedding")]
    head = getattr(policy_net, "policy_head", "embedding")
    m = policy_net.edge_mlp
    trainer = VecPPOTrainer(engine, policy_net.nodes_embedding.weight,
                            [l[0].weight, l[0].bias, l[2].weight, l[2].bias, l[4].weight, l[4].bias],
                            rollout_steps=frames_per_batch, num_epochs=num_epochs, sub_batch_size=sub_batch_size,
                            extra_params=dormant, seed=seed,
                            policy="embedding" if head == "embedding" else "edge_mlp",
                            edge_mlp_params=[m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias],
                            policy_precision={"edge_mlp_bf16": "bf16", "edge_mlp_fp32": "fp32"}.get(head, "x3"))
    log = writer = None
    if log_dir is not None and rank == 0:      # rank 0 alone writes
        os.makedirs(log_dir, exist_ok=True)
        log = open(os.path.join(log_dir, "train_log.jsonl"), "a")
        try:   # optional: the same scalars as TensorBoard events (tensorboard is not part of this image)
            from torch.utils.tensorboard import SummaryWriter
            writer = SummaryWriter(log_dir=log_dir)
        except Exception:  # noqa: BLE001
            writer = None
    trainer.keep_grad = log is not None
    frames = 0
    it = 0
    h = sim.h
    ppo_train.last_eval = {}
    ppo_train.last_trainer = trainer          # introspection hooks for tests / notebooks
    while frames < total_frames:
        t0 = time.perf_counter()
        frames += trainer.collect() // engine.B
        out = trainer.update()
        rec = None
        if log is not None and it % max(1, log_interval) == 0:
            o = out.tolist()
            x0, ag0 = engine.x[0], engine.agents[0]                       # environment 0, like the reference's single env
            done = ag0[:, 8] == 1
            vc = x0[:, h.NUMBER_OF_AGENT] / x0[:, h.MAX_NUMBER_OF_AGENT].clamp(min=1)
            rec = {"global_step": frames,
                   "PPO/avg_episode_return": _episode_returns(trainer.reward.cpu(), trainer.done_frames),
                   "loss/objective": o[0], "loss/value": o[1], "loss/entropy": o[2], "loss/total": o[0] + o[1] + o[2],
                   "approx_kl": o[4], "clip_fraction": o[3], "ESS": o[5],
                   "grad_global_norm": float(trainer.last_grad.norm()) if trainer.last_grad is not None else float("nan"),
                   "transport/avg_vc_ratio": float(vc.mean()), "transport/std_vc_ratio": float(vc.std(unbiased=False)),
                   "iter_seconds": time.perf_counter() - t0}
            if bool(done.any()):
                rec["transport/avg_travel_time"] = float((ag0[done, 3] - ag0[done, 2]).mean())
        # periodic evaluation (src/rl/ppo_trainer.py:147-151): MODE rollout, optionally a sampled one, on eval_env
        if log is not None and eval_env is not None and eval_interval and it % eval_interval == 0:
            rec = rec if rec is not None else {"global_step": frames}
            ev, fr = _evaluate("eval", True, eval_env, policy_module, frames_per_batch)
            rec.update(ev)
            ppo_train.last_eval["eval"] = fr
            if stochastic_eval:
                ev, fr = _evaluate("eval_stochastic", False, eval_env, policy_module, frames_per_batch)
                rec.update(ev)
                ppo_train.last_eval["eval_stochastic"] = fr
        if rec is not None:
            log.write(json.dumps(rec) + "\n")
            log.flush()
            if writer is not None:
                for k, v in rec.items():
                    if k in ("global_step", "iter_seconds"):
                        continue
                    if isinstance(v, list):
                        if k.endswith("_vc"):
                            writer.add_histogram(k, torch.tensor(v), frames)
                    else:
                        writer.add_scalar(k, v, frames)
        it += 1
    trainer.check_flags()
    sim.set_time(engine.time)
    if log is not None:
        log.close()
    if writer is not None:
        writer.close()
    if checkpoint_path is not None and rank == 0:
        try:
            # the reference saves policy_module.state_dict() of torchrl's ProbabilisticActor: keys module.0.module.<net
            # key> (ProbabilisticTensorDictSequential -> TensorDictModule -> the network). Same names here; tensors are
            # cloned so that the file holds the parameters alone, not the flat training buffer they are views of.
            sd = {f"module.0.module.{k}": v.detach().clone() for k, v in policy_net.state_dict().items()}
            torch.save(sd, checkpoint_path)
        except Exception:  # noqa: BLE001 - the reference swallows checkpoint errors too
            pass
    dist_utils.barrier()
    return None
