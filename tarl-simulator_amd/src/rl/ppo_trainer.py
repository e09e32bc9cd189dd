"""ppo_train — same signature as the reference's (src/rl/ppo_trainer.py:12-14); the loop itself is
``tarl_hip.trainer.VecPPOTrainer`` (collector + GAE + clipped PPO loss + Adam, all HIP kernels). The replay-buffer round
trip through the CPU (:133) and per-frame TensorDict bookkeeping are not reproduced. The reference's scalar tags
(:60-88: ``PPO/avg_episode_return``, ``loss/*``, ``approx_kl``, ``clip_fraction``, ``grad_global_norm``,
``transport/*``) go to ``<log_dir>/train_log.jsonl`` — and to a TensorBoard ``SummaryWriter`` as well when
``torch.utils.tensorboard`` is importable. Figures / histograms of the evaluation pass are not reproduced.

Extension: ``num_envs`` (default 1) vectorises the rollout over B environments per GPU; under ``torchrun`` every rank
trains on its own environments and gradients are averaged with one RCCL all-reduce per optimiser step.
"""
from __future__ import annotations

import json
import os
import time

import torch

from .modules import unwrap


def ppo_train(env, policy_module, value_module, *, total_frames=128, frames_per_batch=32, num_epochs=1,
              sub_batch_size=32, device=torch.device("cpu"), checkpoint_path=None, log_dir=None, eval_env=None,
              eval_interval=0, log_interval=1, stochastic_eval=False, num_envs=1, seed=0):
    from tarl_hip.engine import SimEngine
    from tarl_hip.trainer import VecPPOTrainer

    policy_net = unwrap(policy_module, "MPNNPolicyNet")
    value_net = unwrap(value_module, "MPNNValueNetSimple")
    sim = env.simulator
    g = sim.graph
    agents = sim.agent.agent_features
    engine = SimEngine(g.x, g.edge_index, g.edge_attr, sim.Nmax, agents,
                       congestion_constant=getattr(g, "congestion_constant", None), num_envs=num_envs,
                       device=g.x.device, timestep=sim.timestep, seed=seed)
    l = value_net.final_mlp
    dormant = [p for n, p in policy_net.named_parameters() if not n.startswith("nodes_embedding")]
    trainer = VecPPOTrainer(engine, policy_net.nodes_embedding.weight,
                            [l[0].weight, l[0].bias, l[2].weight, l[2].bias, l[4].weight, l[4].bias],
                            rollout_steps=frames_per_batch, num_epochs=num_epochs, sub_batch_size=sub_batch_size,
                            extra_params=dormant, seed=seed)
    log = writer = None
    if log_dir is not None:
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
    while frames < total_frames:
        t0 = time.perf_counter()
        frames += trainer.collect() // engine.B
        out = trainer.update()
        if log is not None and it % max(1, log_interval) == 0:
            o = out.tolist()
            x0, ag0 = engine.x[0], engine.agents[0]                       # environment 0, like the reference's single env
            done = ag0[:, 8] == 1
            vc = x0[:, h.NUMBER_OF_AGENT] / x0[:, h.MAX_NUMBER_OF_AGENT].clamp(min=1)
            rec = {"global_step": frames,
                   "PPO/avg_episode_return": float(trainer.reward.sum(0).mean()),
                   "loss/objective": o[0], "loss/value": o[1], "loss/entropy": o[2], "loss/total": o[0] + o[1] + o[2],
                   "approx_kl": o[4], "clip_fraction": o[3], "ESS": o[5],
                   "grad_global_norm": float(trainer.last_grad.norm()) if trainer.last_grad is not None else float("nan"),
                   "transport/avg_vc_ratio": float(vc.mean()), "transport/std_vc_ratio": float(vc.std(unbiased=False)),
                   "iter_seconds": time.perf_counter() - t0}
            if bool(done.any()):
                rec["transport/avg_travel_time"] = float((ag0[done, 3] - ag0[done, 2]).mean())
            log.write(json.dumps(rec) + "\n")
            log.flush()
            if writer is not None:
                for k, v in rec.items():
                    if k not in ("global_step", "iter_seconds"):
                        writer.add_scalar(k, v, frames)
        it += 1
    sim.set_time(engine.time)
    if log is not None:
        log.close()
    if writer is not None:
        writer.close()
    if checkpoint_path is not None:
        try:
            torch.save(policy_module.state_dict(), checkpoint_path)
        except Exception:  # noqa: BLE001 - the reference swallows checkpoint errors too
            pass
    return None
