"""CPU restatement of the PPO arithmetic configured in src/rl/ppo_trainer.py:35-37,129-145
(TEST INFRASTRUCTURE — see oracle/__init__.py).

**Parity unpinned**: the arithmetic is torchrl 0.5.0's ``GAE`` / ``ClipPPOLoss`` (absent wheel; the reference's tests
pin no advantage or loss value). Restated from the published formulas.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as Fn


# Floor of the advantage's standard deviation in average_gae (csrc/ppo.hip: TARL_ADV_STD_FLOOR — see the note there: SURVEY
# §3.4 says 1e-6, torchrl 0.5.0's GAE may use 1e-4; unverifiable offline, immaterial unless std(A) < 1e-4).
ADV_STD_FLOOR = 1e-6


def gae(reward, value, next_value, done, terminated, gamma=0.99, lmbda=0.95, average_gae=True):
    """torchrl ``GAE(gamma=.99, lmbda=.95, average_gae=True)`` over time-major tensors ``(T, ...)``.

    delta_t = r_t + gamma * V_{t+1} * (1 - terminated_t) - V_t
    A_t     = delta_t + gamma * lmbda * (1 - done_t) * A_{t+1}
    value_target = A + V_t (before normalisation); then A <- (A - mean) / max(std, ADV_STD_FLOOR) with the unbiased std.
    Returns (advantage, value_target)."""
    T = reward.size(0)
    not_term = 1.0 - terminated.to(reward.dtype)
    not_done = 1.0 - done.to(reward.dtype)
    delta = reward + gamma * next_value * not_term - value
    adv = torch.zeros_like(delta)
    run = torch.zeros_like(delta[0])
    for t in range(T - 1, -1, -1):
        run = delta[t] + gamma * lmbda * not_done[t] * run
        adv[t] = run
    target = adv + value
    if average_gae:
        adv = (adv - adv.mean()) / adv.std().clamp_min(ADV_STD_FLOOR)
    return adv, target


def clip_ppo_loss(log_prob_new, log_prob_old, advantage, value, value_target, entropy, clip_epsilon=0.2,
                  entropy_coef=0.01, critic_coef=1.0):
    """torchrl ``ClipPPOLoss(clip_epsilon=.2)`` defaults: entropy bonus 0.01, critic coef 1.0, smooth-L1 critic loss,
    mean reduction. Returns dict(loss_objective, loss_critic, loss_entropy)."""
    lw = log_prob_new - log_prob_old
    ratio = lw.exp()
    gain1 = ratio * advantage
    gain2 = lw.clamp(math.log1p(-clip_epsilon), math.log1p(clip_epsilon)).exp() * advantage
    loss_obj = -torch.min(gain1, gain2).mean()
    loss_critic = critic_coef * Fn.smooth_l1_loss(value, value_target, reduction="none").mean()
    loss_ent = -entropy_coef * entropy.mean()
    return {"loss_objective": loss_obj, "loss_critic": loss_critic, "loss_entropy": loss_ent}


def adam_step(param, grad, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (src/rl/ppo_trainer.py:37), single-tensor form; ``step`` is 1-based. In place."""
    m.lerp_(grad, 1 - beta1)
    v.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(m, denom, value=-lr / bc1)
    return param
