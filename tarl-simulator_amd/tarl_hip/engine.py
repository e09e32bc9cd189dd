"""SimEngine — B independent environments of the traffic simulator over ONE static road graph, all state resident on one
GPU, stepped by the HIP kernels with no host synchronisation.

One step == ``SimulatorEnv._step`` of the reference (src/reinforcement_learning.py:222-309): apply the routing action,
DirectionMPNN + ResponseMPNN, withdraw, insert, reward, advance the clock. The reference has a single environment
(``batch_size=[]``); the batch dimension is this build's way of giving the message-passing kernels enough edges per
launch — every environment follows exactly the reference's semantics.
"""
from __future__ import annotations

import torch

from . import ops

EPISODE_START = 6 * 3600 - 60   # SimulatorEnv._reset: set_time(3600 * 6 - 60)
EPISODE_END = 7 * 3600          # done = time > 7 * 3600


class SimEngine:
    def __init__(self, x, edge_index, edge_attr, Nmax, agent_features, *, congestion_constant=None, num_envs=None,
                 device="cuda", timestep=1, seed=0, plan=None, fused=True, env_base=0):
        """``x`` (N,F) or (B,N,F); ``agent_features`` (A,9) or (B,A,9). 2-D inputs are replicated ``num_envs`` times;
        3-D inputs are used in place (views are kept, so a caller-owned tensor keeps tracking the state).
        ``env_base`` (fused path): global id of this batch's environment 0 — the noise streams of environment b are those
        of global environment ``env_base + b`` under ``seed``, whichever batch (or rank) it is simulated in."""
        dev = torch.device(device)
        self.Nmax = int(Nmax)

        def batched(t, B):
            if t.dim() == 3:                        # caller-owned batch: used in place
                return t if t.is_cuda else t.to(dev, torch.float32)
            if B == 1 and t.is_cuda:                # single environment: a view, the caller's tensor tracks the state
                return t.unsqueeze(0)
            return t.to(dev, torch.float32).unsqueeze(0).repeat(B, 1, 1).contiguous()

        B = x.size(0) if x.dim() == 3 else int(num_envs or 1)
        self._x = batched(x, B)
        self._x_stale = False               # fused path: the packed state is ahead of the reference-layout tensor
        self._packed_stale = False          # unfused step(): x is ahead of the packed state
        self._last_step_time = 0.0
        self.B, self.N, self.F = self._x.shape
        self.agents = batched(agent_features, self.B)
        self.A = self.agents.size(1)
        self.device = self._x.device
        self.edge_index = edge_index.to("cpu", torch.int64)
        self.E = self.edge_index.size(1)
        self.plan = plan if plan is not None else ops.Plan(self.edge_index, self.N)
        self.ec = ops.EdgeConst(edge_attr, self.device)
        self.cc = None if congestion_constant is None else congestion_constant.to(self.device, torch.float32).contiguous()
        self.timestep = int(timestep)
        self.time = EPISODE_START
        self.seed = int(seed)
        self.noise_counter = 0
        # per-step scratch (allocated once; the library never allocates)
        self.chosen = torch.empty((self.B, self.N), dtype=torch.float32, device=self.device)
        self.popped = torch.empty((self.B, self.N), dtype=torch.uint8, device=self.device)
        self.ins_scratch = torch.empty((self.B, 2 * self.A), dtype=torch.int32, device=self.device)
        self.reward = torch.zeros(self.B, dtype=torch.float32, device=self.device)
        self.counts = torch.zeros((self.B, self.N), dtype=torch.float32, device=self.device)
        self.dtt = None
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)    # status word of the unfused kernels
        # fused fast path (csrc/fused.hip): packed hot records + agent SoA mirroring x / agents
        self.fs = ops.FusedState(self.plan, self.B, self.A, self.device, self.Nmax, env_base=env_base) if fused else None
        self.sample_counter = 0
        if self.fs is not None:
            self.resync()

    @property
    def x(self):
        """The state in the reference's layout (B, N, F). On the fused path the packed slot store is authoritative
        between frames; reading ``x`` exports it first (one launch), so what you see is always current."""
        if self._x_stale:
            ops.fused_export(self.plan, self.fs, self._x, self.Nmax, self._last_step_time)
            self._x_stale = False
        return self._x

    def resync(self):
        """Rebuild the fused side buffers from ``x`` / ``agents`` (after construction, reset, or external writes)."""
        if self.fs is not None:
            # tarl_fused_pack re-arms the sticky device status word: whatever an earlier rollout flagged and nobody has
            # read yet (only the non-blocking poll in flight) is raised here instead of being lost. Set-up path: the
            # host synchronisation costs nothing that matters.
            self.fs.check_flags()
            ops.fused_pack(self.plan, self.fs, self.x, self.Nmax, self.agents, self.cc, ec=self.ec)
            self._packed_stale = False
            self.fs.check_flags()

    # -- observation -------------------------------------------------------------------------------------------------
    @property
    def node_features(self):
        """(B, N, 7) view: the observation columns x[:, 3*Nmax:] (TransportationSimulator.state)."""
        return self.x[:, :, 3 * self.Nmax:]

    @property
    def static_node_features(self):
        """Same view WITHOUT refreshing the dynamic columns: for consumers that read only static columns (the live
        policy reads ROAD_INDEX alone) — avoids an export of the packed state."""
        return self._x[:, :, 3 * self.Nmax:]

    def refresh_counts(self):
        self.counts.copy_(self.x[:, :, 3 * self.Nmax + 1])
        return self.counts

    # -- control -----------------------------------------------------------------------------------------------------
    def reset(self):
        """SimulatorEnv._reset: zero FIFOs / counters, clear ON_WAY / DONE, clock = 6 h - 60 s."""
        if self.fs is not None and not self._packed_stale:
            ops.fused_reset(self.plan, self.fs, self.agents)     # packed state stays authoritative; x exported on demand
            self._x_stale = True
        else:
            ops.reset_state(self.x, self.Nmax, self.agents)
            self.resync()
        self._last_step_time = float(EPISODE_START)      # both branches: the first frame's prev_time is the reset clock
        self.time = EPISODE_START
        self.counts.zero_()
        self.reward.zero_()

    def step(self, *, choice=None, action_onehot=None, gumbel=None, want_dtt=False):
        """One env step for all B environments. Noise: explicit ``gumbel`` (B,E) or device Philox keyed by
        (seed, noise_counter). Returns (reward (B,), done: bool). ``self.counts`` holds the new per-node counts."""
        t = float(self.time)
        self._packed_stale = self.fs is not None
        ops.apply_action(self.plan, self.x, self.Nmax, action_onehot=action_onehot, choice=choice)
        self.noise_counter += 1
        self.dtt, _ = ops.core_step(self.plan, self.x, self.Nmax, self.ec, t, congestion_constant=self.cc, gumbel=gumbel,
                                    seed=self.seed, counter=self.noise_counter, want_dtt=want_dtt, chosen=self.chosen,
                                    popped=self.popped, status=self.status)
        ops.withdraw_step(self.plan, self.x, self.Nmax, self.agents, t, want_mask=False)
        ops.insert_step(self.x, self.Nmax, self.agents, t, congestion_constant=self.cc, scratch=self.ins_scratch,
                        reward=self.reward, counts=self.counts)
        self.time += self.timestep           # the reference's equality test compares a view with itself
        return self.reward, self.time > EPISODE_END

    # -- fused fast path: 1 + 3 launches per frame, outputs written straight into caller buffers ---------------------------
    def prepare_policy(self, emb, temperature=1.0):
        """Evaluate the live policy's distribution tables; call once per parameter update."""
        self.tables = ops.fused_policy_prepare(self.plan, self.fs, emb, temperature, getattr(self, "tables", None))

    def check_flags(self):
        """Raise :class:`TarlError` if a kernel flagged a domain exit since the last pack (one host synchronisation)."""
        ops.raise_on_flags(int(self.status.item()))
        if self.fs is not None:
            self.fs.check_flags()

    def frame_fused(self, *, choice=None, log_prob=None, entropy=None, reward=None, counts=None, uniform=None,
                    gumbel=None, dtt=None, popped=None, withdrawn=None, action=None, skip_choice=False):
        """One collector frame (sample + log_prob + choice phase + env step) for all B environments in 4 launches.
        ``choice`` (N, B) int32 and ``counts`` (N, B) fp32 are env-minor. ``action``: an externally sampled action (B, N)
        int32 edge ids (-1: none) instead of the live policy's own sample (state-dependent policies); ``skip_choice``: the
        action is already in the packed state's SELECTED_ROAD bytes (ops.graphdist_rollout(sel8=...)). Returns done."""
        if self._packed_stale:
            self.resync()
        self.sample_counter += 1
        self.noise_counter += 1
        self._x_stale = True
        prev = self._last_step_time
        self._last_step_time = float(self.time)
        tables = getattr(self, "tables", None)
        if action is not None:
            ops.fused_apply_choice(self.plan, self.fs, action)
            tables = None
        if skip_choice:
            tables = None
        ops.fused_frame(self.plan, self.fs, tables, self.agents, self.ec, float(self.time), prev_time=prev,
                        use_cong=self.cc is not None, uniform=uniform, policy_seed=self.seed ^ 0x5DEECE66D,
                        policy_counter=self.sample_counter, gumbel=gumbel, seed=self.seed, counter=self.noise_counter,
                        dtt=dtt, popped=popped, withdrawn=withdrawn, scratch=self.ins_scratch, choice=choice,
                        log_prob=log_prob, entropy=entropy, reward=self.reward if reward is None else reward,
                        counts=counts)
        self.time += self.timestep
        return self.time > EPISODE_END

    @property
    def env_rollout_supported(self):
        """True when one environment's hot records fit a CU's LDS (tarl_rollout_env)."""
        return self.fs is not None and ops.rollout_env_supported(self.plan)

    def _rollout(self, fn, env_minor, T, choice, log_prob, reward, counts, metrics_envs, dtt_node, events, leg, check):
        if self._packed_stale:
            self.resync()
        shp = (T + 1, self.N, self.B) if env_minor else (T + 1, self.B, self.N)
        if counts.dtype != torch.uint8 or tuple(counts.shape) != shp or not counts.is_contiguous():
            raise ValueError(f"counts must be a contiguous uint8 {shp} tensor")
        times = []
        t_clock = self.time
        for _ in range(T):
            times.append(float(t_clock))
            t_clock += self.timestep
        self._x_stale = True
        self._times_dev = fn(self.plan, self.fs, self.tables, self.agents, self.ec, times, use_cong=self.cc is not None,
                             prev_time=self._last_step_time, policy_seed=self.seed ^ 0x5DEECE66D,
                             policy_counter0=self.sample_counter + 1, seed=self.seed, counter0=self.noise_counter + 1,
                             scratch=self.ins_scratch, choice=choice, log_prob=log_prob, reward=reward,
                             counts=counts[1:], metrics_envs=metrics_envs, dtt_node=dtt_node, events=events, leg=leg)
        self.sample_counter += T
        self.noise_counter += T
        self._last_step_time = times[-1]
        self.time = t_clock
        times.append(float(self.time))
        if check:
            self.check_flags()
        return times

    def rollout_policy(self, T, weights, *, bf16=False, temperature, policy_seed, policy_counter0, choice8, log_prob, reward,
                       counts, keep=None, obs_keep=None, metrics_envs=0, dtt_node=None, events=None, leg=None,
                       check=True, precision=None):
        """``T`` frames under the per-edge MLP head (``weights``: ops.EdgeMlpWeights) in one foreign call: per frame
        observation -> logits -> GraphDistribution sample + log-prob -> the simulation frame. ``choice8`` (T,B,N) uint8
        (ENV-MAJOR rank bytes), ``counts`` (T+1,N,B) uint8 (counts[t + 1] = after frame t), ``log_prob`` / ``reward`` (T,B).
        Frame t draws its action with Philox counter ``policy_counter0 + t``. ``precision`` of the rollout's logits: "fp32"
        (fp32 MFMA), "bf16" (= ``bf16=True``) or "x3" (fp32-accurate on the bf16 pipe). Returns the list of clock values."""
        if self._packed_stale:
            self.resync()
        if counts.dtype != torch.uint8 or tuple(counts.shape) != (T + 1, self.N, self.B) or not counts.is_contiguous():
            raise ValueError(f"counts must be a contiguous uint8 {(T + 1, self.N, self.B)} tensor")
        times = []
        t_clock = self.time
        for _ in range(T):
            times.append(float(t_clock))
            t_clock += self.timestep
        self._x_stale = True
        ops.fused_rollout_policy(self.plan, self.fs, self._x, self.agents, self.ec, weights, times,
                                 use_cong=self.cc is not None, bf16=bf16, temperature=temperature, policy_seed=policy_seed,
                                 policy_counter0=policy_counter0, seed=self.seed, counter0=self.noise_counter + 1,
                                 scratch=self.ins_scratch, prev_time=self._last_step_time, keep=keep, obs_keep=obs_keep,
                                 choice8=choice8, log_prob=log_prob, reward=reward, counts=counts[1:],
                                 metrics_envs=metrics_envs, dtt_node=dtt_node, events=events, leg=leg, precision=precision)
        self.sample_counter += T
        self.noise_counter += T
        self._last_step_time = times[-1]
        self.time = t_clock
        times.append(float(self.time))
        if check:
            self.check_flags()
        return times

    def decode_rollout(self, env_minor, *, choice=None, counts=None):
        """The rollout's byte buffers in the formats of the unfused entry points, ENV-MAJOR: ``choice`` (T,N,B) / (T,B,N)
        uint8 -> (T,B,N) int32 edge ids (-1: none); ``counts`` (T',N,B) / (T',B,N) uint8 -> (T',B,N) fp32."""
        out = []
        for buf, key in ((choice, "choice"), (counts, "counts")):
            if buf is None:
                out.append(None)
                continue
            Tn = buf.size(0)
            r = ops.rollout_gather(self.plan, Tn, self.B, env_minor, **{key: buf})[0 if key == "choice" else 1]
            out.append(r.view(Tn, self.B, self.N))
        return out

    def rollout_env(self, T, *, choice, log_prob, reward, counts, metrics_envs=0, dtt_node=None, events=None, leg=None,
                    check=True):
        """Same frames as :meth:`rollout_fused` through ``tarl_rollout_env`` (one workgroup per environment, LDS-resident
        records, a single launch); the buffers are ENV-MAJOR: ``choice`` (T,B,N) uint8, ``counts`` (T+1,B,N) uint8 with
        counts[t + 1] = the counts after frame t; ``log_prob`` (T,B) or None; ``reward`` (T,B)."""
        return self._rollout(ops.rollout_env, False, T, choice, log_prob, reward, counts, metrics_envs, dtt_node, events,
                             leg, check)

    def rollout_fused(self, T, *, choice, log_prob, reward, counts, metrics_envs=0, dtt_node=None, events=None, leg=None,
                      check=True):
        """``T`` consecutive frames with the outputs of frame t written to ``choice[t]`` (T,N,B) uint8 (rank of the chosen
        out-edge; ``ops.rollout_gather`` turns it into edge ids), ``log_prob[t]`` (T,B) or None, ``reward[t]`` (T,B),
        ``counts[t + 1]`` (T+1,N,B) uint8; optional per-step logs: ``leg`` (T,B,2) int32 {departed, arrived}, and for the
        first ``metrics_envs`` environments ``dtt_node`` (T,N,m) fp32 / ``events`` (T,N,m) uint8. Same as T calls of
        :meth:`frame_fused` with the per-frame Python overhead removed. ``check``: read the device status word afterwards
        (one synchronisation) and raise on a domain exit. Returns the list of clock values."""
        return self._rollout(ops.fused_rollout, True, T, choice, log_prob, reward, counts, metrics_envs, dtt_node, events,
                             leg, check)
