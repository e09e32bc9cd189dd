"""VecPPOTrainer — the reference's ``ppo_train`` loop (src/rl/ppo_trainer.py:21-39,129-145) on B vectorised
environments per GPU, every arithmetic step a HIP kernel behind the C ABI:

  HOT LOOP A (T frames): policy logits -> GraphDistribution softmax / sample / log_prob -> env step
  HOT LOOP B (num_epochs): critic over all frames (MFMA) -> GAE -> advantage normalisation (global statistics) ->
                           minibatch of ``sub_batch_size`` frames -> clipped PPO loss fwd+bwd -> one gradient
                           all-reduce (RCCL) -> fused Adam.

Semantics kept from the reference: ONE collector batch per call (``total_frames == frames_per_batch``), the
environment is reset at the start of the batch (``reset_at_each_iter=True``), GAE is recomputed every epoch with the
current critic, one minibatch + one Adam step per epoch, loss = objective + critic + entropy, no gradient clipping.
Multi-GPU: one process per GPU, each with its own environments and minibatch; gradients are averaged.
"""
from __future__ import annotations

import torch

from . import dist_utils, ops
from .flatparams import FlatParams


class StageTimer:
    """HIP-event timing of the update's stages (``bench.py``'s ``update_path`` object): a pair of events on the launch
    stream (torch's current stream — the stream every ``ops`` call enqueues on) around each stage; ``ms()`` synchronises
    and returns {stage: (summed ms, calls)}."""

    def __init__(self):
        self.ev = {}

    class _Span:
        def __init__(self, timer, name):
            self.t, self.name = timer, name

        def __enter__(self):
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

        def __exit__(self, *exc):
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            self.t.ev.setdefault(self.name, []).append((self.a, b))

    def __call__(self, name):
        return StageTimer._Span(self, name)

    def ms(self):
        torch.cuda.synchronize()
        return {k: (sum(a.elapsed_time(b) for a, b in v), len(v)) for k, v in self.ev.items()}


class NoStageTimer:
    def __call__(self, name):
        return self

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


class VecPPOTrainer:
    def __init__(self, engine, emb_param, critic_params, *, rollout_steps, num_epochs=1, sub_batch_size=32, lr=1e-3,
                 gamma=0.99, lmbda=0.95, clip_epsilon=0.2, entropy_coef=0.01, critic_coef=1.0, temperature=1.0,
                 extra_params=(), seed=0, lazy_log_prob=False, rank_offset=True, rollout=None, metrics_envs=1,
                 policy="embedding", edge_mlp_params=None, policy_bf16=False, policy_precision=None):
        """``emb_param``: nn.Parameter (num_nodes, 1) — MPNNPolicyNet.nodes_embedding.weight;
        ``critic_params``: [w1 (64,N+1), b1, w2 (64,64), b2, w3 (1,64), b3] — MPNNValueNetSimple.final_mlp.{0,2,4};
        ``extra_params``: further actor/critic parameters that never receive gradient on the live path (the dormant
        edge MLPs) — they sit in the flat buffer so that the optimiser covers ``loss_module.parameters()`` like the
        reference's does."""
        self.eng = engine
        # policy: "embedding" = the reference's live head (logit = embedding of the target road: state-independent, its
        # distribution is tabulated once per update); "edge_mlp" = the per-edge MLP head it carries as parameters
        # (src/agents/mpnn_agent.py:35-41,227-231): state-DEPENDENT, so every frame evaluates observation -> MLP (MFMA) ->
        # segment softmax -> sample -> log-prob before the simulation step. ``edge_mlp_params`` = [w1, b1, w2, b2, w3, b3]
        # (edge_mlp.{0,2,4}.{weight,bias}; they must also be in ``extra_params`` so that they live in the flat buffer);
        # ``policy_bf16``: rollout logits on the bf16 MFMA path (the update always runs in fp32). ``policy_precision``
        # ("fp32" | "bf16" | "x3") names the rollout kernel outright; the default for a non-bf16 policy is "x3": logits
        # within a few fp32 ulp of the fp32 MFMA kernel's (the north star's 1e-4 contract) at 2.7x its matrix rate.
        self.policy = policy
        self.policy_precision = policy_precision or ("bf16" if policy_bf16 else "x3")
        if self.policy_precision not in ops.EDGE_MLP_PRECISIONS:
            raise ValueError(f"policy_precision must be one of {ops.EDGE_MLP_PRECISIONS}")
        self.policy_bf16 = self.policy_precision == "bf16"
        self.edge_mlp_params = list(edge_mlp_params) if edge_mlp_params is not None else None
        if policy == "edge_mlp":
            if engine.fs is None or self.edge_mlp_params is None or len(self.edge_mlp_params) != 6:
                raise ValueError("policy='edge_mlp' needs the fused engine and the six edge_mlp parameter tensors")
            ids = {id(p) for p in extra_params}
            if not all(id(p) in ids for p in self.edge_mlp_params):
                raise ValueError("edge_mlp_params must be part of extra_params (the optimiser's flat buffer)")
        elif policy != "embedding":
            raise ValueError("policy must be 'embedding' or 'edge_mlp'")
        # lazy_log_prob: do not produce sample_log_prob for every collected frame (as the reference's collector does)
        # but only, exactly, for the frames a minibatch actually reads. Same training result; off by default so that a
        # frame does everything the reference's frame does.
        self.lazy_log_prob = bool(lazy_log_prob)
        self.T = int(rollout_steps)
        self.num_epochs = int(num_epochs)
        self.M = int(sub_batch_size)
        self.lr, self.gamma, self.lmbda = lr, gamma, lmbda
        self.clip_epsilon, self.entropy_coef, self.critic_coef, self.temperature = clip_epsilon, entropy_coef, critic_coef, temperature
        self.rank, self.world = dist_utils.world()
        self.emb_param = emb_param
        self.critic_params = list(critic_params)
        self.flat = FlatParams([emb_param] + self.critic_params + list(extra_params), device=engine.device)
        # replicas start identical: rank 0's initial weights win
        dist_utils.broadcast_(self.flat.flat, src=0)
        B, N, dev = engine.B, engine.N, engine.device
        # which rollout kernel family: "env" = one workgroup per environment, records in LDS, one launch for all frames
        # (tarl_rollout_env; env-major buffers), "frames" = four env-minor launches per frame (tarl_fused_rollout).
        # Default (TARL_ROLLOUT or "auto"): "env" when the graph fits a CU's LDS AND B * N <= 800k (node, environment)
        # pairs — measured crossover on MI355X: N = 256: env 2.1x at B = 1, 2.7x at B = 256, 1.5x at B = 2048, tie at
        # 8192; N = 1024, B = 1024: frames 1.24x; N = 2500: env +19 % at B = 256, frames +13 % at B = 512. One environment
        # keeps one CU busy for the whole frame; the four-launch path spreads the same work over the chip but needs
        # enough environments to fill its lanes and hide four dependent launches.
        import os
        mode = rollout or os.environ.get("TARL_ROLLOUT", "auto")
        if engine.fs is None:
            mode = "unfused"
        elif policy == "edge_mlp":
            mode = "frames+policy"      # per-frame policy evaluation in front of the four-launch frame
        elif mode == "auto":
            mode = "env" if (engine.env_rollout_supported and engine.B * engine.N <= 800_000) else "frames"
        elif mode == "env" and not engine.env_rollout_supported:
            raise ValueError("rollout='env' needs a graph whose hot records fit the LDS (tarl_rollout_env_supported)")
        self.rollout = mode
        self.layout_tag = ops.FUSED_LAYOUT
        # rollout buffers, written directly by the kernels: ENV-MINOR ([frame][node][env]) for "frames"
        self.env_minor = mode in ("frames", "frames+policy")
        shp = (lambda t: (t, N, B)) if self.env_minor else (lambda t: (t, B, N))
        # fused rollouts write one BYTE per (frame, node, env): the count and the rank of the chosen out-edge
        byte = mode in ("frames", "env", "frames+policy")
        self.counts = torch.zeros(shp(self.T + 1), dtype=torch.uint8 if byte else torch.float32, device=dev)
        # "frames+policy": the per-frame sampler (one workgroup per environment) writes its rank bytes env-major
        self.choice = torch.zeros((self.T, B, N) if mode == "frames+policy" else shp(self.T),
                                  dtype=torch.uint8 if byte else torch.int32, device=dev)
        # per-step logs of SimulatorEnv._step, accumulated on the device by the rollout kernels: the leg histogram's
        # (departed, arrived) per frame for every environment, delta_travel_time / pop + withdraw masks per node for the
        # first ``metrics_envs`` environments (the reference logs them for its single environment)
        self.metrics_envs = min(int(metrics_envs), B) if byte else 0
        m = self.metrics_envs
        self.leg = torch.zeros((self.T, B, 2), dtype=torch.int32, device=dev) if byte else None
        self.dtt_node = torch.zeros(shp(self.T)[:1] + ((N, m) if self.env_minor else (m, N)), dtype=torch.float32,
                                    device=dev) if m else None
        self.events = torch.zeros_like(self.dtt_node, dtype=torch.uint8) if m else None
        self._flag_host = torch.zeros(1, dtype=torch.int32).pin_memory() if byte else None
        self._flag_event = None
        self.logp = torch.zeros((self.T, B), dtype=torch.float32, device=dev)
        self.reward = torch.zeros((self.T, B), dtype=torch.float32, device=dev)
        self.times = torch.zeros(self.T + 1, dtype=torch.float32, device=dev)
        self.values = torch.zeros((self.T + 1, B), dtype=torch.float32, device=dev)
        # every rank draws its own minibatches / action noise; rank_offset=False (test hook) makes replicas identical
        off = self.rank if rank_offset else 0
        # minibatch draw: M distinct frames out of T * B (the reference's SamplerWithoutReplacement hands out sub-batches of a
        # shuffled buffer). Drawn on the HOST in O(M) (numpy's Floyd sampler) and copied behind the launches already queued:
        # a device randperm of T * B = 4.2 M keys is seven radix-sort passes + key generation per optimiser step (0.3 ms and a
        # dozen launches per iteration in profiles/r03_default_kernel_stats.csv) for 32 indices
        import numpy as np
        self.np_rng = np.random.Generator(np.random.Philox(key=int(seed) + 7919 * off))
        self.seed = int(seed) + off
        self.sample_counter = 0
        self.last = {}
        self.done_frames = torch.zeros(self.T, dtype=torch.bool)
        self.done_mask = None
        self.obs_idx = None
        self.stage = NoStageTimer()          # bench.py swaps in a StageTimer for its update_path object

    # -- views of the live parameters -----------------------------------------------------------------------------------
    def _emb(self):
        return self.emb_param.data.reshape(-1)

    def _critic(self):
        w1, b1, w2, b2, w3, b3 = (p.data for p in self.critic_params)
        return ops.CriticWeights(w1, b1, w2, b2, w3.reshape(-1), b3)

    def _edge_mlp(self):
        return ops.EdgeMlpWeights(*(p.data for p in self.edge_mlp_params))

    # -- HOT LOOP A, state-dependent policy --------------------------------------------------------------------------------
    @torch.no_grad()
    def _collect_edge_mlp(self):
        """T frames with the per-edge MLP policy: per frame observation (from the packed state) -> edge MLP on MFMA ->
        GraphDistribution sample + log_prob (one launch) -> the three-launch simulation frame with that action, all
        queued by one foreign call per episode segment (tarl_fused_rollout_policy). Nothing is hoisted. The update only
        ever reads the observations of its minibatch frames, and those frames are a random draw that does not depend on
        the data: the draw is made up front and only their observations are kept."""
        from .engine import EPISODE_END
        eng = self.eng
        T, B, N = self.T, eng.B, eng.N
        eng.reset()
        self.counts[0].zero_()
        M = min(self.M, T * B)
        if self.obs_idx is not None:                    # test hook: these frames instead of a random draw
            host = [self.obs_idx.cpu()]
        else:
            host = [self.draw_frames(T * B, M, device=False) for _ in range(self.num_epochs)]
        self._mb_idx = [h.pin_memory().to(eng.device, non_blocking=True) for h in host]
        flat = torch.cat(host)      # (the frame list stays on the host: no device round trip before the rollout)
        order = torch.argsort(flat, stable=True)
        t_sorted = torch.div(flat[order], B, rounding_mode="floor").tolist()
        keep_env = (flat[order] % B).to(torch.int32).pin_memory().to(eng.device, non_blocking=True)
        keep_slot = order.to(torch.int32).pin_memory().to(eng.device, non_blocking=True)
        self.obs_mb = torch.empty((flat.numel(), N, 16), dtype=torch.float32, device=eng.device)
        w = self._edge_mlp()
        pseed = self.seed ^ 0x5DEECE66D
        m = self.metrics_envs
        host_times, done, pos, t0 = [], [False] * T, 0, 0
        while t0 < T:
            # frames until the episode ends (clock past 7 h) or the batch is full: one foreign call
            seg, clock = 0, float(eng.time)
            while t0 + seg < T:
                seg += 1
                clock += eng.timestep
                if clock > EPISODE_END:
                    break
            seg_lo = pos                     # the kept (frame, environment) pairs are sorted by frame
            rel = [0]
            for t in range(t0, t0 + seg):
                while pos < len(t_sorted) and t_sorted[pos] == t:
                    pos += 1
                rel.append(pos - seg_lo)
            keep = (rel, keep_env[seg_lo:pos], keep_slot[seg_lo:pos]) if pos > seg_lo else None
            sl = slice(t0, t0 + seg)
            times = eng.rollout_policy(seg, w, precision=self.policy_precision, temperature=self.temperature, policy_seed=pseed,
                                       policy_counter0=self.sample_counter + 1, choice8=self.choice[sl],
                                       log_prob=self.logp[sl], reward=self.reward[sl],
                                       counts=self.counts[t0:t0 + seg + 1], keep=keep, obs_keep=self.obs_mb,
                                       metrics_envs=m, dtt_node=self.dtt_node[sl] if m else None,
                                       events=self.events[sl] if m else None, leg=self.leg[sl], check=False)
            host_times += times[:-1]
            self.sample_counter += seg
            t0 += seg
            if eng.time > EPISODE_END:
                done[t0 - 1] = True
                if t0 < T:
                    eng.reset()
                    self.counts[t0].zero_()
        host_times.append(float(eng.time))
        self.times.copy_(torch.tensor(host_times, dtype=torch.float32))
        self.done_frames = torch.tensor(done, dtype=torch.bool)
        self.done_mask = (self.done_frames.to(eng.device, torch.uint8).view(T, 1).expand(T, B).contiguous()
                          if any(done) else None)
        self._flag_host.copy_(eng.fs.flags, non_blocking=True)      # polled at the next collect / checked at the end
        self._flag_event = torch.cuda.Event()
        self._flag_event.record()
        self._epoch = 0
        return T * B

    # -- HOT LOOP A -------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def collect(self):
        """T frames for all B environments (SyncDataCollector with reset_at_each_iter=True, ExplorationType.RANDOM).
        An episode that ends inside the batch (clock past 7 h: ``done = terminated``, src/reinforcement_learning.py:273-276)
        is followed by a reset, like the collector's auto-reset: the rollout is split at that frame, the reset observation
        becomes the next frame's observation and ``done_frames`` marks the frame for GAE."""
        from .engine import EPISODE_END
        if self.policy == "edge_mlp":
            return self._collect_edge_mlp()
        eng = self.eng
        eng.reset()
        emb = self._emb()
        T = self.T
        host_times, done = [], [False] * T
        if eng.fs is not None:
            # fused path: every output lands directly in the rollout buffers (no copies)
            self.counts[0].zero_()
            eng.prepare_policy(emb, self.temperature)      # once per parameter update, not per frame
            # sample_log_prob is only ever read for the <= sub_batch_size frames of each minibatch: keep the behaviour
            # policy's parameters and evaluate it (exactly, with the unfused kernel) for those frames at update time
            self.emb_rollout = emb.clone()
            self.poll_flags()
            run = eng.rollout_env if self.rollout == "env" else eng.rollout_fused
            t = 0
            while t < T:
                # frames until the episode ends (the frame whose step pushes the clock past EPISODE_END is the last)
                left = (EPISODE_END - eng.time) // eng.timestep + 1
                n = int(min(T - t, max(1, left)))
                sl = slice(t, t + n)
                host_times += run(n, choice=self.choice[sl], log_prob=None if self.lazy_log_prob else self.logp[sl],
                                  reward=self.reward[sl], counts=self.counts[t:t + n + 1],
                                  metrics_envs=self.metrics_envs,
                                  dtt_node=None if self.dtt_node is None else self.dtt_node[sl],
                                  events=None if self.events is None else self.events[sl], leg=self.leg[sl],
                                  check=False)[:-1]
                t += n
                if eng.time > EPISODE_END:
                    done[t - 1] = True
                    if t < T:
                        eng.reset()
                        self.counts[t].zero_()         # the reset observation is frame t's observation
            host_times.append(float(eng.time))
            # the device status word travels to pinned host memory behind the rollout; it is looked at when it has
            # arrived (no stall of the launch pipeline) and by check_flags() at the caller's synchronisation points
            self._flag_host.copy_(eng.fs.flags, non_blocking=True)
            self._flag_event = torch.cuda.Event()
            self._flag_event.record()
        else:
            for t in range(T):
                self.counts[t].copy_(eng.counts)
                host_times.append(float(eng.time))
                logits = ops.policy_edge_logits(eng.plan, eng.node_features, emb)
                proba = ops.graphdist_softmax(eng.plan, logits, self.temperature)
                self.sample_counter += 1
                _, choice = ops.graphdist_sample(eng.plan, proba, seed=self.seed ^ 0x5DEECE66D,
                                                 counter=self.sample_counter, want_onehot=False, want_choice=True)
                lp, _ = ops.graphdist_logprob_entropy(eng.plan, proba, choice=choice, want_entropy=False)
                self.choice[t].copy_(choice)
                self.logp[t].copy_(lp)
                reward, is_done = eng.step(choice=choice)
                self.reward[t].copy_(reward)
                if is_done:
                    done[t] = True
                    if t + 1 < T:
                        eng.reset()
            self.counts[T].copy_(eng.counts)
            host_times.append(float(eng.time))
        self.times.copy_(torch.tensor(host_times, dtype=torch.float32))
        self.done_frames = torch.tensor(done, dtype=torch.bool)
        # (T, B) mask for tarl_gae, only when an episode actually ended inside the batch
        self.done_mask = (self.done_frames.to(eng.device, torch.uint8).view(T, 1).expand(T, eng.B).contiguous()
                          if any(done) else None)
        return T * eng.B

    def draw_frames(self, n, M, device=True):
        """M distinct flat frame indices t * B + b out of n, uniform, int64 (host draw; device=True: asynchronous copy)."""
        idx = torch.from_numpy(self.np_rng.choice(n, size=M, replace=False, shuffle=True).astype("int64"))
        return idx.pin_memory().to(self.eng.device, non_blocking=True) if device else idx

    def poll_flags(self):
        """Raise if a finished rollout flagged a domain exit (non-blocking)."""
        if self._flag_event is not None and self._flag_event.query():
            self._flag_event = None
            if int(self._flag_host[0]) != 0:
                self.eng.check_flags()

    def check_flags(self):
        """Blocking form of :meth:`poll_flags` (call at a synchronisation point, e.g. the end of training)."""
        self._flag_event = None
        self.eng.check_flags()

    # -- HOT LOOP B -------------------------------------------------------------------------------------------------------
    def advantages(self):
        """GAE(gamma, lmbda, average_gae=True) with the current critic over all (T+1)*B observations."""
        eng = self.eng
        T, B, N = self.T, eng.B, eng.N
        cw = self._critic()
        with self.stage("critic_all_frames"):
            if self.env_minor and B % 128 == 0:
                v = ops.critic_forward_slabs(cw, self.counts, self.times)           # reads [frame][node][env] bytes as is
            elif self.env_minor and self.counts.dtype == torch.uint8:    # odd batch sizes: the count bytes as fp32 rows first
                _, rows = ops.rollout_gather(eng.plan, T + 1, B, True, counts=self.counts)
                v, _, _ = ops.critic_forward(cw, rows, self.times, rows_per_time=B)
            elif self.env_minor:
                rows = self.counts.permute(0, 2, 1).contiguous().view((T + 1) * B, N)
                v, _, _ = ops.critic_forward(cw, rows, self.times, rows_per_time=B)
            else:
                v, _, _ = ops.critic_forward(cw, self.counts.view((T + 1) * B, N), self.times, rows_per_time=B)
        self.values = v.view(T + 1, B)
        # done = terminated (src/reinforcement_learning.py:296): no bootstrap across an episode end
        with self.stage("gae"):
            adv, target = ops.gae(self.reward, self.values[:T], self.values[1:], done=self.done_mask,
                                  terminated=self.done_mask, gamma=self.gamma, lmbda=self.lmbda)
            stats = ops.advantage_stats(adv)
            dist_utils.allreduce_sum_(stats)          # global mean / std over all ranks' frames
            ops.advantage_normalize_(adv, stats)
        return adv, target

    def minibatch_step(self, adv, target, idx=None):
        """One minibatch + one Adam step. ``idx`` (test hook): flat frame indices t * B + b instead of a random draw."""
        eng = self.eng
        T, B, N, E = self.T, eng.B, eng.N, eng.E
        M = min(self.M, T * B)
        if self.policy == "edge_mlp":       # the draw was made before the rollout (its observations were kept)
            k = self._epoch
            idx = self._mb_idx[k]
            M = idx.numel()
            off = sum(i.numel() for i in self._mb_idx[:k])
            obs_mb = self.obs_mb[off:off + M]
            self._epoch += 1
        elif idx is None:
            idx = self.draw_frames(T * B, M)
        else:
            idx = idx.to(eng.device)
            M = idx.numel()
        st = self.stage
        with st("minibatch_gather"):
            if self.policy == "edge_mlp":
                _, counts_mb = ops.rollout_gather(eng.plan, T, B, True, idx, counts=self.counts[:T])     # env-minor bytes
                choice_mb, _ = ops.rollout_gather(eng.plan, T, B, False, idx, choice=self.choice)        # env-major bytes
            elif self.rollout == "unfused":
                counts_mb = self.counts[:T].view(T * B, N).index_select(0, idx)
                choice_mb = self.choice.view(T * B, N).index_select(0, idx)
            else:   # one launch: the sampled frames' action bytes -> edge ids, count bytes -> fp32 rows
                choice_mb, counts_mb = ops.rollout_gather(eng.plan, T, B, self.env_minor, idx, choice=self.choice,
                                                          counts=self.counts[:T])
            nf = eng.static_node_features[:1].expand(M, N, 7)
            if eng.fs is not None and self.lazy_log_prob:   # behaviour log-prob of the sampled frames, rollout-time parameters
                p_old = ops.graphdist_softmax(eng.plan, ops.policy_edge_logits(eng.plan, nf, self.emb_rollout),
                                              self.temperature)
                lp_old, _ = ops.graphdist_logprob_entropy(eng.plan, p_old, choice=choice_mb, want_entropy=False)
            else:
                lp_old = self.logp.view(-1).index_select(0, idx)
            adv_mb = adv.view(-1).index_select(0, idx)
            tgt_mb = target.view(-1).index_select(0, idx)
            time_mb = self.times[:T].index_select(0, torch.div(idx, B, rounding_mode="floor"))
        # actor forward (the live policy reads only the static ROAD_INDEX column: broadcast one observation over M rows)
        with st("actor_logits_fwd"):
            if self.policy == "edge_mlp":
                wmlp = self._edge_mlp()
                logits = ops.policy_edge_mlp(eng.plan, obs_mb, eng.ec, wmlp)          # fp32 MFMA
            else:
                logits = ops.policy_edge_logits(eng.plan, nf, self._emb())
        with st("graphdist_fwd"):
            proba = ops.graphdist_softmax(eng.plan, logits, self.temperature)
            lp_new, ent = ops.graphdist_logprob_entropy(eng.plan, proba, choice=choice_mb)
        cw = self._critic()
        with st("critic_fwd"):
            # split-K while the minibatch is far from filling the chip with 128-row MFMA tiles (M = 4 096: 32 workgroups walking
            # all N columns alone took 743 us, bench.py's update_path; spread over the columns: see DESIGN §4.7)
            value, h1, h2 = ops.critic_forward(cw, counts_mb, time_mb, 1, keep_hidden=True, split_k=M <= 16384)
        scale = 1.0 / self.world
        with st("ppo_loss"):
            out, g_lp, g_ent, g_val = ops.ppo_loss(lp_new, lp_old, adv_mb, value, tgt_mb, ent,
                                                   clip_epsilon=self.clip_epsilon, entropy_coef=self.entropy_coef,
                                                   critic_coef=self.critic_coef, grad_scale=scale)
        # backward
        self.flat.zero_grad()
        with st("graphdist_bwd"):
            g_logits = ops.graphdist_logprob_entropy_bwd(eng.plan, proba, self.temperature, choice=choice_mb,
                                                         grad_log_prob=g_lp, grad_entropy=g_ent, log_prob_fwd=lp_new)
        with st("actor_logits_bwd"):
            if self.policy == "edge_mlp":
                gm = [self.flat.grad_view(p) for p in self.edge_mlp_params]
                ops.policy_edge_mlp_bwd(eng.plan, obs_mb, eng.ec, wmlp, g_logits,
                                        (gm[0], gm[1], gm[2], gm[3], gm[4].view(-1), gm[5]))
            else:
                g_emb = ops.policy_edge_logits_bwd(eng.plan, nf, g_logits, self.emb_param.numel())
                self.flat.grad_view(self.emb_param).add_(g_emb.view_as(self.emb_param))
        gw = [self.flat.grad_view(p) for p in self.critic_params]
        with st("critic_bwd"):
            ops.critic_backward(cw, counts_mb, time_mb, 1, h1, h2, g_val,
                                (gw[0], gw[1], gw[2], gw[3], gw[4].view(-1), gw[5]))
        with st("grad_allreduce"):
            self.flat.allreduce_grads()               # ONE all-reduce of the fused gradient buffer (RCCL over xGMI)
        self.last_grad = self.flat.grad.clone() if getattr(self, "keep_grad", False) else None
        with st("adam"):
            self.flat.adam_step(lr=self.lr)
        return out

    def update(self):
        out = None
        for _ in range(self.num_epochs):
            adv, target = self.advantages()
            out = self.minibatch_step(adv, target)
        self.last = {"losses": out}
        return out

    def train_iteration(self):
        frames = self.collect()
        self.update()
        return frames
