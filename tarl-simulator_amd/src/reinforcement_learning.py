"""GraphDistribution and SimulatorEnv (reference: src/reinforcement_learning.py).

GraphDistribution: one categorical per source node over its out-edges. The reference re-sorts the static
``edge_index`` and rebuilds boundary masks on every construction; here the structure is a cached static plan and the
arithmetic is ``tarl_graphdist_*`` (segment softmax, inverse-CDF sample reproducing the reference's global
double-accumulated cumsum + fp32 rebase, log_prob / entropy with a hand-written backward).

SimulatorEnv: the torchrl-shaped environment (``_reset`` / ``_step`` / ``rollout``) around TransportationSimulator;
one step enqueues ~10 kernels and never synchronises the host.
"""
from __future__ import annotations

import time

import torch
from torch.distributions import Distribution

from ._compat import EnvBase, Spec, TensorDict, TensorDictBase, cached_plan, require_cuda
from .transportation_simulator import TransportationSimulator


class _LogProbEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, plan, temperature, action):
        from tarl_hip import ops
        l2 = logits.detach().contiguous()
        proba = ops.graphdist_softmax(plan, l2, temperature)
        lp, ent = ops.graphdist_logprob_entropy(plan, proba, action_onehot=action, want_logprob=action is not None)
        ctx.plan, ctx.temperature, ctx.action, ctx.proba, ctx.lp = plan, temperature, action, proba, lp
        if lp is None:
            lp = torch.zeros_like(ent)
        return lp, ent

    @staticmethod
    def backward(ctx, g_lp, g_ent):
        from tarl_hip import ops
        g = ops.graphdist_logprob_entropy_bwd(ctx.plan, ctx.proba, ctx.temperature, action_onehot=ctx.action,
                                              grad_log_prob=None if ctx.action is None else g_lp.contiguous(),
                                              grad_entropy=g_ent.contiguous(), log_prob_fwd=ctx.lp)
        return g, None, None, None


class GraphDistribution(Distribution):
    arg_constraints = {}

    def __init__(self, logits: torch.Tensor, edge_index: torch.Tensor, temperature: float = 1.0):
        super().__init__(validate_args=False)
        from tarl_hip import ops
        require_cuda(logits, "logits")
        self.logits = logits
        self.edge_index = edge_index
        self.temperature = float(temperature)
        self.plan = cached_plan(edge_index, None)   # node count = max id + 1, derived once per topology
        self.nb_nodes = self.plan.num_groups
        self.proba = ops.graphdist_softmax(self.plan, logits.detach().contiguous(), self.temperature)
        self._mode = None
        self._uniform = None

    @property
    def deterministic_sample(self):
        if self._mode is None:
            from tarl_hip import ops
            self._mode, _ = ops.graphdist_mode(self.plan, self.proba)
        return self._mode

    @property
    def mode(self):
        return self.deterministic_sample

    def inject_uniform(self, u: torch.Tensor):
        """Use these ``nb_nodes`` uniforms for the next sample (the reference draws ``torch.rand(nb_nodes)``)."""
        self._uniform = u.to(self.proba.device, torch.float32).contiguous()

    def sample(self, sample_shape=torch.Size()):
        """int64 one-hot over the edges, exactly one edge per source node (unbatched, like the reference)."""
        from tarl_hip import ops
        if len(tuple(sample_shape)) or self.proba.dim() != 1:
            raise NotImplementedError("batched sampling is not defined by the reference either (SURVEY Q9)")
        u = self._uniform if self._uniform is not None else torch.rand(self.nb_nodes, device=self.proba.device)
        self._uniform = None
        onehot, _ = ops.graphdist_sample(self.plan, self.proba, uniform=u)
        return onehot

    def log_prob(self, action: torch.Tensor):
        a = action.to(torch.int64).contiguous()
        return _LogProbEntropy.apply(self.logits, self.plan, self.temperature, a)[0]

    def entropy(self):
        return _LogProbEntropy.apply(self.logits, self.plan, self.temperature, None)[1].flatten()


class SimulatorEnv(EnvBase):
    """RL environment around the traffic simulator; observation keys ``node_features (N,7)``, ``edge_features (E,1)``,
    ``agent_index (N,)``, ``time (1,)``; action = int64/bool one-hot over the edges."""

    def __init__(self, device: str = "cpu", timestep_size: int = 1, start_time: int = 0, scenario: str = "Easy",
                 torch_compile: bool = False):
        super().__init__(device=device)
        self.simulator = TransportationSimulator(device=device, torch_compile=torch_compile)
        self.simulator.load_network(scenario=scenario)
        self.simulator.config_parameters(timestep_size=timestep_size, start_time=start_time)
        g = self.simulator.graph
        self.num_edge = g.edge_index.size(1)
        self.num_node = g.x.size(0)
        self.num_obs = 7
        self.reward_spec = Spec((1,), torch.float32, device)
        self.action_spec = Spec((self.num_edge,), torch.bool, device)
        self.observation_spec = {"node_features": Spec((self.num_node, self.num_obs), torch.float32, device),
                                 "edge_features": Spec((self.num_edge, 1), torch.float32, device),
                                 "agent_index": Spec((self.num_node,), torch.int64, device),
                                 "time": Spec((1,), torch.float32, device)}
        self.state = self.simulator.state()

    def _set_seed(self, seed):
        self.rng = torch.Generator(device=self.device)
        self.rng.manual_seed(seed)
        return seed

    def _obs(self, extra=None):
        x, edge_attr, _, agent_index = self.simulator.state()
        d = {"node_features": x, "edge_features": edge_attr, "agent_index": agent_index,
             "time": torch.tensor([self.simulator.time], dtype=torch.float32, device=self.device)}
        d.update(extra or {})
        return TensorDict(d, batch_size=[])

    def _reset(self, tensordict: TensorDictBase = None) -> TensorDictBase:
        sim = self.simulator
        sim.reset()
        sim.inserting_time = sim.choice_time = sim.core_time = sim.withdraw_time = 0
        sim.leg_histogram_values, sim.road_optimality_values = [], []
        sim.on_way_before = sim.done_before = 0
        sim.model_core.response_mpnn.update_history = []
        sim.set_time(3600 * 6 - 60)
        sim.agent.reset()
        return self._obs({"terminated": torch.tensor([False]), "done": torch.tensor([False])})

    def _step(self, tensordict: TensorDictBase):
        from tarl_hip import ops
        sim, h = self.simulator, self.simulator.h
        g = sim.graph
        action = tensordict["action"]
        require_cuda(g.x, "graph.x")
        b = time.time()
        plan = cached_plan(g.edge_index, g.x.size(0))
        ops.apply_action(plan, g.x, h.Nmax, action_onehot=action.to(g.x.device, torch.int64).contiguous())
        e = time.time(); sim.choice_time += e - b; b = e
        sim.graph = sim.model_core(g)
        e = time.time(); sim.core_time += e - b; b = e
        g.x = sim.agent.withdraw_agent_from_network(g, h)
        e = time.time(); sim.withdraw_time += e - b; b = e
        g.x = sim.agent.insert_agent_into_network(g, h)
        sim.inserting_time += time.time() - b
        reward = (-torch.sum(g.x[:, h.NUMBER_OF_AGENT])).flatten()
        sim.set_time(sim.time + sim.timestep)       # always advances
        done = torch.tensor(sim.time > 7 * 3600)
        sim._log_step()
        return self._obs({"reward": reward, "terminated": done, "done": done})

    # torchrl-style public API used by the runner -----------------------------------------------------------------------
    def reset(self):
        return self._reset()

    def step(self, tensordict):
        out = self._step(tensordict)
        tensordict["next"] = out
        return tensordict

    def rollout(self, max_steps, policy=None, break_when_any_done=True):
        """Roll the policy for ``max_steps`` frames; returns the list of frames (each with a "next" entry)."""
        td = self._reset()
        frames = []
        for _ in range(int(max_steps)):
            td = policy(td) if policy is not None else td
            td = self.step(td)
            frames.append(td)
            nxt = td["next"]
            if break_when_any_done and bool(nxt["done"]):
                break
            td = TensorDict({k: v for k, v in nxt.items() if k not in ("reward",)}, batch_size=[])
        return frames
