"""Runner / RunnerArgs — CLI-facing orchestration (reference: src/runner.py). ``mpnn`` and ``mpnn+ppo`` run on the HIP
path; ``random`` and ``dijkstra`` run the classical loop on the same kernels (``dijkstra``: all-pairs next-hop table by
``tarl_apsp`` instead of networkx)."""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path

import torch


@dataclass
class RunnerArgs:
    algo: str
    scenario: str
    mode: str
    timestep_size: int = 1
    start_end_time: list = (0, 86400)
    epochs: int = 1
    rollout_steps: int = 32
    seed: int = 0
    device: str = "cpu"
    output_dir: str = "runs"
    profile: bool = False
    torch_compile: bool = False
    steps: int = None          # README / BASELINE use --steps; the reference CLI lacks it
    num_envs: int = 1
    policy_head: str = "embedding"


class Runner:
    def __init__(self, args: RunnerArgs):
        self.args = args
        # one process per GPU under torchrun: join the process group (backend nccl = RCCL) and bind this rank's device
        # BEFORE any GPU work; a single process (WORLD_SIZE unset) skips all of it
        from tarl_hip import dist_utils
        self.rank, self.world, local = dist_utils.init_from_env()
        # the path runs on the GPU only: "cuda" on ROCm is the HIP device
        if torch.cuda.is_available():
            local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
            self.device = torch.device("cuda", local)
        else:
            self.device = torch.device(args.device)
        torch.manual_seed(args.seed)          # identical initial weights on every rank (also broadcast by the trainer)

    def close(self, failed=False):
        """Leave the process group (if this process joined one). ``failed``: this rank is on its way out with an
        exception — no barrier (the peers are in other collectives), and a communicator that cannot be destroyed cleanly
        is left to the launcher, which kills the job when this process exits non-zero."""
        import torch.distributed as dist
        from tarl_hip import dist_utils
        if self.world > 1 and dist.is_initialized():
            if failed:
                try:
                    dist.destroy_process_group()
                except Exception:      # noqa: BLE001 — the original exception is the one to report
                    pass
                return
            dist_utils.barrier()
            dist.destroy_process_group()

    def setup(self):
        from .reinforcement_learning import SimulatorEnv
        from .transportation_simulator import TransportationSimulator
        from .agents.base import Agents, DijkstraAgents
        a = self.args
        if a.algo in {"dijkstra", "random"}:
            self.simulator = TransportationSimulator(str(self.device), torch_compile=a.torch_compile)
            self.simulator.load_network(scenario=a.scenario)
            self.agent = self.simulator.agent = (DijkstraAgents if a.algo == "dijkstra" else Agents)(str(self.device))
            self.agent.load(scenario=a.scenario)
            self.simulator.config_parameters(timestep_size=a.timestep_size, start_time=a.start_end_time[0])
            self.agent.set_time(a.start_end_time[0])
        elif a.algo in {"mpnn", "mpnn+ppo"}:
            from .agents.mpnn_agent import MPNNPolicyNet, MPNNValueNetSimple
            self.env = SimulatorEnv(device=str(self.device), timestep_size=a.timestep_size,
                                    start_time=a.start_end_time[0], scenario=a.scenario, torch_compile=a.torch_compile)
            g, h = self.env.simulator.graph, self.env.simulator.h
            free_flow = g.x[:, h.FREE_FLOW_TIME_TRAVEL][g.edge_index[1]]
            self.policy_net = MPNNPolicyNet(g.edge_index, g.x.size(0), free_flow, device=str(self.device))
            self.policy_net.policy_head = a.policy_head
            self.policy_net.load(a.scenario)
            self.value_net = MPNNValueNetSimple(g.edge_index, g.x.size(0), device=str(self.device))
            self.value_net.load(a.scenario)
            self.env.simulator.agent = self.policy_net     # the policy IS the population store used by the env
        else:
            raise ValueError(f"Unknown algorithm {a.algo}")

    def _actor(self, return_log_prob):
        from .reinforcement_learning import GraphDistribution
        from .rl.modules import ProbabilisticActor, TensorDictModule
        inner = TensorDictModule(self.policy_net, in_keys=["node_features", "edge_features", "agent_index"],
                                 out_keys=["logits"])
        return ProbabilisticActor(module=inner, spec=self.env.action_spec, distribution_class=GraphDistribution,
                                  in_keys=["logits"],
                                  distribution_kwargs={"edge_index": self.env.simulator.graph.edge_index},
                                  return_log_prob=return_log_prob)

    def train(self):
        a = self.args
        if not (a.algo == "mpnn+ppo" and a.mode == "train"):
            raise RuntimeError("Training is only supported for algo 'mpnn+ppo'")
        from .rl.modules import TensorDictModule, ValueOperator
        from .rl.ppo_trainer import ppo_train
        policy_module = self._actor(return_log_prob=True)
        value_module = ValueOperator(TensorDictModule(self.value_net,
                                                      in_keys=["node_features", "edge_features", "agent_index", "time"],
                                                      out_keys=["value"]),
                                     in_keys=["node_features", "edge_features", "agent_index", "time"])
        # the evaluation environment of the reference's Runner.train (src/runner.py:111-118): a second SimulatorEnv that
        # shares the policy (= the population store)
        from .reinforcement_learning import SimulatorEnv
        eval_env = SimulatorEnv(device=str(self.device), timestep_size=a.timestep_size, start_time=a.start_end_time[0],
                                scenario=a.scenario, torch_compile=a.torch_compile)
        eval_env.simulator.agent = self.policy_net
        out = Path(a.output_dir)
        if self.rank == 0:
            out.mkdir(parents=True, exist_ok=True)
        # rank 0 alone writes the checkpoint and the logs; every rank takes part in the training collectives
        ppo_train(self.env, policy_module, value_module, total_frames=a.rollout_steps,
                  frames_per_batch=a.rollout_steps, num_epochs=a.epochs, device=self.device,
                  checkpoint_path=(out / "policy.pt") if self.rank == 0 else None,
                  log_dir=str(out) if self.rank == 0 else None, eval_env=eval_env, eval_interval=1,
                  num_envs=a.num_envs, seed=a.seed)

    def eval(self):
        a = self.args
        n = a.steps if a.steps is not None else (a.start_end_time[1] - a.start_end_time[0]) // a.timestep_size
        if a.algo in {"dijkstra", "random"}:
            for _ in range(n):
                self.simulator.run()
            sim, agent = self.simulator, self.agent
        else:
            with torch.no_grad():
                self.env.rollout(n, self._actor(return_log_prob=False), break_when_any_done=False)
            sim, agent = self.env.simulator, self.env.simulator.agent
        mask = agent.agent_features[:, agent.DONE] == 1
        tt = agent.agent_features[mask, agent.ARRIVAL_TIME] - agent.agent_features[mask, agent.DEPARTURE_TIME]
        avg = float(tt.mean()) if bool(mask.any()) else float("nan")
        total = sim.inserting_time + sim.choice_time + sim.core_time + sim.withdraw_time
        if self.rank != 0:      # every rank evaluated its replica; one summary / one set of metric tables
            return {"steps": n, "arrived": int(mask.sum()), "avg_travel_time": avg}
        print("\n=== Simulation Summary ===")
        print(f"{'Steps:':25} {n:10d}")
        print(f"{'Agents arrived:':25} {int(mask.sum()):10d}")
        print(f"{'Average travel time:':25} {avg:10.2f} s")
        for label, v in (("Agent Insertion time:", sim.inserting_time), ("Route Choice time:", sim.choice_time),
                         ("Core Model time:", sim.core_time), ("Agent Withdrawal time:", sim.withdraw_time)):
            print(f"{label:25} {v:10.2f} s   (host enqueue time; kernels run asynchronously)")
        print("-" * 42)
        print(f"{'Total simulation time:':25} {total:10.2f} s")
        # the reference's eval report (src/runner.py:166-174, 219-226): phase-time pie, node metrics, leg histogram, road
        # optimality, and the simulated daily counts against the MSA assignment's expected flows
        out_dir = Path(a.output_dir)
        try:
            sim.plot_computation_time(str(out_dir))
            sim.compute_node_metrics(str(out_dir))
            sim.plot_leg_histogram(str(out_dir))
            sim.plot_road_optimality(str(out_dir))
            if sim.graph.x.size(0) <= 4096:          # all-pairs table per MSA iteration: keep it to mid-size graphs
                from .algorithms.user_equilibrium_msa import run_msa
                expected = run_msa(sim.graph, agent)
                out_dir.mkdir(parents=True, exist_ok=True)
                with open(out_dir / "msa_expected_flows.csv", "w") as f:
                    f.write("road,expected_hourly_flow\n")
                    f.writelines(f"{r},{v}\n" for r, v in expected.items())
                sim.plot_daily_counts(expected, str(out_dir))
            import matplotlib.pyplot as plt
            plt.close("all")
        except Exception as exc:  # noqa: BLE001 - analysis output must not fail the run
            print(f"metric tables / figures skipped: {exc}")
        return {"steps": n, "arrived": int(mask.sum()), "avg_travel_time": avg}
