"""Command line of the MI355X build. The flag set is the reference CLI's contract (main.py of the reference: --algo,
--scenario, --mode, --timestep_size, --start-end-time, --epochs, --rollout-steps, --seed, --device, --output-dir,
--profile, --torch-compile) plus ``--steps`` (used by the reference's README but missing from its parser) and
``--num-envs`` (vectorised environments per GPU) and ``--policy-head``."""
import argparse
import os
import sys

# before anything imports torch (the HIP runtime reads it once): dmabuf IPC, which RCCL needs on hosts without legacy IPC —
# also when the ranks were started by an external `torchrun main.py` whose environment lacks it. A launcher's value wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from src.runner import Runner, RunnerArgs  # noqa: E402

ALGOS = ("dijkstra", "random", "mpnn", "mpnn+ppo")

# (flag, argparse keyword arguments)
OPTIONS = (
    ("--algo", dict(choices=ALGOS, default="dijkstra", help="routing agent")),
    ("--scenario", dict(type=str, default="Easy",
                        help="data/<scenario>/ (MATSim XML), save/<scenario>/ cache, or synthetic-<edges>-<agents>[-seed]")),
    ("--mode", dict(choices=("eval", "train"), default="eval")),
    ("--timestep_size", dict(type=int, default=1, help="seconds per simulation step")),
    ("--start-end-time", dict(type=int, nargs=2, default=[0, 86400], metavar=("START", "END"))),
    ("--epochs", dict(type=int, default=1, help="PPO minibatch steps on the collected batch")),
    ("--rollout-steps", dict(type=int, default=32, help="frames collected per environment")),
    ("--seed", dict(type=int, default=0)),
    ("--device", dict(type=str, default="cpu", help="kept for compatibility: the path always runs on the GPU")),
    ("--output-dir", dict(type=str, default="runs")),
    ("--profile", dict(action="store_true")),
    ("--torch-compile", dict(action="store_true", help="accepted and ignored: the kernels are hand-written HIP")),
    ("--steps", dict(type=int, default=None, help="number of eval steps (overrides start/end time)")),
    ("--num-envs", dict(type=int, default=1, help="vectorised environments per GPU for mpnn+ppo training")),
    ("--policy-head", dict(choices=("embedding", "edge_mlp", "edge_mlp_fp32", "edge_mlp_bf16"), default="embedding",
                           help="mpnn / mpnn+ppo: the reference's live embedding head, or the per-edge MLP head it keeps "
                                "as parameters (state-dependent; edge_mlp: rollout logits at fp32 accuracy on the bf16 matrix "
                                "pipe — operands in exact bf16 pieces —, edge_mlp_fp32: on the fp32 matrix pipe, exact fp32 "
                                "products, edge_mlp_bf16: bf16 logits; the PPO update runs in fp32)")),
)


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="TARL routing experiments on MI355X (classical agents, MPNN policy, PPO)")
    for flag, kw in OPTIONS:
        parser.add_argument(flag, **kw)
    return parser


def main(argv=None):
    """Under ``torchrun --nproc-per-node N main.py --algo mpnn+ppo --mode train ...`` every rank joins the process group
    (RCCL), binds its own GPU, trains on its own rollouts (engine seed + rank) with averaged gradients, and only rank 0
    writes logs / checkpoints / metric tables."""
    ns = build_parser().parse_args(argv)
    runner = Runner(RunnerArgs(**vars(ns)))       # joins the process group when launched by torchrun
    try:
        runner.setup()
        if ns.mode == "train":
            runner.train()
        runner.eval()
    except BaseException:
        # a rank that fails must not enter a collective its peers are not in (they may sit in a gradient all-reduce or in
        # ppo_train's trailing barrier): leave the group WITHOUT a barrier and exit non-zero so the launcher tears the
        # job down instead of waiting for the communicator's timeout
        runner.close(failed=True)
        raise
    else:
        runner.close()


if __name__ == "__main__":
    main()
