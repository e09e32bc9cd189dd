"""Unified entry point (reference: main.py) — same flags, plus ``--steps`` (used by the reference's README but missing
from its CLI) and ``--num-envs`` (vectorised environments per GPU)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from src.runner import Runner, RunnerArgs  # noqa: E402


def main(argv=None):
    p = argparse.ArgumentParser(description="Unified runner for classical and RL experiments (MI355X hot path)")
    p.add_argument("--algo", choices=["dijkstra", "random", "mpnn", "mpnn+ppo"], default="dijkstra")
    p.add_argument("--scenario", type=str, default="Easy",
                   help="save/<scenario>/ cache of the reference, or synthetic-<edges>-<agents>[-seed]")
    p.add_argument("--mode", choices=["eval", "train"], default="eval")
    p.add_argument("--timestep_size", type=int, default=1)
    p.add_argument("--start-end-time", type=int, nargs=2, default=[0, 86400])
    p.add_argument("--epochs", type=int, default=1)
    p.add_argument("--rollout-steps", type=int, default=32)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--device", type=str, default="cpu")
    p.add_argument("--output-dir", type=str, default="runs")
    p.add_argument("--profile", action="store_true")
    p.add_argument("--torch-compile", action="store_true")
    p.add_argument("--steps", type=int, default=None, help="number of eval steps (overrides start/end time)")
    p.add_argument("--num-envs", type=int, default=1, help="vectorised environments per GPU for mpnn+ppo training")
    args = p.parse_args(argv)
    runner = Runner(RunnerArgs(**vars(args)))
    runner.setup()
    if args.mode == "train":
        runner.train()
    runner.eval()


if __name__ == "__main__":
    main()
