"""Inert stand-in for torchrl==0.5.0: only what src/reinforcement_learning.py needs to be importable."""
