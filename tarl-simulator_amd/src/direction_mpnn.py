"""DirectionMPNN — first message-passing round of the traffic-flow step (reference: src/direction_mpnn.py).

Every road offers its head-of-queue agent to the downstream road that agent selected; every downstream road admits at
most one of the offers (Gumbel-max over the turn probabilities) and enqueues it. Same constructor, attributes,
``forward`` signature and in-place contract as the reference class; the arithmetic is ``tarl_direction_step``.
"""
from __future__ import annotations

from typing import Optional

import torch

from ._compat import MessagePassingBase, cached_edge_const, cached_plan, require_cuda
from .feature_helpers import FeatureHelpers


class DirectionMPNN(MessagePassingBase, FeatureHelpers):
    def __init__(self, Nmax=100, time: int = 0):
        MessagePassingBase.__init__(self)
        FeatureHelpers.__init__(self, Nmax=Nmax)
        self.time = time
        self.Nmax = Nmax
        self.road_optimality_data = None     # {"delta_travel_time": Tensor(E)} after every forward
        self.noise_seed = None               # device Philox seed; None -> torch.initial_seed()
        self._noise_counter = 0
        self._uniform = None                 # one-shot injected noise (parity runs)
        self._status = None                  # device status word: a count reached Nmax in an earlier forward
        self._status_host = None             # pinned mirror of it, filled by a non-blocking copy behind every forward
        self._status_event = None

    def set_time(self, time):
        self.time = time
        self._poll_status()

    # A count that reaches Nmax leaves the reference's defined domain: its update (src/direction_mpnn.py:172-191) then
    # writes slot ``Nmax`` of the id block — column 0 of the arrival block —, the arrival into the departure block and
    # the departure into MAX_NUMBER_OF_AGENT, all inside the row: SILENT corruption of the FIFO, and an IndexError only
    # several steps later (count >= Nmax + 7). The kernels flag the first such count in a device status word. The word
    # travels to pinned host memory behind every forward (no host synchronisation on the step path) and is looked at when
    # it has arrived — at the next forward / set_time — or, blocking, by check().
    def _poll_status(self, block=False):
        ev = self._status_event
        if ev is None:
            return
        if block:
            ev.synchronize()
        elif not ev.query():
            return
        self._status_event = None
        if int(self._status_host[0]) != 0:
            self._status.zero_()
            self._status_host.zero_()
            raise IndexError("DirectionMPNN: a FIFO count reached Nmax — outside the reference's defined domain (its update "
                             "silently overwrites the neighbouring FIFO blocks there and raises IndexError a few steps on)")

    def check(self):
        """Blocking check of the status word (call at the end of a run; one host synchronisation)."""
        self._poll_status(block=True)

    def inject_uniform(self, u: torch.Tensor):
        """Use these ``E`` uniforms for the next forward instead of device noise (the reference draws them with
        ``torch.rand_like`` from the global generator, src/direction_mpnn.py:137)."""
        self._uniform = u

    def forward(self, x, edge_index, edge_attr, critical_number: Optional[torch.Tensor] = None,
                congestion_constant: Optional[torch.Tensor] = None):
        """Mutates ``x`` (R, F) in place and returns it."""
        from tarl_hip import ops
        require_cuda(x, "x")
        plan = cached_plan(edge_index, x.size(0))
        ec = cached_edge_const(edge_attr, x.device)
        if self._status is None or self._status.device != x.device:
            self._status = torch.zeros(1, dtype=torch.int32, device=x.device)
            self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._status_event = None
        self._poll_status()
        gumbel = None
        if self._uniform is not None:
            gumbel = ops.gumbel_from_uniform_cpu(self._uniform).to(x.device)
            self._uniform = None
        self._noise_counter += 1
        seed = torch.initial_seed() if self.noise_seed is None else self.noise_seed
        cc = None if congestion_constant is None else congestion_constant.to(torch.float32).contiguous()
        dtt, _ = ops.direction_step(plan, x, self.Nmax, ec, self.time, congestion_constant=cc, gumbel=gumbel,
                                    seed=seed & 0x7FFFFFFFFFFFFFFF, counter=self._noise_counter, status=self._status)
        self.road_optimality_data = {"delta_travel_time": dtt.view(-1)}
        self._status_host.copy_(self._status, non_blocking=True)
        self._status_event = torch.cuda.Event()
        self._status_event.record()
        return x
