"""oracle/ — CPU restatement of the reference's MPNN+PPO routing hot path.

TEST INFRASTRUCTURE, NOT PRODUCT. Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the checker / reported baseline. The product path
(``tarl-simulator_amd/``) never imports it and has no CPU fallback.

What it restates (plain torch on CPU, fp32, the reference's own algorithmic structure: full-row ``x_i/x_j`` gathers,
sequential ``index_add_`` aggregation, per-call sort in the distribution):

* ``oracle.sim``    — DirectionMPNN / ResponseMPNN / SimulationCoreModel / insert / withdraw / env step
                      (reference ``src/direction_mpnn.py``, ``src/response_mpnn.py``, ``src/simulation_core_model.py``,
                      ``src/agents/base.py:244-403``, ``src/reinforcement_learning.py:222-309``)
* ``oracle.dist``   — GraphDistribution (``src/reinforcement_learning.py:15-96``)
* ``oracle.nets``   — live policy / critic forward (``src/agents/mpnn_agent.py:117-231,428-450``)
* ``oracle.ppo``    — GAE, clipped PPO loss, Adam as configured in ``src/rl/ppo_trainer.py:35-37,129-145``

Third-party arithmetic that the reference delegates to absent wheels is restated from the pinned versions'
published semantics: torch-geometric 2.5.0 (``MessagePassing.propagate``), torch-scatter 2.1.2
(``scatter_add/max/softmax``), torchrl 0.5.0 (``GAE``, ``ClipPPOLoss``).

Pinning status
--------------
* sim / dist / nets: PINNED against the reference's own source executed in the build container under stand-ins for
  the absent wheels (``oracle/refshim``, ``oracle/make_golden.py`` -> ``tests/golden/*.npz``) and against the integer
  facts of the reference's tests (``tests/conftest.py:45-91`` Braess fixture, ``tests/agents_test.py:12-73``).
  The stand-ins are this build's reading of the third-party semantics (cannot be checked against the wheels offline).
* ppo (GAE / ClipPPOLoss): **parity unpinned** — the arithmetic *is* torchrl 0.5.0, which is absent, and the
  reference's tests pin no loss or advantage value. Restated from the published formulas only.

Randomness is an explicit input everywhere (uniform / Gumbel tensors); when omitted the functions draw from torch's
global generator in the same order and count as the reference does.
"""
