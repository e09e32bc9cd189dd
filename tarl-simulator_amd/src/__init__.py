"""Drop-in mirror of the reference's ``src`` package for the MPNN + PPO routing hot path: same module
paths, class names, signatures and in-place conventions; every tensor operation is a hand-written HIP kernel reached
through the C ABI (tarl_hip). Put ``tarl-simulator_amd/`` first on ``sys.path`` instead of the reference checkout."""
