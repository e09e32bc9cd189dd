"""CPU: every entry point of a multi-process GPU job puts HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC, what RCCL needs on
hosts without legacy IPC) into the environment BEFORE torch is imported — also when the ranks come from an external
``torchrun bench.py`` / ``torchrun main.py`` whose environment lacks it — and never overrides a launcher's own value."""
import os
import subprocess
import sys

import pytest

from conftest import PKG, ROOT

PROBE = """
import os, sys
sys.path[:0] = [{root!r}, {pkg!r}]
assert "torch" not in sys.modules
import {mod}
print("IPC=" + os.environ["HSA_ENABLE_IPC_MODE_LEGACY"])
"""


@pytest.mark.parametrize("mod", ["bench", "main", "tarl_hip.dist_utils"])
@pytest.mark.parametrize("preset", [None, "1"])
def test_entry_points_default_the_ipc_mode_before_torch(mod, preset):
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    if preset is not None:
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = preset
    out = subprocess.run([sys.executable, "-c", PROBE.format(root=ROOT, pkg=PKG, mod=mod)], capture_output=True, text=True,
                         timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "IPC=" + (preset or "0")


def test_the_default_is_set_above_the_torch_import():
    """Source order, not just the end state: the setdefault line precedes the first torch import in each file."""
    for rel in ("bench.py", os.path.join("tarl-simulator_amd", "main.py"),
                os.path.join("tarl-simulator_amd", "tarl_hip", "dist_utils.py")):
        src = open(os.path.join(ROOT, rel)).read()
        k = src.index('os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")')
        later = [src.find(pat) for pat in ("\nimport torch", "\nfrom src.runner import")]
        assert all(p == -1 or p > k for p in later), rel
