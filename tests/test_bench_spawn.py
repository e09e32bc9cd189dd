"""CPU: ``python bench.py --gpus N`` without a launcher starts its own ranks — as a CHILD process, from a parent that never
touches the GPU (on the GPU pool a process that has initialised the GPU must not be replaced by another program, and the
parent has no business holding a device anyway). The child here is a stand-in; the real thing runs in
tests/test_gpu_bench_contract.py::test_bench_starts_its_own_ranks."""
import importlib
import io
import os
import sys

import pytest

from conftest import ROOT


class _FakeProc:
    def __init__(self, lines, rc):
        self.stdout = io.StringIO("".join(lines))
        self._rc = rc

    def wait(self):
        return self._rc


@pytest.mark.parametrize("rc,lines,expect", [
    (0, ['noise from a rank\n', '{"metric": "ppo_env_steps_per_sec", "n_gpus": 4}\n'], 0),
    (1, ['{"metric": "ppo_env_steps_per_sec", "n_gpus": 4}\n'], 1),          # a rank failed: the command fails
    (0, ['no json at all\n'], 1),                                            # rank 0 printed nothing: not a success
])
def test_parent_spawns_torchrun_and_never_touches_the_gpu(monkeypatch, capsys, rc, lines, expect):
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    import subprocess
    import torch
    seen = {}

    def fake_popen(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw.get("env")
        return _FakeProc(lines, rc)

    def boom(*a, **k):
        raise AssertionError("the spawning parent queried the GPU")
    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.setattr(torch.cuda, "is_available", boom)
    monkeypatch.setattr(torch.cuda, "set_device", boom)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == expect
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]          # the same flags reach every rank
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr()
    assert out.out.count('{"metric"') == (1 if any(l.startswith('{"metric"') for l in lines) else 0)
    assert "noise from a rank" not in out.out                                       # everything else goes to stderr


def test_under_a_launcher_nothing_is_spawned(monkeypatch):
    """With WORLD_SIZE in the environment (torchrun, the driver's launch line) main() goes straight on as a rank."""
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    called = []
    monkeypatch.setattr(bench, "spawn_ranks", lambda args: called.append(args))
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    from tarl_hip import dist_utils
    monkeypatch.setattr(dist_utils, "init_from_env", lambda backend=None: (_ for _ in ()).throw(RuntimeError("rank path reached")))
    with pytest.raises(RuntimeError, match="rank path reached"):
        bench.main()
    assert not called
