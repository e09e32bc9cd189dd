"""GPU: bench.py's one JSON line carries the contract's keys, and the roofline objects are fractions of peak.
(A reduced size and a short CPU sample keep this quick; the driver's run uses the defaults.)"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_line_contract():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--envs", "512", "--cpu-seconds", "1", "--congested-steps", "1", "--policy-envs", "128",
                          "--policy-steps", "1"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k, v in (("metric", "ppo_env_steps_per_sec"), ("unit", "env-steps/s"), ("n_gpus", 1), ("steps", 2), ("warmup", 1),
                 ("higher_is_better", True), ("scaling", "weak"), ("vs_baseline", None), ("data", "synthetic")):
        assert d[k] == v, (k, d[k])
    assert d["value"] > 0 and d["ms_per_step"] > 0 and isinstance(d["dtype"], str)
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["rollout_kernels"] == "frames"
    assert abs(d["value"] - 512 * 256 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6     # env-steps of the step / its time
    for key in ("roofline", "roofline_direction", "roofline_insert"):
        r = d[key]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert 0.0 <= r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert r["avg_launch_us"] > 0 and r["bytes_basis"] in ("pmc_counters", "compulsory")
        # the committed PMC record is for the default size: at another size the compulsory bytes stand in, and say so
        assert (r["traffic"] is None) == (r["bytes_basis"] == "compulsory")
    assert d["roofline"]["frac"] > 0.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "env-steps/s" and c["sample"]
    assert d["congested_regime"]["value"] > 0
    assert d["state_dependent_policy"]["bf16"]["value"] > 0 and d["state_dependent_policy"]["fp32"]["value"] > 0
