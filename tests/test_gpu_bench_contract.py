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
    det = os.path.join("gpurun_out", "bench_details_contract_test.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--envs", "512", "--cpu-seconds", "1", "--congested-steps", "1", "--policy-envs", "128",
                          "--policy-steps", "1", "--update-epochs", "2", "--update-sub-batch", "256", "--update-steps", "1",
                          "--config5-envs", "0", "--details", det], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    # the driver's record keeps the tail of the line: numbers only, below 4 KB (prose lives in the sidecar)
    assert len(lines[0]) < 4096, len(lines[0])
    d = json.loads(lines[0])
    for k, v in (("metric", "ppo_env_steps_per_sec"), ("unit", "env-steps/s"), ("n_gpus", 1), ("steps", 2), ("warmup", 1),
                 ("higher_is_better", True), ("scaling", "weak"), ("vs_baseline", None), ("data", "synthetic")):
        assert d[k] == v, (k, d[k])
    assert set(d) >= {"config", "roofline", "roofline_direction", "roofline_insert", "config5", "update_path", "congested_regime",
                      "state_dependent_policy", "value_rollout_only", "cpu_baseline", "msgpass_pair_edges_per_sec"}
    assert d["value"] > 0 and d["ms_per_step"] > 0 and isinstance(d["dtype"], str)
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["rollout_kernels"] == "frames"
    assert abs(d["value"] - 512 * 256 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-4     # env-steps of the step / its time
    r = d["roofline"]                                   # the contract's object, in full
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] == "k_fused_rows"
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 5e-4
    assert r["avg_launch_us"] > 0 and r["bytes_basis"] in ("pmc", "compulsory")
    # the committed PMC record is for the default size: at another size the compulsory bytes stand in, and say so
    assert (r["traffic"] is None) == (r["bytes_basis"] == "compulsory")
    for holder in (d, d["congested_regime"]):          # the other kernels / workloads: the same numbers under fewer keys
        for key in ("roofline_direction", "roofline_insert") + (("roofline",) if holder is not d else ()):
            q = holder[key]
            assert 0.0 <= q["frac"] <= 1.0 and q["avg_launch_us"] > 0 and q["us_all"] > 0 and abs(q["frac"] - q["achieved"] / 8000.0) < 5e-4
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "env-steps/s" and c["sample"]
    assert d["congested_regime"]["value"] > 0 and d["value_rollout_only"] >= d["value"] * 0.9
    p = d["state_dependent_policy"]
    for tag in ("fp32", "fp32_x3", "bf16"):            # fp32 = exact fp32 products on the fp32 matrix pipe; fp32_x3 = bf16 pieces
        assert p[tag]["value"] > 0 and p[tag]["roofline"]["bound"] == "mfma" and 0 < p[tag]["roofline"]["frac"] <= 1.0
    assert "issued_frac" in p["fp32_x3"]["roofline"] and p["envs_per_gpu"] == 128
    u = d["update_path"]
    assert u["epochs"] == 2 and u["sub_batch"] == 256 and u["value"] > 0 and 0 < u["update_frac"] < 1
    assert {"critic_all_frames", "graphdist_bwd", "critic_bwd", "ppo_loss", "grad_allreduce", "adam"} <= set(u["stage_us"])
    assert all(v > 0 for v in u["stage_us"].values()) and u["slowest_minibatch_stage"] in u["stage_us"]
    assert d["config5"] is None                                                                        # skipped here (--config5-envs 0)
    assert 1 <= c["all_cores"]["cores"] <= (os.cpu_count() or 1) and c["all_cores"]["value"] > 0      # physical cores of one socket
    assert c["all_cores"]["logical_cpus_of_the_host"] == (os.cpu_count() or 1)
    assert c["value_rollout_only"] >= c["value"] > 0 and c["update_seconds"] > 0                       # update included in value
    assert d["per_rank"]["timed_seconds"] == [pytest.approx(d["timed_seconds"], rel=0.2)] and len(d["per_rank"]["setup_seconds"]) == 1
    assert d["world_size_seen_by_backend"] == 1 and d["replica_param_max_abs_diff"] == 0.0
    # the sidecar carries the prose and the verbose forms
    det_d = json.load(open(os.path.join(ROOT, det)))
    assert "what" in det_d["roofline"] and det_d["roofline"]["compulsory_bytes_per_launch"] > 0
    assert "stages" in det_d["update_path"] and "sample" in det_d["cpu_baseline"] and "note" in det_d["state_dependent_policy"]


def test_bench_starts_its_own_ranks():
    """``python bench.py --gpus 2`` with no launcher around it: the parent (which never touches the GPU) starts the two
    ranks through torch.distributed.run, relays rank 0's ONE JSON line and exits with the children's status. Here the two
    ranks share the box's single GPU, so the process group runs on gloo (RCCL refuses two ranks on one device); on an
    8-GPU node the same command shape runs one rank per GPU over RCCL."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TARL_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                          "--envs", "128", "--cpu-seconds", "0", "--congested-window", "0", "--policy-envs", "0"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size_seen_by_backend"] == 2 and d["dist_backend"] == "gloo"
    assert d["replica_param_max_abs_diff"] == 0.0      # after the timed Adam steps both replicas still hold rank 0's bits
    assert len(d["per_rank"]["timed_seconds"]) == 2 and all(v > 0 for v in d["per_rank"]["timed_seconds"])
    assert abs(d["value"] - 2 * 128 * 256 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-4      # both ranks' frames / slowest rank
    assert "cpu_baseline" not in d                                                             # N = 1 only
    # a failing rank makes the whole command fail (no silent half-result)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--envs", "128", "--rollout-steps", "0", "--cpu-seconds", "0", "--congested-window", "0", "--policy-envs", "0"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith('{"metric"')]
