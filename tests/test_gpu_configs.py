"""GPU: the BASELINE.json configurations that the other test files only touch shortened or not at all.

* config 5 (100 000 route edges = 25x250 torus, 25 000 roads, 262 144 agents): the fused frame == the unfused kernels,
  frame by frame with device Philox noise on both paths, the merged rollout launcher == the frame loop, the domain
  invariants of tests/test_gpu_properties.py, and the banked accumulators (12 500 workgroups serve one environment).
* config 3 (`main.py --algo mpnn+ppo --mode train`, 1 024 route edges, 1 024 agents) at the FULL ``--rollout-steps 256``
  through both rollout kernel families (LDS-resident `tarl_rollout_env` and the four-launch `tarl_fused_rollout`): the
  two trainings end with bit-identical weights.
* the launch-geometry guard: a graph with more node chunks than one grid dimension holds is refused loudly."""
import importlib
import json
import math
import sys

import pytest
import torch

from test_gpu_properties import _check_invariants

pytestmark = pytest.mark.gpu


def _config5(B, fused, seed=9, t0=21540, t1=21600):
    from tarl_hip import synth
    from tarl_hip.engine import SimEngine
    net = synth.torus_network(25, 250)
    assert net.edge_index.size(1) == 100_000 and net.num_roads == 25_000
    pops = torch.stack([synth.population(262_144, net.num_roads, seed=70 + b, t0=t0, t1=t1) for b in range(B)])
    eng = SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr, net.Nmax,
                    pops.cuda(), congestion_constant=net.congestion_constant, seed=seed, fused=fused)
    eng.reset()
    return net, eng


def test_config5_fused_equals_unfused_and_invariants():
    from tarl_hip import ops
    B, frames = 3, 24
    net, e1 = _config5(B, False)
    _, e2 = _config5(B, True)
    assert not e2.env_rollout_supported          # 25 000 roads do not fit one CU's LDS: only the four-launch path applies
    N = e1.N
    emb = torch.randn(N, generator=torch.Generator().manual_seed(3)).cuda()
    e2.prepare_policy(emb)
    ch2 = torch.empty((N, B), dtype=torch.int32, device="cuda")
    lp2, rw2 = torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
    c2 = torch.empty((N, B), device="cuda")
    rws, cts = [], [torch.zeros((B, N), device="cuda")]
    for s in range(frames):
        logits = ops.policy_edge_logits(e1.plan, e1.node_features, emb)
        p = ops.graphdist_softmax(e1.plan, logits)
        _, ch1 = ops.graphdist_sample(e1.plan, p, seed=e2.seed ^ 0x5DEECE66D, counter=s + 1, want_onehot=False,
                                      want_choice=True)
        lp1, _ = ops.graphdist_logprob_entropy(e1.plan, p, choice=ch1, want_entropy=False)
        e1.step(choice=ch1)
        e2.frame_fused(choice=ch2, log_prob=lp2, reward=rw2, counts=c2)
        assert torch.equal(ch1, ch2.t()), f"actions frame {s}"
        assert torch.allclose(lp1, lp2, rtol=1e-5, atol=1e-3), f"log-prob frame {s}"      # sums of 25 000 terms
        assert torch.equal(e1.agents, e2.agents), f"agents frame {s}"
        assert torch.equal(e1.reward, rw2) and torch.equal(e1.counts, c2.t()) and e1.time == e2.time
        if s % 6 == 0 or s == frames - 1:
            assert torch.equal(e1.x, e2.x), f"state frame {s}"
        rws.append(rw2.clone())
        cts.append(c2.t().clone())
    assert float(e2.agents[:, :, 7].sum()) > 1000          # the departure window is inside the 24 frames
    _check_invariants(net, e2, torch.stack(rws), torch.stack(cts))


def test_config5_rollout_launcher_equals_frame_loop():
    B, T = 2, 20
    net, e1 = _config5(B, True)
    _, e2 = _config5(B, True)
    N = e1.N
    emb = torch.randn(N, generator=torch.Generator().manual_seed(4)).cuda()
    e1.prepare_policy(emb)
    e2.prepare_policy(emb)
    ch = torch.zeros((T, N, B), dtype=torch.uint8, device="cuda")
    ct = torch.zeros((T + 1, N, B), dtype=torch.uint8, device="cuda")
    lp, rw = torch.zeros((T, B), device="cuda"), torch.zeros((T, B), device="cuda")
    e1.rollout_fused(T, choice=ch, log_prob=lp, reward=rw, counts=ct)
    ch, ct = e1.decode_rollout(True, choice=ch, counts=ct)          # (T, B, N) edge ids / fp32 counts
    ch2 = torch.empty((N, B), dtype=torch.int32, device="cuda")
    lp2, rw2, c2 = torch.empty(B, device="cuda"), torch.empty(B, device="cuda"), torch.empty((N, B), device="cuda")
    for s in range(T):
        e2.frame_fused(choice=ch2, log_prob=lp2, reward=rw2, counts=c2)
        assert torch.equal(ch[s], ch2.t()) and torch.equal(lp[s], lp2) and torch.equal(rw[s], rw2), f"frame {s}"
        assert torch.equal(ct[s + 1], c2.t()), f"counts frame {s}"
    assert torch.equal(e1.x, e2.x) and torch.equal(e1.agents, e2.agents)
    _check_invariants(net, e1, rw, ct)


def test_too_many_node_chunks_is_refused():
    """A ring of 300 000 roads = 75 000 node chunks (four rows per workgroup) > the 65 535 a grid's y extent holds:
    tarl_check_fused_core says so."""
    from tarl_hip import lib, ops
    N = 300_000
    src = torch.arange(N)
    ei = torch.stack([src, (src + 1) % N])
    plan = ops.Plan(ei, N)
    fs = ops.FusedState(plan, 1, 2, "cuda", 3)
    x = torch.zeros((1, N, 3 * 3 + 7), device="cuda")
    ag = torch.zeros((1, 2, 9), device="cuda")
    ec = ops.EdgeConst(torch.full((N, 1), 1.0), "cuda")
    with pytest.raises(lib.TarlError, match="node chunks"):
        ops.fused_pack(plan, fs, x, 3, ag, ec=ec)


@pytest.mark.parametrize("num_envs", [1, 3])
def test_config3_full_length_both_rollout_kernels(tmp_path, monkeypatch, capsys, num_envs):
    from conftest import PKG
    sys.path.insert(0, PKG)
    monkeypatch.chdir(tmp_path)
    main = importlib.import_module("main").main
    from src.runner import Runner
    created = []
    orig_setup = Runner.setup

    def spy_setup(self):
        orig_setup(self)
        created.append(self)
    monkeypatch.setattr(Runner, "setup", spy_setup)
    weights = {}
    for mode in ("env", "frames"):
        monkeypatch.setenv("TARL_ROLLOUT", mode)
        out = tmp_path / mode
        main(["--algo", "mpnn+ppo", "--mode", "train", "--scenario", "synthetic-1024-1024", "--rollout-steps", "256",
              "--epochs", "2", "--steps", "8", "--num-envs", str(num_envs), "--output-dir", str(out), "--seed", "3"])
        assert "Simulation Summary" in capsys.readouterr().out
        r = created[-1]
        weights[mode] = (r.policy_net.nodes_embedding.weight.detach().clone(),
                         [p.detach().clone() for p in r.value_net.final_mlp.parameters()])
        logs = [json.loads(l) for l in open(out / "train_log.jsonl")]
        assert logs and logs[-1]["global_step"] == 256 and all(math.isfinite(v) for v in logs[-1].values()
                                                               if isinstance(v, float))
        assert logs[-1]["PPO/avg_episode_return"] < 0            # 256 frames of a filling network: agents were on the road
    assert torch.equal(weights["env"][0], weights["frames"][0])
    for a, b in zip(weights["env"][1], weights["frames"][1]):
        assert torch.equal(a, b)
