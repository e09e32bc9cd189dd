"""GPU parity (through the C ABI) for GraphDistribution and the live policy logits: HIP kernels vs golden vectors of the
reference and vs the CPU oracle. Integer outputs (sampled actions, mode) bit-exact given identical probabilities and
noise; probabilities / log-probs / entropies / gradients within 1e-4 fp32 (tolerance stated per assert)."""
import math

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4   # BASELINE.json north_star: MPNN logits and PPO losses within 1e-4 fp32


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tarl_hip import ops as _ops
    return _ops


def dev(t):
    return t.cuda()


@pytest.mark.parametrize("name", ["dist_small", "dist_mid"])
def test_graphdist_golden(ops, name):
    g = load_golden(name)
    ei = g["edge_index"]
    N = int(ei.max()) + 1
    plan = ops.Plan(ei, N)
    assert plan.num_groups == g["nb_nodes"] and not plan.src_sorted
    proba = ops.graphdist_softmax(plan, dev(g["logits"]))
    assert torch.allclose(proba.cpu(), g["proba"], atol=1e-6, rtol=1e-5)
    lp, ent = ops.graphdist_logprob_entropy(plan, proba, action_onehot=dev(g["a0"]))
    assert abs(lp.item() - float(g["lp0"])) < TOL and abs(ent.item() - float(g["entropy"])) < TOL
    # integer routing: identical probabilities + identical noise => identical int64 one-hot actions
    pref = dev(g["proba"])
    for k in range(4):
        onehot, choice = ops.graphdist_sample(plan, pref, uniform=dev(g[f"u{k}"]), want_choice=True)
        assert onehot.dtype == torch.int64 and torch.equal(onehot.cpu(), g[f"a{k}"])
        picked = torch.nonzero(g[f"a{k}"]).view(-1)
        assert torch.equal(torch.sort(choice.cpu().long())[0], torch.sort(picked)[0])
        lpk, _ = ops.graphdist_logprob_entropy(plan, pref, choice=choice)
        assert abs(lpk.item() - float(g[f"lp{k}"])) < TOL
    mode, _ = ops.graphdist_mode(plan, pref)
    assert torch.equal(mode.cpu(), g["mode"])
    lp_bad, _ = ops.graphdist_logprob_entropy(plan, pref, action_onehot=dev(g["bad"]))
    assert lp_bad.item() == -math.inf
    # batched forward + backward
    pb = ops.graphdist_softmax(plan, dev(g["logits_b"]))
    assert torch.allclose(pb.cpu(), g["proba_b"], atol=1e-6, rtol=1e-5)
    acts = dev(g["acts_b"])
    lpb, entb = ops.graphdist_logprob_entropy(plan, pb, action_onehot=acts)
    assert torch.allclose(lpb.cpu(), g["lp_b"], atol=TOL, rtol=0) and torch.allclose(entb.cpu(), g["ent_b"], atol=TOL, rtol=0)
    w = dev(g["w_b"])
    g_lp = ops.graphdist_logprob_entropy_bwd(plan, pb, 1.0, action_onehot=acts, grad_log_prob=w, log_prob_fwd=lpb)
    g_en = ops.graphdist_logprob_entropy_bwd(plan, pb, 1.0, action_onehot=acts, grad_entropy=w)
    assert torch.allclose(g_lp.cpu(), g["grad_lp_b"], atol=TOL, rtol=0)
    assert torch.allclose(g_en.cpu(), g["grad_ent_b"], atol=TOL, rtol=0)


def test_graphdist_vs_oracle_large_and_temperature(ops):
    """10k-edge torus, batch of logits, temperature != 1; sampled actions must equal the oracle's when both use the
    GPU's probabilities (the global double-accumulated cumsum + fp32 rebase of the reference is what is being checked:
    at node 2 499 the thresholds are quantised to 2^-12)."""
    from oracle import dist
    from tarl_hip import synth
    net = synth.torus_network(25, 25)
    ei, N, E = net.edge_index, net.num_roads, net.edge_index.size(1)
    plan = ops.Plan(ei, N)
    assert plan.src_sorted
    gen = torch.Generator().manual_seed(2)
    logits = torch.randn((3, E), generator=gen) * 2
    for T in (1.0, 0.7):
        p = ops.graphdist_softmax(plan, dev(logits), T)
        for b in range(3):
            d = dist.GraphDist(logits[b], ei, T)
            assert torch.allclose(p[b].cpu(), d.proba, atol=1e-6, rtol=1e-5)
            d2 = dist.GraphDist(logits[b], ei, T, proba=p[b].cpu())   # integer check on the GPU's probabilities
            u = torch.rand(N, generator=gen)
            onehot, choice = ops.graphdist_sample(plan, p[b].contiguous(), uniform=dev(u), want_choice=True)
            assert torch.equal(onehot.cpu(), d2.sample(u))
            lp, ent = ops.graphdist_logprob_entropy(plan, p[b].contiguous(), choice=choice)
            assert abs(lp.item() - d2.log_prob(onehot.cpu()).item()) < 2e-2      # sum of 2 500 logs, fp32 order
            assert abs(lp.item() - d2.log_prob(onehot.cpu()).item()) / abs(lp.item()) < 1e-5
            assert abs(ent.item() - d2.entropy().item()) / abs(ent.item()) < 1e-5


def test_graphdist_device_sampler_statistics(ops):
    """uniform=NULL: Philox on device. Frequencies follow the probabilities; deterministic in (seed, counter)."""
    from tarl_hip import synth
    net = synth.torus_network(2, 2)
    ei, N, E = net.edge_index, net.num_roads, net.edge_index.size(1)
    plan = ops.Plan(ei, N)
    logits = torch.randn(E, generator=torch.Generator().manual_seed(4))
    B = 8192
    p = ops.graphdist_softmax(plan, dev(logits.unsqueeze(0).repeat(B, 1).contiguous()))
    a1, _ = ops.graphdist_sample(plan, p, seed=3, counter=9)
    a2, _ = ops.graphdist_sample(plan, p, seed=3, counter=9)
    a3, _ = ops.graphdist_sample(plan, p, seed=3, counter=10)
    assert torch.equal(a1, a2) and not torch.equal(a1, a3)
    assert bool((a1.view(B, N, 4).sum(-1) == 1).all())
    freq = a1.float().mean(0).cpu()
    assert torch.allclose(freq, p[0].cpu(), atol=0.03)


def test_policy_logits_golden_and_grad(ops):
    g = load_golden("nets")
    ei = g["edge_index"]
    N = g["node_features"].size(0)
    plan = ops.Plan(ei, N)
    emb = dev(g["pol__nodes_embedding__weight"].reshape(-1).contiguous())
    logits = ops.policy_edge_logits(plan, dev(g["node_features"]), emb)
    assert torch.equal(logits.cpu(), g["logits"])
    lb = ops.policy_edge_logits(plan, dev(g["node_features_b"]), emb)
    assert torch.equal(lb.cpu(), g["logits_b"])
    # backward vs autograd of the oracle expression
    from oracle import nets
    w = g["pol__nodes_embedding__weight"].clone().requires_grad_(True)
    gl = torch.randn(g["logits_b"].shape, generator=torch.Generator().manual_seed(1))
    (nets.policy_logits(g["node_features_b"], ei, w) * gl).sum().backward()
    ge = ops.policy_edge_logits_bwd(plan, dev(g["node_features_b"]), dev(gl), emb.numel())
    assert torch.allclose(ge.cpu(), w.grad.view(-1), atol=1e-5, rtol=1e-5)


def test_rollout_with_gpu_sampling_matches_golden(ops):
    """Policy logits -> softmax -> sample -> log_prob all on the GPU inside the env loop, fed with the reference's noise:
    the sampled actions equal the reference's at every one of the 90 steps (so the whole trajectory stays bit-exact)."""
    g = load_golden("env_het")
    Nmax, ei = g["Nmax"], g["edge_index"]
    N = g["x_init"].size(0)
    plan = ops.Plan(ei, N)
    ec = ops.EdgeConst(g["edge_attr"], "cuda")
    cc = dev(g["congestion_constant"])
    emb = dev(g["w_emb"])
    x, ag = dev(g["x_init"].clone()), dev(g["agents0"].clone())
    ops.reset_state(x, Nmax, ag)
    t = g["time0"]
    for s in range(g["T"]):
        logits = ops.policy_edge_logits(plan, x[:, 3 * Nmax:], emb)
        p = ops.graphdist_softmax(plan, logits)
        onehot, choice = ops.graphdist_sample(plan, p, uniform=dev(g["u_sample"][s]), want_choice=True)
        assert torch.equal(onehot.cpu(), g["action"][s]), f"action differs at step {s}"
        lp, _ = ops.graphdist_logprob_entropy(plan, p, choice=choice)
        assert abs(lp.item() - g["log_prob"][s].item()) < TOL
        ops.apply_action(plan, x, Nmax, choice=choice)
        ops.core_step(plan, x, Nmax, ec, t, congestion_constant=cc,
                      gumbel=dev(ops.gumbel_from_uniform_cpu(g["u_dir"][s])))
        ops.withdraw_step(plan, x, Nmax, ag, t)
        ops.insert_step(x, Nmax, ag, t, congestion_constant=cc)
        t += 1
        assert torch.equal(x.cpu(), g["x"][s]) and torch.equal(ag.cpu(), g["agents"][s])


@pytest.mark.parametrize("case", ["golden_mid", "torus_philox", "torus_hot", "ring_philox"])
def test_graphdist_rollout_one_launch_equals_the_chain(ops, case):
    """tarl_graphdist_rollout (softmax -> sample -> log_prob in one launch, probabilities never materialised) against the
    three-launch chain: identical actions, rank bytes, SELECTED_ROAD bytes and bit-identical log-probs — on the golden
    graph (unsorted edges, nodes without out-edges), on the 10k-edge torus with device noise, and with logits hot enough
    that some nodes draw nothing (cumulative sum below u: the action is then not one edge per node, log_prob = -inf).
    ring_philox: 1 030 nodes with two out-edges each and device noise: the register sampler shares one Philox block between
    the four lanes that draw consecutive indices when an environment's first index is a multiple of four — here every
    second environment's is not (1 030 = 2 mod 4) and takes the block-per-node path: both must equal the chain."""
    from tarl_hip import synth
    gen = torch.Generator().manual_seed(11)
    if case == "golden_mid":
        g = load_golden("dist_mid")
        ei = g["edge_index"]
        N = int(ei.max()) + 1
        B, T = 5, 0.8
        logits = torch.randn((B, ei.size(1)), generator=gen) * 3
    elif case == "ring_philox":
        N, B, T = 1030, 6, 0.7
        src = torch.arange(N).repeat_interleave(2)
        dst = torch.stack([(torch.arange(N) + 1) % N, (torch.arange(N) + 7) % N], 1).reshape(-1)
        ei = torch.stack([src, dst])
        logits = torch.randn((B, ei.size(1)), generator=gen) * 2
    else:
        net = synth.torus_network(25, 25)
        ei, N = net.edge_index, net.num_roads
        B, T = (7, 1.3) if case == "torus_philox" else (4, 1.0)
        logits = torch.randn((B, ei.size(1)), generator=gen) * (2 if case == "torus_philox" else 30)
    plan = ops.Plan(ei, N)
    E, G = ei.size(1), plan.num_groups
    logits = dev(logits)
    uniform = None
    if case not in ("torus_philox", "ring_philox"):
        uniform = torch.rand((B, G), generator=gen)
        if case == "torus_hot":
            uniform[:, ::7] = 0.99999994            # the largest fp32 below 1: beyond a rounded-down cumulative sum
        uniform = dev(uniform)
    p = ops.graphdist_softmax(plan, logits, T)
    _, ch = ops.graphdist_sample(plan, p, uniform=uniform, seed=5, counter=17, want_onehot=False, want_choice=True)
    lp, _ = ops.graphdist_logprob_entropy(plan, p, choice=ch, want_entropy=False)
    prev = torch.randint(0, 3, (N, B), generator=gen).to(torch.uint8)
    sel_ref = dev(prev.clone())
    import ctypes
    from tarl_hip import lib as _lib
    f = _lib.FusedStruct()             # tarl_fused_apply_choice only touches sel8: drive it through its C entry
    f.sel8 = sel_ref.data_ptr()
    _lib.check(_lib.load().tarl_fused_apply_choice(plan.handle, ctypes.byref(f), B, ch.data_ptr(), _lib.current_stream()))
    ch2 = torch.full((B, N), -7, dtype=torch.int32, device="cuda")
    c8 = torch.zeros((B, N), dtype=torch.uint8, device="cuda")
    sel = dev(prev.clone())
    lp2 = ops.graphdist_rollout(plan, logits, T, uniform=uniform, seed=5, counter=17, choice=ch2, choice8=c8, sel8=sel)
    assert torch.equal(ch2, ch)
    assert torch.equal(lp2, lp), (lp2 - lp).abs().max()
    assert torch.equal(sel, sel_ref) and torch.equal(c8, sel_ref.t())
    if case == "torus_hot":
        assert bool((ch < 0).any()) and bool(torch.isinf(lp).any())
        assert bool(((c8 & 0x80) != 0).any())
    else:
        assert bool(torch.isfinite(lp).all())
    # only the log-prob (the trainer's use when the state is updated elsewhere)
    assert torch.equal(ops.graphdist_rollout(plan, logits, T, uniform=uniform, seed=5, counter=17), lp)
