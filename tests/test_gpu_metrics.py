"""GPU: what SimulatorEnv._step logs per step (src/reinforcement_learning.py:278-294) — the leg histogram's (departed,
arrived) counts, DirectionMPNN's delta_travel_time (src/direction_mpnn.py:94-96) and the pop / withdraw masks behind
compute_node_metrics — accumulated on the device INSIDE the rollouts (tarl_fused_rollout, tarl_rollout_env), against the
frame API's per-frame outputs (delta_travel_time (B, E), popped / withdrawn (B, N)), which tests/test_gpu_fused.py checks
bit-exactly against the unfused kernels and, through them, the reference's golden rollouts."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engines(B, W=5, H=5, A=900):
    from tarl_hip import synth
    from tarl_hip.engine import SimEngine
    net = synth.torus_network(W, H, heterogeneous=True, seed=3)
    N = net.num_roads
    pops = torch.stack([synth.population(A, N, seed=b, t0=21540, t1=21570) for b in range(B)])
    mk = lambda: SimEngine(net.x.cuda().unsqueeze(0).repeat(B, 1, 1).contiguous(), net.edge_index, net.edge_attr,
                           net.Nmax, pops.clone().cuda(), congestion_constant=net.congestion_constant, seed=9)
    emb = torch.randn(N, generator=torch.Generator().manual_seed(5)).cuda()
    out = []
    for _ in range(2):
        e = mk()
        e.reset()
        e.prepare_policy(emb)
        out.append(e)
    return net, out[0], out[1]


@pytest.mark.parametrize("mode,m_env", [("frames", 3), ("frames", 5), ("env", 2), ("env", 5)])
def test_rollout_logs_equal_the_frame_api(mode, m_env):
    B, T = 5, 60
    net, e1, e2 = _engines(B)
    N, E = e1.N, e1.E
    src = net.edge_index[0].cuda()
    # frame by frame, with the per-frame side outputs of the frame API
    dtt = torch.empty((B, E), device="cuda")
    pop = torch.empty((B, N), dtype=torch.uint8, device="cuda")
    wd = torch.empty((B, N), dtype=torch.uint8, device="cuda")
    dtt_ref, ev_ref, leg_ref = [], [], []
    on_way0 = e1.agents[:, :, 7].sum(1)
    done0 = e1.agents[:, :, 8].sum(1)
    for t in range(T):
        e1.frame_fused(dtt=dtt, popped=pop, withdrawn=wd)
        node_d = torch.zeros((B, N), device="cuda")
        node_d[:, src] = dtt                       # delta_travel_time is a property of the edge's SOURCE road
        dtt_ref.append(node_d)
        ev_ref.append(pop | (wd << 1))
        on_way, done = e1.agents[:, :, 7].sum(1), e1.agents[:, :, 8].sum(1)
        # the reference's leg-histogram row: departures = d(on_way) + d(done), arrivals = d(done)
        leg_ref.append(torch.stack([on_way - on_way0 + done - done0, done - done0], dim=1))
        on_way0, done0 = on_way, done
    dtt_ref, ev_ref, leg_ref = torch.stack(dtt_ref), torch.stack(ev_ref), torch.stack(leg_ref)     # (T, B, ...)
    # one call, logs written by the rollout kernels
    env_minor = mode == "frames"
    shp = (lambda t, k: (t, N, k)) if env_minor else (lambda t, k: (t, k, N))
    ch = torch.zeros(shp(T, B), dtype=torch.uint8, device="cuda")
    ct = torch.zeros(shp(T + 1, B), dtype=torch.uint8, device="cuda")
    rw = torch.zeros((T, B), device="cuda")
    dn = torch.full(shp(T, m_env), -1.0, device="cuda")
    ev = torch.full(shp(T, m_env), 255, dtype=torch.uint8, device="cuda")
    leg = torch.full((T, B, 2), -1, dtype=torch.int32, device="cuda")
    run = e2.rollout_fused if env_minor else e2.rollout_env
    run(T, choice=ch, log_prob=None, reward=rw, counts=ct, metrics_envs=m_env, dtt_node=dn, events=ev, leg=leg)
    assert torch.equal(e1.x, e2.x) and torch.equal(e1.agents, e2.agents)
    if env_minor:
        dn, ev = dn.permute(0, 2, 1), ev.permute(0, 2, 1)
    assert torch.equal(dn, dtt_ref[:, :m_env]), "delta_travel_time per node"
    assert torch.equal(ev, ev_ref[:, :m_env]), "pop / withdraw masks"
    assert torch.equal(leg.float(), leg_ref), "leg histogram (departed, arrived)"
    assert int(leg[:, :, 0].sum()) > 0 and int(leg[:, :, 1].sum()) > 0 and int(ev.sum()) > 0 and float(dn.sum()) >= 0


def test_trainer_keeps_the_logs_and_hourly_counts_follow():
    """VecPPOTrainer collects the logs during training; the hourly per-road departure counts of compute_node_metrics
    (src/transportation_simulator.py:563-670) follow from the event masks by a sum over the frames of each hour."""
    from src.agents.mpnn_agent import MPNNPolicyNet, MPNNValueNetSimple
    from tarl_hip.trainer import VecPPOTrainer
    net, eng, _ = _engines(4)
    N = eng.N
    torch.manual_seed(0)
    pol = MPNNPolicyNet(net.edge_index, N, None, device="cuda")
    val = MPNNValueNetSimple(net.edge_index, N, device="cuda")
    l = val.final_mlp
    tr = VecPPOTrainer(eng, pol.nodes_embedding.weight, [l[0].weight, l[0].bias, l[2].weight, l[2].bias, l[4].weight, l[4].bias],
                       rollout_steps=80, sub_batch_size=8, metrics_envs=2,
                       extra_params=[p for n, p in pol.named_parameters() if not n.startswith("nodes_embedding")])
    tr.train_iteration()
    assert tr.leg.shape == (80, 4, 2) and tr.events.shape[0] == 80 and tr.dtt_node.shape[0] == 80
    ev = tr.events.permute(0, 2, 1) if tr.env_minor else tr.events              # (T, m, N)
    hours = (tr.times[:80] // 3600).long()
    moved = ((ev & 1) + ((ev >> 1) & 1)).float()                                  # pops + withdrawals per (frame, env, road)
    hourly = torch.zeros((int(hours.max()) + 1, 2, N), device="cuda").index_add_(0, hours, moved)
    assert float(hourly.sum()) == float(moved.sum()) > 0
    # departures / arrivals add up to the agents' flags at the end of the rollout
    assert torch.equal(tr.leg[:, :, 1].sum(0).float(), eng.agents[:, :, 8].sum(1))
    assert torch.equal((tr.leg[:, :, 0] - tr.leg[:, :, 1]).sum(0).float(), eng.agents[:, :, 7].sum(1))
