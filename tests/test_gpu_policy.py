"""-m gpu: the fused rollout-time policy step (dcd_policy_features + GEMM + dcd_policy_cell_heads) against the
PyTorch MultigridNetwork it reads its weights from (reference semantics: models/multigrid_models.py:109-167,
models/common.py:149-156, models/distributions.py:18-56).

Tolerance: fp32 round-off of differently ordered sums — |diff| <= 3e-6 on values / log-probs / hidden state
(|h|,|c| < 1 by construction; logits of a freshly initialised policy head are O(1e-2))."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ATOL = 3e-6


def _net(seed=0):
    from envs import spaces
    from models import MultigridNetwork
    torch.manual_seed(seed)
    obs = {"image": spaces.Box(0, 255, (3, 5, 5), "uint8"), "direction": spaces.Box(0, 3, (1,), "uint8")}
    net = MultigridNetwork(obs, spaces.Discrete(7), scalar_fc=5, scalar_dim=4, recurrent_arch="lstm", recurrent_hidden_size=256)
    with torch.no_grad():                                            # un-degenerate the heads (init gain 0.01, zero biases)
        for p in net.parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
        net.actor[1].linear.weight.mul_(60.0)
    return net.to(DEV)


def _inputs(B, seed=1):
    g = torch.Generator().manual_seed(seed)
    image = (torch.randint(0, 11, (B, 3, 5, 5), generator=g).float() / 10.0).to(DEV)
    direction = torch.randint(0, 4, (B, 1), generator=g).float().to(DEV)
    hc = (torch.rand(B, 512, generator=g) * 2 - 1).to(DEV)
    masks = (torch.rand(B, 1, generator=g) > 0.2).float().to(DEV)
    return image, direction, hc, masks


def _fused(net, B, image, direction, hc, masks, deterministic, counter=0, seed=11):
    from dcd_hip.policy import StudentActor
    assert StudentActor.supported(net)
    fa = StudentActor(net, B, seed=seed, env_index_base=1000)
    fa.counter = counter
    out = dict(hc=torch.zeros(B, 512, device=DEV), value=torch.zeros(B, 1, device=DEV),
               action=torch.zeros(B, 1, dtype=torch.int64, device=DEV), logp=torch.zeros(B, 1, device=DEV),
               log_dist=torch.zeros(B, 7, device=DEV), u=torch.zeros(B, device=DEV))
    fa.act_into(image, direction, hc, masks, out["hc"], out["value"], out["action"], out["logp"], out["log_dist"],
                deterministic=deterministic, u_out=out["u"])
    torch.cuda.synchronize()
    return fa, out


@pytest.mark.parametrize("B", [1, 37, 4096])
def test_fused_act_matches_module(B):
    net = _net()
    image, direction, hc, masks = _inputs(B)
    with torch.no_grad():
        value, action, log_dist, (h, c) = net.act({"image": image, "direction": direction}, (hc[:, :256], hc[:, 256:]), masks,
                                                  deterministic=True)
    fa, out = _fused(net, B, image, direction, hc, masks, deterministic=True)
    assert (out["hc"][:, :256] - h).abs().max().item() <= ATOL and (out["hc"][:, 256:] - c).abs().max().item() <= ATOL
    assert (out["value"] - value).abs().max().item() <= ATOL
    assert (out["log_dist"] - log_dist).abs().max().item() <= ATOL
    assert torch.allclose(out["log_dist"].exp().sum(-1), torch.ones(B, device=DEV), atol=1e-5)
    top2 = log_dist.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5                          # arg max unambiguous at fp32 round-off
    assert clear.float().mean().item() > 0.9
    assert torch.equal(out["action"][clear], action[clear])
    assert torch.equal(out["logp"], out["log_dist"].gather(-1, out["action"]))


def test_sampling_is_inverse_cdf_of_its_own_distribution():
    B = 4096
    net = _net(3)
    image, direction, hc, masks = _inputs(B, 5)
    fa, out = _fused(net, B, image, direction, hc, masks, deterministic=False, counter=7)
    p = out["log_dist"].double().exp().cpu().numpy()
    cum = np.cumsum(p, axis=1)
    a = out["action"].cpu().numpy()[:, 0]
    u = out["u"].double().cpu().numpy()
    assert (u >= 0).all() and (u < 1).all() and 0.45 < u.mean() < 0.55
    lo = np.where(a > 0, cum[np.arange(B), np.maximum(a - 1, 0)], 0.0)
    hi = cum[np.arange(B), a]
    assert (u >= lo - 1e-6).all() and (u < hi + 1e-6).all()
    assert torch.equal(out["logp"], out["log_dist"].gather(-1, out["action"]))
    # a different counter / seed / env gives different draws; the same triple repeats
    _, again = _fused(net, B, image, direction, hc, masks, deterministic=False, counter=7)
    _, other = _fused(net, B, image, direction, hc, masks, deterministic=False, counter=8)
    assert torch.equal(again["u"], out["u"]) and not torch.equal(other["u"], out["u"])
    assert len(np.unique(out["u"].cpu().numpy())) > 0.99 * B


def test_sampling_frequencies():
    """One observation replicated over 4096 envs x 8 counters: empirical action frequencies vs probabilities."""
    B = 4096
    net = _net(4)
    image, direction, hc, masks = (t[:1].expand(B, *t.shape[1:]).contiguous() for t in _inputs(1, 9))
    counts = np.zeros(7)
    for k in range(8):
        _, out = _fused(net, B, image, direction, hc, masks, deterministic=False, counter=k)
        counts += np.bincount(out["action"].cpu().numpy()[:, 0], minlength=7)
    p = out["log_dist"][0].double().exp().cpu().numpy()
    n = counts.sum()
    z = (counts - n * p) / np.sqrt(n * p * (1 - p) + 1e-12)
    assert np.abs(z).max() < 5.0, (counts / n, p)


def test_act_storage_fills_the_reference_fields_and_tracks_optimizer_steps():
    from algos import RolloutStorage
    from envs import spaces
    from dcd_hip.policy import StudentActor
    B, T = 64, 4
    net = _net(6)
    obs = {"image": spaces.Box(0, 255, (3, 5, 5), "uint8"), "direction": spaces.Box(0, 3, (1,), "uint8")}
    st = RolloutStorage(net, T, B, obs, spaces.Discrete(7), 256, recurrent_arch="lstm", use_proper_time_limits=True, device=DEV)
    image, direction, hc, masks = _inputs(B, 2)
    st.obs["image"][1].copy_(image); st.obs["direction"][1].copy_(direction)
    st.recurrent_hidden_states[1].copy_(hc); st.masks[1].copy_(masks)
    fa = StudentActor.try_create(net, B, seed=1)
    fa.act_storage(st, 1, deterministic=True)
    with torch.no_grad():
        value, action, log_dist, (h, c) = net.act(st.get_obs(1), st.get_recurrent_hidden_state(1), st.masks[1], deterministic=True)
    assert (st.value_preds[1] - value).abs().max().item() <= ATOL
    assert (st.action_log_dist[1] - log_dist).abs().max().item() <= ATOL
    assert (st.recurrent_hidden_states[2] - torch.cat([h, c], -1)).abs().max().item() <= ATOL
    assert torch.equal(st.action_log_probs[1], st.action_log_dist[1].gather(-1, st.actions[1]))
    # an optimiser step changes the weights in place: the next act must see them (LSTM copy refreshed, heads read in place)
    opt = torch.optim.SGD(net.parameters(), lr=0.5)
    v, lp, ent, _ = net.evaluate_actions(st.get_obs(1), st.get_recurrent_hidden_state(1), st.masks[1], st.actions[1])
    (v.mean() + lp.mean()).backward()
    opt.step()
    fa.refresh()
    fa.act_storage(st, 1, deterministic=True)
    with torch.no_grad():
        value2, _, log_dist2, (h2, c2) = net.act(st.get_obs(1), st.get_recurrent_hidden_state(1), st.masks[1], deterministic=True)
    assert (value2 - value).abs().max().item() > 1e-4                 # the step did change the function
    assert (st.value_preds[1] - value2).abs().max().item() <= ATOL
    assert (st.recurrent_hidden_states[2] - torch.cat([h2, c2], -1)).abs().max().item() <= ATOL


@pytest.mark.parametrize("u8", [False, True])
def test_value_storage_is_get_value_of_the_last_observation(u8):
    """The bootstrap value after a rollout (adversarial_runner.py:586-590) by the fused launches: equals the module's
    get_value on obs[T], leaves every rollout field of the storage alone."""
    from algos import RolloutStorage
    from envs import spaces
    from dcd_hip.policy import StudentActor
    B, T = 96, 3
    net = _net(8)
    obs = {"image": spaces.Box(0, 255, (3, 5, 5), "uint8"), "direction": spaces.Box(0, 3, (1,), "uint8")}
    st = RolloutStorage(net, T, B, obs, spaces.Discrete(7), 256, recurrent_arch="lstm", use_proper_time_limits=True, device=DEV)
    image, direction, hc, masks = _inputs(B, 4)
    if u8:
        st.enable_u8_obs()
        st.obs_u8["image"][T].copy_((image * 10).round().permute(0, 2, 3, 1).to(torch.uint8))
        st.obs_u8["direction"][T].copy_(direction.view(-1).to(torch.uint8))
        st.mark_obs_stale()
    else:
        st.obs["image"][T].copy_(image); st.obs["direction"][T].copy_(direction)
    st.recurrent_hidden_states[T].copy_(hc); st.masks[T].copy_(masks)
    before = {k: getattr(st, k).clone() for k in ("actions", "action_log_probs", "action_log_dist", "recurrent_hidden_states")}
    fa = StudentActor.try_create(net, B, seed=1)
    v = fa.value_storage(st, T)
    with torch.no_grad():
        want = net.get_value(st.get_obs(-1), st.get_recurrent_hidden_state(-1), st.masks[-1])
    assert v.data_ptr() == st.value_preds[-1].data_ptr()
    assert (st.value_preds[-1] - want).abs().max().item() <= ATOL
    for k, b in before.items():
        assert torch.equal(getattr(st, k), b), k
    st.compute_returns(v, True, 0.995, 0.95)                         # value already in place: no copy onto itself
    assert torch.isfinite(st.returns).all()


def test_unsupported_networks_fall_back_to_the_pytorch_modules():
    from envs import spaces
    from models import MultigridNetwork
    from dcd_hip.policy import StudentActor
    obs = {"image": spaces.Box(0, 255, (3, 15, 15), "uint8"), "time_step": spaces.Box(0, 27, (1,), "uint8"),
           "random_z": spaces.Box(0, 1, (50,), "float32")}
    teacher = MultigridNetwork(obs, spaces.Discrete(169), conv_filters=128, scalar_fc=10, scalar_dim=28, random_z_dim=50,
                               recurrent_arch="lstm", recurrent_hidden_size=256).to(DEV)
    assert not StudentActor.supported(teacher) and StudentActor.try_create(teacher, 8) is None


# ---- fused masked LSTM sequence (PPO update path) ---------------------------------------------------
@pytest.mark.parametrize("T,N", [(1, 5), (7, 33), (48, 256)])
def test_fused_lstm_sequence_matches_composed_ops(T, N, monkeypatch):
    """MaskedLSTMSequence (GEMM + dcd_lstm_seq_*_step per time step) vs the same recurrence composed from PyTorch
    ops with autograd (RNN._lstm_cuda_unfused, itself checked against nn.LSTM chunking on the CPU): outputs,
    final state and every gradient.  Tolerance: fp32 round-off of the differently fused pointwise math,
    relative to each tensor's scale (gradients sum over T*N terms)."""
    from models.common import RNN
    torch.manual_seed(T * 100 + N)
    K, H = 149, 256
    rnn = RNN(K, H, arch="lstm").to(DEV)
    x = torch.randn(T * N, K, device=DEV, requires_grad=True)
    h0 = (torch.rand(N, H, device=DEV) * 2 - 1).requires_grad_()
    c0 = (torch.rand(N, H, device=DEV) * 2 - 1).requires_grad_()
    masks = (torch.rand(T * N, 1, device=DEV) > 0.15).float()
    w_out = torch.randn(T * N, H, device=DEV)
    w_h, w_c = torch.randn(N, H, device=DEV), torch.randn(N, H, device=DEV)

    def run(fused):
        monkeypatch.setenv("DCD_FUSED_LSTM", "1" if fused else "0")
        for p in list(rnn.parameters()) + [x, h0, c0]:
            p.grad = None
        y, (hT, cT) = rnn(x, (h0, c0), masks)
        ((y * w_out).sum() + (hT * w_h).sum() + (cT * w_c).sum()).backward()
        return [y.detach(), hT.detach(), cT.detach(), x.grad.clone(), h0.grad.clone(), c0.grad.clone()] + \
               [p.grad.clone() for p in rnn.parameters()]

    ref, got = run(False), run(True)
    names = ["y", "hT", "cT", "dx", "dh0", "dc0"] + [n for n, _ in rnn.named_parameters()]
    for n, a, b in zip(names, ref, got):
        scale = max(a.abs().max().item(), 1e-6)
        assert (a - b).abs().max().item() <= 2e-5 * scale + 1e-6, (n, (a - b).abs().max().item(), scale)


def test_fused_lstm_sequence_without_final_state_grads(monkeypatch):
    """evaluate_actions ignores the returned state: its grads arrive as None."""
    from models.common import RNN
    torch.manual_seed(0)
    rnn = RNN(20, 256, arch="lstm").to(DEV)
    x = torch.randn(6 * 8, 20, device=DEV)
    masks = torch.ones(6 * 8, 1, device=DEV)
    masks[16:24] = 0
    h0 = c0 = torch.zeros(8, 256, device=DEV)
    outs = []
    for fused in ("0", "1"):
        monkeypatch.setenv("DCD_FUSED_LSTM", fused)
        rnn.zero_grad()
        y, _ = rnn(x, (h0, c0), masks)
        y.pow(2).sum().backward()
        outs.append([y.detach()] + [p.grad.clone() for p in rnn.parameters()])
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() <= 2e-5 * max(a.abs().max().item(), 1e-6) + 1e-6


# ---- incremental input projection of the teacher's PPO batch --------------------------------------------
def test_incremental_teacher_projection_matches_dense_path(monkeypatch):
    """MultigridNetwork._incremental_projection (dense GEMM at t = 0 + per-position products of the changed conv
    features, cumsum over time) vs the dense conv -> Linear path on a real 27-step teacher rollout: same values,
    log-probs, entropy and parameter gradients up to fp32 summation order (the 21 632-term dot products of the dense
    path and the incremental sums round differently: 1e-4 relative to each tensor's scale)."""
    from tests.test_gpu_runner import build, make_args
    args = make_args(ued_algo="paired", num_processes=48)
    runner, venv = build(args)
    teacher = runner.agents["adversary_env"]
    runner._teacher_rollout(teacher, runner.adversary_env_rollout_steps)
    st, net = teacher.storage, teacher.algo.actor_critic
    T, N = st.actions.shape[:2]
    obs = {k: v[:-1].reshape(T * N, *v.shape[2:]) for k, v in st.obs.items()}
    hxs = st.get_recurrent_hidden_state(0)
    masks, actions = st.masks[:-1].reshape(T * N, 1), st.actions.reshape(T * N, 1)
    assert net._incremental_eligible(obs["image"], N)
    w = torch.randn(T * N, 1, device=DEV)

    def run(flag, obs=obs):
        monkeypatch.setenv("DCD_INCREMENTAL_PROJECTION", flag)
        net.zero_grad()
        v, lp, ent, _ = net.evaluate_actions(obs, hxs, masks, actions)
        ((v * w).sum() + (lp * w).sum() + ent).backward()
        return [v.detach(), lp.detach(), ent.detach()] + [p.grad.clone() for p in net.parameters()]

    dense, inc = run("0"), run("1")                                  # batched form: one GEMM over the padded positions
    monkeypatch.setattr(type(net), "INCREMENTAL_MAX_PADDING", 0.0)
    loop = run("1")                                                  # per-position form (skewed position histograms)
    monkeypatch.undo()
    names = ["value", "log_prob", "entropy"] + [n for n, _ in net.named_parameters()]
    for other in (inc, loop):
        for n, a, b in zip(names, dense, other):
            scale = max(a.abs().max().item(), 1e-6)
            assert (a - b).abs().max().item() <= 1e-4 * scale + 1e-6, (n, (a - b).abs().max().item(), scale)
    # the rollout starts every env from the same empty grid (projected once); with one env's first frame altered the
    # t = 0 block goes through the dense GEMM again
    obs2 = {k: v.clone() for k, v in obs.items()}
    obs2["image"][1, 0, 4, 5] = 3.0
    for n, a, b in zip(names, run("0", obs2), run("1", obs2)):
        scale = max(a.abs().max().item(), 1e-6)
        assert (a - b).abs().max().item() <= 1e-4 * scale + 1e-6, (n, (a - b).abs().max().item(), scale)
    # the path really ran: the projection is formed and is sparse on a constructive teacher's rollout
    xp = net._incremental_projection(obs["image"], N)
    assert xp is not None and xp.shape == (T * N, 1024)
    # a batch whose frames are unrelated is not sparse: dense fallback
    noise = torch.rand_like(obs["image"])
    assert net._incremental_projection(noise, N) is None
    venv.close()


def test_split_k_linear_matches_plain_linear():
    """models.common.linear on a PPO-sized batch (weight gradient as a batched GEMM over row slices + sum) vs
    F.linear: same output, input / weight / bias gradients up to fp32 summation order."""
    import torch.nn.functional as F
    from models.common import linear, matmul_tn
    torch.manual_seed(0)
    for R, k, m in [(65536, 256, 32), (131072 + 4096, 27, 16), (70000, 32, 7), (65536 + 3, 64, 300), (110592, 32, 169)]:
        x = torch.randn(R, k, device=DEV, requires_grad=True)
        w = torch.randn(m, k, device=DEV, requires_grad=True)
        b = torch.randn(m, device=DEV, requires_grad=True)
        g = torch.randn(R, m, device=DEV)
        outs = []
        for fn in (F.linear, linear):
            for t in (x, w, b):
                t.grad = None
            y = fn(x, w, b)
            y.backward(g)
            outs.append([y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone()])
        for a, c in zip(*outs):
            assert (a - c).abs().max().item() <= 2e-5 * max(a.abs().max().item(), 1.0)
        a2, b2 = torch.randn(R, m, device=DEV), torch.randn(R, k, device=DEV)
        ref = a2.t() @ b2
        assert (matmul_tn(a2, b2) - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_grid_actor_matches_teacher_module():
    """GridActor (dcd_policy_grid_features + one stacked GEMM + the module's own heads) vs MultigridNetwork.act on a
    teacher rollout's observations: value, normalised logits and (h', c') within fp32 round-off of the 21 948-term
    dot products (1e-5 relative to each tensor's scale); deterministic actions equal where the top-2 gap is clear."""
    from dcd_hip.policy import GridActor
    from tests.test_gpu_runner import build, make_args
    args = make_args(ued_algo="paired", num_processes=40)
    runner, venv = build(args)
    teacher = runner.agents["adversary_env"]
    net, st = teacher.algo.actor_critic, teacher.storage
    assert GridActor.supported(net)
    runner._teacher_rollout(teacher, runner.adversary_env_rollout_steps)       # itself on the GridActor path
    ga = GridActor(net, 40)
    for step in (0, 5, 26):
        obs, hc, masks = st.get_obs(step), st.recurrent_hidden_states[step], st.masks[step]
        with torch.no_grad():
            v0, a0, l0, (h0, c0) = net.act(obs, st.get_recurrent_hidden_state(step), masks, deterministic=True)
            v1, a1, l1, (h1, c1) = ga.act(obs, hc, masks, deterministic=True)
        for n, a, b in (("value", v0, v1), ("logits", l0, l1), ("h", h0, h1), ("c", c0, c1)):
            assert (a - b).abs().max().item() <= 1e-5 * max(a.abs().max().item(), 1.0), (step, n, (a - b).abs().max().item())
        top2 = l0.topk(2, dim=-1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-4
        assert torch.equal(a0[clear], a1[clear])
    assert st.actions.min() >= 0 and st.actions.max() < 169 and torch.isfinite(st.value_preds).all()
    venv.close()


def test_grid_actor_carried_projection_matches_dense_operand(monkeypatch):
    """Teacher acting with the image projection carried from step to step (first frame projected once, then only the
    conv positions that see a changed cell) vs the dense 21 948-wide operand at every step, over all 27 steps of a real
    teacher episode: value, logits and (h', c') within 2e-5 of each tensor's scale at every step (the carried sum adds
    <= 18 products per step in fp32), and the carried path really was taken."""
    from dcd_hip.policy import GridActor
    from tests.test_gpu_runner import build, make_args
    args = make_args(ued_algo="paired", num_processes=40)
    runner, venv = build(args)
    teacher = runner.agents["adversary_env"]
    net, st = teacher.algo.actor_critic, teacher.storage
    runner._teacher_rollout(teacher, runner.adversary_env_rollout_steps)
    ga, gd = GridActor(net, 40), GridActor(net, 40)
    ga.begin_episode()
    assert ga._incremental
    worst = 0.0
    for step in range(runner.adversary_env_rollout_steps):
        obs, hc, masks = st.get_obs(step), st.recurrent_hidden_states[step], st.masks[step]
        with torch.no_grad():
            v0, a0, l0, (h0, c0) = gd.act(obs, hc, masks, deterministic=True)
            v1, a1, l1, (h1, c1) = ga.act(obs, hc, masks, deterministic=True, first=step == 0,
                                          prev_image=st.obs["image"][step - 1] if step else None)
        assert ga._incremental and ga._xp_img is not None, step
        for n, a, b in (("value", v0, v1), ("logits", l0, l1), ("h", h0, h1), ("c", c0, c1)):
            err = (a - b).abs().max().item() / max(a.abs().max().item(), 1.0)
            worst = max(worst, err)
            assert err <= 2e-5, (step, n, err)
        top2 = l0.topk(2, dim=-1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-4
        assert torch.equal(a0[clear], a1[clear])
    # first frames that differ between envs: the episode falls back to the dense operand
    gf = GridActor(net, 40)
    gf.begin_episode()
    obs = {k: v.clone() for k, v in st.get_obs(0).items()}
    obs["image"][3, 0, 5, 5] = 2.0
    with torch.no_grad():
        gf.act(obs, st.recurrent_hidden_states[0], st.masks[0], deterministic=True, first=True)
    assert not gf._incremental
    # DCD_INCREMENTAL_ACT=0: never carried
    monkeypatch.setenv("DCD_INCREMENTAL_ACT", "0")
    g0 = GridActor(net, 40)
    g0.begin_episode()
    assert not g0._incremental
    venv.close()


@pytest.mark.parametrize("B", [40, 333])
def test_grid_actor_device_carried_projection_matches_dense_operand(B):
    """The device-resident carried projection (dcd_grid_delta_rows + tiled batched GEMM + dcd_gather_sum_rows, raw u8
    observations, nothing read back) against the dense operand over a real 27-step teacher episode: value, logits, (h', c')
    within 2e-5 of scale at every step; the (env, slot) -> row table is a bijection onto the bins in env order; a step
    that changes two cells raises through check_carry_errors; a first frame differing in ONE cell stays exact."""
    from dcd_hip import DcdHipError
    from dcd_hip.policy import GridActor
    from tests.test_gpu_runner import build, make_args
    args = make_args(ued_algo="paired", num_processes=B)
    runner, venv = build(args)
    teacher = runner.agents["adversary_env"]
    net, st = teacher.algo.actor_critic, teacher.storage
    assert st.obs_u8 is not None                                       # u8 rollout buffers are the default
    T = runner.adversary_env_rollout_steps
    runner._teacher_rollout(teacher, T)
    img8 = st.obs_u8["image"]
    ga, gd = GridActor(net, B), GridActor(net, B)
    ga.begin_episode()
    for step in range(T):
        obs, hc, masks = st.rollout_obs(step), st.recurrent_hidden_states[step], st.masks[step]
        assert ga.device_carry_supported(obs["image"])
        with torch.no_grad():
            v0, a0, l0, (h0, c0) = gd.act(obs, hc, masks, deterministic=True)
            v1, a1, l1, (h1, c1) = ga.act(obs, hc, masks, deterministic=True, first=step == 0,
                                          prev_image=img8[step - 1] if step else None)
        for n, a, b in (("value", v0, v1), ("logits", l0, l1), ("h", h0, h1), ("c", c0, c1)):
            err = (a - b).abs().max().item() / max(a.abs().max().item(), 1.0)
            assert err <= 2e-5, (step, n, err)
        top2 = l0.topk(2, dim=-1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-4
        assert torch.equal(a0[clear], a1[clear])
        # the binning: rows of (env, slot) pairs are distinct, lie in their position's bin, ranks follow env order
        d = ga._d
        cell, src, count = d["cell"].cpu().numpy(), d["src"].cpu().numpy(), d["count"].cpu().numpy()
        tile = d["tile"].cpu().numpy()
        changed = (img8[step] != (img8[step - 1] if step else img8[step][:1])).any(-1).flatten(1).cpu().numpy()
        assert ((cell >= 0) == changed.any(1)).all() and (changed.sum(1) <= 1).all()
        R = d["R"]
        base = np.concatenate([[0], np.cumsum((count + R - 1) // R * R)])
        rows = src[src >= 0]
        assert len(np.unique(rows)) == len(rows) == count.sum()
        for b in range(B):
            for k in range(9):
                if src[b, k] >= 0:
                    cr, cq = divmod(int(cell[b]), 15)
                    p = (cr - k // 3) * 13 + (cq - k % 3)
                    assert base[p] <= src[b, k] < base[p] + count[p] and tile[src[b, k] // R] == p
        last = {}
        for b in range(B):                                            # env order within a bin
            for k in range(9):
                if src[b, k] >= 0:
                    p = int(tile[src[b, k] // R])
                    assert src[b, k] == base[p] + last.get(p, 0)
                    last[p] = last.get(p, 0) + 1
    ga.check_carry_errors()
    # two changed cells in one step: error word -> exception
    g2 = GridActor(net, B)
    g2.begin_episode()
    two = img8[0].clone()
    two[1, 3, 3, 0] = 2
    two[1, 7, 9, 0] = 2
    obs0 = dict(st.rollout_obs(0))
    with torch.no_grad():
        g2.act(obs0, st.recurrent_hidden_states[0], st.masks[0], deterministic=True, first=True)
        g2.act(dict(obs0, image=two), st.recurrent_hidden_states[0], st.masks[0], deterministic=True, prev_image=img8[0])
    with pytest.raises(DcdHipError):
        g2.check_carry_errors()
    # a first frame that differs from env 0's in one cell is carried exactly
    g1 = GridActor(net, B)
    g1.begin_episode()
    one = img8[0].clone()
    one[2, 4, 6, 0] = 2
    with torch.no_grad():
        v1 = g1.act(dict(obs0, image=one), st.recurrent_hidden_states[0], st.masks[0], deterministic=True, first=True)[0]
        v0 = gd.act(dict(obs0, image=one), st.recurrent_hidden_states[0], st.masks[0], deterministic=True)[0]
    g1.check_carry_errors()
    assert (v0 - v1).abs().max().item() <= 2e-5 * max(v0.abs().max().item(), 1.0)
    venv.close()


@pytest.mark.parametrize("F,G,n_tiles,n_used", [(128, 1024, 9, 7), (64, 256, 3, 3), (128, 1024, 2, 0)])
def test_grid_tile_gemm_matches_fp32_fma_chain(F, G, n_tiles, n_used):
    """dcd_grid_tile_gemm (v_mfma_f32_32x32x2_f32): y[t] = rows[t] @ wp[tile_pos[t]] for t < n_used — against the same
    product in fp64 (|err| <= 2e-6 x sum |a b|: one fp32 rounding per product and per addition over K <= 128) and, exactly,
    against the k-ordered fp32 fmaf chain the instruction is specified to compute (checked on a sample of elements);
    tiles >= n_used keep their old contents."""
    import ctypes as C
    from dcd_hip import check, lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(F + n_tiles)
    P, R = 5, lib().dcd_grid_delta_tile_rows()
    rows = (torch.rand(n_tiles * R, F, generator=g) * 2 - 1).to(DEV)
    wp = (torch.rand(P, F, G, generator=g) * 2 - 1).to(DEV)
    tile = torch.randint(0, P, (n_tiles,), generator=g).to(torch.int32).to(DEV)
    used = torch.tensor([n_used], dtype=torch.int32, device=DEV)
    y = torch.full((n_tiles * R, G), 7.0, device=DEV)
    check(lib().dcd_grid_tile_gemm(ptr(rows), ptr(wp), ptr(tile, torch.int32), ptr(used, torch.int32), ptr(y), C.c_int32(n_tiles),
                                   C.c_int32(F), C.c_int32(G), stream_ptr()))
    torch.cuda.synchronize()
    a = rows.view(n_tiles, R, F).double()
    w = wp[tile.long()].double()
    want = torch.bmm(a, w)
    bound = torch.bmm(a.abs(), w.abs()) * 2e-6 + 1e-30
    got = y.view(n_tiles, R, G)
    assert ((got[:n_used].double() - want[:n_used]).abs() <= bound[:n_used]).all()
    assert (got[n_used:] == 7.0).all()
    if n_used:
        rows_h, wp_h, tile_h, got_h = rows.cpu().numpy(), wp.cpu().numpy(), tile.cpu().numpy(), got.cpu().numpy()
        rng = np.random.default_rng(1)
        for _ in range(64):
            t, r, c = int(rng.integers(n_used)), int(rng.integers(R)), int(rng.integers(G))
            acc = np.float32(0)
            for k in range(F):                                       # fmaf(a_k, b_k, acc): the product is exact in fp64, one rounding
                acc = np.float32(np.float64(rows_h[t * R + r, k]) * np.float64(wp_h[tile_h[t], k, c]) + np.float64(acc))
            assert got_h[t, r, c] == acc, (t, r, c)


def test_paired_student_actors_equal_two_single_actors():
    """dcd_policy_features_u8_pair + one batched GEMM + dcd_policy_cell_heads_pair (both PAIRED students per launch) write
    exactly what two single-agent act_storage calls write: every policy field of both storages, bit for bit."""
    from algos import RolloutStorage
    from envs import spaces
    from dcd_hip.policy import PairedStudentActors, StudentActor
    B, T = 200, 3
    nets = [_net(11), _net(12)]
    obs = {"image": spaces.Box(0, 255, (3, 5, 5), "uint8"), "direction": spaces.Box(0, 3, (1,), "uint8")}
    def storages():
        out = []
        for k, net in enumerate(nets):
            st = RolloutStorage(net, T, B, obs, spaces.Discrete(7), 256, recurrent_arch="lstm", use_proper_time_limits=True, device=DEV)
            st.enable_u8_obs()
            g = torch.Generator().manual_seed(100 + k)
            st.obs_u8["image"].copy_(torch.randint(0, 11, st.obs_u8["image"].shape, generator=g).to(torch.uint8))
            st.obs_u8["direction"].copy_(torch.randint(0, 4, st.obs_u8["direction"].shape, generator=g).to(torch.uint8))
            st.recurrent_hidden_states[0].copy_(torch.rand(B, 512, generator=g) * 2 - 1)
            st.masks.copy_((torch.rand(T + 1, B, 1, generator=g) > 0.2).float())
            out.append(st)
        return out
    buf = PairedStudentActors.buffers(nets, B)
    paired = [StudentActor(n, B, seed=5 + k, env_index_base=64, buffers=PairedStudentActors.half(buf, k)) for k, n in enumerate(nets)]
    single = [StudentActor(n, B, seed=5 + k, env_index_base=64) for k, n in enumerate(nets)]
    pa = PairedStudentActors(paired, buf)
    sp, ss = storages(), storages()
    for f in paired + single:
        f.begin_rollout(T)
    for step in range(T):
        pa.act_storage(sp, step)
        for f, st in zip(single, ss):
            f.act_storage(st, step)
    torch.cuda.synchronize()
    for a, b in zip(sp, ss):
        for k in ("value_preds", "actions", "action_log_probs", "action_log_dist", "recurrent_hidden_states"):
            assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert not torch.equal(sp[0].actions, sp[1].actions)             # two different policies / streams


def _teacher_setup(B):
    from tests.test_gpu_runner import build, make_args
    runner, venv = build(make_args(ued_algo="paired", num_processes=B))
    teacher = runner.agents["adversary_env"]
    with torch.no_grad():                                            # un-degenerate the policy head (init gain 0.01)
        teacher.algo.actor_critic.actor[1].linear.weight.mul_(80.0)
        teacher.algo.actor_critic.actor[1].linear.bias.add_(0.3 * torch.randn_like(teacher.algo.actor_critic.actor[1].linear.bias))
    runner._teacher_rollout(teacher, runner.adversary_env_rollout_steps)
    return runner, venv, teacher


def _wide_heads(ga, gates, hc, masks, deterministic, counter=0, seed=3):
    import ctypes as C
    from dcd_hip import check, lib, ptr, stream_ptr
    B, NA = ga.B, ga._dims["NA"]
    out = dict(hc=torch.zeros(B, 512, device=DEV), value=torch.zeros(B, 1, device=DEV), action=torch.zeros(B, 1, dtype=torch.int64, device=DEV),
               logp=torch.zeros(B, 1, device=DEV), log_dist=torch.zeros(B, NA, device=DEV), u=torch.zeros(B, device=DEV))
    check(lib().dcd_policy_wide_cell_heads(C.byref(ga._struct), ptr(ga.packed), ptr(gates), ptr(hc), ptr(masks), ptr(out["hc"]), ptr(out["value"]),
                                           ptr(out["action"], torch.int64), ptr(out["logp"]), ptr(out["log_dist"]), ptr(out["u"]),
                                           C.c_uint32(seed), C.c_uint32(counter), None, C.c_int64(500), C.c_int32(int(deterministic)),
                                           C.c_int32(B), stream_ptr()))
    torch.cuda.synchronize()
    return out


def test_wide_cell_heads_match_the_teacher_module_and_sample_their_own_distribution():
    """dcd_policy_wide_cell_heads (169 actions) on the gates of a real teacher step: (h', c'), value and normalised logits
    within 3e-6 of the PyTorch cell + heads, arg max equal where unambiguous, log-prob = log_dist[action]; sampled actions are
    the inverse CDF (index order) of the kernel's own distribution at its own uniform; frequencies over 32k draws |z| < 5."""
    from dcd_hip.policy import GridActor
    B = 256
    runner, venv, teacher = _teacher_setup(B)
    net, st = teacher.algo.actor_critic, teacher.storage
    ga = GridActor(net, B)
    assert ga.fused_heads and ga._dims["NA"] == 169
    for step in (0, 5, 26):
        obs, hc, masks = st.rollout_obs(step), st.recurrent_hidden_states[step], st.masks[step]
        with torch.no_grad():
            v0, a0, l0, (h0, c0) = ga._act_dense(obs, hc, masks, deterministic=True)
        out = _wide_heads(ga, ga.gates, hc, masks, deterministic=True)
        assert (out["hc"][:, :256] - h0).abs().max().item() <= ATOL and (out["hc"][:, 256:] - c0).abs().max().item() <= ATOL
        assert (out["value"] - v0).abs().max().item() <= ATOL
        assert (out["log_dist"] - l0).abs().max().item() <= 2e-5       # logits O(1) after the head rescale: 1e-5 relative
        top2 = l0.topk(2, dim=-1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-4
        assert clear.float().mean().item() > 0.9 and torch.equal(out["action"][clear], a0[clear])
        assert torch.equal(out["logp"], out["log_dist"].gather(-1, out["action"]))
        smp = _wide_heads(ga, ga.gates, hc, masks, deterministic=False, counter=step)
        p = smp["log_dist"].double().exp().cpu().numpy()
        cum = np.cumsum(p, axis=1)
        a = smp["action"].cpu().numpy()[:, 0]
        u = smp["u"].double().cpu().numpy()
        lo = np.where(a > 0, cum[np.arange(B), np.maximum(a - 1, 0)], 0.0)
        assert (u >= lo - 2e-6).all() and (u < cum[np.arange(B), a] + 2e-6).all() and 0.4 < u.mean() < 0.6
        assert torch.equal(smp["logp"], smp["log_dist"].gather(-1, smp["action"]))
        assert len(np.unique(a)) > 20                                 # a spread-out 169-way distribution, not one action
    # one env replicated: frequencies
    gates1 = ga.gates[:1].expand(B, -1).contiguous()
    hc1, m1 = st.recurrent_hidden_states[26][:1].expand(B, -1).contiguous(), st.masks[26][:1].expand(B, -1).contiguous()
    counts = np.zeros(169)
    for k in range(128):
        o = _wide_heads(ga, gates1, hc1, m1, deterministic=False, counter=k)
        counts += np.bincount(o["action"].cpu().numpy()[:, 0], minlength=169)
    p = o["log_dist"][0].double().exp().cpu().numpy()
    n = counts.sum()
    z = (counts - n * p) / np.sqrt(n * p * (1 - p) + 1e-12)
    assert np.abs(z).max() < 5.0
    venv.close()


def test_grid_actor_act_storage_equals_act_plus_insert():
    """GridActor.act_storage (carried projection + dcd_policy_grid_rest_features + one GEMM + dcd_policy_wide_cell_heads, straight into
    the storage) against GridActor.act (same carried projection, PyTorch cell / heads) + insert_policy over a real 27-step
    teacher episode, deterministic: value / log-dist / log-prob / hidden state within 2e-5 of scale, actions equal where the
    arg max is unambiguous; the sampled path fills valid actions and the log-prob of the action it drew."""
    from dcd_hip.policy import GridActor
    B = 96
    runner, venv, teacher = _teacher_setup(B)
    net, st = teacher.algo.actor_critic, teacher.storage
    T = runner.adversary_env_rollout_steps
    ga, gf = GridActor(net, B), GridActor(net, B)
    ga.begin_episode(); gf.begin_episode()
    gf.begin_rollout(T)
    keep = {k: getattr(st, k).clone() for k in ("value_preds", "actions", "action_log_probs", "action_log_dist", "recurrent_hidden_states")}
    img8 = st.obs_u8["image"]
    for step in range(T):
        hc, masks = keep["recurrent_hidden_states"][step], st.masks[step]
        st.recurrent_hidden_states[step].copy_(hc)
        with torch.no_grad():
            v, a, l, (h, c) = ga.act(st.rollout_obs(step), hc, masks, deterministic=True, first=step == 0,
                                     prev_image=img8[step - 1] if step else None)
        gf.act_storage(st, step, deterministic=True)
        for name, x, y in (("value", v, st.value_preds[step]), ("log_dist", l, st.action_log_dist[step]),
                           ("h", h, st.recurrent_hidden_states[step + 1][:, :256]), ("c", c, st.recurrent_hidden_states[step + 1][:, 256:])):
            assert (x - y).abs().max().item() <= 2e-5 * max(x.abs().max().item(), 1.0), (step, name)
        top2 = l.topk(2, dim=-1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-4
        assert torch.equal(st.actions[step][clear], a[clear])
        assert torch.equal(st.action_log_probs[step], st.action_log_dist[step].gather(-1, st.actions[step]))
    gf.check_carry_errors()
    gf.act_storage(st, T - 1)                                          # sampled
    assert st.actions[T - 1].min() >= 0 and st.actions[T - 1].max() < 169
    assert torch.equal(st.action_log_probs[T - 1], st.action_log_dist[T - 1].gather(-1, st.actions[T - 1]))
    venv.close()
