"""-m gpu: the fused PPO-batch kernels of the student (csrc/ppo_kernels.hip, models/fused_ppo.py) against the same
`evaluate_actions` composed from PyTorch ops on the device (DCD_FUSED_PPO=0: GEMM conv, Linear layers, FixedCategorical — the
path tests/test_gpu_ppo_vs_reference.py pinned to the reference's numbers before these kernels existed; that file now runs
THROUGH them).  Every output and every parameter gradient, fp32 and raw-u8 observations, row counts that do not fill the
kernels' tiles (16-row feature tiles, 128-row head tiles, partial workgroups of the backward passes).

Tolerance: 2e-5 x the tensor's scale (max |reference|) — differently ordered fp32 sums over up to T*N rows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


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
        net.actor[1].linear.weight.mul_(40.0)
    return net.to(DEV)


def _batch(T, N, seed):
    g = torch.Generator().manual_seed(seed)
    img8 = torch.randint(0, 11, (T * N, 5, 5, 3), generator=g, dtype=torch.uint8)
    dir8 = torch.randint(0, 4, (T * N,), generator=g, dtype=torch.uint8)
    lut = torch.from_numpy((np.arange(256, dtype=np.float64) / 10.0).astype(np.float32))
    image = lut[img8.long()].permute(0, 3, 1, 2).contiguous()
    masks = (torch.rand(T * N, 1, generator=g) > 0.1).float()
    h0 = (torch.rand(N, 256, generator=g) * 2 - 1, torch.rand(N, 256, generator=g) * 2 - 1)
    action = torch.randint(0, 7, (T * N, 1), generator=g)
    w = [torch.randn(T * N, 1, generator=g), torch.randn(T * N, 1, generator=g), float(torch.randn((), generator=g))]
    to = lambda t: t.to(DEV)
    return dict(img8=to(img8), dir8=to(dir8), image=to(image), direction=to(dir8.float().view(-1, 1)), masks=to(masks),
                h0=(to(h0[0]), to(h0[1])), action=to(action), w=[to(w[0]), to(w[1]), w[2]])


def _run(net, b, u8, fused, monkeypatch):
    monkeypatch.setenv("DCD_FUSED_PPO", "1" if fused else "0")
    net.zero_grad(set_to_none=True)
    inputs = {"image_u8": b["img8"], "direction_u8": b["dir8"]} if (u8 and fused) else {"image": b["image"], "direction": b["direction"]}
    value, logp, ent, (hT, cT) = net.evaluate_actions(inputs, b["h0"], b["masks"], b["action"])
    loss = (value * b["w"][0]).sum() + (logp * b["w"][1]).sum() + ent * b["w"][2] * value.shape[0] + hT.sum() * 0.5 + cT.sum() * 0.25
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    return value.detach(), logp.detach(), ent.detach(), hT.detach(), grads


def _close(a, b, what, rel=2e-5):
    scale = float(b.abs().max()) + 1e-12
    err = float((a - b).abs().max())
    assert err <= rel * scale + 1e-7, f"{what}: max |diff| {err:.3g} at scale {scale:.3g}"


@pytest.mark.parametrize("T,N", [(1, 3), (5, 3), (32, 37), (64, 64), (7, 300)])
@pytest.mark.parametrize("u8", [False, True])
def test_fused_evaluate_actions_equals_composed_ops(T, N, u8, monkeypatch):
    net = _net(T)
    b = _batch(T, N, seed=11 * T + N)
    ref = _run(net, b, u8, False, monkeypatch)
    out = _run(net, b, u8, True, monkeypatch)
    for k, name in enumerate(("value", "log_prob", "entropy", "h_T")):
        _close(out[k], ref[k], name)
    assert set(out[4]) == set(ref[4])
    for k in ref[4]:
        _close(out[4][k], ref[4][k], "grad " + k)


def test_fused_path_is_the_one_taken_and_falls_back_when_unsupported(monkeypatch):
    from envs import spaces
    from models import MultigridNetwork, fused_ppo
    net = _net(1)
    b = _batch(4, 8, 3)
    calls = {"n": 0}
    orig = fused_ppo.evaluate_actions

    def spy(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    monkeypatch.setattr(fused_ppo, "evaluate_actions", spy)
    monkeypatch.setenv("DCD_FUSED_PPO", "1")
    net.evaluate_actions({"image": b["image"], "direction": b["direction"]}, b["h0"], b["masks"], b["action"])
    assert calls["n"] == 1
    with torch.no_grad():                                             # discard_grad's forward-only form
        net.evaluate_actions({"image": b["image"], "direction": b["direction"]}, b["h0"], b["masks"], b["action"])
    assert calls["n"] == 2
    obs = {"image": spaces.Box(0, 255, (3, 5, 5), "uint8"), "direction": spaces.Box(0, 3, (1,), "uint8")}
    wide = MultigridNetwork(obs, spaces.Discrete(7), conv_filters=8, scalar_fc=5, scalar_dim=4, recurrent_arch="lstm").to(DEV)
    assert not fused_ppo.supported(wide)
    wide.evaluate_actions({"image": b["image"], "direction": b["direction"]}, b["h0"], b["masks"], b["action"])
    assert calls["n"] == 2


def test_unused_outputs_and_repeated_backward_of_one_update(monkeypatch):
    """entropy_coef = 0 leaves the entropy out of the loss; a value-only loss leaves the actor without upstream gradients."""
    monkeypatch.setenv("DCD_FUSED_PPO", "1")
    net = _net(2)
    b = _batch(6, 5, 9)
    for use in ((1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0)):
        got = {}
        for fused in (False, True):
            monkeypatch.setenv("DCD_FUSED_PPO", "1" if fused else "0")
            net.zero_grad(set_to_none=True)
            value, logp, ent, _ = net.evaluate_actions({"image": b["image"], "direction": b["direction"]}, b["h0"], b["masks"], b["action"])
            (use[0] * (value * b["w"][0]).sum() + use[1] * (logp * b["w"][1]).sum() + use[2] * ent * 30.0).backward()
            got[fused] = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in net.named_parameters()}
        for k in got[False]:
            _close(got[True][k], got[False][k], f"{use} grad {k}")


@pytest.mark.parametrize("R", [65536, 131072 + 77, 1265777])
def test_split_k_weight_gradient_with_a_row_count_no_power_of_two_divides(R):
    """matmul_tn(a, b) = a^T b for tall operands: 256-way split of the largest multiple of 256 rows + the product of the rows
    left over (the teacher's conv-window rows come in odd counts: 1 265 777 in a 4096 x 27 batch)."""
    from models import common
    g = torch.Generator().manual_seed(R % 1000)
    a = torch.randn(R, 128, generator=g).to(DEV)
    b = torch.randn(R, 27, generator=g).to(DEV)
    ref = (a.double().t() @ b.double())
    out = common.matmul_tn(a, b)
    assert out.shape == (128, 27)
    scale = float(ref.abs().max())
    assert float((out.double() - ref).abs().max()) <= 2e-5 * scale
    # ... and it is at least as close to the float64 product as the plain fp32 GEMM
    plain = a.t() @ b
    assert float((out.double() - ref).abs().max()) <= 2 * float((plain.double() - ref).abs().max()) + 1e-6 * scale


def test_u8_minibatch_feeds_the_composed_path_through_its_lazy_fp32_view(monkeypatch):
    """DCD_FUSED_PPO=0 on a u8 storage: the recurrent generator still hands over raw bytes (algos/storage.py::LazyU8Obs) and the
    composed network asks for `image` / `direction` — materialised per minibatch by dcd_obs_u8_to_f32 — so the update equals the
    reference's numbers exactly as the fp32 storage's does."""
    from algos import ACAgent, PPO
    from algos.storage import LazyU8Obs
    from tests import ppo_fixtures as F
    monkeypatch.setenv("DCD_FUSED_PPO", "0")
    g = F.load("ppo.npz")
    net = F.student_net(DEV)
    st = F.student_storage(net, g, DEV, u8=True, wide=False)
    adv = st.returns[:-1] - st.value_preds[:-1]
    batch = next(st.recurrent_generator(adv, 1))
    assert isinstance(batch[0], LazyU8Obs) and "image" not in batch[0]
    assert torch.equal(batch[0]["image"].view(12, 4, 3, 5, 5), st.obs["image"][:-1]) or batch[0]["image"].shape == (48, 3, 5, 5)
    algo = PPO(net, clip_param=0.2, ppo_epoch=2, num_mini_batch=1, value_loss_coef=0.5, entropy_coef=0.01, lr=1e-4, eps=1e-5,
               max_grad_norm=0.5, clip_value_loss=True, log_grad_norm=True)
    torch.manual_seed(2)
    vl, al, ent, info = ACAgent(algo, st).update()
    F.check_update(net, g, vl, al, ent, info, atol_w=2e-6)
