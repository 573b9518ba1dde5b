"""CPU: MultigridNetwork, RolloutStorage (host pieces) and PPO against the reference's outputs captured
in tests/golden/ppo.npz (reference modules imported in the build container, seeded).

Tolerances (fp32): network forward / value predictions rtol 1e-5; PPO losses rtol 1e-4;
parameters after the update atol 2e-6 (Adam lr 1e-4 on fp32 weights).
"""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")
T, B = 12, 4


def _net(g):
    from envs import spaces
    from models import MultigridNetwork
    obs = {"image": spaces.Box(0, 255, (3, 5, 5), "uint8"), "direction": spaces.Box(0, 3, (1,), "uint8")}
    net = MultigridNetwork(obs, spaces.Discrete(7), scalar_fc=5, scalar_dim=4, recurrent_arch="lstm", recurrent_hidden_size=256)
    net.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w_") and not k.startswith("w_after_")})
    return net, obs


def _storage(net, obs_space, g, proper=True):
    from algos import RolloutStorage
    from envs import spaces
    st = RolloutStorage(net, T, B, obs_space, spaces.Discrete(7), 256, recurrent_arch="lstm", use_proper_time_limits=proper)
    st.obs["image"].copy_(torch.from_numpy(g["img"]))
    st.obs["direction"].copy_(torch.from_numpy(g["direction"]))
    st.truncated_obs["image"].copy_(torch.from_numpy(g["trunc_img"]))
    st.truncated_obs["direction"].copy_(torch.from_numpy(g["trunc_dir"]))
    st.masks.copy_(torch.from_numpy(g["masks"]))
    st.bad_masks.copy_(torch.from_numpy(g["bad_masks"]))
    st.rewards.copy_(torch.from_numpy(g["rewards"]))
    st.recurrent_hidden_states.copy_(torch.from_numpy(g["hidden"]))
    st.actions.copy_(torch.from_numpy(g["act_actions"]))
    st.value_preds[:-1].copy_(torch.from_numpy(g["act_values"]))
    st.value_preds[-1].copy_(torch.from_numpy(g["next_value"]))
    st.action_log_dist.copy_(torch.from_numpy(g["act_logdist"]))
    st.action_log_probs.copy_(torch.from_numpy(g["action_log_probs"]))
    st.returns.copy_(torch.from_numpy(g["returns"]))
    return st


def test_same_seed_gives_reference_initial_weights():
    from envs import spaces
    from models import MultigridNetwork
    g = np.load(os.path.join(G, "ppo.npz"))
    torch.set_num_threads(1)
    torch.manual_seed(0)
    obs = {"image": spaces.Box(0, 255, (3, 5, 5), "uint8"), "direction": spaces.Box(0, 3, (1,), "uint8")}
    net = MultigridNetwork(obs, spaces.Discrete(7), scalar_fc=5, scalar_dim=4, recurrent_arch="lstm", recurrent_hidden_size=256)
    assert sum(p.numel() for p in net.parameters()) == int(g["n_params"]) == 433953
    for k, v in net.state_dict().items():
        np.testing.assert_allclose(v.numpy(), g["w_" + k], rtol=0, atol=1e-6, err_msg=k)


def test_teacher_net_shapes_match_reference():
    from envs import spaces
    from models import MultigridNetwork
    s = np.load(os.path.join(G, "teacher_shapes.npz"))
    adv = {"image": spaces.Box(0, 255, (3, 15, 15), "uint8"), "time_step": spaces.Box(0, 27, (1,), "uint8"),
           "random_z": spaces.Box(0, 1.0, (50,), "float32")}
    net = MultigridNetwork(adv, spaces.Discrete(169), conv_filters=128, scalar_fc=10, scalar_dim=28, random_z_dim=50,
                           recurrent_arch="lstm", recurrent_hidden_size=256)
    assert sum(p.numel() for p in net.parameters()) == int(s["n_params"]) == 22502732
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == tuple(s[k]), k


def test_act_and_evaluate_match_reference():
    g = np.load(os.path.join(G, "ppo.npz"))
    torch.set_num_threads(1)
    net, obs_space = _net(g)
    hid = torch.from_numpy(g["hidden"])
    for t in range(T):
        obs = {"image": torch.from_numpy(g["img"][t]), "direction": torch.from_numpy(g["direction"][t])}
        with torch.no_grad():
            v, a, ld, h = net.act(obs, (hid[t, :, :256], hid[t, :, 256:]), torch.from_numpy(g["masks"][t]), deterministic=True)
        np.testing.assert_allclose(v.numpy(), g["act_values"][t], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ld.numpy(), g["act_logdist"][t], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(torch.cat(h, -1).numpy(), g["hidden"][t + 1], rtol=1e-5, atol=1e-6)
    st = _storage(net, obs_space, g)
    flat = lambda x: x[:-1].reshape(T * B, *x.shape[2:])
    with torch.no_grad():
        ev, elp, eent, _ = net.evaluate_actions({"image": flat(st.obs["image"]), "direction": flat(st.obs["direction"])},
                                                st.get_recurrent_hidden_state(0), flat(st.masks), st.actions.reshape(T * B, 1))
    np.testing.assert_allclose(ev.numpy(), g["eval_values"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(elp.numpy(), g["eval_logp"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(eent.item(), g["eval_entropy"], rtol=1e-5)


def test_truncated_value_preds_match_reference():
    g = np.load(os.path.join(G, "ppo.npz"))
    torch.set_num_threads(1)
    net, obs_space = _net(g)
    st = _storage(net, obs_space, g)
    out = st._compute_truncated_value_preds().numpy()
    np.testing.assert_allclose(out, g["truncated_value_preds"], rtol=1e-5, atol=1e-6)
    counts = (g["bad_masks"][..., 0] == 0).sum(0)
    assert (counts == 1).any() and (counts >= 2).any()          # both the skipped and the forwarded case


def test_ppo_update_matches_reference():
    from algos import ACAgent, PPO
    g = np.load(os.path.join(G, "ppo.npz"))
    torch.set_num_threads(1)
    net, obs_space = _net(g)
    st = _storage(net, obs_space, g)
    algo = PPO(net, clip_param=0.2, ppo_epoch=2, num_mini_batch=1, value_loss_coef=0.5, entropy_coef=0.01, lr=1e-4,
               eps=1e-5, max_grad_norm=0.5, clip_value_loss=True, log_grad_norm=True)
    agent = ACAgent(algo, st)
    torch.manual_seed(2)
    vl, al, ent, info = agent.update()
    np.testing.assert_allclose(vl, g["ppo_value_loss"], rtol=1e-4)
    np.testing.assert_allclose(al, g["ppo_action_loss"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(ent, g["ppo_entropy"], rtol=1e-4)
    np.testing.assert_allclose(info["grad_norms"], g["ppo_grad_norms"], rtol=1e-3)
    for k, v in net.state_dict().items():
        np.testing.assert_allclose(v.numpy(), g["w_after_" + k], rtol=0, atol=2e-6, err_msg=k)
    # after_update carried the last step over (storage.py:195-203)
    assert torch.equal(st.masks[0], st.masks[-1]) and torch.equal(st.obs["image"][0], st.obs["image"][-1])
    # discard_grad: losses computed, parameters frozen (robust PLR, ppo.py:129-130)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    st.returns.copy_(torch.from_numpy(g["returns"]))
    torch.manual_seed(3)
    agent.update(discard_grad=True)
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k]), k


def test_lz_complexity_and_agent_facade():
    from algos.storage import lempel_ziv_complexity
    assert lempel_ziv_complexity("") == 0 and lempel_ziv_complexity("aaaa") == 2 and lempel_ziv_complexity("abab") == 3


def test_gemm_conv_and_manual_lstm_formulas_match_torch_modules():
    """The CUDA-path formulas (im2col GEMM conv, explicit masked LSTM) evaluated on CPU tensors equal the
    nn.Conv2d / nn.LSTM path the reference uses (fp32 tolerance)."""
    from models.common import Conv2d_tf, RNN
    torch.manual_seed(0)
    conv = Conv2d_tf(3, 16, kernel_size=3, stride=1, padding="valid")
    x = torch.rand(37, 3, 5, 5)
    np.testing.assert_allclose(conv._valid_conv_gemm(x).detach().numpy(), conv(x).detach().numpy(), rtol=1e-5, atol=1e-6)
    rnn = RNN(149, 256, "lstm")
    T, N = 9, 5
    xs = torch.randn(T * N, 149)
    masks = (torch.rand(T * N, 1) > 0.3).float()
    h0 = (torch.randn(N, 256), torch.randn(N, 256))
    y_ref, s_ref = rnn(xs, h0, masks)
    y, s = rnn._lstm_cuda(xs, h0, masks, single_step=False)
    np.testing.assert_allclose(y.detach().numpy(), y_ref.detach().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(s[1].detach().numpy(), s_ref[1].detach().numpy(), rtol=1e-5, atol=2e-6)
    y1_ref, _ = rnn(xs[:N], h0, masks[:N])
    y1, _ = rnn._lstm_cuda(xs[:N], h0, masks[:N], single_step=True)
    np.testing.assert_allclose(y1.detach().numpy(), y1_ref.detach().numpy(), rtol=1e-5, atol=2e-6)
