"""CPU: the host-side PPO / storage / network code against the two round-4 fixtures generated from the imported reference —
ppo_wide.npz (student, T = 32, B = 32, two minibatches) and ppo_teacher.npz (the 22.5 M-parameter PAIRED teacher over a
constructive tape, T = 27, B = 8).  The `-m gpu` twin (tests/test_gpu_ppo_vs_reference.py) runs the same checks through the
device branches (GEMM conv, split-K weight gradients, fused LSTM node, carried projection, u8 rollout buffers).

Tolerances: per-step network outputs rtol 1e-5 / atol 2e-6; losses rtol 1e-4 (north_star); gradient norms rtol 1e-3; weights
after the update atol 2e-6 (two Adam steps of <= 1e-4 each)."""
import numpy as np
import torch

from tests import ppo_fixtures as F


def test_student_wide_act_truncated_values_and_update_match_reference():
    from algos import ACAgent, PPO
    torch.set_num_threads(2)
    g = F.load("ppo_wide.npz")
    net = F.student_net("cpu")
    st = F.student_storage(net, g, "cpu", u8=False, wide=True)
    T = st.num_steps
    h = st.recurrent_hidden_states[0]
    for t in range(T):
        with torch.no_grad():
            v, a, ld, hc = net.act(st.get_obs(t), (h[:, :256], h[:, 256:]), st.masks[t], deterministic=True)
        np.testing.assert_allclose(v.numpy(), g["act_values"][t], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(ld.numpy(), g["act_logdist"][t], rtol=1e-5, atol=2e-6)
        h = torch.cat(hc, -1)
        if t == 0:
            np.testing.assert_allclose(h.numpy(), g["hidden1"], rtol=1e-5, atol=2e-6)
        st.recurrent_hidden_states[t + 1].copy_(h)
    np.testing.assert_allclose(h.numpy(), g["hiddenT"], rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(st._compute_truncated_value_preds().numpy(), g["truncated_value_preds"], rtol=1e-5, atol=2e-6)
    algo = PPO(net, clip_param=0.2, ppo_epoch=2, num_mini_batch=2, value_loss_coef=0.5, entropy_coef=0.0, lr=1e-4, eps=1e-5,
               max_grad_norm=0.5, clip_value_loss=True, log_grad_norm=True)
    torch.manual_seed(2)
    vl, al, ent, info = ACAgent(algo, st).update()
    F.check_update(net, g, vl, al, ent, info, atol_w=2e-6)


def test_teacher_act_and_update_match_reference():
    from algos import ACAgent, PPO
    torch.set_num_threads(4)
    g = F.load("ppo_teacher.npz")
    net, w0 = F.teacher_net("cpu")
    assert sum(p.numel() for p in net.parameters()) == 22502732
    st = F.teacher_storage(net, g, "cpu", u8=False)
    T = st.num_steps
    h = st.recurrent_hidden_states[0]
    for t in range(T):
        with torch.no_grad():
            v, a, ld, hc = net.act(st.get_obs(t), (h[:, :256], h[:, 256:]), st.masks[t], deterministic=True)
        np.testing.assert_allclose(v.numpy(), g["act_values"][t], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(ld.numpy(), g["act_logdist"][t], rtol=1e-5, atol=2e-6)
        h = torch.cat(hc, -1)
    np.testing.assert_allclose(h.numpy(), g["hiddenT"], rtol=1e-5, atol=5e-6)
    algo = PPO(net, clip_param=0.2, ppo_epoch=2, num_mini_batch=1, value_loss_coef=0.5, entropy_coef=0.0, lr=1e-4, eps=1e-5,
               max_grad_norm=0.5, clip_value_loss=True, log_grad_norm=True)
    torch.manual_seed(6)
    vl, al, ent, info = ACAgent(algo, st).update()
    F.check_update(net, g, vl, al, ent, info, atol_w=2e-6, w0=w0, atol_action=2e-7)
