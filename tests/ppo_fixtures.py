"""Builders shared by the CPU and `-m gpu` PPO-vs-reference tests: network + RolloutStorage on a device, filled from the
reference-generated fixtures tests/golden/{ppo,ppo_wide,ppo_teacher}.npz (tests/golden/make_golden.py: gen_ppo, gen_ppo_wide,
gen_ppo_teacher — the reference's own `MultigridNetwork`, `RolloutStorage`, `PPO` imported in the build container)."""
import os
import sys

import numpy as np
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
if G not in sys.path:
    sys.path.insert(0, G)


def load(name):
    return np.load(os.path.join(G, name))


def student_spaces():
    from envs import spaces
    return {"image": spaces.Box(0, 255, (3, 5, 5), "uint8"), "direction": spaces.Box(0, 3, (1,), "uint8")}, spaces.Discrete(7)


def teacher_spaces():
    from envs import spaces
    return ({"image": spaces.Box(0, 255, (3, 15, 15), "uint8"), "time_step": spaces.Box(0, 27, (1,), "uint8"),
             "random_z": spaces.Box(0, 1.0, (50,), "float32")}, spaces.Discrete(169))


def student_net(device, weights_from="ppo.npz"):
    from models import MultigridNetwork
    g = load(weights_from)
    obs, act = student_spaces()
    net = MultigridNetwork(obs, act, scalar_fc=5, scalar_dim=4, recurrent_arch="lstm", recurrent_hidden_size=256)
    net.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w_") and not k.startswith("w_after_")})
    return net.to(device)


def teacher_net(device):
    """The 22.5 M-parameter teacher with det_weights' host-independent initial weights (what gen_ppo_teacher loaded into the
    reference's module)."""
    import det_weights
    from models import MultigridNetwork
    obs, act = teacher_spaces()
    net = MultigridNetwork(obs, act, conv_filters=128, scalar_fc=10, scalar_dim=28, random_z_dim=50, recurrent_arch="lstm",
                           recurrent_hidden_size=256)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    w0 = det_weights.fill_state_dict(shapes, gains={"actor.1.linear.weight": 0.01})
    net.load_state_dict({k: torch.from_numpy(v) for k, v in w0.items()})
    return net.to(device), w0


def chw10(im8):
    """VecPreprocessImageWrapper (obs_wrappers.py:88-115) on raw [.., x, y, c] bytes: k / 10.0 in fp64, rounded to fp32, CHW."""
    nd = im8.ndim
    return torch.from_numpy((im8.astype(np.float64) / 10.0).astype(np.float32).transpose(*range(nd - 3), nd - 1, nd - 3, nd - 2).copy())


def _fill_common(st, g, device):
    T = st.num_steps
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    st.masks.copy_(tt(g["masks"]))
    st.bad_masks.copy_(tt(g["bad_masks"]))
    st.rewards.copy_(tt(g["rewards"]))
    st.actions.copy_(tt(g["act_actions"]))
    st.value_preds[:-1].copy_(tt(g["act_values"]))
    st.value_preds[-1].copy_(tt(g["next_value"]))
    st.action_log_dist.copy_(tt(g["act_logdist"]))
    st.action_log_probs.copy_(tt(g["action_log_probs"]))
    st.returns.copy_(tt(g["returns"]))
    return tt


def student_storage(net, g, device, u8, wide):
    """Storage after the reference's rollout: wide=False -> ppo.npz (T = 12, B = 4, fp32 observation tape k / 10 stored as
    floats), wide=True -> ppo_wide.npz (T = 32, B = 32, raw bytes).  u8: the product's default rollout buffers (raw bytes,
    fp32 derived on demand by dcd_obs_u8_to_f32) — only possible on a HIP device."""
    from algos import RolloutStorage
    obs, act = student_spaces()
    T, B = (int(g["T"]), int(g["B"])) if wide else (12, 4)
    st = RolloutStorage(net, T, B, obs, act, 256, recurrent_arch="lstm", use_proper_time_limits=True, device=device)
    tt = _fill_common(st, g, device)
    if wide:
        img8, dir8, timg8, tdir8 = g["img8"], g["dir8"], g["trunc_img8"], g["trunc_dir8"]
    else:                                  # the small fixture holds k / 10 as floats in CHW: recover the bytes exactly
        back = lambda a: np.rint(a.astype(np.float64) * 10.0).astype(np.uint8).transpose(0, 1, 3, 4, 2)
        img8, timg8 = back(g["img"]), back(g["trunc_img"])
        dir8, tdir8 = g["direction"][..., 0].astype(np.uint8), g["trunc_dir"][..., 0].astype(np.uint8)
        assert np.array_equal(chw10(img8).numpy(), g["img"])
    if u8:
        assert st.enable_u8_obs()
        st.obs_u8["image"].copy_(tt(img8))
        st.obs_u8["direction"].copy_(tt(dir8))
        st.truncated_obs_u8["image"].copy_(tt(timg8))
        st.truncated_obs_u8["direction"].copy_(tt(tdir8))
        st.mark_obs_stale()
    else:
        st._obs["image"].copy_(chw10(img8).to(device))
        st._obs["direction"].copy_(tt(dir8.astype(np.float32)[..., None]))
        st.truncated_obs["image"].copy_(chw10(timg8).to(device))
        st.truncated_obs["direction"].copy_(tt(tdir8.astype(np.float32)[..., None]))
    if wide:
        st.recurrent_hidden_states[0].copy_(tt(g["hidden0"]))
    else:
        st.recurrent_hidden_states.copy_(tt(g["hidden"]))
    return st


def teacher_storage(net, g, device, u8):
    from algos import RolloutStorage
    obs, act = teacher_spaces()
    T, B = int(g["T"]), int(g["B"])
    st = RolloutStorage(net, T, B, obs, act, 256, recurrent_arch="lstm", use_proper_time_limits=False, device=device)
    tt = _fill_common(st, g, device)
    enc = g["enc"]
    if u8:
        assert st.enable_u8_obs()
        st.obs_u8["image"].copy_(tt(enc))
        st.mark_obs_stale()
    else:
        st._obs["image"].copy_(chw10(enc).to(device))
    st._obs["time_step"].copy_(torch.arange(T + 1, dtype=torch.float32, device=device).view(T + 1, 1, 1).expand(T + 1, B, 1))
    st._obs["random_z"].copy_(tt(g["random_z"]))
    st.recurrent_hidden_states[0].copy_(tt(g["hidden0"]))
    return st


def check_update(net, g, vl, al, ent, info, atol_w, w0=None, rtol_loss=1e-4, atol_action=1e-7, rtol_norm=1e-3, pre=""):
    """The reference's losses / gradient norms / post-update weights (ppo.py:61-146).  pre="": its fp32 CPU run; pre="f64_": the
    same reference code evaluated in float64 (make_golden._as_double).  Returns the largest weight deviation."""
    np.testing.assert_allclose(vl, g[pre + "ppo_value_loss"], rtol=rtol_loss)
    np.testing.assert_allclose(al, g[pre + "ppo_action_loss"], rtol=rtol_loss, atol=atol_action)
    np.testing.assert_allclose(ent, g[pre + "ppo_entropy"], rtol=rtol_loss)
    np.testing.assert_allclose(info["grad_norms"], g[pre + "ppo_grad_norms"], rtol=rtol_norm)
    worst = 0.0
    for k, v in net.state_dict().items():
        v = v.detach().cpu().numpy()
        if pre + "w_after_" + k in g.files:
            ref = g[pre + "w_after_" + k]
            worst = max(worst, float(np.abs(v - ref).max()))
            np.testing.assert_allclose(v, ref, rtol=0, atol=atol_w, err_msg=k)
        else:                               # the teacher's LSTM input weight: column sample + fp64 sums of its change
            cols = g["big_cols"]
            worst = max(worst, float(np.abs(v[:, cols] - g[pre + "w_after_big_sample"]).max()))
            np.testing.assert_allclose(v[:, cols], g[pre + "w_after_big_sample"], rtol=0, atol=atol_w, err_msg=k + " (column sample)")
            d = v.astype(np.float64) - w0[k].astype(np.float64)
            scale = float(g[pre + "big_delta_abs_mean"])
            # a row sums 21 692 changes of mean size `scale`, a column 1 024: deviations allowed at 1e-3 of the summed magnitude
            np.testing.assert_allclose(d.sum(1), g[pre + "big_delta_rowsum"], rtol=0, atol=1e-3 * scale * d.shape[1], err_msg=k + " (row sums)")
            np.testing.assert_allclose(d.sum(0), g[pre + "big_delta_colsum"], rtol=0, atol=1e-3 * scale * d.shape[0], err_msg=k + " (column sums)")
            np.testing.assert_allclose(np.abs(d).mean(), scale, rtol=1e-3)
    return worst
