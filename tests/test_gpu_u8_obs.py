"""-m gpu: raw u8 observations in the students' rollout buffers (RolloutStorage u8 mode, DCD_U8_OBS): the /10 and HWC->CHW
of VecPreprocessImageWrapper (obs_wrappers.py:88-115) moved into the consumers must give bit-identical results to the
engine's own preprocessed fp32 emission."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _obs_pair(B=700, exotic=True, V=5):
    """One engine step with BOTH emissions on levels with every object type."""
    import dcd_hip
    from tests import mg_diff
    eng = dcd_hip.MultiGridEngine(B, device=DEV, n_clutter=25, choose_goal_last=1, agent_view_size=V)
    enc = mg_diff.random_encodings(np.random.default_rng(1), B, 15, exotic=exotic)
    eng.reset_to_encoding(torch.from_numpy(enc).to(DEV))
    a = dcd_hip.engine.random_actions(B, 0, 9, 0, 7, DEV)
    return eng, eng.step(a, 0, eng.step_buffers(u8=True, f32=True))["obs"]


@pytest.mark.parametrize("V", [5, 7, 3])
def test_conversion_kernel_equals_the_engines_fp32_emission(V):
    from dcd_hip import check, lib, ptr, stream_ptr
    eng, o = _obs_pair(V=V)
    img = torch.full_like(o["image_f32"], -1.0)
    d = torch.full_like(o["dir_f32"], -1.0)
    check(lib().dcd_obs_u8_to_f32(ptr(o["image_u8"]), ptr(o["dir_u8"]), ptr(img), ptr(d), C.c_int64(eng.B), C.c_int32(V), stream_ptr()))
    assert torch.equal(img, o["image_f32"]) and torch.equal(d, o["dir_f32"])
    assert o["image_u8"].max() >= 8                                   # goals / lava / agents in view: more than walls


def test_policy_features_from_bytes_equal_features_from_fp32():
    from dcd_hip.policy import StudentActor
    from envs import spaces
    from models import MultigridNetwork
    eng, o = _obs_pair(B=333)
    obs = {"image": spaces.Box(0, 255, (3, 5, 5), "uint8"), "direction": spaces.Box(0, 3, (1,), "uint8")}
    torch.manual_seed(0)
    net = MultigridNetwork(obs, spaces.Discrete(7), scalar_fc=5, scalar_dim=4, recurrent_arch="lstm", recurrent_hidden_size=256).to(DEV)
    outs = []
    for img, d in ((o["image_f32"], o["dir_f32"]), (o["image_u8"], o["dir_u8"])):
        act = StudentActor(net, eng.B, seed=3)
        g = torch.Generator(device=DEV).manual_seed(1)
        hc = torch.randn((eng.B, 512), device=DEV, generator=g)
        masks = (torch.rand((eng.B, 1), device=DEV, generator=g) > 0.2).float()
        res = dict(hc=torch.zeros_like(hc), value=torch.zeros((eng.B, 1), device=DEV), action=torch.zeros((eng.B, 1), dtype=torch.int64, device=DEV),
                   lp=torch.zeros((eng.B, 1), device=DEV), ld=torch.zeros((eng.B, 7), device=DEV))
        act.act_into(img, d, hc, masks, res["hc"], res["value"], res["action"], res["lp"], res["ld"])
        outs.append((act.x.clone(), res))
    assert torch.equal(outs[0][0], outs[1][0])                          # the GEMM operand rows, bit for bit
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k


def test_runner_rollouts_identical_with_u8_and_fp32_buffers(monkeypatch):
    """A PAIRED update with the students' buffers in u8 mode == the same update with fp32 buffers: every storage tensor
    (the derived fp32 observations included), truncated-observation value predictions and losses."""
    from tests.test_gpu_runner import _storage_tensors, build, make_args
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DCD_U8_OBS", mode)
        args = make_args(ued_algo="paired", num_processes=128, num_steps=70, env_name="MultiGrid-MiniGoalLastAdversarial-v0")   # 50-step limit: truncations
        runner, venv = build(args)
        st = runner.agents["agent"].storage
        assert (st.obs_u8 is not None) == (mode == "1")
        stats = runner.run()
        assert (runner.agents["adversary_env"].storage.obs_u8 is not None) == (mode == "1")       # the teacher's full-grid image too
        got[mode] = ({n: {k: v.clone() for k, v in _storage_tensors(runner.agents[n].storage).items() if not k.startswith("_")}
                      for n in ("agent", "adversary_agent", "adversary_env")},
                     {k: stats[k] for k in ("agent_value_loss", "adversary_value_loss", "agent_pg_loss", "mean_env_return")},
                     {k: v.clone() for k, v in st.obs.items()}, st.truncated_value_preds.clone(),
                     {k: v.clone() for k, v in runner.agents["adversary_env"].storage.obs.items()})
        venv.close()
    a, b = got["1"], got["0"]
    assert (a[3] != got["1"][0]["agent"]["value_preds"]).any()          # truncated observations were evaluated
    assert torch.equal(a[3], b[3])
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k                          # derived fp32 observations == engine-written ones
    for k in a[4]:
        assert torch.equal(a[4][k], b[4][k]), ("teacher", k)
    for n in a[0]:
        for k in a[0][n]:
            if k in b[0][n] and "obs" not in k:
                assert torch.equal(a[0][n][k], b[0][n][k]), (n, k)
    assert a[1] == b[1]


def test_grid_actor_on_raw_bytes_equals_grid_actor_on_fp32():
    """The teacher's rollout-time policy step reading the raw full-grid bytes == the same step on the preprocessed fp32
    image over a whole teacher episode: bit for bit on the dense operand (dcd_policy_grid_features_u8, exact k/10 table);
    the carried projection runs on bytes as the device-resident kernel chain and on fp32 as PyTorch index ops — the
    same sum in a different order: within 2e-5 of each tensor's scale."""
    from dcd_hip.policy import GridActor
    from tests.test_gpu_runner import build, make_args
    runner, venv = build(make_args(ued_algo="paired", num_processes=48))
    teacher = runner.agents["adversary_env"]
    net, st = teacher.algo.actor_critic, teacher.storage
    assert st.obs_u8 is not None
    runner._teacher_rollout(teacher, runner.adversary_env_rollout_steps)
    for carried in (False, True):
        g32, g8 = GridActor(net, 48), GridActor(net, 48)
        if carried:
            g32.begin_episode(); g8.begin_episode()
        for step in range(runner.adversary_env_rollout_steps):
            hc, masks = st.recurrent_hidden_states[step], st.masks[step]
            kw32 = dict(first=step == 0, prev_image=st.obs["image"][step - 1] if step else None) if carried else {}
            kw8 = dict(first=step == 0, prev_image=st.obs_u8["image"][step - 1] if step else None) if carried else {}
            with torch.no_grad():
                r32 = g32.act(st.get_obs(step), hc, masks, deterministic=True, **kw32)
                r8 = g8.act(st.rollout_obs(step), hc, masks, deterministic=True, **kw8)
            assert st.rollout_obs(step)["image"].dtype == torch.uint8
            if carried:
                assert g32._incremental and g8._incremental
            for i, (x, y) in enumerate(zip((r32[0], r32[1], r32[2], r32[3][0], r32[3][1]), (r8[0], r8[1], r8[2], r8[3][0], r8[3][1]))):
                if not carried:
                    assert torch.equal(x, y), (carried, step)
                elif i != 1:                                             # (the arg max may flip on a 1e-7 logit tie)
                    assert (x - y).abs().max().item() <= 2e-5 * max(x.abs().max().item(), 1.0), (carried, step, i)
        if carried:
            g8.check_carry_errors()
    venv.close()
