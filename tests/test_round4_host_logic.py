"""CPU: host-side pieces added in round 4 that need no GPU — the lazy u8 minibatch observations, the start-up-environment guard
of the concurrent-update switch, the shape test of the fused PPO path, the remainder rule of the split-K contraction."""
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lazy_u8_obs_materialises_the_reference_view_once_and_only_on_demand():
    from algos.storage import LazyU8Obs
    calls = []

    def convert(src, dst, sl):                                       # stand-in for dcd_obs_u8_to_f32: k / 10.0, HWC -> CHW
        calls.append(1)
        dst["image"][sl] = torch.from_numpy((src["image"].numpy().astype(np.float64) / 10.0).astype(np.float32)).permute(0, 3, 1, 2)
        dst["direction"][sl] = src["direction"].float().view(-1, 1)
    img8 = torch.randint(0, 11, (6, 5, 5, 3), dtype=torch.uint8)
    dir8 = torch.randint(0, 4, (6,), dtype=torch.uint8)
    o = LazyU8Obs(img8, dir8, convert)
    assert "image_u8" in o and "image" not in o and not calls       # the fused path asks for the bytes: nothing converted
    assert o.get("random_z") is None and o.get("missing", 7) == 7
    img = o["image"]
    assert calls == [1] and img.shape == (6, 3, 5, 5) and torch.equal(img[2, 1], (img8[2, :, :, 1].double() / 10.0).float())
    assert o.get("direction").shape == (6, 1) and calls == [1]      # both entries came from the one conversion
    assert "image" in o and torch.equal(o["image"], img)
    try:
        o["time_step"]
        raise AssertionError("unknown key must raise")
    except KeyError:
        pass


def test_startup_env_reads_the_environment_the_process_started_with():
    code = ("import os, sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from envs.runners.adversarial_runner import _startup_env\n"
            "os.environ['TENSILE_STREAMK_DATA_PARALLEL'] = '1'; os.environ['DCD_PROBE_LATE'] = 'x'\n"
            "print(_startup_env('TENSILE_STREAMK_DATA_PARALLEL'), _startup_env('DCD_PROBE_EARLY'), _startup_env('DCD_PROBE_LATE'))\n"
            % (ROOT, os.path.join(ROOT, "dcd-isaac_amd")))
    env = {k: v for k, v in os.environ.items() if k != "TENSILE_STREAMK_DATA_PARALLEL"}
    env["DCD_PROBE_EARLY"] = "yes"
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split() == ["None", "yes", "None"]               # values set from Python after start-up are not seen


def test_fused_ppo_path_is_for_the_registered_student_only():
    from envs import spaces
    from models import MultigridNetwork, fused_ppo
    obs = {"image": spaces.Box(0, 255, (3, 5, 5), "uint8"), "direction": spaces.Box(0, 3, (1,), "uint8")}
    student = MultigridNetwork(obs, spaces.Discrete(7), scalar_fc=5, scalar_dim=4, recurrent_arch="lstm", recurrent_hidden_size=256)
    assert fused_ppo.supported(student)
    assert not fused_ppo.supported(MultigridNetwork(obs, spaces.Discrete(7), conv_filters=8, scalar_fc=5, scalar_dim=4, recurrent_arch="lstm"))
    assert not fused_ppo.supported(MultigridNetwork(obs, spaces.Discrete(7), scalar_fc=5, scalar_dim=4, recurrent_arch="gru"))
    assert not fused_ppo.supported(MultigridNetwork(obs, spaces.Discrete(12), scalar_fc=5, scalar_dim=4, recurrent_arch="lstm"))
    adv = {"image": spaces.Box(0, 255, (3, 15, 15), "uint8"), "time_step": spaces.Box(0, 27, (1,), "uint8"), "random_z": spaces.Box(0, 1.0, (50,), "float32")}
    teacher = MultigridNetwork(adv, spaces.Discrete(169), conv_filters=128, scalar_fc=10, scalar_dim=28, random_z_dim=50, recurrent_arch="lstm")
    assert not fused_ppo.supported(teacher)
    # on CPU tensors evaluate_actions takes the composed path whatever the shape (the product's fused path needs the device)
    T, N = 3, 2
    inputs = {"image": torch.rand(T * N, 3, 5, 5), "direction": torch.randint(0, 4, (T * N, 1)).float()}
    v, lp, ent, _ = student.evaluate_actions(inputs, (torch.zeros(N, 256), torch.zeros(N, 256)), torch.ones(T * N, 1), torch.randint(0, 7, (T * N, 1)))
    assert v.shape == (T * N, 1) and lp.shape == (T * N, 1) and ent.dim() == 0


def test_matmul_tn_on_cpu_is_the_plain_product():
    from models import common
    a, b = torch.randn(1000, 7), torch.randn(1000, 5)
    assert torch.equal(common.matmul_tn(a, b), a.t() @ b)
