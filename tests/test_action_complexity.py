"""CPU: dcd_action_complexity (csrc/host_utils.cpp, host memory) against the Python statement of
RolloutStorage.get_action_complexity / get_batched_action_complexity (algos/storage.py:329-369) and of the
`lempel_ziv_complexity` package's phrase count."""
import numpy as np
import torch

from algos.storage import RolloutStorage, lempel_ziv_complexity


def test_lz76_known_answers():
    # Kaspar & Schuster's example parses 1|0|01|1110|1100|1010|... ; the phrase-set definition used by the package:
    assert lempel_ziv_complexity("") == 0
    assert lempel_ziv_complexity("0000") == 2            # 0 | 00 | 0 (already a phrase)
    assert lempel_ziv_complexity("0 1 0 1") == 5         # '0',' ','1',' 0',' 1'
    assert lempel_ziv_complexity("1001111011000010") == 8


def _storage(T, B, n_actions, seed):
    from envs import spaces
    obs = {"image": spaces.Box(0, 255, (3, 5, 5), "uint8"), "direction": spaces.Box(0, 3, (1,), "uint8")}
    st = RolloutStorage(None, T, B, obs, spaces.Discrete(n_actions), 4, recurrent_arch="lstm")
    g = torch.Generator().manual_seed(seed)
    st.actions.copy_(torch.randint(0, n_actions, (T, B, 1), generator=g))
    st.masks.copy_((torch.rand(T + 1, B, 1, generator=g) > 0.06).float())
    st.masks[-1] = 0
    return st


def _python_reference(st):
    tot, n, per = 0, 0, {}
    for b, s in st._episode_action_strings():
        c = lempel_ziv_complexity(s)
        tot, n = tot + c, n + 1
        per.setdefault(b, []).append(c)
    return tot / max(n, 1), per


def test_c_action_complexity_equals_python_loop():
    for T, B, A, seed in [(64, 9, 7, 0), (200, 5, 169, 1), (3, 4, 7, 2), (256, 16, 3, 3)]:
        st = _storage(T, B, A, seed)
        if seed == 2:
            st.masks[0, 1] = 0                                       # a done at t = 0: an empty first segment counts as a trajectory
        ref, per = _python_reference(st)
        assert st.get_action_complexity() == ref
        got = st.get_batched_action_complexity()
        assert got.shape == (B, 1)
        for b in range(B):
            assert abs(got[b, 0].item() - np.float32(np.mean(per[b]))) < 1e-6


def test_threaded_action_complexity_equals_single_thread(monkeypatch):
    """Envs are split over host threads (>= 64 envs each): same totals and per-env values as one thread and as the loop."""
    st = _storage(40, 331, 7, 5)
    monkeypatch.setenv("DCD_HOST_THREADS", "1")
    one, one_env = st.get_action_complexity(), st.get_batched_action_complexity()
    monkeypatch.setenv("DCD_HOST_THREADS", "5")
    many, many_env = st.get_action_complexity(), st.get_batched_action_complexity()
    assert one == many and torch.equal(one_env, many_env)
    ref, _ = _python_reference(st)
    assert many == ref
