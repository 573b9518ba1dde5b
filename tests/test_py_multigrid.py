"""CPU: the pure-Python object-per-cell MultiGrid (oracle/py_multigrid.py — the 'py-subproc' CPU baseline
stand-in) against the C oracle: same encodings + start dirs + action tape -> identical obs / reward / done."""
import numpy as np

from oracle import orc
from oracle.py_multigrid import PyMultiGridEnv, PySubprocVecEnv
from tests import mg_diff


def test_python_env_matches_c_oracle():
    rng = np.random.default_rng(0)
    B, S = 6, 15
    m = orc.OracleMg(num_envs=B, max_steps=60, max_episode_steps=60)
    enc = mg_diff.random_encodings(rng, B, S, exotic=True)
    o = m.reset_to_encoding(enc)
    dirs = m.get_state()[:, 10]
    envs = [PyMultiGridEnv(S, 5, 60, 60) for _ in range(B)]
    for b, e in enumerate(envs):
        ob = e.set_level(enc[b], dirs[b])
        assert np.array_equal(ob["image"], o["image_u8"][b]) and ob["direction"][0] == o["dir_u8"][b]
    for t in range(400):
        a = rng.choice(7, size=B, p=[0.15, 0.15, 0.5, 0.05, 0.05, 0.05, 0.05])
        out = m.step(a)
        for b, e in enumerate(envs):
            ob, r, d, info = e.step(int(a[b]))
            if d:
                ob = e.reset_agent()
            assert bool(d) == bool(out["done"][b]), (t, b)
            assert np.float32(r) == out["reward"][b]
            assert np.array_equal(ob["image"], out["obs"]["image_u8"][b]), (t, b)
            assert ("truncated" in info) == bool(out["info"][b] & 2)


def test_subproc_vec_env_roundtrip():
    rng = np.random.default_rng(1)
    n = 3
    m = orc.OracleMg(num_envs=n)
    enc = mg_diff.random_encodings(rng, n, 15)
    o = m.reset_to_encoding(enc)
    v = PySubprocVecEnv(n)
    try:
        v.set_levels(enc, m.get_state()[:, 10])
        for t in range(20):
            a = rng.integers(0, 3, n)
            obs, rew, done, infos = v.step_env(a)
            out = m.step(a)
            assert np.array_equal(obs["image"], out["obs"]["image_f32"])
            assert np.array_equal(done, out["done"].astype(bool))
    finally:
        v.close()
