"""Differential driver: runs two MultiGrid engines (oracle / hostsim / HIP) through the same
seeded call sequence and compares every output and the full observable state after each call.

Used by tests/test_hostsim_vs_oracle.py (CPU) and tests/test_gpu_env_parity.py (GPU).
"""
import numpy as np


def assert_same(a, b, what):
    if isinstance(a, dict):
        assert set(a) == set(b), (what, set(a) ^ set(b))
        for k in a:
            assert_same(a[k], b[k], f"{what}.{k}")
        return
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    if a.dtype.kind == "f":
        same = a.view(np.uint32) == b.view(np.uint32)
    else:
        same = a == b
    if not same.all():
        bad = np.argwhere(~same)
        raise AssertionError(f"{what}: {len(bad)} mismatches, first at {bad[0].tolist()}: "
                             f"{a[tuple(bad[0])]} vs {b[tuple(bad[0])]}")


def compare_state(x, y, what):
    assert_same(x.get_encodings(), y.get_encodings(), what + ":encodings")
    assert_same(x.get_state(), y.get_state(), what + ":state")
    assert_same(x.get_metrics(), y.get_metrics(), what + ":metrics")
    assert x.poll_errors() == y.poll_errors(), what + ":errors"


def random_encodings(rng, n, S, p_wall=0.15, exotic=False):
    """Valid level encodings [n,S,S,3] with one agent and one goal; optionally doors/keys/etc."""
    enc = np.zeros((n, S, S, 3), np.uint8)
    enc[..., 0] = 1
    enc[:, 0, :, :] = (2, 5, 0)
    enc[:, -1, :, :] = (2, 5, 0)
    enc[:, :, 0, :] = (2, 5, 0)
    enc[:, :, -1, :] = (2, 5, 0)
    for b in range(n):
        cells = [(x, y) for x in range(1, S - 1) for y in range(1, S - 1)]
        rng.shuffle(cells)
        ax, ay = cells.pop()
        gx, gy = cells.pop()
        enc[b, ax, ay] = (10, 0, rng.integers(0, 4))
        enc[b, gx, gy] = (8, 1, 0)
        for (x, y) in cells:
            u = rng.random()
            if u < p_wall:
                enc[b, x, y] = (2, rng.integers(0, 6) if exotic else 5, 0)
            elif exotic and u < p_wall + 0.12:
                t = rng.choice([3, 4, 5, 6, 7, 9])
                if t == 4:
                    enc[b, x, y] = (4, rng.integers(0, 6), rng.integers(0, 3))
                elif t == 9:
                    enc[b, x, y] = (9, 0, 0)
                else:
                    enc[b, x, y] = (t, rng.integers(0, 6), 0)
    return enc


def drive(x, y, rng, n_calls=60, steps_per_call=40, exotic=False, action_fn=None, check_every=1):
    """x, y: engines with the OracleMg call set.  Returns number of episodes seen."""
    B, S = x.B, x.S
    interior = (S - 2) ** 2
    n_adv = x.cfg.n_clutter + 2
    seeds = rng.integers(0, 2 ** 31, B)
    x.set_seed(seeds)
    y.set_seed(seeds)
    episodes = 0
    t_global = 0
    term_x = term_y = None

    def teacher_build():
        assert_same(x.reset(), y.reset(), "reset")
        for k in range(n_adv):
            loc = rng.integers(0, interior, B)
            assert_same(x.step_adversary(loc), y.step_adversary(loc), f"step_adversary[{k}]")
        compare_state(x, y, "after teacher")

    teacher_build()
    assert_same(x.reset_agent(), y.reset_agent(), "reset_agent")
    for call in range(n_calls):
        kind = rng.choice(["steps", "steps", "steps", "teacher", "random", "encoding", "encoding_idx", "mutate",
                           "replay", "steps_rr", "steps_last", "fixed_level"])
        what = f"call{call}:{kind}"
        if kind == "teacher":
            teacher_build()
            assert_same(x.reset_agent(), y.reset_agent(), what)
        elif kind == "random":
            assert_same(x.reset_random(), y.reset_random(), what)
        elif kind == "encoding":
            enc = random_encodings(rng, B, S, exotic=exotic)
            assert_same(x.reset_to_encoding(enc), y.reset_to_encoding(enc), what)
        elif kind == "encoding_idx":
            n = int(rng.integers(1, B + 1))
            idx = rng.permutation(B)[:n]
            enc = random_encodings(rng, n, S, exotic=exotic)
            assert_same(x.reset_to_encoding(enc, idx), y.reset_to_encoding(enc, idx), what)
        elif kind == "fixed_level":
            enc = random_encodings(rng, B, S, exotic=exotic)
            assert_same(x.reset_to_fixed_level(enc), y.reset_to_fixed_level(enc), what)
        elif kind == "mutate":
            ne = int(rng.integers(1, 8))
            assert_same(x.mutate_level(ne), y.mutate_level(ne), what)
        elif kind == "replay":
            acts = rng.integers(0, interior, (B, n_adv))
            lens = np.full((B,), n_adv, np.int32)
            assert_same(x.reset_to_actions(acts, lens), y.reset_to_actions(acts, lens), what)
        else:
            for s in range(steps_per_call):
                if action_fn is not None:
                    a = action_fn(t_global)
                else:
                    # forward-biased so that goals get reached
                    a = rng.choice(7, size=B, p=[0.15, 0.15, 0.5, 0.05, 0.05, 0.05, 0.05])
                flags = 0
                if kind == "steps_rr":
                    flags |= 1
                if kind == "steps_last" and s == steps_per_call - 1:
                    flags |= 2
                ox = x.step(a, flags, term_x)
                oy = y.step(a, flags, term_y)
                term_x, term_y = ox["term_obs"], oy["term_obs"]
                assert_same(ox, oy, f"{what}[{s}]")
                episodes += int((ox["info"] & 1).sum())
                t_global += 1
                if check_every and (s % check_every == 0):
                    compare_state(x, y, f"{what}[{s}]")
        compare_state(x, y, what)
    return episodes
