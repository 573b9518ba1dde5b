"""Platform-independent network weights for fixtures too large to store their initial weights in (the PAIRED teacher: 22.5 M
parameters = 90 MB).  `torch`'s initialisers go through LAPACK QR (orthogonal_) whose rounding may differ between hosts; this
fills every tensor of a state dict from an integer hash of (tensor index, element index) — uint64 arithmetic in numpy, exact
everywhere — scaled to the variance the reference's initialiser would give (uniform on [-a, a], a = sqrt(3 / fan_in) * gain),
so activations and gradients have realistic magnitudes.  Used by tests/golden/make_golden.py (with the REFERENCE's module) and
by the GPU tests (with ours): same names, same shapes, same numbers."""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(x):
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def uniform(k, n):
    """n doubles in [0, 1) for tensor number k."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        x = idx * np.uint64(0x9E3779B97F4A7C15) + np.uint64(k + 1) * np.uint64(0xD1B54A32D192ED03)
        x = _splitmix(x)
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def fill_state_dict(shapes, gains=None):
    """{name: float32 array} for an ordered {name: shape}.  gains: optional {substring: gain} (first match wins); biases
    (1-D tensors) get amplitude 0.02 so that no term of the forward pass is identically zero."""
    out = {}
    for k, (name, shape) in enumerate(shapes.items()):
        shape = tuple(int(s) for s in shape)
        n = int(np.prod(shape)) if shape else 1
        u = uniform(k, n) * 2.0 - 1.0
        if len(shape) <= 1:
            a = 0.02
        else:
            fan_in = int(np.prod(shape[1:]))
            gain = 1.0
            for sub, g in (gains or {}).items():
                if sub in name:
                    gain = g
                    break
            a = np.sqrt(3.0 / fan_in) * gain
        out[name] = (u * a).astype(np.float32).reshape(shape)
    return out


def teacher_tape(rng, T, B, S=15):
    """A constructive teacher's observation tape, restated from SURVEY A.1 (adversarial.py:452-539): an empty walled grid,
    then one placement per step — walls on empty cells for steps 0..T-3, the goal at step T-2 (overwrites), the agent at step
    T-1 (a random free cell when its cell holds the goal).  Returns (enc u8[T+1,B,S,S,3] indexed [x][y][c], loc i64[T,B]):
    consecutive frames differ in at most one cell, as the carried input projection assumes."""
    enc = np.zeros((T + 1, B, S, S, 3), np.uint8)
    g = np.zeros((B, S, S, 3), np.uint8)
    g[..., 0] = 1                                     # empty (1,0,0)
    for e in (0, S - 1):
        g[:, e, :, :] = (2, 5, 0)                     # wall (2,5,0)
        g[:, :, e, :] = (2, 5, 0)
    enc[0] = g
    loc = rng.integers(0, (S - 2) ** 2, (T, B))
    for t in range(T):
        for b in range(B):
            x, y = int(loc[t, b]) % (S - 2) + 1, int(loc[t, b]) // (S - 2) + 1
            if t == T - 2:
                g[b, x, y] = (8, 1, 0)
            elif t == T - 1:
                if g[b, x, y, 0] == 8:
                    free = np.argwhere(g[b, 1:-1, 1:-1, 0] == 1)
                    x, y = (free[rng.integers(len(free))] + 1).tolist()
                g[b, x, y] = (10, 0, rng.integers(0, 4))
            elif g[b, x, y, 0] == 1:
                g[b, x, y] = (2, 5, 0)
        enc[t + 1] = g
    return enc, loc
