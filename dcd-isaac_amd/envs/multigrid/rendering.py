"""Level images from grid encodings (reference: Grid.render / Grid.render_tile, envs/multigrid/multigrid.py:159-261,
Agent.render :140-153; drawn with gym-minigrid's rendering primitives, which are recalled here — gym-minigrid is not
in the reference tree, so pixels are a functional restatement, not a pinned one).

Host-side and cold: `venv.get_images()` serves `train.py`'s level screenshots (train.py:204-232).  A tile is drawn
once per (type, colour, state) at 3 x 3 supersampling and cached, as in `render_tile`; the level image is a gather of
tiles.  Tile: black with grey lines on its left and top edges; Wall: filled; Floor: half-intensity fill; Goal: green;
Lava: orange with three black waves; Key / Ball / Box / Door: the gym-minigrid glyphs; Agent: a triangle pointing
along its direction in AGENT_COLOURS[0].
"""
import math

import numpy as np

TILE_PIXELS = 32
COLORS = [(255, 0, 0), (0, 255, 0), (0, 0, 255), (112, 39, 195), (255, 255, 0), (100, 100, 100)]   # red green blue purple yellow grey
AGENT_COLOUR = (60, 182, 234)                                                                    # multigrid.py:46-52, agent 0
_cache = {}


def _coords(n):
    c = (np.arange(n) + 0.5) / n
    return np.meshgrid(c, c)                                         # xf[y, x], yf[y, x]


def _fill(img, mask, color):
    img[mask] = color


def _rect(xf, yf, x0, x1, y0, y1):
    return (xf >= x0) & (xf <= x1) & (yf >= y0) & (yf <= y1)


def _circle(xf, yf, cx, cy, r):
    return (xf - cx) ** 2 + (yf - cy) ** 2 <= r * r


def _line(xf, yf, x0, y0, x1, y1, r):
    dx, dy = x1 - x0, y1 - y0
    t = np.clip(((xf - x0) * dx + (yf - y0) * dy) / (dx * dx + dy * dy), 0, 1)
    return (xf - (x0 + t * dx)) ** 2 + (yf - (y0 + t * dy)) ** 2 <= r * r


def _triangle(xf, yf, a, b, c):
    def side(p, q):
        return (xf - q[0]) * (p[1] - q[1]) - (p[0] - q[0]) * (yf - q[1])
    d1, d2, d3 = side(a, b), side(b, c), side(c, a)
    neg, pos = (d1 < 0) | (d2 < 0) | (d3 < 0), (d1 > 0) | (d2 > 0) | (d3 > 0)
    return ~(neg & pos)


def render_tile(obj_type, color, state, tile_size=TILE_PIXELS, subdivs=3):
    key = (int(obj_type), int(color), int(state), tile_size)
    if key in _cache:
        return _cache[key]
    n = tile_size * subdivs
    xf, yf = _coords(n)
    img = np.zeros((n, n, 3), dtype=np.float64)
    _fill(img, _rect(xf, yf, 0, 0.031, 0, 1), (100, 100, 100))
    _fill(img, _rect(xf, yf, 0, 1, 0, 0.031), (100, 100, 100))
    c = np.array(COLORS[min(int(color), 5)], dtype=np.float64)
    t = int(obj_type)
    if t == 2:                                                       # wall
        _fill(img, _rect(xf, yf, 0, 1, 0, 1), c)
    elif t == 3:                                                     # floor
        _fill(img, _rect(xf, yf, 0.031, 1, 0.031, 1), c / 2)
    elif t == 4:                                                     # door: open / closed / locked
        if state == 0:
            _fill(img, _rect(xf, yf, 0.88, 1.00, 0.00, 1.00), c)
            _fill(img, _rect(xf, yf, 0.92, 0.96, 0.04, 0.96), (0, 0, 0))
        else:
            _fill(img, _rect(xf, yf, 0.00, 1.00, 0.00, 1.00), c)
            if state == 2:
                _fill(img, _rect(xf, yf, 0.06, 0.94, 0.06, 0.94), 0.45 * c)
                _fill(img, _rect(xf, yf, 0.52, 0.75, 0.50, 0.56), c)
            else:
                _fill(img, _rect(xf, yf, 0.04, 0.96, 0.04, 0.96), (0, 0, 0))
                _fill(img, _rect(xf, yf, 0.08, 0.92, 0.08, 0.92), c)
                _fill(img, _rect(xf, yf, 0.12, 0.88, 0.12, 0.88), (0, 0, 0))
                _fill(img, _circle(xf, yf, 0.75, 0.50, 0.08), c)
    elif t == 5:                                                     # key
        _fill(img, _rect(xf, yf, 0.50, 0.63, 0.31, 0.88), c)
        _fill(img, _rect(xf, yf, 0.38, 0.50, 0.59, 0.66), c)
        _fill(img, _rect(xf, yf, 0.38, 0.50, 0.81, 0.88), c)
        _fill(img, _circle(xf, yf, 0.56, 0.28, 0.190), c)
        _fill(img, _circle(xf, yf, 0.56, 0.28, 0.064), (0, 0, 0))
    elif t == 6:                                                     # ball
        _fill(img, _circle(xf, yf, 0.5, 0.5, 0.31), c)
    elif t == 7:                                                     # box
        _fill(img, _rect(xf, yf, 0.12, 0.88, 0.12, 0.88), c)
        _fill(img, _rect(xf, yf, 0.18, 0.82, 0.18, 0.82), (0, 0, 0))
        _fill(img, _rect(xf, yf, 0.16, 0.84, 0.47, 0.53), c)
    elif t == 8:                                                     # goal
        _fill(img, _rect(xf, yf, 0, 1, 0, 1), COLORS[1])
    elif t == 9:                                                     # lava
        _fill(img, _rect(xf, yf, 0, 1, 0, 1), (255, 128, 0))
        for i in range(3):
            ylo, yhi = 0.3 + 0.2 * i, 0.4 + 0.2 * i
            for (x0, y0, x1, y1) in ((0.1, ylo, 0.3, yhi), (0.3, yhi, 0.5, ylo), (0.5, ylo, 0.7, yhi), (0.7, yhi, 0.9, ylo)):
                _fill(img, _line(xf, yf, x0, y0, x1, y1, 0.03), (0, 0, 0))
    elif t == 10:                                                    # agent: triangle rotated by 90 degrees per direction
        th = -0.5 * math.pi * int(state)                             # rotate_fn samples the un-rotated shape at the inverse rotation
        xr = 0.5 + (xf - 0.5) * math.cos(th) - (yf - 0.5) * math.sin(th)
        yr = 0.5 + (yf - 0.5) * math.cos(th) + (xf - 0.5) * math.sin(th)
        _fill(img, _triangle(xr, yr, (0.12, 0.19), (0.87, 0.50), (0.12, 0.81)), AGENT_COLOUR)
    tile = img.reshape(tile_size, subdivs, tile_size, subdivs, 3).mean(axis=(1, 3)).astype(np.uint8)      # downsample
    _cache[key] = tile
    return tile


def render_level(enc, tile_size=TILE_PIXELS):
    """enc: uint8 [S, S, 3] indexed [x][y][c] (Grid.encode + the agent cell) -> uint8 [S*tile, S*tile, 3] (row = y)."""
    S = enc.shape[0]
    img = np.zeros((S * tile_size, S * tile_size, 3), dtype=np.uint8)
    for y in range(S):
        for x in range(S):
            t, c, s = enc[x, y]
            img[y * tile_size:(y + 1) * tile_size, x * tile_size:(x + 1) * tile_size] = render_tile(t, c, s, tile_size)
    return img


def render_levels(encodings, tile_size=TILE_PIXELS):
    return [render_level(e, tile_size) for e in np.asarray(encodings)]
