"""Level screenshots (reference: multigrid.py:159-261 Grid.render_tile / Grid.render, util save_images, train.py:204-232).
Parity unpinned: gym-minigrid's drawing primitives are not in the reference tree and the reference holds no image
fixture, so these tests check the documented geometry (tile size, grid lines, colours, agent orientation), not pixels
of a reference render."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dcd-isaac_amd"))

from envs.multigrid import rendering as R  # noqa: E402


def test_tiles():
    empty = R.render_tile(1, 0, 0)
    assert empty.shape == (32, 32, 3) and empty.dtype == np.uint8
    assert (empty[:, 0] == 100).all() and (empty[0, :] == 100).all()          # grid lines on the left and top edges
    assert (empty[2:, 2:] == 0).all()
    wall = R.render_tile(2, 5, 0)
    assert (wall == 100).all()                                                 # grey fill
    goal = R.render_tile(8, 1, 0)
    assert (goal[5:, 5:] == (0, 255, 0)).all()
    lava = R.render_tile(9, 0, 0)
    assert tuple(lava[2, 2]) == (255, 128, 0) and (lava.reshape(-1, 3).sum(1) == 0).any()
    assert R.render_tile(1, 0, 0) is empty                                     # cached


def test_agent_orientation():
    cx = {}
    for d in range(4):
        t = R.render_tile(10, 0, d).astype(int)
        m = t.sum(-1) > 0                                                     # antialiased edge pixels included
        m[0, :] = m[:, 0] = False                                             # grid lines
        ys, xs = np.nonzero(m)
        assert len(xs) > 100
        cx[d] = (xs.mean() - 15.5, ys.mean() - 15.5)
        tip = {0: xs.max(), 1: ys.max(), 2: 31 - xs.min(), 3: 31 - ys.min()}[d]   # the apex reaches 0.87 along DIR_TO_VEC[d]
        assert 26 <= tip <= 28
    # the centroid lies on the blunt side; consecutive directions are quarter turns of each other
    assert cx[0][0] < 0 and abs(cx[0][1]) < 0.6 and cx[1][1] < 0 and cx[2][0] > 0 and cx[3][1] > 0


def test_level_and_save(tmp_path):
    S = 6
    enc = np.zeros((S, S, 3), dtype=np.uint8)
    enc[..., 0] = 1
    enc[0, :, 0] = enc[-1, :, 0] = enc[:, 0, 0] = enc[:, -1, 0] = 2
    enc[0, :, 1] = enc[-1, :, 1] = enc[:, 0, 1] = enc[:, -1, 1] = 5
    enc[4, 2] = (8, 1, 0)                                                      # goal at x=4, y=2
    enc[1, 3] = (10, 0, 3)
    img = R.render_level(enc)
    assert img.shape == (S * 32, S * 32, 3)
    assert (img[2 * 32 + 8:3 * 32, 4 * 32 + 8:5 * 32] == (0, 255, 0)).all()    # row block = y, column block = x
    assert (img[:32] == 100).all()
    assert (img[3 * 32:4 * 32, 32:64] == np.array(R.AGENT_COLOUR)).all(-1).any()
    from util import save_images
    p = str(tmp_path / "shots.png")
    save_images([img, img, img], p, normalize=True)
    from PIL import Image
    back = np.asarray(Image.open(p))
    assert back.shape == (S * 32 + 4, 3 * (S * 32 + 2) + 2, 3)
    np.testing.assert_array_equal(back[2:2 + S * 32, 2:2 + S * 32], (255.0 * img / 255).astype(np.uint8))
    assert (back[:2] == 0).all()
