"""Pure-Python, object-per-cell MultiGrid + a pipe-based subprocess vec-env: the "reference-shaped" CPU
stand-in of SURVEY §8(d) / BASELINE.md §3 (`py-subproc`).

TEST / BASELINE INFRASTRUCTURE ONLY (like everything under oracle/): never imported by dcd-isaac_amd/.

The reference's own CPU path (envs/wrappers/parallel_wrappers.py:19-145 workers + gym_minigrid objects)
cannot run here (gym, gym_minigrid, baselines absent).  This file restates its COST STRUCTURE, not its
code: one Python object per grid cell, `slice` + repeated `rotate_left` + per-cell `encode` for the 5x5
view (multigrid.py:300-338,977-1055), a dict observation per step, and one `multiprocessing` worker per
env exchanging pickled (cmd, data) tuples over pipes with auto-reset on done.  Transition semantics follow
the same spec as oracle/mg_oracle.c and are checked against it (tests/test_py_multigrid.py): given a level
encoding, start direction and action stream the trajectories are identical (the goal respawn position,
which needs the shared Philox stream, is the one thing not modelled — it is unobservable after the
auto-reset).
"""
import multiprocessing as mp

import numpy as np

DIR_TO_VEC = [(1, 0), (0, 1), (-1, 0), (0, -1)]
OBJ = dict(unseen=0, empty=1, wall=2, floor=3, door=4, key=5, ball=6, box=7, goal=8, lava=9, agent=10)


class WorldObj:
    def __init__(self, type, color=0, state=0):
        self.type, self.color, self.state = type, color, state

    def encode(self):
        st = self.state if self.type in ("door", "agent") else 0
        return (OBJ[self.type], self.color, st)

    def can_overlap(self):
        return self.type in ("goal", "floor", "lava") or (self.type == "door" and self.state == 0)


def decode(t, c, s):
    names = {v: k for k, v in OBJ.items()}
    n = names[int(t)]
    if n in ("unseen", "empty"):
        return None
    if n == "goal":
        return WorldObj("goal", 1)
    if n == "lava":
        return WorldObj("lava", 0)
    if n == "agent":
        return WorldObj("agent", 0, int(s))
    if n == "door":
        return WorldObj("door", int(c), 0 if s == 0 else (2 if s == 2 else 1))
    return WorldObj(n, int(c))


class Grid:
    def __init__(self, w, h):
        self.width, self.height = w, h
        self.grid = [None] * (w * h)

    def get(self, i, j):
        return self.grid[j * self.width + i]

    def set(self, i, j, v):
        self.grid[j * self.width + i] = v

    def slice(self, top_x, top_y, w, h):
        g = Grid(w, h)
        for j in range(h):
            for i in range(w):
                x, y = top_x + i, top_y + j
                g.set(i, j, self.get(x, y) if 0 <= x < self.width and 0 <= y < self.height else WorldObj("wall", 5))
        return g

    def rotate_left(self):
        g = Grid(self.height, self.width)
        for i in range(self.width):
            for j in range(self.height):
                v = self.get(i, j)
                if v is not None and v.type == "agent":
                    v = WorldObj("agent", v.color, (v.state - 1) % 4)
                g.set(j, g.height - 1 - i, v)
        return g

    def encode(self):
        a = np.zeros((self.width, self.height, 3), dtype=np.uint8)
        for i in range(self.width):
            for j in range(self.height):
                v = self.get(i, j)
                a[i, j] = (1, 0, 0) if v is None else v.encode()
        return a


class PyMultiGridEnv:
    """Single-agent adversarial MultiGrid level played by a student (step / reset_agent / auto-reset)."""

    def __init__(self, size=15, view=5, max_steps=250, max_episode_steps=250):
        self.size, self.view, self.max_steps, self.max_episode_steps = size, view, max_steps, max_episode_steps
        self.grid = Grid(size, size)
        self.carrying = None

    def set_level(self, enc, start_dir):
        S = self.size
        self.grid = Grid(S, S)
        self.start_pos = None
        for i in range(S):
            for j in range(S):
                v = decode(*enc[i, j])
                if v is not None and v.type == "agent":
                    self.start_pos = (i, j)
                self.grid.set(i, j, v)
        self.start_dir = int(start_dir)
        self.agent_pos = None
        return self.reset_agent()

    def reset_agent(self):
        if self.agent_pos is not None:
            self.grid.set(self.agent_pos[0], self.agent_pos[1], None)
        self.agent_pos, self.agent_dir = self.start_pos, self.start_dir
        self.grid.set(self.start_pos[0], self.start_pos[1], WorldObj("agent", 0, self.agent_dir))
        self.carrying, self.done, self.step_count, self.elapsed = None, False, 0, 0
        return self.gen_obs()

    def gen_obs(self):
        V, (ax, ay), d = self.view, self.agent_pos, self.agent_dir
        top = [(ax, ay - V // 2), (ax - V // 2, ay), (ax - V + 1, ay - V // 2), (ax - V // 2, ay - V + 1)][d]
        g = self.grid.slice(top[0], top[1], V, V)
        for _ in range(d + 1):
            g = g.rotate_left()
        g.set(V // 2, V - 1, self.carrying)
        return {"image": g.encode(), "direction": [d]}

    def step(self, action):
        self.step_count += 1
        reward = 0
        dx, dy = DIR_TO_VEC[self.agent_dir]
        fx, fy = self.agent_pos[0] + dx, self.agent_pos[1] + dy
        fwd = self.grid.get(fx, fy)
        if action == 0 or action == 1:
            self.agent_dir = (self.agent_dir + (3 if action == 0 else 1)) % 4
            self.grid.set(self.agent_pos[0], self.agent_pos[1], WorldObj("agent", 0, self.agent_dir))
        elif action == 2:
            if fwd is not None and fwd.type == "goal":
                self.done = True
                reward = 1 - 0.9 * (self.step_count / self.max_steps)
            elif fwd is not None and fwd.type == "lava":
                self.done = True
            elif fwd is None or fwd.can_overlap():
                self.grid.set(self.agent_pos[0], self.agent_pos[1], None)
                self.agent_pos = (fx, fy)
                self.grid.set(fx, fy, WorldObj("agent", 0, self.agent_dir))
        elif action == 3:
            if fwd is not None and fwd.type in ("key", "ball", "box") and self.carrying is None:
                self.carrying = fwd
                self.grid.set(fx, fy, None)
        elif action == 4:
            if fwd is None and self.carrying is not None:
                self.grid.set(fx, fy, self.carrying)
                self.carrying = None
        elif action == 5:
            if fwd is not None and fwd.type == "door":
                if fwd.state == 2:
                    if self.carrying is not None and self.carrying.type == "key" and self.carrying.color == fwd.color:
                        fwd.state = 0
                else:
                    fwd.state = 1 if fwd.state == 0 else 0
            elif fwd is not None and fwd.type == "box":
                self.grid.set(fx, fy, None)
        obs = self.gen_obs()
        done = self.done or self.step_count >= self.max_steps
        info = {}
        self.elapsed += 1
        if self.elapsed >= self.max_episode_steps:
            info["truncated"] = not done
            info["truncated_obs"] = obs
            done = True
        return obs, reward, done, info


def _worker(remote, parent_remote, size, view, max_steps, max_episode_steps):
    parent_remote.close()
    env = PyMultiGridEnv(size, view, max_steps, max_episode_steps)
    try:
        while True:
            cmd, data = remote.recv()
            if cmd == "step_env":
                ob, r, d, info = env.step(data)
                if d:
                    ob = env.reset_agent()
                remote.send((ob, r, d, info))
            elif cmd == "set_level":
                remote.send(env.set_level(*data))
            elif cmd == "close":
                remote.close()
                break
    except (KeyboardInterrupt, EOFError):
        pass


class PySubprocVecEnv:
    """One worker process per env, pickled tuples over pipes (parallel_wrappers.py:103-145,294-311)."""

    def __init__(self, n, size=15, view=5, max_steps=250, max_episode_steps=250, context="fork"):
        ctx = mp.get_context(context)
        self.remotes, work = zip(*[ctx.Pipe() for _ in range(n)])
        self.ps = [ctx.Process(target=_worker, args=(w, r, size, view, max_steps, max_episode_steps), daemon=True)
                   for w, r in zip(work, self.remotes)]
        for p in self.ps:
            p.start()
        for w in work:
            w.close()
        self.n = n

    def set_levels(self, encs, dirs):
        for r, e, d in zip(self.remotes, encs, dirs):
            r.send(("set_level", (e, d)))
        return [r.recv() for r in self.remotes]

    def step_env(self, actions):
        for r, a in zip(self.remotes, actions):
            r.send(("step_env", int(a)))
        res = [r.recv() for r in self.remotes]
        obs, rews, dones, infos = zip(*res)
        image = np.stack([o["image"] for o in obs]) / 10.0          # VecPreprocessImageWrapper (obs_wrappers.py:88-115)
        image = image.transpose(0, 3, 1, 2).astype(np.float32)
        direction = np.stack([o["direction"] for o in obs]).astype(np.float32)
        return {"image": image, "direction": direction}, np.stack(rews), np.stack(dones), infos

    def close(self):
        for r in self.remotes:
            try:
                r.send(("close", None))
            except Exception:
                pass
        for p in self.ps:
            p.join(timeout=2)
