"""Numpy-in / numpy-out adapter giving the HIP engine the OracleMg call set (tests only)."""
import numpy as np
import torch

import dcd_hip
from oracle import orc


def _np(d):
    if isinstance(d, dict):
        return {k: _np(v) for k, v in d.items()}
    return d.cpu().numpy()


class HipMg:
    def __init__(self, device="cuda:0", engine=None, **cfg):
        """engine: wrap an existing MultiGridEngine (a replica view) instead of creating one."""
        full = dict(orc.DEFAULT_CFG)
        full.update(cfg)
        self.cfg = orc.make_config(**full)
        self.B, self.S, self.V, self.Z = full["num_envs"], full["size"], full["agent_view_size"], full["random_z_dim"]
        self.dev = torch.device(device)
        n = full.pop("num_envs")
        self.eng = engine if engine is not None else dcd_hip.MultiGridEngine(n, device=device, **full)

    def replica(self, k):
        cfg = {f[0]: getattr(self.cfg, f[0]) for f in self.cfg._fields_ if f[0] != "reserved"}
        return HipMg(device=str(self.dev), engine=self.eng.replica(k), **cfg)

    def clone_levels(self, src):
        self.eng.clone_levels(src.eng)

    def step_pair(self, other, a0, a1, flags=0, term0=None, term1=None):
        outs = []
        for h, term in ((self, term0), (other, term1)):
            o = h.eng.step_buffers()
            if term is not None:
                for k, v in term.items():
                    o["term_obs"][k].copy_(torch.as_tensor(v))
            outs.append(o)
        self.eng.step_pair(other.eng, self._t(a0, torch.int64), self._t(a1, torch.int64), flags, outs[0], outs[1])
        return _np(outs[0]), _np(outs[1])

    def _t(self, a, dtype):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.dev)

    def set_seed(self, seeds):
        self.eng.set_seed(self._t(seeds, torch.int64))

    def poll_errors(self):
        return self.eng.poll_errors()

    def reset(self):
        o = _np(self.eng.reset())
        o.pop("done")
        return o

    def step_adversary(self, loc):
        return _np(self.eng.step_adversary(self._t(loc, torch.int64)))

    def reset_agent(self):
        return _np(self.eng.reset_agent())

    def reset_random(self):
        return _np(self.eng.reset_random())

    def reset_to_encoding(self, enc, idx=None):
        i = None if idx is None else self._t(idx, torch.int32)
        return _np(self.eng.reset_to_encoding(self._t(enc, torch.uint8), i))

    def reset_to_fixed_level(self, enc, idx=None):
        i = None if idx is None else self._t(idx, torch.int32)
        return _np(self.eng.reset_to_fixed_level(self._t(enc, torch.uint8), i))

    def reset_to_actions(self, actions, lens, idx=None):
        i = None if idx is None else self._t(idx, torch.int32)
        return _np(self.eng.reset_to_actions(self._t(actions, torch.int64), self._t(lens, torch.int32), i))

    def mutate_level(self, num_edits):
        return _np(self.eng.mutate_level(num_edits))

    def step(self, actions, flags=0, term_init=None):
        out = self.eng.step_buffers()
        if term_init is not None:
            for k, v in term_init.items():
                out["term_obs"][k].copy_(torch.as_tensor(v))
        self.eng.step(self._t(actions, torch.int64), flags, out)
        return _np(out)

    def get_encodings(self):
        return _np(self.eng.get_encodings())

    def get_metrics(self):
        return _np(self.eng.get_metrics())

    def get_state(self):
        return _np(self.eng.get_state())
