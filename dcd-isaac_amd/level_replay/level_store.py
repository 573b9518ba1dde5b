"""LevelStore — seed <-> level map with parent lineage (reference: level_replay/level_store.py:16-110).

Levels are hashable values: action strings for constructive teachers, `bytes` of the uint8 grid
encoding for editor / reset_random levels (adversarial_runner.py:142-150, 362-376).
"""
import numpy as np


class LevelStore:
    def __init__(self, max_size=None, data_info={}):
        self.max_size = max_size
        self.seed2level = {}
        self.level2seed = {}
        self.seed2parent = {}
        self.next_seed = 1
        self.levels = set()
        self.data_info = data_info

    def __len__(self):
        return len(self.levels)

    def _insert(self, level, parent_seed=None):
        if level is None:
            return None
        if level in self.levels:
            return self.level2seed[level]
        if self.max_size is not None:
            while len(self.levels) >= self.max_size:            # FIFO eviction
                self._remove(next(iter(self.seed2level)))
        seed = self.next_seed
        self.seed2level[seed] = level
        self.seed2parent[seed] = ([] if parent_seed is None
                                  else self.seed2parent[parent_seed] + [self.seed2level[parent_seed]])
        self.level2seed[level] = seed
        self.levels.add(level)
        self.next_seed += 1
        return seed

    def insert(self, level, parent_seeds=None):
        if hasattr(level, "__iter__") and not isinstance(level, (str, bytes)):
            return [self._insert(l, None if parent_seeds is None else parent_seeds[i]) for i, l in enumerate(level)]
        return self._insert(level)

    def _remove(self, level_seed):
        if level_seed is None or level_seed < 0:
            return
        level = self.seed2level.pop(level_seed)
        self.levels.remove(level)
        del self.level2seed[level]
        del self.seed2parent[level_seed]

    def remove(self, level_seed):
        if hasattr(level_seed, "__iter__"):
            for s in level_seed:
                self._remove(s)
        else:
            self._remove(level_seed)

    def reconcile_seeds(self, level_seeds):
        new = set(level_seeds)
        if len(new) == 1 and -1 in new:
            return
        for seed in set(self.seed2level) - new:
            self._remove(seed)

    def get_level(self, level_seed):
        level = self.seed2level[level_seed]
        if self.data_info and self.data_info.get("numpy", False):
            level = np.frombuffer(level, dtype=self.data_info["dtype"]).reshape(*self.data_info["shape"])
        return level
