"""Writes tests/golden/mg_train_configs.json: for every shipped MultiGrid training configuration of the
reference (train_scripts/grid_configs/minigrid/*/*.json — grid-search DATA files, one value per flag), the
flag -> value map that `train_scripts/make_cmd.py` turns into `python -m train --k=v ...`.  Run in the build
container (needs /root/reference); the fixture travels, the reference does not."""
import glob
import json
import os

REF = "/root/reference/train_scripts/grid_configs/minigrid"
out = {}
for f in sorted(glob.glob(os.path.join(REF, "*", "*.json"))):
    grid = json.load(open(f))["grid"]
    assert all(len(v) == 1 for v in grid.values()), f
    out[os.path.relpath(f, REF)] = {k: v[0] for k, v in grid.items()}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mg_train_configs.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print(path, len(out), "configs")
