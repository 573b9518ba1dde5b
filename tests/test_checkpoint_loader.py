"""CPU: util.load_checkpoint_safe — train.py's loader for `model.tar` files (its own on --checkpoint, a FOREIGN one on
--xpid_finetune, reference train.py:129-141).  It must rebuild everything a runner checkpoint holds
(adversarial_runner.py:203-216: state dicts, deques of returns, the pickled LevelSampler / LevelStore with their numpy arrays
and spaces) and must refuse — without executing — a file that names anything else."""
import collections
import os
import pickle

import numpy as np
import pytest
import torch


def _state():
    from envs import spaces
    from level_replay import LevelSampler, LevelStore
    obs = spaces.Dict({"image": spaces.Box(0, 255, (3, 5, 5))})
    ls = LevelSampler([], obs, spaces.Discrete(7), num_actors=4, strategy="positive_value_loss", sample_full_distribution=True,
                      seed_buffer_size=6, score_transform="rank", staleness_coef=0.3, rho=0.5, replay_prob=0.5)
    ls.observe_external_unseen_sample([1, 2, 3], [True, False, True])
    st = LevelStore(data_info={"numpy": True, "dtype": np.uint8, "shape": (3, 3, 3)})
    st.insert([np.arange(27, dtype=np.uint8).tobytes()])
    returns = collections.deque(maxlen=10)
    returns.extend([1.0, 2.5, np.float64(3.0)])
    net = torch.nn.Linear(3, 2)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    net(torch.ones(1, 3)).sum().backward()
    opt.step()
    return {"agent_state_dict": {"agent": net.state_dict()}, "optimizer_state_dict": {"agent": opt.state_dict()}, "agent_returns": returns,
            "num_updates": 3, "latest_env_stats": collections.defaultdict(float, {"a": 1.5}), "level_store": st, "level_samplers": {"agent": ls}}


def test_runner_checkpoint_round_trips(tmp_path):
    from util import load_checkpoint_safe, safe_checkpoint
    sd = _state()
    path = os.path.join(tmp_path, "model.tar")
    safe_checkpoint({"runner_state_dict": sd}, path)
    out = load_checkpoint_safe(path, map_location="cpu")["runner_state_dict"]
    assert list(out["agent_returns"]) == [1.0, 2.5, 3.0] and out["agent_returns"].maxlen == 10
    assert out["num_updates"] == 3 and out["latest_env_stats"]["a"] == 1.5 and out["latest_env_stats"]["missing"] == 0.0
    for k, v in sd["agent_state_dict"]["agent"].items():
        assert torch.equal(out["agent_state_dict"]["agent"][k], v)
    assert out["optimizer_state_dict"]["agent"]["param_groups"][0]["lr"] == 1e-4
    ls, ref = out["level_samplers"]["agent"], sd["level_samplers"]["agent"]
    assert type(ls).__name__ == "LevelSampler" and np.array_equal(ls.seed_scores, ref.seed_scores) and ls.staging_seed_set == ref.staging_seed_set
    assert out["level_store"].seed2level == sd["level_store"].seed2level


class _Hostile:
    def __reduce__(self):
        return (os.system, ("echo hostile-checkpoint-ran > /dev/null",))


@pytest.mark.parametrize("payload", [_Hostile(), pickle.Unpickler, print])
def test_foreign_globals_are_refused_not_executed(tmp_path, payload):
    from util import load_checkpoint_safe
    path = os.path.join(tmp_path, "evil.tar")
    torch.save({"runner_state_dict": {"agent_state_dict": {"agent": {}}, "x": payload}}, path)
    with pytest.raises(pickle.UnpicklingError, match="allow-list"):
        load_checkpoint_safe(path, map_location="cpu")


class _Marker:
    """reduce -> open(path, 'w'): the side effect a hostile stream would have if any of its pickles were executed."""
    def __init__(self, path):
        self.path = path

    def __reduce__(self):
        return (open, (self.path, "w"))


@pytest.mark.parametrize("form", ["raw_header", "legacy_save"])
def test_hostile_legacy_format_streams_are_refused_not_executed(tmp_path, form):
    """torch.load's LEGACY (non-zip) path reads its header pickles and members through `pickle_module.load(f)`, not through
    `pickle_module.Unpickler`: both must go through the allow-list.  `raw_header`: a file whose FIRST pickle is
    the payload (what torch reads as the magic number); `legacy_save`: a well-formed `_use_new_zipfile_serialization=False`
    file with the payload inside the object pickle."""
    from util import load_checkpoint_safe
    marker = os.path.join(tmp_path, "ran")
    path = os.path.join(tmp_path, "evil_legacy.tar")
    if form == "raw_header":
        with open(path, "wb") as f:
            pickle.dump(_Marker(marker), f, protocol=2)
            pickle.dump(1001, f, protocol=2)
    else:
        torch.save({"runner_state_dict": {"x": _Marker(marker)}}, path, _use_new_zipfile_serialization=False)
    with pytest.raises(pickle.UnpicklingError, match="allow-list"):
        load_checkpoint_safe(path, map_location="cpu")
    assert not os.path.exists(marker)


def test_benign_legacy_format_checkpoint_still_loads(tmp_path):
    from util import load_checkpoint_safe
    path = os.path.join(tmp_path, "legacy.tar")
    net = torch.nn.Linear(3, 2)
    torch.save({"runner_state_dict": {"agent_state_dict": {"agent": net.state_dict()}, "num_updates": 7}}, path,
               _use_new_zipfile_serialization=False)
    out = load_checkpoint_safe(path, map_location="cpu")["runner_state_dict"]
    assert out["num_updates"] == 7 and torch.equal(out["agent_state_dict"]["agent"]["weight"], net.state_dict()["weight"])
