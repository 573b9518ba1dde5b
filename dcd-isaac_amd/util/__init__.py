"""Factories and helpers with the reference's names (reference: util/__init__.py:59-252)."""
import argparse
import csv
import os
import random
import shutil

import numpy as np
import torch

from envs.registration import make as gym_make
from envs.wrappers import HipAdversarialVecEnv
from .make_agent import make_agent  # noqa: F401


class DotDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__

    def __init__(self, dct=None):
        super().__init__()
        for k, v in (dct or {}).items():
            self[k] = DotDict(v) if hasattr(v, "keys") else v


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def array_to_csv(a):
    return ",".join(str(x) for x in a)


def cprint(condition, *args, **kwargs):
    if condition:
        print(*args, **kwargs)


def seed(seed):
    """train.py:60-61: torch / numpy / python RNGs."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def safe_checkpoint(state_dict, path, index=None, archive_interval=None):
    """Atomic torch.save (reference :59-69): write *_tmp, os.replace, optional archive copy."""
    filename, ext = os.path.splitext(path)
    tmp = f"{filename}_tmp{ext}"
    torch.save(state_dict, tmp)
    os.replace(tmp, path)
    if index is not None and archive_interval is not None and archive_interval > 0 and index % archive_interval == 0:
        shutil.copy(path, f"{filename}_{index}{ext}")


def _checkpoint_allow_list():
    """(module, name) pairs a runner checkpoint may name: torch's own weights-only list (tensor rebuilders, storages, dtypes,
    OrderedDict ...) + the containers, numpy types and the four host-side classes adversarial_runner.py:203-216 pickles."""
    allow = set()
    try:
        from torch._weights_only_unpickler import _get_allowed_globals
        for full in _get_allowed_globals():
            mod, _, name = full.rpartition(".")
            allow.add((mod, name))
    except Exception:                                            # noqa: BLE001  (private helper: fall back to the essentials)
        allow |= {("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_parameter"), ("collections", "OrderedDict"),
                  ("torch", "FloatStorage"), ("torch", "LongStorage"), ("torch", "IntStorage"), ("torch", "ByteStorage"),
                  ("torch", "DoubleStorage"), ("torch", "BoolStorage"), ("torch", "HalfStorage"), ("torch.storage", "UntypedStorage")}
    allow |= {("collections", n) for n in ("deque", "defaultdict", "OrderedDict")}
    for mod in ("builtins", "__builtin__"):                       # protocol <= 2 pickles name the Python-2 module
        allow |= {(mod, n) for n in ("set", "frozenset", "float", "int", "list", "dict", "tuple", "str", "bool", "bytes", "complex", "slice",
                                     "range", "object", "bytearray")}
    allow |= {("copy_reg", "_reconstructor"), ("copyreg", "_reconstructor")}     # object.__new__(cls) of an allow-listed cls
    for mod in ("numpy", "numpy.core.multiarray", "numpy._core.multiarray", "numpy.core.numeric", "numpy._core.numeric"):
        allow |= {(mod, n) for n in ("ndarray", "dtype", "_reconstruct", "scalar", "_frombuffer", "float64", "float32", "int64", "int32",
                                     "int16", "int8", "uint64", "uint32", "uint16", "uint8", "bool_", "bool")}
    allow |= {("numpy.dtypes", n) for n in ("Float64DType", "Float32DType", "Int64DType", "Int32DType", "Int16DType", "Int8DType", "UInt64DType",
                                             "UInt32DType", "UInt16DType", "UInt8DType", "BoolDType", "ObjectDType", "BytesDType", "StrDType")}
    allow |= {("level_replay.level_sampler", "LevelSampler"), ("level_replay.level_store", "LevelStore"),
              ("level_replay.sharded", "ShardedLevelSampler"), ("level_replay.sharded", "ShardedLevelStore"),
              ("envs.spaces", "Box"), ("envs.spaces", "Discrete"), ("envs.spaces", "Dict")}
    return allow


class _AllowListPickle:
    """A `pickle_module` for torch.load whose Unpickler resolves ONLY allow-listed globals (the restricted-unpickler recipe of the
    pickle documentation).  torch.load(weights_only=True) cannot be used: its unpickler refuses to rebuild a non-empty
    collections.deque (`agent_returns`, adversarial_runner.py:206), which every runner checkpoint holds."""
    import pickle as _pickle
    __name__ = "dcd_allow_list_pickle"
    dump, dumps = _pickle.dump, _pickle.dumps
    Pickler, UnpicklingError, PickleError = _pickle.Pickler, _pickle.UnpicklingError, _pickle.PickleError
    HIGHEST_PROTOCOL, DEFAULT_PROTOCOL = _pickle.HIGHEST_PROTOCOL, _pickle.DEFAULT_PROTOCOL
    _allowed = None

    class Unpickler(_pickle.Unpickler):
        def find_class(self, module, name):
            if _AllowListPickle._allowed is None:
                _AllowListPickle._allowed = _checkpoint_allow_list()
            if (module, name) not in _AllowListPickle._allowed:
                import pickle
                raise pickle.UnpicklingError(f"checkpoint names {module}.{name}, which is not on the allow-list of "
                                             f"util.load_checkpoint_safe: refused (nothing from the file was executed)")
            return super().find_class(module, name)

    # torch.load's LEGACY (non-zip) path reads the header pickles (magic number, protocol, sys_info) and the tar-format members
    # through `pickle_module.load(f)` — the module-level functions, not `Unpickler`: they must be restricted as well.
    @staticmethod
    def load(file, **kwargs):
        return _AllowListPickle.Unpickler(file, **kwargs).load()

    @staticmethod
    def loads(data, **kwargs):
        import io
        return _AllowListPickle.Unpickler(io.BytesIO(data), **kwargs).load()


def load_checkpoint_safe(path, map_location=None):
    """torch.load of a `model.tar` that executes nothing the file names beyond an allow-list: tensors, containers, numpy
    arrays / scalars and the host-side classes a runner checkpoint pickles next to the state dicts (adversarial_runner.py:
    203-216: LevelSampler, LevelStore — the module paths are the reference's, so its checkpoints name the same classes — and
    the observation / action spaces they hold).  A file that names anything else raises pickle.UnpicklingError."""
    return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_AllowListPickle)


def save_images(images, path=None, normalize=False, channels_first=False, nrow=8, padding=2):
    """Tile `images` (list of HWC uint8 arrays) into one grid image and write it as PNG (reference: util/__init__.py
    save_images -> torchvision make_grid/save_image, 8 per row, 2-pixel black padding; torchvision is not a
    dependency here, the grid is assembled in numpy and written with PIL)."""
    import numpy as np
    if path is None or len(images) == 0:
        return
    arr = [np.asarray(im) for im in images]
    if channels_first:
        arr = [a.transpose(1, 2, 0) for a in arr]
    if normalize:
        arr = [(255.0 * (a - a.min()) / max(float(a.max() - a.min()), 1e-5)) for a in arr]
    arr = [np.clip(a, 0, 255).astype(np.uint8) for a in arr]
    h, w = arr[0].shape[:2]
    n = len(arr)
    cols = min(nrow, n)
    rows = (n + cols - 1) // cols
    canvas = np.zeros((rows * (h + padding) + padding, cols * (w + padding) + padding, 3), dtype=np.uint8)
    for i, a in enumerate(arr):
        r, c = divmod(i, cols)
        y0, x0 = padding + r * (h + padding), padding + c * (w + padding)
        canvas[y0:y0 + h, x0:x0 + w] = a
    from PIL import Image
    Image.fromarray(canvas).save(path)


def get_obs_at_index(obs, i):
    if isinstance(obs, dict):
        return {k: obs[k][i] for k in obs.keys()}
    return obs[i]


def set_obs_at_index(obs, obs_, i):
    if isinstance(obs, dict):
        for k in obs.keys():
            obs[k][i] = obs_[k].squeeze(0)
    else:
        obs[i] = obs_[0].squeeze(0)


def is_discrete_actions(env, adversary=False):
    space = env.adversary_action_space if adversary else env.action_space
    return space.__class__.__name__ == "Discrete"


def is_dense_reward_env(env_name):
    return env_name.startswith("CarRacing")


def _make_env(args):
    if not args.env_name.startswith("MultiGrid"):
        raise ValueError(f"{args.env_name}: only the MultiGrid path is built (SURVEY §8)")
    kwargs = {"seed": args.seed}
    if args.singleton_env:
        kwargs["fixed_environment"] = True
    if vars(args).get("use_global_critic") or vars(args).get("use_global_policy"):
        raise NotImplementedError("MultiGridFullyObsWrapper (global critic) is outside the hot path")
    return gym_make(args.env_name, **kwargs)


def create_parallel_env(args, adversary=True, device=None, env_index_base=0):
    """(venv, ued_venv) as util/__init__.py:184-220; the whole wrapper stack is one HipAdversarialVecEnv."""
    if device is None:
        device = "cuda:0"
    make_fn = lambda: _make_env(args)
    venv = HipAdversarialVecEnv([make_fn] * args.num_processes, adversary=adversary, device=device,
                                env_index_base=env_index_base, engine_seed=args.seed,
                                normalize_returns=bool(vars(args).get("normalize_returns", False)))
    if args.singleton_env:
        seeds = [args.seed] * args.num_processes
    else:
        seeds = [env_index_base + i for i in range(args.num_processes)]
    venv.set_seed(seeds)
    return venv, venv


def make_plr_args(args, obs_space, action_space):
    return dict(seeds=[], obs_space=obs_space, action_space=action_space, num_actors=args.num_processes,
                strategy=args.level_replay_strategy, replay_schedule=args.level_replay_schedule,
                score_transform=args.level_replay_score_transform, temperature=args.level_replay_temperature,
                eps=args.level_replay_eps, rho=args.level_replay_rho, replay_prob=args.level_replay_prob,
                alpha=args.level_replay_alpha, staleness_coef=args.staleness_coef,
                staleness_transform=args.staleness_transform, staleness_temperature=args.staleness_temperature,
                sample_full_distribution=args.train_full_distribution,
                seed_buffer_size=args.level_replay_seed_buffer_size,
                seed_buffer_priority=args.level_replay_seed_buffer_priority,
                use_dense_rewards=is_dense_reward_env(args.env_name), gamma=args.gamma)
