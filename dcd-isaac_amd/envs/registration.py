"""Environment registry with the reference's surface: register / make / spec
(reference: envs/registration.py:36-173, itself a copy of gym's registry).

`make(id, **kwargs)` instantiates the entry-point class with the registered kwargs and, when the
spec carries `max_episode_steps`, records the TimeLimit on the returned object
(`env._max_episode_steps`, reference :122-124).  For the MultiGrid ids the returned object is a
*level-family description* (envs.multigrid.adversarial.AdversarialEnv): the transition itself runs
batched on the GPU inside envs.wrappers.HipAdversarialVecEnv, which reads that description.
"""
import importlib
import re

env_id_re = re.compile(r"^(?:[\w:-]+\/)?([\w:.-]+)-v(\d+)$")


class Error(Exception):
    pass


class UnregisteredEnv(Error):
    pass


class DeprecatedEnv(Error):
    pass


def load(name):
    mod_name, attr_name = name.split(":")
    return getattr(importlib.import_module(mod_name), attr_name)


class EnvSpec:
    def __init__(self, id, entry_point=None, reward_threshold=None, kwargs=None, nondeterministic=False,
                 tags=None, max_episode_steps=None):
        match = env_id_re.search(id)
        if not match:
            raise Error(f"Attempted to register malformed environment ID: {id}. "
                        f"(Currently all IDs must be of the form {env_id_re.pattern}.)")
        self.id = id
        self.entry_point = entry_point
        self.reward_threshold = reward_threshold
        self.nondeterministic = nondeterministic
        self.tags = {} if tags is None else tags
        self.tags["wrapper_config.TimeLimit.max_episode_steps"] = max_episode_steps
        self.max_episode_steps = max_episode_steps
        self._env_name = match.group(1)
        self._kwargs = {} if kwargs is None else kwargs

    def make(self, **kwargs):
        if self.entry_point is None:
            raise Error(f"Attempting to make deprecated env {self.id}. "
                        "(HINT: is there a newer registered version of this env?)")
        merged = dict(self._kwargs)
        merged.update(kwargs)
        cls = self.entry_point if callable(self.entry_point) else load(self.entry_point)
        env = cls(**merged)
        env.spec = self
        return env

    def __repr__(self):
        return f"EnvSpec({self.id})"


class EnvRegistry:
    def __init__(self):
        self.env_specs = {}

    def make(self, path, **kwargs):
        spec = self.spec(path)
        env = spec.make(**kwargs)
        if spec.max_episode_steps is not None and not spec.tags.get("vnc"):
            env._max_episode_steps = spec.max_episode_steps          # the TimeLimit wrapper of the reference
        return env

    def all(self):
        return self.env_specs.values()

    def spec(self, path):
        if ":" in path:
            mod_name, _, id = path.partition(":")
            try:
                importlib.import_module(mod_name)
            except ImportError:
                raise Error(f"A module ({mod_name}) was specified for the environment but was not found")
        else:
            id = path
        match = env_id_re.search(id)
        if not match:
            raise Error(f"Attempted to look up malformed environment ID: {id.encode('utf-8')}. "
                        f"(Currently all IDs must be of the form {env_id_re.pattern}.)")
        try:
            return self.env_specs[id]
        except KeyError:
            env_name = match.group(1)
            matching = [k for k, v in self.env_specs.items() if env_name == v._env_name]
            if matching:
                raise DeprecatedEnv(f"Env {id} not found (valid versions include {matching})")
            raise UnregisteredEnv(f"No registered env with id: {id}")

    def register(self, id, **kwargs):
        if id in self.env_specs:
            raise Error(f"Cannot re-register id: {id}")
        self.env_specs[id] = EnvSpec(id, **kwargs)


registry = EnvRegistry()


def register(id, **kwargs):
    return registry.register(id, **kwargs)


def make(id, **kwargs):
    return registry.make(id, **kwargs)


def spec(id):
    return registry.spec(id)
