"""ACAgent — (algo, storage) pair behind the reference's facade (reference: algos/agent.py:7-50)."""

_MODEL_CALLS = ("act", "get_value", "evaluate_actions")


class ACAgent:
    def __init__(self, algo, storage):
        self.algo = algo
        self.storage = storage

    @property
    def actor_critic(self):
        return self.algo.actor_critic

    def __getattr__(self, name):                       # act / get_value forwarded to the network
        if name in _MODEL_CALLS:
            return getattr(self.algo.actor_critic, name)
        raise AttributeError(name)

    def update(self, discard_grad=False):
        info = self.algo.update(self.storage, discard_grad=discard_grad)
        self.storage.after_update()
        return info

    def insert(self, *args, **kwargs):
        return self.storage.insert(*args, **kwargs)

    def to(self, device):
        self.algo.actor_critic.to(device)
        self.storage.to(device)
        return self

    def train(self):
        self.algo.actor_critic.train()

    def eval(self):
        self.algo.actor_critic.eval()

    def random(self):
        self.algo.actor_critic.random = True

    def process_action(self, action):
        fn = getattr(self.algo.actor_critic, "process_action", None)
        return fn(action) if fn is not None else action

    @property
    def is_recurrent(self):
        return self.algo.actor_critic.is_recurrent
