from . import adversarial  # noqa: F401  (registers the MultiGrid-*Adversarial* ids)
