from .adversarial_runner import AdversarialRunner  # noqa: F401
