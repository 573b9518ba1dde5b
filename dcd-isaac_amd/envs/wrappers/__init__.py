from .hip_vec_env import HipAdversarialVecEnv  # noqa: F401

# The reference builds ParallelAdversarialVecEnv -> VecMonitor -> VecNormalize -> VecPreprocessImageWrapper
# (util/__init__.py:184-220); here that whole stack is the single class above.
ParallelAdversarialVecEnv = HipAdversarialVecEnv
