"""-m gpu: VecNormalize(ret=True) fused on the device (dcd_return_normalize; envs/wrappers/vec_normalize.py:37-45) against
the numpy restatement over real MultiGrid steps: normalised rewards, running returns and running moments."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B", [64, 1500])
def test_normalised_rewards_follow_the_wrapper(B):
    from oracle import orc
    from arguments import parser
    from util import create_parallel_env
    args = parser.parse_args([])
    args.env_name, args.num_processes, args.normalize_returns = "MultiGrid-MiniGoalLastAdversarial-v0", B, True
    venv, _ = create_parallel_env(args, device="cuda:0")
    assert venv.normalize_returns
    raw_env, _ = create_parallel_env(parser.parse_args(["--env_name", args.env_name, "--num_processes", str(B)]), device="cuda:0")
    ref = orc.VecNormalizeRet(B)
    rng = np.random.default_rng(0)
    for v in (venv, raw_env):
        v.reset_random()
    saw_reward = False
    for t in range(120):
        a = torch.from_numpy(rng.choice(7, size=B, p=[0.15, 0.15, 0.5, 0.05, 0.05, 0.05, 0.05]))
        _, rew, done, _ = venv.step_env(a)
        _, raw, done2, _ = raw_env.step_env(a)
        assert np.array_equal(done, done2)
        want = ref.step(raw.cpu().numpy()[:, 0], done)
        saw_reward |= bool((raw > 0).any())
        np.testing.assert_allclose(rew.cpu().numpy()[:, 0], want.astype(np.float32), rtol=1e-6, atol=0)
        if t == 60:                                              # reset_agent zeroes the running returns (vec_normalize.py:61-64)
            venv.reset_agent(); raw_env.reset_agent(); ref.reset()
    assert saw_reward
    np.testing.assert_allclose(venv._ret.cpu().numpy(), ref.ret, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(venv.ret_rms.cpu().numpy(), [ref.mean, ref.var, ref.count], rtol=1e-11)
    venv.close(); raw_env.close()
