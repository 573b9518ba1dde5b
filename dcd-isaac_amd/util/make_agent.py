"""Agent factory with the reference's signature (reference: util/make_agent.py:15-58,160-244)."""
from algos import ACAgent, PPO, RolloutStorage
from models import MultigridNetwork


def model_for_multigrid_agent(env, agent_type="agent", recurrent_arch=None, recurrent_hidden_size=256,
                              use_global_critic=False, use_global_policy=False):
    if use_global_critic or use_global_policy:
        raise NotImplementedError("global critic/policy nets are off in every shipped MultiGrid config (SURVEY §2 #8)")
    if agent_type == "adversary_env":
        obs_space, act_space = env.adversary_observation_space, env.adversary_action_space
        return MultigridNetwork(observation_space=obs_space, action_space=act_space, conv_filters=128, scalar_fc=10,
                                scalar_dim=int(obs_space["time_step"].high[0]) + 1,
                                random_z_dim=obs_space["random_z"].shape[0], recurrent_arch=recurrent_arch,
                                recurrent_hidden_size=recurrent_hidden_size)
    obs_space = env.observation_space
    return MultigridNetwork(observation_space=obs_space, action_space=env.action_space, scalar_fc=5,
                            scalar_dim=int(obs_space["direction"].high[0]) + 1, recurrent_arch=recurrent_arch,
                            recurrent_hidden_size=recurrent_hidden_size)


def model_for_env_agent(env_name, env, agent_type="agent", recurrent_arch=None, recurrent_hidden_size=256, **kw):
    assert agent_type in ["agent", "adversary_agent", "adversary_env"]
    if not env_name.startswith("MultiGrid"):
        raise ValueError(f"Unsupported environment {env_name}: only the MultiGrid path is built (SURVEY §8).")
    return model_for_multigrid_agent(env, agent_type, recurrent_arch, recurrent_hidden_size,
                                     kw.get("use_global_critic", False), kw.get("use_global_policy", False))


def make_agent(name, env, args, device="cpu", process_group=None):
    is_adversary_env = "env" in name
    v = vars(args)
    if is_adversary_env:
        obs_space, act_space = env.adversary_observation_space, env.adversary_action_space
        num_steps = int(obs_space["time_step"].high[0])
        recurrent_arch = args.recurrent_adversary_env and args.recurrent_arch
        entropy_coef, ppo_epoch = args.adv_entropy_coef, args.adv_ppo_epoch
        num_mini_batch, max_grad_norm = args.adv_num_mini_batch, args.adv_max_grad_norm
        use_popart = v.get("adv_use_popart", False)
    else:
        obs_space, act_space = env.observation_space, env.action_space
        num_steps = args.num_steps
        recurrent_arch = args.recurrent_agent and args.recurrent_arch
        entropy_coef, ppo_epoch = args.entropy_coef, args.ppo_epoch
        num_mini_batch, max_grad_norm = args.num_mini_batch, args.max_grad_norm
        use_popart = v.get("use_popart", False)
    actor_critic = model_for_env_agent(args.env_name, env, name, recurrent_arch=recurrent_arch,
                                       recurrent_hidden_size=args.recurrent_hidden_size,
                                       use_global_critic=v.get("use_global_critic", False),
                                       use_global_policy=v.get("use_global_policy", False))
    use_proper_time_limits = (hasattr(env, "get_max_episode_steps") and env.get_max_episode_steps() is not None
                              and v.get("handle_timelimits", False))
    if args.algo != "ppo":
        raise ValueError(f"Unsupported RL algorithm {args.algo}.")
    actor_critic.to(device)
    algo = PPO(actor_critic=actor_critic, clip_param=args.clip_param, ppo_epoch=ppo_epoch, num_mini_batch=num_mini_batch,
               value_loss_coef=args.value_loss_coef, entropy_coef=entropy_coef, lr=args.lr, eps=args.eps,
               max_grad_norm=max_grad_norm, clip_value_loss=args.clip_value_loss, log_grad_norm=args.log_grad_norm,
               process_group=process_group)
    storage = RolloutStorage(model=actor_critic, num_steps=num_steps, num_processes=args.num_processes,
                             observation_space=obs_space, action_space=act_space,
                             recurrent_hidden_state_size=args.recurrent_hidden_size, recurrent_arch=args.recurrent_arch,
                             use_proper_time_limits=use_proper_time_limits, use_popart=use_popart, device=device)
    return ACAgent(algo=algo, storage=storage)
