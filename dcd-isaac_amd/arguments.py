"""Command-line flags with the reference's names and defaults (reference: arguments.py:14-522), so a
reference command line / make_cmd.py grid drives this engine unchanged.  Flags that only concern
CarRacing / BipedalWalker are accepted (defaults kept) but those domains are out of scope."""
import argparse


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


# (name, type, default)
_FLAGS = [
    # PPO / returns
    ("algo", str, "ppo"), ("lr", float, 1e-4), ("eps", float, 1e-5), ("alpha", float, 0.99), ("gamma", float, 0.995),
    ("use_gae", bool, True), ("gae_lambda", float, 0.95), ("entropy_coef", float, 0.0), ("adv_entropy_coef", float, 0.0),
    ("value_loss_coef", float, 0.5), ("max_grad_norm", float, 0.5), ("adv_max_grad_norm", float, 0.5),
    ("normalize_returns", bool, False), ("adv_normalize_returns", bool, False), ("use_popart", bool, False),
    ("adv_use_popart", bool, False), ("seed", int, 1), ("num_processes", int, 32), ("num_steps", int, 256),
    ("ppo_epoch", int, 5), ("adv_ppo_epoch", int, 5), ("num_mini_batch", int, 1), ("adv_num_mini_batch", int, 1),
    ("clip_param", float, 0.2), ("clip_value_loss", bool, True), ("clip_reward", float, None),
    ("adv_clip_reward", float, None), ("num_env_steps", int, 500000),
    # architecture
    ("recurrent_arch", str, "lstm"), ("recurrent_agent", bool, True), ("recurrent_adversary_env", bool, False),
    ("recurrent_hidden_size", int, 256),
    # UED
    ("ued_algo", str, "paired"), ("protagonist_plr", bool, False), ("antagonist_plr", bool, False),
    ("use_reset_random_dr", bool, False),
    # PLR
    ("use_plr", bool, False), ("level_replay_strategy", str, "value_l1"), ("level_replay_eps", float, 0.05),
    ("level_replay_score_transform", str, "rank"), ("level_replay_temperature", float, 0.1),
    ("level_replay_schedule", str, "proportionate"), ("level_replay_rho", float, 1.0),
    ("level_replay_prob", float, 0.0), ("level_replay_alpha", float, 1.0), ("staleness_coef", float, 0.3),
    ("staleness_transform", str, "power"), ("staleness_temperature", float, 1.0),
    ("train_full_distribution", bool, True), ("level_replay_seed_buffer_size", int, 4000),
    ("level_replay_seed_buffer_priority", str, "replay_support"), ("reject_unsolvable_seeds", bool, False),
    ("no_exploratory_grad_updates", bool, False),
    # ACCEL
    ("use_editor", bool, False), ("level_editor_prob", float, 0.0), ("level_editor_method", str, "random"),
    ("base_levels", str, "batch"), ("num_edits", int, 0),
    # logging / checkpoints
    ("xpid_finetune", str, None), ("model_finetune", str, "model"), ("no_cuda", bool, False), ("xpid", str, "latest"),
    ("log_dir", str, "~/logs/dcd/"), ("log_interval", int, 1), ("checkpoint_interval", int, 100),
    ("archive_interval", int, 0), ("checkpoint_basis", str, "num_updates"), ("weight_log_interval", int, 0),
    ("screenshot_interval", int, 5000), ("screenshot_batch_size", int, 1), ("render", bool, False),
    ("checkpoint", bool, False), ("disable_checkpoint", bool, False), ("log_grad_norm", bool, False),
    ("log_action_complexity", bool, False), ("log_replay_complexity", bool, False),
    ("log_plr_buffer_stats", bool, False), ("verbose", bool, False),
    # eval
    ("test_interval", int, 250), ("test_num_episodes", int, 10), ("test_num_processes", int, 2),
    ("test_env_names", str, "MultiGrid-SixteenRooms-v0,MultiGrid-Labyrinth-v0,MultiGrid-Maze-v0"),
    # env
    ("env_name", str, "MultiGrid-GoalLastAdversarial-v0"), ("handle_timelimits", bool, False),
    ("singleton_env", bool, False), ("use_global_critic", bool, False), ("use_global_policy", bool, False),
    # other domains (accepted, unused)
    ("grayscale", bool, False), ("crop_frame", bool, False), ("reward_shaping", bool, False),
    ("num_action_repeat", int, 1), ("frame_stack", int, 1), ("num_control_points", int, 12),
    ("min_rad_ratio", float, 0.333333333), ("max_rad_ratio", float, 1.0), ("use_skip", bool, False),
    ("choose_start_pos", bool, False), ("use_sketch", bool, True), ("use_categorical_adv", bool, False),
    ("sparse_rewards", bool, False), ("num_goal_bins", int, 1),
]

parser = argparse.ArgumentParser(description="DCD rollout-and-regret engine (MI355X)")
for _name, _type, _default in _FLAGS:
    if _type is bool:
        parser.add_argument(f"--{_name}", type=str2bool, nargs="?", const=True, default=_default)
    else:
        parser.add_argument(f"--{_name}", type=_type, default=_default)
# additions of this engine
parser.add_argument("--device", type=str, default=None, help="HIP device (default cuda:LOCAL_RANK)")
parser.add_argument("--wall_limit_s", type=float, default=0.0,
                    help="stop after this many seconds of training: checkpoint and exit (resume with --checkpoint true)")
