"""PPO-clip with clipped value loss (reference: algos/ppo.py:19-146), plus the data-parallel hooks
the reference does not have: when a `torch.distributed` process group is passed, advantage
normalisation uses global moments, losses/entropies are averaged over the global minibatch and
gradients are all-reduced (RCCL over xGMI on MI355X) before clipping — so N ranks × B envs equal
one process with N·B envs.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim


class PPO:
    def __init__(self, actor_critic, clip_param, ppo_epoch, num_mini_batch, value_loss_coef, entropy_coef,
                 lr=None, eps=None, max_grad_norm=None, clip_value_loss=True, log_grad_norm=False,
                 process_group=None):
        self.actor_critic = actor_critic
        self.clip_param = clip_param
        self.ppo_epoch = ppo_epoch
        self.num_mini_batch = num_mini_batch
        self.clip_value_loss = clip_value_loss
        self.value_loss_coef = value_loss_coef
        self.entropy_coef = entropy_coef
        self.max_grad_norm = max_grad_norm
        self.optimizer = optim.Adam(actor_critic.parameters(), lr=lr, eps=eps)
        self.log_grad_norm = log_grad_norm
        self.process_group = process_group
        # a group of ONE rank runs every exchange step too (each is then the identity, bit for bit): the RCCL path on a one-GPU box
        self.distributed = process_group is not None
        self._flat_grad = None
        if self.distributed:                                         # ranks seed differently (env streams): start from rank 0's weights
            import torch.distributed as dist
            src = dist.get_global_rank(process_group, 0)
            for t in list(actor_critic.parameters()) + list(actor_critic.buffers()):
                dist.broadcast(t.data, src=src, group=process_group)

    # ---- distributed helpers ----------------------------------------------------------------
    @property
    def world_size(self):
        if self.process_group is None:
            return 1
        import torch.distributed as dist
        return dist.get_world_size(self.process_group)

    def _all_reduce(self, t):
        if self.process_group is not None:
            import torch.distributed as dist
            dist.all_reduce(t, group=self.process_group)
        return t

    def _normalize_advantages(self, adv):
        """(adv - mean) / (std + 1e-5) with unbiased std (ppo.py:67-69), moments over ALL ranks: every rank contributes
        (n, mean, std) of its own slice — torch's own fp32 reductions, the single-process arithmetic — as one row of a
        [world, 3] fp64 tensor (ONE all-reduce = an all-gather), and the rows are merged exactly as parallel-variance
        merging prescribes: mean = sum n_i mean_i / N, M2 = sum (n_i - 1) var_i + sum n_i (mean_i - mean)^2.  With one rank
        the merge returns that rank's mean and std bit for bit, so a 1-rank group equals the run without a group."""
        if not self.distributed:
            return (adv - adv.mean()) / (adv.std() + 1e-5)
        import torch.distributed as dist
        rows = torch.zeros((self.world_size, 3), dtype=torch.float64, device=adv.device)
        rows[dist.get_rank(self.process_group)] = torch.stack([torch.tensor(float(adv.numel()), dtype=torch.float64, device=adv.device),
                                                               adv.mean().double(), adv.std().double()])
        self._all_reduce(rows)
        n, m, sd = rows[:, 0], rows[:, 1], rows[:, 2]
        total = n.sum()
        mean = (n * m).sum() / total
        m2 = ((n - 1) * sd * sd).sum() + (n * (m - mean) ** 2).sum()
        std = (m2 / (total - 1)).clamp_min(0).sqrt()
        return (adv - mean.float()) / (std.float() + 1e-5)

    # ---- gradient exchange (no counterpart in the reference: one process there) ------------------------------------
    # Every parameter's .grad is a VIEW into one persistent flat buffer (no cat / copy-back per minibatch: two passes over
    # the 90 MB teacher gradient saved).  The buffer is cut into buckets of <= BUCKET_BYTES; a bucket's all-reduce is
    # launched ASYNCHRONOUSLY from a post-accumulate-grad hook as soon as the backward pass has produced every gradient
    # that lies in it (RCCL runs it on its own stream, overlapping the rest of the backward), and the three loss
    # statistics of the minibatch ride in the tail of the buffer — one exchange per minibatch, nothing else.
    BUCKET_BYTES = 24 << 20                       # teacher: 90 MB -> 4 buckets (its LSTM input weight alone is 89 MB)

    def _setup_flat_grads(self):
        params = [p for p in self.actor_critic.parameters() if p.requires_grad]
        # reverse order: the backward pass reaches the heads first, the conv last -> buckets complete front to back
        order = list(reversed(params))
        total = sum(p.numel() for p in order)
        flat = torch.zeros(total + 3, dtype=order[0].dtype, device=order[0].device)      # + value / action loss, entropy
        off, spans = 0, []
        for p in order:
            n = p.numel()
            p.grad = flat[off:off + n].view_as(p)
            spans.append((off, off + n))
            off += n
        per = max(1, self.BUCKET_BYTES // flat.element_size())
        nb = max(1, (total + per - 1) // per)
        bounds = [(b * per, min((b + 1) * per, total)) for b in range(nb)]
        bounds[-1] = (bounds[-1][0], total + 3)                                          # the statistics travel with the last bucket
        need = [sum(1 for (a, e) in spans if a < hi and e > lo) for (lo, hi) in bounds]
        touching = [[b for b, (lo, hi) in enumerate(bounds) if a < hi and e > lo] for (a, e) in spans]
        self._flat_grad, self._flat_params, self._bucket_bounds = flat, order, bounds
        self._bucket_need, self._bucket_of = need, touching
        self._bucket_left, self._handles, self._launched = list(need), [], [False] * nb
        if self.distributed:
            for i, p in enumerate(order):
                p.register_post_accumulate_grad_hook(lambda _p, i=i: self._grad_ready(i))

    def _launch_bucket(self, b):
        import torch.distributed as dist
        lo, hi = self._bucket_bounds[b]
        self._launched[b] = True
        self._handles.append(dist.all_reduce(self._flat_grad[lo:hi], group=self.process_group, async_op=True))

    def _grad_ready(self, i):
        """Post-accumulate hook of parameter i (same order on every rank: same graph): a bucket goes out once its last
        gradient is in.  The final bucket also carries the loss statistics and is launched by `_sync_gradients`."""
        last = len(self._bucket_bounds) - 1
        for b in self._bucket_of[i]:
            self._bucket_left[b] -= 1
            if self._bucket_left[b] == 0 and b != last:
                self._launch_bucket(b)

    def _zero_grad(self):
        if self._flat_grad is None:
            self._setup_flat_grads()
        off = 0
        for p in self._flat_params:                                   # re-attach a .grad someone set to None / replaced
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self._flat_grad[off:off + n].data_ptr():
                p.grad = self._flat_grad[off:off + n].view_as(p)
            off += n
        self._flat_grad.zero_()
        self._bucket_left, self._handles = list(self._bucket_need), []
        self._launched = [False] * len(self._bucket_bounds)

    def _sync_gradients(self, stats):
        """Launches what the hooks did not (the final bucket with the statistics in its tail; buckets whose parameters
        received no gradient), waits, averages.  Returns the job-wide means of `stats` (value loss, action loss,
        entropy) — they cost no collective of their own."""
        if not self.distributed:
            return stats
        self._flat_grad[-3:].copy_(stats)
        for b in range(len(self._bucket_bounds)):
            if not self._launched[b]:
                self._launch_bucket(b)
        for h in self._handles:
            h.wait()
        self._flat_grad /= self.world_size
        return self._flat_grad[-3:].clone()

    def _grad_norm(self):
        total = 0.0
        for p in self.actor_critic.parameters():
            if p.grad is not None:
                total += p.grad.data.norm(2).item() ** 2
        return total ** 0.5

    def _losses(self, rollouts, obs_b, hxs_b, actions_b, value_preds_b, return_b, masks_b, old_logp_b, adv_b):
        """[value loss, action loss, entropy] of one minibatch — the expressions of `update`, for its forward-only form."""
        values, logp, dist_entropy, _ = self.actor_critic.evaluate_actions(obs_b, hxs_b, masks_b, actions_b)
        ratio = torch.exp(logp - old_logp_b)
        surr1 = ratio * adv_b
        surr2 = torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param) * adv_b
        action_loss = -torch.min(surr1, surr2).mean()
        if rollouts.use_popart:
            self.actor_critic.popart.update(return_b, group=self.process_group)
            return_b = self.actor_critic.popart.normalize(return_b)
        if self.clip_value_loss:
            clipped = value_preds_b + (values - value_preds_b).clamp(-self.clip_param, self.clip_param)
            value_loss = 0.5 * torch.max((values - return_b).pow(2), (clipped - return_b).pow(2)).mean()
        else:
            value_loss = F.smooth_l1_loss(values, return_b)
        return torch.stack([value_loss, action_loss, dist_entropy])

    def _sync_stats(self, stats):
        """Job-wide means of the three loss statistics when no gradient exchange carries them."""
        if not self.distributed:
            return stats
        self._all_reduce(stats)
        return stats / self.world_size

    # ---- update -------------------------------------------------------------------------------
    def update(self, rollouts, discard_grad=False):
        value_preds = rollouts.denorm_value_preds if rollouts.use_popart else rollouts.value_preds      # ppo.py:62-65
        advantages = rollouts.returns[:-1] - value_preds[:-1]
        advantages = self._normalize_advantages(advantages)
        value_loss_epoch = action_loss_epoch = dist_entropy_epoch = 0.0
        grad_norms = []
        # discard_grad (robust PLR's exploratory updates, ppo.py:129-130): the reference runs the backward pass and throws the
        # gradients away.  Nothing observable depends on them unless their norm is logged — the parameters stay, the losses come
        # from the forward pass — so the backward pass, the gradient exchange and the clipping are skipped and the forward pass
        # runs without an autograd graph.  Every epoch's forward pass still runs (its minibatch permutation sums the same terms
        # in another order) and the generator still draws its permutations: losses, PopArt state and the position of torch's
        # generator are what they would have been (tests/test_models_ppo_vs_reference.py, test_gpu_runner.py).
        forward_only = bool(discard_grad and not self.log_grad_norm)
        for _ in range(self.ppo_epoch):
            gen = (rollouts.recurrent_generator(advantages, self.num_mini_batch) if self.actor_critic.is_recurrent
                   else rollouts.feed_forward_generator(advantages, self.num_mini_batch))
            for (obs_b, hxs_b, actions_b, value_preds_b, return_b, masks_b, old_logp_b, adv_b) in gen:
                if forward_only:
                    with torch.no_grad():
                        v, a, e = self._sync_stats(self._losses(rollouts, obs_b, hxs_b, actions_b, value_preds_b, return_b, masks_b,
                                                                old_logp_b, adv_b)).tolist()
                    value_loss_epoch += v
                    action_loss_epoch += a
                    dist_entropy_epoch += e
                    continue
                values, logp, dist_entropy, _ = self.actor_critic.evaluate_actions(obs_b, hxs_b, masks_b, actions_b)
                ratio = torch.exp(logp - old_logp_b)
                surr1 = ratio * adv_b
                surr2 = torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param) * adv_b
                action_loss = -torch.min(surr1, surr2).mean()
                if rollouts.use_popart:                              # ppo.py:102-104: targets in PopArt's normalised space
                    self.actor_critic.popart.update(return_b, group=self.process_group)   # job-wide batch moments
                    return_b = self.actor_critic.popart.normalize(return_b)
                if self.clip_value_loss:
                    clipped = value_preds_b + (values - value_preds_b).clamp(-self.clip_param, self.clip_param)
                    value_loss = 0.5 * torch.max((values - return_b).pow(2), (clipped - return_b).pow(2)).mean()
                else:
                    value_loss = F.smooth_l1_loss(values, return_b)
                self._zero_grad()
                loss = value_loss * self.value_loss_coef + action_loss - dist_entropy * self.entropy_coef
                loss.backward()
                stats = self._sync_gradients(torch.stack([value_loss.detach(), action_loss.detach(), dist_entropy.detach()]))
                if self.log_grad_norm:
                    grad_norms.append(self._grad_norm())
                if self.max_grad_norm is not None and self.max_grad_norm > 0:
                    nn.utils.clip_grad_norm_(self.actor_critic.parameters(), self.max_grad_norm)
                if not discard_grad:
                    self.optimizer.step()
                v, a, e = stats.tolist()
                value_loss_epoch += v
                action_loss_epoch += a
                dist_entropy_epoch += e
        n = self.ppo_epoch * self.num_mini_batch
        info = {"grad_norms": grad_norms} if self.log_grad_norm else {}
        return value_loss_epoch / n, action_loss_epoch / n, dist_entropy_epoch / n, info
