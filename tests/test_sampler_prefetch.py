"""CPU: LevelSampler.prefetch_next_update — the next update's replay decision and first draws computed speculatively on a host
thread (under the previous update's PPO kernels) must leave EXACTLY what the sequential calls leave: same decisions, same seeds,
same staleness, same position of the global np.random stream (reference order: adversarial_runner.py:616-622 -> level_sampler.py:
606-680); and a speculation that anything overtook (np.random consumed, scores / staleness / buffer changed) must be dropped."""
import numpy as np
import pytest

from oracle import orc


class _Space:
    n = 7


def _sampler(B, N=60, **kw):
    from level_replay import LevelSampler
    base = dict(num_actors=B, strategy="positive_value_loss", score_transform="rank", temperature=0.3, staleness_coef=0.3, rho=0.5,
                replay_prob=0.8, alpha=1.0, gamma=0.995)
    base.update(kw)
    s = LevelSampler(list(range(1, N + 1)), None, _Space(), **base)
    rng = np.random.default_rng(1)
    s.seed_scores = rng.random(N)
    s.seed_staleness = rng.integers(0, 40, N).astype(np.float64)
    s.unseen_seed_weights = (rng.random(N) < 0.2).astype(np.float64)
    return s


def _state(s):
    return (s.seed_staleness.copy(), s.seed_scores.copy(), s.unseen_seed_weights.copy(), np.random.get_state()[1].copy(),
            np.random.get_state()[2])


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("kw", [{}, dict(score_transform="power", staleness_transform="power"), dict(staleness_coef=0.0),
                                dict(sample_full_distribution=True, seed_buffer_size=60, replay_prob=0.5)])
def test_prefetched_updates_equal_the_sequential_order(kw):
    B = 48
    runs = []
    for prefetch in (False, True):
        np.random.seed(11)
        s = _sampler(B, **kw)
        if kw.get("sample_full_distribution"):                       # a full working buffer
            s.seeds = np.arange(1, 61)
            s.seed2index = {int(x): i for i, x in enumerate(s.seeds)}
            s.working_seed_set = set(int(x) for x in s.seeds)
            s.working_seed_buffer_size = 60
            s.unseen_seed_weights[:] = 0
        log, hits = [], 0
        for it in range(12):
            if prefetch:
                assert s.prefetch_next_update(B) or kw.get("never")
            if it % 4 == 3:                                          # something overtakes the speculation ...
                if it % 8 == 3:
                    np.random.rand()                                 # ... the global stream moved
                else:
                    s.seed_scores[it % 7] += 0.25                    # ... the scores moved
            d = s.sample_replay_decision()
            seeds = s.sample_replay_levels(B) if d else None
            extra = s.sample_replay_levels(3)                        # per-step draws of the rollout: never prefetched
            log.append((d, seeds, extra, _state(s)))
            s.seed_scores[(3 * it) % 60] = np.random.rand()          # the update's effect on the sampler
        runs.append((log, getattr(s, "prefetch_hits", 0)))
    (a, _), (b, hits) = runs
    assert hits >= 8                                                  # the speculation was used (decision + draws of most updates)
    for x, y in zip(a, b):
        assert x[0] == y[0] and x[1] == y[1] and x[2] == y[2] and _same(x[3], y[3])
    assert any(x[0] for x in a)
    if kw.get("sample_full_distribution"):
        assert not all(x[0] for x in a)                             # and some updates were not replay updates


def test_a_mismatching_request_drops_the_speculation():
    np.random.seed(5)
    s = _sampler(16)
    ref = _sampler(16)
    np.random.seed(5)
    s.prefetch_next_update(16)
    # the runner asks for something else than was speculated (another count, no staleness update): the result must be the plain one
    d0 = s.sample_replay_decision()
    st = np.random.get_state()
    got = s.sample_replay_levels(5)
    np.random.seed(5)
    assert ref.sample_replay_decision() == d0
    assert np.array_equal(np.random.get_state()[1], st[1]) and np.random.get_state()[2] == st[2]
    assert ref.sample_replay_levels(5) == got and np.array_equal(ref.seed_staleness, s.seed_staleness)
    assert "_prefetch" not in s.__dict__


def test_prefetch_state_is_not_pickled():
    import pickle
    s = _sampler(8)
    s.prefetch_next_update(8)
    t = pickle.loads(pickle.dumps(s))
    assert "_prefetch" not in t.__dict__ and "_shadow_rng" not in t.__dict__
    s._drop_prefetch()


@pytest.mark.parametrize("chunk,extra", [(512, 400), (16, 400), (16, 37), (7, 0)])
def test_streamed_per_step_draws_equal_the_sequential_order(chunk, extra, monkeypatch):
    """The per-episode re-sampling of a replay rollout asks for a few draws per step: with `stream_extra` they come from the
    speculated continuation of the same sequential stream — across chunk boundaries (snapshot + replay of the remainder on
    settling), beyond the speculated amount (the rest is drawn in program order) and with zero-size requests in between."""
    from level_replay import LevelSampler
    monkeypatch.setattr(LevelSampler, "STREAM_CHUNK", chunk)
    B = 32
    runs = []
    for prefetch in (False, True):
        np.random.seed(21)
        s = _sampler(B, N=80)
        req = np.random.default_rng(3)
        log = []
        for it in range(8):
            if prefetch:
                s.prefetch_next_update(B, stream_extra=extra)
            if it == 5:
                np.random.rand()                                     # overtaken before the decision: everything is dropped
            d = s.sample_replay_decision()
            seeds = s.sample_replay_levels(B) if d else None
            steps = []
            if d:
                for t in range(40):
                    n = int(req.integers(0, 6)) if t % 3 else 0      # ragged counts, many empty steps
                    if it == 3 and t == 20:
                        np.random.rand()                             # overtaken in the middle of the rollout: the stream must let go
                    steps.append(s.sample_replay_levels(n))
            w = s.sample_weights().copy()                            # a reader: sees the staleness after exactly the draws made
            log.append((d, seeds, steps, w, _state(s)))
            s.seed_scores[(5 * it) % 80] = np.random.rand()
            s.after_update()
        runs.append((log, getattr(s, "stream_hits", 0)))
    (a, _), (b, hits) = runs
    if extra >= 100:
        assert hits > 100                                            # served from the stream
    for x, y in zip(a, b):
        assert x[0] == y[0] and x[1] == y[1] and x[2] == y[2], "decisions / seeds"
        assert np.array_equal(x[3], y[3]) and _same(x[4], y[4])


@pytest.mark.parametrize("chunk,extra", [(512, 300), (16, 300), (16, 40)])
def test_peeked_then_popped_prefix_equals_per_step_requests(chunk, extra, monkeypatch):
    """The device replay queue's use of the stream (adversarial_runner._queued_replay_rollout): peek the next draws without
    consuming, let the rollout take a prefix, then pop exactly that prefix in ONE request — seeds, staleness, weights and the
    position of np.random equal the run that asks a few per step.  peek beyond the speculation returns what is there (or None)."""
    from level_replay import LevelSampler
    monkeypatch.setattr(LevelSampler, "STREAM_CHUNK", chunk)
    B = 32
    runs = []
    for bulk in (False, True):
        np.random.seed(33)
        s = _sampler(B, N=80)
        req = np.random.default_rng(8)
        log = []
        for it in range(6):
            s.prefetch_next_update(B, stream_extra=extra)
            d = s.sample_replay_decision()
            seeds = s.sample_replay_levels(B) if d else None
            got = []
            if d:
                counts = [int(req.integers(0, 5)) for _ in range(30)]
                total = min(sum(counts), extra)
                if bulk:
                    peek = s.peek_stream(extra, extra)                     # (waits for the whole speculation: deterministic test)
                    assert peek is not None and len(peek) == extra
                    assert len(s.peek_stream(1, 5)) == 5
                    before = _state(s)
                    assert s.peek_stream(extra + 1, extra + 1) is None      # more than was speculated: not served
                    assert _same(before, _state(s))                       # peeking moved nothing
                    total = min(total, len(peek))
                    got = s.sample_replay_levels(total)
                    assert got == peek[:total]
                else:
                    left = total
                    for c in counts:
                        c = min(c, left)
                        got += s.sample_replay_levels(c)
                        left -= c
            w = s.sample_weights().copy()
            log.append((d, seeds, got, w, _state(s)))
            s.seed_scores[(7 * it) % 80] = np.random.rand()
            s.after_update()
        runs.append(log)
    assert sum(1 for x in runs[0] if x[0]) >= 2
    for x, y in zip(*runs):
        assert x[0] == y[0] and x[1] == y[1] and x[2] == y[2]
        assert np.array_equal(x[3], y[3]) and _same(x[4], y[4])
