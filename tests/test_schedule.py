"""Host-side bookkeeping of the sampling schedule (text2human_amd/schedule.py): CPU, numpy only."""
import numpy as np
import pytest

from text2human_amd import schedule


def _random_schedule(B, T, steps, seed):
    """step of every token as the reference's loop would assign it (uniform thresholds 1/t)"""
    rng = np.random.default_rng(seed)
    step = np.zeros(B * T, dtype=np.int64)
    for t in range(steps, 0, -1):
        hit = (rng.random(B * T) < np.float32(1.0) / np.float32(t)) & (step == 0)
        step[hit] = t
    assert (step > 0).all()
    return step


@pytest.mark.parametrize('B,T,steps', [(8, 512, 256), (1, 512, 5), (3, 64, 1), (32, 512, 256)])
def test_compact_rounds_cover_every_row_once_and_keep_each_samples_order(B, T, steps):
    step = _random_schedule(B, T, steps, seed=B * 7 + steps)
    order, start, round_steps = schedule.group_rounds(step, B, T, compact=True)
    assert sorted(order.tolist()) == list(range(B * T))
    R = len(start) - 1
    assert start[0] == 0 and start[-1] == B * T and round_steps.shape == (R, B)
    for b in range(B):
        mine = np.unique(step[b * T:(b + 1) * T])[::-1]
        assert round_steps[:len(mine), b].tolist() == mine.tolist()      # descending t, no gaps
        assert (round_steps[len(mine):, b] == 0).all()
    for r in range(R):
        rows = order[start[r]:start[r + 1]]
        assert (np.diff(rows) > 0).all()                                # sorted by row inside a round
        assert (step[rows] == round_steps[r, rows // T]).all()
    st = schedule.stats(round_steps, steps)
    assert st['sample_steps_needed'] == sum(len(np.unique(step[b * T:(b + 1) * T])) for b in range(B))
    assert st['rounds'] == max(len(np.unique(step[b * T:(b + 1) * T])) for b in range(B)) <= steps
    if steps == 256 and T == 512:   # P(no change) = (1 - 1/t)^(masked) ~ exp(-2): 13.5 % of the pairs
        assert 0.84 < st['sample_steps_needed'] / st['sample_steps_possible'] < 0.89


def test_synchronous_rounds_are_the_reference_loop_minus_empty_steps():
    B, T, steps = 4, 128, 60
    step = _random_schedule(B, T, steps, seed=3)
    order, start, round_steps = schedule.group_rounds(step, B, T, compact=False)
    active = np.unique(step)[::-1]
    assert len(start) - 1 == len(active)
    for r, t in enumerate(active):
        rows = order[start[r]:start[r + 1]]
        assert (step[rows] == t).all() and len(rows) == (step == t).sum()
        for b in range(B):
            assert round_steps[r, b] == (t if (step[b * T:(b + 1) * T] == t).any() else 0)


def test_draw_offsets_follow_the_generator():
    steps, H = 6, 18
    mask = np.zeros(steps + 1, dtype=np.int64)
    mask[6], mask[5], mask[3], mask[1] = 0b101, 0, 1 << 17, 0b11
    rand_off, expo_off, final = schedule.draw_offsets(mask, steps, 40, 4, 4096, H)
    assert rand_off[6] == 40 and expo_off[6, 0] == 44 and expo_off[6, 2] == 44 + 4096
    assert rand_off[5] == 44 + 2 * 4096 and (expo_off[5] == -1).all()
    assert rand_off[4] == rand_off[5] + 4 and rand_off[3] == rand_off[4] + 4
    assert expo_off[3, 17] == rand_off[3] + 4 and rand_off[2] == rand_off[3] + 4 + 4096
    assert expo_off[1, 0] == rand_off[1] + 4 and expo_off[1, 1] == rand_off[1] + 4 + 4096
    assert final == rand_off[1] + 4 + 2 * 4096
    assert (expo_off[[6, 3, 1]] >= 0).sum() == 5


def test_group_rounds_rejects_rows_without_a_step():
    with pytest.raises(ValueError):
        schedule.group_rounds(np.array([1, 0, 2, 1]), 1, 4)


def test_padded_tables_repeat_a_row_of_the_same_round():
    B, T, steps = 3, 64, 20
    step = _random_schedule(B, T, steps, seed=11)
    order, start, _ = schedule.group_rounds(step, B, T, compact=True)
    vals = (order * 7 + 3).astype(np.int64)
    widest = int(np.diff(start).max())
    maxr = -(-widest // 16) * 16
    rows_tbl, val_tbl = schedule.padded_tables(order, vals, start, maxr)
    assert rows_tbl.shape == val_tbl.shape == (len(start) - 1, maxr) and rows_tbl.dtype == np.int32
    for r in range(len(start) - 1):
        mine = order[start[r]:start[r + 1]]
        assert rows_tbl[r, :len(mine)].tolist() == mine.tolist()
        assert set(rows_tbl[r, len(mine):].tolist()) <= {int(mine[-1])}         # padding = a row of THIS round
        assert (val_tbl[r] == rows_tbl[r].astype(np.int64) * 7 + 3).all()        # values travel with their rows
    with pytest.raises(ValueError):
        schedule.padded_tables(order, vals, start, widest - 1)


def test_seed_as_int64_roundtrip():
    """torch's initial_seed() is a uint64; the graph path keeps it in an int64 device word."""
    import torch
    for u in (0, 5, 2**63 - 1, 2**63, 2**64 - 3):
        i = schedule.as_int64(u)
        assert -2**63 <= i < 2**63 and (i & 0xFFFFFFFFFFFFFFFF) == u
        t = torch.empty(1, dtype=torch.int64).fill_(i)      # (fill_(u) itself raises for u >= 2^63)
        assert int(t.view(torch.uint8).numpy().view('<u8')[0]) == u


def test_round_tables_replay_visits_every_row_in_its_own_round():
    """schedule.RoundTables = the host bookkeeping of graph replay with the CPU statement of t2h_schedule_advance."""
    B, T, steps = 5, 64, 30
    step = _random_schedule(B, T, steps, seed=21)
    order, start, round_steps = schedule.group_rounds(step, B, T, compact=True)
    offs = (order * 11 + 5).astype(np.int64)
    maxr = -(-int(np.diff(start).max()) // 16) * 16
    tb = schedule.RoundTables(order, offs, start, maxr)
    seen, rounds = np.zeros(B * T, dtype=np.int64), []

    def body(rows, vals):
        r = len(rounds)
        rounds.append(rows)
        assert rows.shape == (maxr, ) and (vals == rows.astype(np.int64) * 11 + 5).all()
        uniq = np.unique(rows)
        assert (step[uniq] == round_steps[r, uniq // T]).all()     # every listed row belongs to THIS round
        seen[uniq] += 1

    assert tb.replay(body) == tb.n_rounds - 1 == len(rounds) - 1
    assert (seen == 1).all()
    with pytest.raises(IndexError):
        tb.advance()


@pytest.mark.parametrize('B,T,steps', [(8, 512, 256), (32, 512, 256), (5, 64, 9)])
def test_leave_order_makes_the_running_samples_a_prefix(B, T, steps):
    """schedule.leave_order: with the samples sorted by their number of rounds, the samples still running in round r
    are the first k_r of the batch, k_r never grows, and the (sample, round) pairs run = the pairs that change a token."""
    step = _random_schedule(B, T, steps, seed=B + steps)
    perm, n_act = schedule.leave_order(step, B, T)
    assert sorted(perm.tolist()) == list(range(B)) and (np.diff(n_act) <= 0).all()
    orig_row = (perm[:, None] * T + np.arange(T)[None, :]).reshape(-1)
    order, start, round_steps = schedule.group_rounds(step[orig_row], B, T, compact=True)
    active = (round_steps > 0).sum(1)
    assert (np.diff(active) <= 0).all() and active[0] == B and active[-1] >= 1
    for r in range(len(active)):
        assert (round_steps[r, :active[r]] > 0).all() and (round_steps[r, active[r]:] == 0).all()
        rows = order[start[r]:start[r + 1]]
        assert (rows // T < active[r]).all()                      # a round only lists rows of running samples
    st = schedule.stats(round_steps, steps, active)
    assert st['sample_steps_launched'] == st['sample_steps_needed'] <= st['rounds'] * B
    if steps == 256:
        assert st['sample_steps_launched'] < st['rounds'] * B       # somebody finishes before the slowest sample


def _toy_token(state_b, row_in_sample, draw_row, step, head):
    """Stand-in for (transformer + categorical draw): depends on the WHOLE current state of the sample -- like
    attention -- and on the identity of the noise element (the row of the reference's draw, the step, the head)."""
    h = (int(state_b.sum()) * 1000003 + int((state_b * (np.arange(len(state_b)) + 1)).sum())) & 0x7FFFFFFF
    return (h ^ (draw_row * 2654435761) ^ (step * 40503) ^ (head * 97)) % 1024


def _reference_loop(step, tex, B, T, steps, mask_id):
    """models/sample_model.py:279-317 with the toy model: every step, all samples, tokens of that step sampled from the
    state BEFORE the step's writes."""
    x = np.full((B, T), mask_id, dtype=np.int64)
    for t in range(steps, 0, -1):
        rows = np.nonzero(step == t)[0]
        new = [(r, _toy_token(x[r // T], r % T, r, t, int(tex[r]))) for r in rows]
        for r, v in new:
            x[r // T, r % T] = v + 1024 * int(tex[r])
    return x


def _rounds_loop(step, tex, B, T, mask_id, shrink):
    """engine.sample_tokens' host logic: (reordered) compact rounds; a round evaluates only the running prefix."""
    perm = np.arange(B)
    if shrink:
        perm, _ = schedule.leave_order(step, B, T)
    orig_row = (perm[:, None] * T + np.arange(T)[None, :]).reshape(-1)
    step_p, tex_p = step[orig_row], tex[orig_row]
    order, start, round_steps = schedule.group_rounds(step_p, B, T, compact=True)
    active = (round_steps > 0).sum(1)
    rng_rows = orig_row[order]
    x = np.full((B, T), mask_id, dtype=np.int64)
    for r in range(len(start) - 1):
        lo, hi = int(start[r]), int(start[r + 1])
        k = int(active[r]) if shrink else B
        before = x.copy()                                   # the round's "hidden states": one evaluation, then writes
        for i in range(lo, hi):
            row = int(order[i])
            assert row // T < k
            x[row // T, row % T] = _toy_token(before[row // T], row % T, int(rng_rows[i]), int(step_p[row]),
                                              int(tex_p[row])) + 1024 * int(tex_p[row])
    return x[np.argsort(perm)]                              # back in the caller's order (engine.in_batch_order)


@pytest.mark.parametrize('B,T,steps,seed', [(4, 32, 16, 0), (7, 16, 40, 1), (3, 64, 256, 2), (1, 8, 3, 3)])
def test_compact_and_shrinking_rounds_give_the_reference_loops_tokens(B, T, steps, seed):
    """With a toy model whose output depends on a sample's whole current state (and on which element of the reference's
    noise tensor a row uses), the compact rounds -- with and without finished samples leaving the batch -- end in the
    same tokens as the reference's synchronous loop: samples never interact, so each may walk its own steps."""
    step = _random_schedule(B, T, steps, seed)
    tex = np.random.default_rng(seed + 100).integers(0, 18, B * T)
    want = _reference_loop(step, tex, B, T, steps, mask_id=18432)
    assert (want != 18432).all()
    for shrink in (False, True):
        got = _rounds_loop(step, tex, B, T, 18432, shrink)
        assert (got == want).all(), f'shrink={shrink}'
