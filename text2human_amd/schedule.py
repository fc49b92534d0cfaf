"""The unmasking schedule of `sample_fn` (models/sample_model.py:279-317), computed BEFORE the
transformer runs.

In the reference loop the tokens that change at step t are `rand([B,T]) < 1/t` minus the ones
already unmasked, and the only other consumer of the generator is one full `[B*T, n_class]`
`exponential_` draw per head that has a changed token at that step.  Nothing of this depends on
the transformer's logits: the whole schedule -- which row changes at which step, and the
generator offset of every draw -- is a function of the seed and of the texture map.  Two
consequences, both used by engine.sample_tokens:

* no host round trip inside the sampling loop (the reference's data-dependent `if`,
  sample_model.py:301-302, is resolved up front);
* a sample whose tokens do not change at a step never has its logits read there, its `x_t` does
  not move, and samples never interact (attention is per sample): every sample can walk through
  ITS OWN active steps.  Round r evaluates, for every sample, its r-th active step -- about
  0.865 * steps rounds of the full batch instead of `steps`, with the same tokens (up to fp32
  summation-order ties in the last layer's few-rows tail, whose tile depends on the row count).

This module is host-side integer bookkeeping (numpy); the draws themselves are reproduced on the
device (t2h_unmask_schedule / t2h_sample_heads) or taken from an explicit noise source.
"""
import numpy as np


def as_int64(u64):
    """A uint64 (torch's `initial_seed()` after `torch.seed()` / a negative `manual_seed`) as the int64 with the
    same bits: what an int64 device tensor can hold; the kernels read the word back as uint64."""
    u = int(u64) & 0xFFFFFFFFFFFFFFFF
    return u - (1 << 64) if u >= (1 << 63) else u


def draw_offsets(head_mask, steps, offset0, rand_inc, expo_inc, n_heads):
    """Generator offsets of the reference's draws.

    head_mask[t] (t = steps .. 1): bit h set iff head h samples at step t.  Returns
    (rand_off int64 [steps + 1], expo_off int64 [steps + 1, n_heads] (-1 where the head draws
    nothing), final offset): the generator holds rand_off[t] before the `rand` of step t and
    expo_off[t, h] before head h's `exponential_` of that step (heads in ascending order)."""
    head_mask = np.asarray(head_mask, dtype=np.int64)
    rand_off = np.zeros(steps + 1, dtype=np.int64)
    expo_off = np.full((steps + 1, n_heads), -1, dtype=np.int64)
    cur = int(offset0)
    for t in range(steps, 0, -1):
        rand_off[t] = cur
        cur += int(rand_inc)
        m = int(head_mask[t])
        for h in range(n_heads):
            if (m >> h) & 1:
                expo_off[t, h] = cur
                cur += int(expo_inc)
    return rand_off, expo_off, cur


def group_rounds(step_of_row, B, T, compact=True):
    """Rounds of the sampling loop.

    step_of_row int [B*T]: the step (>= 1) at which each token row changes.  compact=True: round r
    of sample b is its r-th active step (descending t); compact=False: round r is the r-th step (of
    any sample) that changes a token, for every sample alike -- the reference's own loop minus the
    steps that change nothing.  Returns
      order        int64 [B*T]   row ids sorted by (round, row),
      start        int64 [R + 1] order[start[r]:start[r + 1]] are the rows of round r,
      round_steps  int32 [R, B]  the step sample b is at in round r (0 = idle in that round)."""
    steps = np.asarray(step_of_row, dtype=np.int64).reshape(B, T)
    if steps.min() < 1:
        raise ValueError('every token row must have a step >= 1')
    if compact:
        uniq = [np.unique(steps[b]) for b in range(B)]            # ascending
    else:
        g = np.unique(steps)
        uniq = [g] * B
    n_rounds = max(len(u) for u in uniq)
    rnd = np.empty((B, T), dtype=np.int64)
    round_steps = np.zeros((n_rounds, B), dtype=np.int32)
    for b in range(B):
        u = uniq[b]
        rnd[b] = len(u) - 1 - np.searchsorted(u, steps[b])       # descending t = ascending round
        if compact:
            round_steps[:len(u), b] = u[::-1]
        else:
            present = np.isin(u, steps[b])
            round_steps[:len(u), b] = np.where(present, u, 0)[::-1]
    flat = rnd.reshape(-1)
    order = np.lexsort((np.arange(B * T), flat)).astype(np.int64)
    start = np.searchsorted(flat[order], np.arange(n_rounds + 1)).astype(np.int64)
    return order, start, round_steps


def leave_order(step_of_row, B, T):
    """Order of the samples for a compact schedule in which FINISHED samples leave the batch: samples sorted by the
    number of rounds they take part in (= their distinct steps), descending, ties in index order.  With the batch in
    this order the samples still active in round r are exactly the first k_r, so the transformer of round r runs on
    the first k_r * T rows only (about 2 % of the evaluations at B = 8, 3.5 % at B = 32, are (sample, round) pairs of
    samples that have no step left).  -> (perm int64 [B]: new position -> original sample, rounds per sample in the
    new order)"""
    steps = np.asarray(step_of_row, dtype=np.int64).reshape(B, T)
    n_act = np.array([len(np.unique(steps[b])) for b in range(B)], dtype=np.int64)
    perm = np.argsort(-n_act, kind='stable').astype(np.int64)
    return perm, n_act[perm]


def padded_tables(order, per_row, start, maxr):
    """The schedule as fixed-width tables for graph replay (engine.RoundGraph, t2h_schedule_advance):
    row r of the result = the rows of round r, padded to `maxr` entries with copies of the round's LAST
    row (sampling a row twice writes the same token twice), and the same for the per-row values
    (generator offsets).  -> (rows int32 [R, maxr], values int64 [R, maxr])"""
    n_rounds = len(start) - 1
    rows_tbl = np.empty((n_rounds, maxr), dtype=np.int32)
    val_tbl = np.empty((n_rounds, maxr), dtype=np.int64)
    for r in range(n_rounds):
        lo, hi = int(start[r]), int(start[r + 1])
        k = hi - lo
        if not 0 < k <= maxr:
            raise ValueError(f'round {r} has {k} rows, tables are {maxr} wide')
        rows_tbl[r, :k], val_tbl[r, :k] = order[lo:hi], per_row[lo:hi]
        rows_tbl[r, k:], val_tbl[r, k:] = order[hi - 1], per_row[hi - 1]
    return rows_tbl, val_tbl


class RoundTables:
    """Host side of graph replay (engine.RoundGraph.run): the run's schedule as padded tables plus the round
    cursor the captured graph advances.  `advance()` is the CPU statement of t2h_schedule_advance -- copy round
    *ctr of the tables into the staging rows / values, then ctr += 1 -- so the bookkeeping (how many replays, what
    each replay sees, that the padding only repeats rows of the same round) can be exercised without a GPU
    (tests/test_schedule.py, the 2-rank gloo run of tests/test_bench_dist.py)."""

    def __init__(self, order, per_row, start, maxr, per_row32=None):
        self.rows_tbl, self.val_tbl = padded_tables(order, per_row, start, maxr)
        # second per-row channel (int32: the rows of the reference's noise tensor when the samples were reordered)
        self.aux32_tbl = (padded_tables(order, np.asarray(per_row32, dtype=np.int64), start, maxr)[1].astype(np.int32)
                          if per_row32 is not None else None)
        self.n_rounds, self.maxr = len(start) - 1, int(maxr)
        self.ctr = 0

    def advance(self):
        if self.ctr >= self.n_rounds:
            raise IndexError(f'round cursor {self.ctr} past the {self.n_rounds} rounds of this run')
        r = self.ctr
        self.ctr += 1
        return self.rows_tbl[r].copy(), self.val_tbl[r].copy()

    def replay(self, body, first_eager=True):
        """The launch pattern of RoundGraph.run: round 0 eagerly on the run that captures (first_eager), then one
        replay per remaining round; body(rows, values) is called once per round.  -> number of replays."""
        self.ctr = 0
        for _ in range(self.n_rounds):
            body(*self.advance())
        return self.n_rounds - (1 if first_eager and self.n_rounds else 0)


def stats(round_steps, steps, active=None):
    """Evaluation counts for the bench line: (sample, step) pairs the reference evaluates, pairs that
    change a token (the ones whose logits are read), rounds launched, and the (sample, round) pairs the
    transformer was actually run for (`active[r]` samples in round r; default: the whole batch)."""
    n_rounds, B = round_steps.shape
    launched = int(n_rounds * B) if active is None else int(np.asarray(active).sum())
    return dict(rounds=int(n_rounds), steps=int(steps), batch=int(B),
                sample_steps_possible=int(B * steps),
                sample_steps_needed=int((round_steps > 0).sum()),
                sample_steps_launched=launched)
