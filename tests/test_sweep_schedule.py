"""The scheduled temporally blocked sweep against the dense sweep on the CPU: oracle/sweep_schedule.py (a restatement of
csrc/rows_device.h's schedule and mf_engine.hip's scheduled branch) driven by the engines' own host bookkeeping
(reviews4rec_amd.engine._SweepSchedule) through random loops of training steps on and off the schedule, flushes,
period changes and reads -- every element must receive every step's update exactly once, in order, and end with the
dense sweep's bits."""
import random

import numpy as np
import pytest

from oracle import sweep_schedule as S


def adam32(p, m, v, g, step, lr=np.float32(0.002), wd=np.float32(1e-6), b1=0.9, b2=0.999, eps=np.float32(1e-8)):
    """One element of torch.optim.Adam(lr, weight_decay) in float32 (oracle/optim.adam_step's arithmetic)."""
    f = np.float32
    g = f(g + wd * p)
    m = f(f(b1) * m + f(1.0 - b1) * g)
    v = f(f(b2) * v + f(f(1.0 - b2) * g) * g)
    bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
    denom = f(f(np.sqrt(v)) * f(1.0 / np.sqrt(bc2)) + eps)
    return f(p - f(lr / bc1) * f(m / denom)), m, v


def test_schedule_visits_every_chunk_once_per_period():
    for period in range(1, S.MF_TB_MAX + 1):
        for nch in (1, 2, 7, 8, 9, 63, 64, 65, 1000):
            seen = np.zeros(nch, dtype=np.int64)
            for now in range(100, 100 + period):
                due = [S.due_chunk(q, now, period) for q in range(-(-nch // period))]
                due = [c for c in due if c < nch]
                assert len(set(due)) == len(due)
                for c in due:
                    assert S.prev_visit(c, now, period) == now            # due now
                    assert S.prev_visit(c, now - 1, period) == now - period
                    seen[c] += 1
                for c in range(nch):                                       # and nobody else is
                    if c not in due:
                        assert S.prev_visit(c, now, period) < now
            assert (seen == 1).all(), (period, nch)


@pytest.mark.parametrize('width,chunk,seed', [(4, 16, 0), (5, 16, 1), (3, 8, 2), (8, 8, 3)])
def test_scheduled_sweep_is_the_dense_sweep_element_by_element(width, chunk, seed):
    from reviews4rec_amd.engine import _SweepSchedule

    class Host(_SweepSchedule):                      # the engines' bookkeeping, nothing else
        has_tables, _ws, sweep_period, step_count = True, object(), 8, 0

    rnd = random.Random(seed)
    rng = np.random.default_rng(seed)
    rows = 37
    p0 = rng.standard_normal((rows, width)).astype(np.float32)
    lazy, host = S.LazyTable(p0, chunk, adam32), Host()
    dense_p, dense_m, dense_v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for _ in range(60):
        act = rnd.random()
        if act < 0.08:                                # a flush (evaluation, state_dict ...)
            step = host._pending()
            if step is not None:
                lazy.step([], [], step, host._tb_period, host._tb_base, 1, inc=0)
                host._tb_base = step
            np.testing.assert_array_equal(lazy.p, dense_p)
            continue
        if act < 0.16:
            host.sweep_period = rnd.choice([1, 2, 3, 5, 8])
        defer = rnd.random() < 0.85
        B = rnd.choice([1, 3, 6])
        ids = [rnd.randrange(rows) if rnd.random() < 0.7 else rows - 1 for _ in range(B)]
        grads = rng.standard_normal((B, width)).astype(np.float32)
        host.step_count += 1
        now = host.step_count
        period, base, sweep_all, want = host._schedule(defer)
        # the forward reads current rows
        seen = lazy.read_rows(ids, now, period, base)
        np.testing.assert_array_equal(seen, dense_p[ids])
        lazy.step(ids, grads, now, period, base, sweep_all)
        host._scheduled(sweep_all, want, now)
        # the dense sweep: every element, this step's update (rows summed in batch order)
        G = np.zeros_like(dense_p)
        for k, r in enumerate(ids):
            G[r] = G[r] + grads[k]
        for r in range(rows):
            for c in range(width):
                dense_p[r, c], dense_m[r, c], dense_v[r, c] = adam32(dense_p[r, c], dense_m[r, c], dense_v[r, c], G[r, c], now)
    step = host._pending()
    if step is not None:
        lazy.step([], [], step, host._tb_period, host._tb_base, 1, inc=0)
    np.testing.assert_array_equal(lazy.p, dense_p)
    np.testing.assert_array_equal(lazy.m, dense_m)
    np.testing.assert_array_equal(lazy.v, dense_v)
    for e, log in enumerate(lazy.log):                # every step's update exactly once, in order
        assert log == list(range(1, host.step_count + 1)), (e, log[:12])
