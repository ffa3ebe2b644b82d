"""The temporally blocked ID-table sweep on a schedule, restated (CPU oracle of csrc/rows_device.h + mf_engine.hip's
scheduled branch + engine._SweepSchedule's bookkeeping).

TEST INFRASTRUCTURE -- never imported by the product package.

What it restates (this repo's own design, not a reference file: the reference's optimiser is the DENSE sweep,
main.py:94-96 -> torch.optim.Adam over every row of the ID tables every step; oracle/optim.adam_step).  The claim under
test: visiting chunk c only at the steps s with (c % P + c / P + s) % P == 0 and letting whoever needs an element newer
apply the missing gradient-zero updates on the way gives every element EXACTLY the dense sweep's sequence of updates --
each step's update once, in step order, with that step's scalars.  `LazyTable` applies updates through a caller-given
`update(p, m, v, g, step)` and logs, per element, the steps applied; tests/test_sweep_schedule.py drives it with the
engine's own host bookkeeping and compares with the dense sweep (numpy float32 Adam: bit equality, and the logs)."""
import numpy as np

MF_TB_MAX = 8


def prev_visit(c, t, period):
    """rows_device.h tb_prev_visit: the last step <= t at which the schedule visits chunk c."""
    ph = (c % period + c // period) % period
    return t - (ph + t) % period


def due_chunk(q, now, period):
    """rows_device.h tb_due_chunk: the one chunk of block q (chunks q * period ...) visited at step `now`."""
    return q * period + (period - (q + now) % period) % period


class LazyTable:
    """One ID table [rows, width] under the scheduled sweep: p, m, v; rlast per row; the chunk tags."""

    def __init__(self, p, chunk, update):
        self.p = p.astype(np.float32).copy()
        self.m = np.zeros_like(self.p)
        self.v = np.zeros_like(self.p)
        self.rows, self.width = self.p.shape
        self.chunk = chunk
        self.nch = -(-self.p.size // chunk)
        self.rlast = np.zeros(self.rows, dtype=np.int64)
        self.ctag = np.zeros(self.nch, dtype=np.int64)
        self.update = update
        self.log = [[] for _ in range(self.p.size)]           # steps applied, per element

    def _current(self, e, now, period, base):
        """through which step is flat element e current before step `now`'s own update (tb_current)"""
        return max(base, prev_visit(e // self.chunk, now - 1, period), int(self.rlast[e // self.width]))

    def _apply(self, e, lo, hi, g_last=None):
        """updates of steps lo + 1 .. hi on flat element e (gradient zero, except g_last at step hi)"""
        assert hi - lo <= MF_TB_MAX, 'more pending updates than a visit can apply'
        P, M, V = self.p.reshape(-1), self.m.reshape(-1), self.v.reshape(-1)
        for s in range(lo + 1, hi + 1):
            g = np.float32(g_last) if (g_last is not None and s == hi) else np.float32(0.0)
            P[e], M[e], V[e] = self.update(P[e], M[e], V[e], g, s)
            self.log[e].append(s)

    def read_rows(self, ids, now, period, base):
        """what the forward of step `now` sees: the named rows as of step now - 1, nothing written back"""
        out = np.empty((len(ids), self.width), dtype=np.float32)
        P, M, V = self.p.reshape(-1), self.m.reshape(-1), self.v.reshape(-1)
        for k, r in enumerate(ids):
            for c in range(self.width):
                e = int(r) * self.width + c
                p, m, v = P[e], M[e], V[e]
                for s in range(self._current(e, now, period, base) + 1, now):
                    p, m, v = self.update(p, m, v, np.float32(0.0), s)
                out[k, c] = p
        return out

    def step(self, ids, grads, now, period, base, sweep_all, inc=1):
        """the update launch of step `now` (inc = 0: a flush through `now`, no rows named)"""
        touched = {}
        for k, r in enumerate(ids):                            # entry waves: a row's entries summed in batch order
            touched.setdefault(int(r), np.zeros(self.width, dtype=np.float32))
            touched[int(r)] = touched[int(r)] + grads[k].astype(np.float32)
        for r in touched:                                      # forward / register: tags
            e0 = r * self.width
            self.ctag[e0 // self.chunk] = now
            self.ctag[(e0 + self.width - 1) // self.chunk] = now
        # the sweep: the due chunks (all of them when sweep_all)
        if sweep_all:
            visit = range(self.nch)
        else:
            visit = [c for c in (due_chunk(q, now, period) for q in range(-(-self.nch // period))) if c < self.nch]
        for c in visit:
            vis = max(base, prev_visit(c, now - inc, period))
            recent = self.ctag[c] > vis
            for e in range(c * self.chunk, min((c + 1) * self.chunk, self.p.size)):
                r = e // self.width
                if inc and r in touched:
                    continue                                   # its entry wave owns it
                cur = max(vis, int(self.rlast[r])) if recent else vis
                if not recent:
                    assert self.rlast[r] <= vis                # (what the chunk tag promises)
                self._apply(e, cur, now)
        for r, g in touched.items():                           # entry waves: catch up, then the gradient update
            for c in range(self.width):
                e = r * self.width + c
                self._apply(e, self._current(e, now, period, base), now, g_last=g[c])
            self.rlast[r] = now
