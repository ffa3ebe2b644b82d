#!/usr/bin/env python3
"""Device timeline of a rocprofv3 --kernel-trace run (rocpd .db): every idle gap longer than GAP_US starts a
'region'; for each region print the first N dispatches (start / duration / gap to the previous end, in us) --
what the first steps behind a bench fence look like next to the steady state.
python tools/kernel_timeline.py <db> [N=24] [GAP_US=40]"""
import sqlite3
import sys


def main(db_path, n=24, gap_us=40.0):
    db = sqlite3.connect(db_path)
    names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = 'kernels' if 'kernels' in names else None
    if view is None:
        print('no kernels view; have:', names)
        return
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    s, e = ('start', 'end') if 'start' in cols else ('start_timestamp', 'end_timestamp')
    rows = db.execute('select name, %s, %s from kernels order by %s' % (s, e, s)).fetchall()
    rows = [(nm.split('(')[0].replace('r4r::', '').replace('void ', '')[:34], a / 1e3, b / 1e3) for nm, a, b in rows]
    regions, cur = [], []
    for i, r in enumerate(rows):
        if i and r[1] - rows[i - 1][2] > gap_us:
            regions.append(cur)
            cur = []
        cur.append(r)
    regions.append(cur)
    for k, reg in enumerate(regions):
        if len(reg) < 40:
            continue
        t0 = reg[0][1]
        span = reg[-1][2] - t0
        print('--- region %d: %d dispatches, %.1f us' % (k, len(reg), span))
        prev = None
        for nm, a, b in reg[:n]:
            print('  %-34s start %8.1f  dur %6.1f  gap %5.1f' % (nm, a - t0, b - a, 0.0 if prev is None else a - prev))
            prev = b
        print('  gemm durations:', ' '.join('%.0f' % (b - a) for nm, a, b in reg if 'gemm' in nm))
        # steady state: median per-kernel duration over the second half of the region
        half = reg[len(reg) // 2:]
        by = {}
        for nm, a, b in half:
            by.setdefault(nm, []).append(b - a)
        print('  second half medians:', ', '.join('%s %.1f' % (nm, sorted(v)[len(v) // 2]) for nm, v in by.items()))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24, float(sys.argv[3]) if len(sys.argv) > 3 else 40.0)
