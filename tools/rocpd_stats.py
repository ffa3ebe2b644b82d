#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd (.db) result into the per-kernel stats table that
`rocprofv3 --kernel-trace --stats` summarises, as CSV (for profiles/)."""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = db.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall()
    with open(out_path, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDuration(us)', 'AverageDuration(us)', 'Percentage'])
        for name, calls, total, avg, pct in rows:
            w.writerow([name.split('(')[0] if name.startswith('r4r::') else name[:120], calls,
                        round(total, 3), round(avg, 3), round(pct, 3)])
    print('wrote', out_path, len(rows), 'kernels')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
