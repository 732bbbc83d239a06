#!/usr/bin/env python
"""Dump the kernel summary of a rocprofv3 (rocpd sqlite) result: python tools/rocpd_summary.py x.db [out.txt]"""
import glob
import os
import sqlite3
import sys


def main(db, out=None):
    if os.path.isdir(db):   # rocprofv3 -d <dir>: take the newest database below it
        dbs = sorted(glob.glob(os.path.join(db, '**', '*.db'), recursive=True), key=os.path.getmtime)
        db = dbs[-1]
    c = sqlite3.connect(db)
    rows = list(c.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    lines = ['%-110s %8s %14s %12s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', '%')]
    for name, calls, tot, avg, pct in rows:
        lines.append('%-110s %8d %14.1f %12.1f %7.2f' % (name[:110], calls, tot / 1e3 if tot > 1e6 else tot, avg / 1e3 if tot > 1e6 else avg, pct))
    try:
        cur = c.execute('select * from counters_collection limit 1')
        cols = [d[0] for d in cur.description]
        if 'counter_name' in cols:
            lines.append('')
            lines.append('PMC counters (per dispatch, summed over dimensions):')
            q = ('select kernel_name, counter_name, count(*), sum(value)/count(*) from '
                 '(select dispatch_id, kernel_name, counter_name, sum(value) as value from counters_collection '
                 ' group by dispatch_id, kernel_name, counter_name) group by kernel_name, counter_name')
            for r in c.execute(q):
                lines.append('%-90s %-22s n=%-5d avg=%.6g' % (r[0][:90], r[1], r[2], r[3]))
    except Exception as e:  # no counters in this run
        pass
    txt = '\n'.join(lines)
    print(txt)
    if out:
        open(out, 'w').write(txt + '\n')


if __name__ == '__main__':
    main(*sys.argv[1:3])
