#!/usr/bin/env python
"""Where the device waits inside a step: the idle gaps in front of every kernel of a rocprofv3 --kernel-trace result (rocpd
sqlite), per (previous kernel -> kernel) pair on the queue that runs the step kernel, and the other queues' busy time.

    python tools/rocpd_timeline.py <dir or .db> [out.txt] [skip_first_n_step_kernels]
"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.replace('void ', '').replace('odr::', '')
    return name.split('(')[0][:46]


def main(db, out=None, skip=2):
    if os.path.isdir(db):
        db = sorted(glob.glob(os.path.join(db, '**', '*.db'), recursive=True), key=os.path.getmtime)[-1]
    c = sqlite3.connect(db)
    cols = [d[0] for d in c.execute('select * from kernels limit 1').description]
    qcol = 'stream_id' if 'stream_id' in cols else ('queue_id' if 'queue_id' in cols else None)
    rows = list(c.execute('select name, start, end%s from kernels order by start' % ((', ' + qcol) if qcol else '')))
    if not qcol:
        rows = [r + (0, ) for r in rows]
    byq = defaultdict(list)
    for r in rows:
        byq[r[3]].append(r)
    main_q = max(byq, key=lambda q: sum(1 for r in byq[q] if 'k_step_grid' in r[0] or 'k_step_leeway' in r[0]))
    ks = byq[main_q]
    steps = [k for k, r in enumerate(ks) if 'k_step_grid' in r[0] or 'k_step_leeway' in r[0]]
    lines = ['columns of the kernel view: %s; queues: %s' % (qcol, {q: len(v) for q, v in byq.items()})]
    if len(steps) <= skip + 1:
        lines.append('too few step kernels')
    else:
        a, b = steps[skip], steps[-1]
        nst = len(steps) - 1 - skip
        span = ks[b][1] - ks[a][1]
        gaps, busy, each = defaultdict(lambda: [0, 0.0]), defaultdict(lambda: [0, 0.0]), defaultdict(list)
        for k in range(a, b):
            busy[short(ks[k][0])][0] += 1
            busy[short(ks[k][0])][1] += ks[k][2] - ks[k][1]
            g = ks[k + 1][1] - max(r[2] for r in ks[max(a, k - 8):k + 1])
            key = short(ks[k][0]) + ' -> ' + short(ks[k + 1][0])
            gaps[key][0] += 1
            gaps[key][1] += max(0, g)
            each[key].append(max(0, g) / 1e3)
        lines.append('steady part: %d steps, %.1f us per step (start of step kernel to start of step kernel)' % (nst, span / nst / 1e3))
        lines.append('busy on the step queue, us per step:')
        for kname, (cnt, tot) in sorted(busy.items(), key=lambda x: -x[1][1]):
            lines.append('  %-48s %6.2f calls/step %9.1f' % (kname, cnt / nst, tot / nst / 1e3))
        lines.append('  %-48s %26.1f' % ('sum', sum(t for _, t in busy.values()) / nst / 1e3))
        lines.append('idle in front of a kernel on the step queue, us per step:')
        for key, (cnt, tot) in sorted(gaps.items(), key=lambda x: -x[1][1])[:24]:
            lines.append('  %-96s %6.2f /step %8.1f' % (key, cnt / nst, tot / nst / 1e3))
        lines.append('  %-96s %20.1f' % ('sum', sum(t for _, t in gaps.values()) / nst / 1e3))
        for key, _ in sorted(gaps.items(), key=lambda x: -x[1][1])[:3]:
            lines.append('  every occurrence of %s, us: %s' % (key, ' '.join('%.0f' % g for g in each[key])))
        t0, t1 = ks[a][1], ks[b][1]
        for q, v in byq.items():
            if q == main_q:
                continue
            ob = defaultdict(float)
            for r in v:
                if r[1] >= t0 and r[2] <= t1:
                    ob[short(r[0])] += r[2] - r[1]
            if ob:
                lines.append('queue %s, busy us per step: %s' % (q, {k: round(t / nst / 1e3, 1) for k, t in ob.items()}))
    txt = '\n'.join(lines)
    print(txt)
    if out:
        open(out, 'w').write(txt + '\n')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, int(sys.argv[3]) if len(sys.argv) > 3 else 2)
