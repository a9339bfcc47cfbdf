"""Concurrency / gap analysis of a rocprofv3 kernel trace (rocpd sqlite output).

usage: python tools/timeline.py <results.db> [skip_fraction]
Prints, for the last (1 - skip_fraction) of the dispatches: wall time, per-kernel launch count, average
duration and busy/wall ratio, the idle fraction (no kernel running) and the average number of kernels in flight.
"""
import collections
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    rows = db.execute("select name,start,end,queue_id,stream_id from kernels order by start").fetchall()
    rows = rows[int(len(rows) * skip):]
    lo = rows[0][1]
    hi = max(r[2] for r in rows)
    wall = hi - lo
    short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "").replace("amk::", "")[:34] or "(unnamed)"
    busy, cnt = collections.Counter(), collections.Counter()
    for n, s, e, _q, _st in rows:
        busy[short(n)] += e - s
        cnt[short(n)] += 1
    print(f"dispatches {len(rows)}  wall {wall / 1e6:.3f} ms  queues {len(set(r[3] for r in rows))}  "
          f"streams {len(set(r[4] for r in rows))}")
    for k, v in busy.most_common():
        print(f"  {k:<36s} n={cnt[k]:5d}  avg {v / cnt[k] / 1e3:8.1f} us  busy/wall {v / wall:5.2f}")
    ev = sorted([(r[1], 1) for r in rows] + [(r[2], -1) for r in rows])
    cur, last, idle, acc = 0, lo, 0, 0
    for t, d in ev:
        if cur == 0:
            idle += t - last
        acc += cur * (t - last)
        last = t
        cur += d
    print(f"idle fraction {idle / wall:.3f}   average kernels in flight {acc / wall:.2f}")


if __name__ == "__main__":
    main()
