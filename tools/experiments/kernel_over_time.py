"""Durations of chosen kernels over the run (rocprofv3 rocpd sqlite): 20 time buckets, per kernel the mean / max duration in each.
usage: python tools/experiments/kernel_over_time.py <results.db> name_fragment [name_fragment ...]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
t0, t1 = rows[0][1], rows[-1][2]
nb = 20
for frag in sys.argv[2:]:
    b = [[] for _ in range(nb)]
    for n, s, e in rows:
        if frag in n:
            b[min(nb - 1, int((s - t0) * nb / (t1 - t0)))].append((e - s) / 1e3)
    print(frag, "(bucket: n mean max us)")
    print("  " + "  ".join(f"{len(x)}:{(sum(x) / len(x)) if x else 0:.0f}/{max(x) if x else 0:.0f}" for x in b))
