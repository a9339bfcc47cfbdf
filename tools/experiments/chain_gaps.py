"""Per-stream timeline of a rocprofv3 kernel trace (rocpd sqlite): for every stream, the kernels of one launch chain in order with
their start offsets -- where a chain waits.  usage: python tools/experiments/chain_gaps.py <results.db> [stream index] [t0_frac]"""
import re, sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end,queue_id,stream_id from kernels order by start").fetchall()
which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.7
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "").replace("amk::", "").replace("(anonymous namespace)::", "")[:40]
streams = collections.Counter(r[4] for r in rows)
print("streams (kernels each):", dict(streams))
busy = sorted(streams, key=lambda k: -streams[k])
sid = busy[min(which, len(busy) - 1)]
rs = [r for r in rows if r[4] == sid]
rs = rs[int(len(rs) * frac):][:70]
t0 = rs[0][1]
prev_end = t0
for n, s, e, q, st in rs:
    print(f"  +{(s - t0) / 1e3:10.1f} us  gap {(s - prev_end) / 1e3:9.1f}  dur {(e - s) / 1e3:8.1f}  {short(n)}")
    prev_end = e
