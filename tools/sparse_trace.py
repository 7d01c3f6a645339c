"""Timeline of the last steps of a chain() loop out of a rocprofv3 rocpd database (tools: sqlite3 only).
usage: python tools/sparse_trace.py results.db [nrows]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = [r for r in db.execute("select name, start, end, stream_id, queue_id from kernels order by start") if "mrg" in r[0]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 45
t0 = rows[-n][1]
for name, s, e, st, q in rows[-n:]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  s{st} q{q} {name[:64]}")
d = collections.defaultdict(list)
for name, s, e, st, q in rows[len(rows) // 2:]:
    d[name[:64]].append((e - s) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:66s} n={len(v):4d} avg={sum(v) / len(v):8.1f} us")
