"""Timeline of a short timed region of bench.py out of a rocprofv3 rocpd database: start-to-start intervals of the level-0
launch, and what follows the last small-levels launch.  python tools/region_trace.py results.db [nsteps]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = [r for r in db.execute("select name, start, end from kernels order by start") if "mrg" in r[0]]
l0 = [i for i, r in enumerate(rows) if "chess_v1_pyr_kernel" in r[0]]
sel = l0[-n:]
t0 = rows[sel[0]][1]
prev = None
for k, i in enumerate(sel):
    s, e = rows[i][1], rows[i][2]
    print(f"step {k:2d}: L0 start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  start-to-start {((s - prev) / 1e3) if prev else 0:7.1f}")
    prev = s
last_multi = max(i for i, r in enumerate(rows) if "multi" in r[0])
tm = rows[last_multi][2]
print(f"last small-levels launch ends at {(tm - t0) / 1e3:.1f} us; after it:")
for r in rows[last_multi + 1:]:
    print(f"   {(r[1] - tm) / 1e3:8.1f} .. {(r[2] - tm) / 1e3:8.1f} us  {r[0][:60]}")
