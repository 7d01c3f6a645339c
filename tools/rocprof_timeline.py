#!/usr/bin/env python3
"""Timeline of this repo's kernels out of a rocprofv3 --kernel-trace run (rocpd .db): the last few steps of the run,
one line per kernel with its queue / stream, start relative to the first line, duration and the gap to the previous
kernel on the same queue.  python tools/rocprof_timeline.py <results.db> [steps = 4] [marker = pyramid_fast|chess_v1_pyr]"""
import sqlite3
import sys


def main(path, steps=4, marker=None):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
    rows = c.execute(f"select name, start, end, {qcol or '0'}, grid_x, grid_y from kernels where name like '%mrg::%' order by start").fetchall()
    if not rows:
        print("no mrg:: kernels"); return
    if marker is None:
        marker = "pyramid_fast" if any("pyramid_fast" in r[0] for r in rows[-200:]) else "chess_v1_pyr"
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) <= steps:
        first = 0
    else:
        first = marks[-steps - 1]
    t0 = rows[first][1]
    last_end = {}
    print(f"# {path}; columns {qcol}; marker {marker}")
    for name, st, en, q, gx, gy in rows[first:]:
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("mrg::", "")
        gap = (st - last_end[q]) / 1e3 if q in last_end else float("nan")
        last_end[q] = en
        print(f"q{q:<3} {(st - t0) / 1e3:9.1f} us  +{(en - st) / 1e3:7.1f} us  gap {gap:7.1f}  {short}  [{gx}x{gy}]")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4, sys.argv[3] if len(sys.argv) > 3 else None)
