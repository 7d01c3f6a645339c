#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd .db) for this repo's kernels.

    python tools/rocprof_summary.py gpurun_out/<run>/<name>_results.db > profiles/<round>_<what>.txt

One row per (kernel, grid): calls, average / min / max duration, registers, LDS.
Only `mrg::` kernels are listed in detail; everything else (torch's synthetic
frame generator, memsets, copies) is folded into one line.
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, grid_x, grid_y, grid_z, workgroup_x, duration, vgpr_count, sgpr_count, lds_size "
                     "from kernels").fetchall()
    groups, other = {}, [0, 0.0]
    for name, gx, gy, gz, wx, dur, vg, sg, lds in rows:
        if "mrg::" not in name:
            other[0] += 1
            other[1] += dur
            continue
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        groups.setdefault((short, gx, gy, gz, wx, vg, sg, lds), []).append(dur)
    print(f"# {path}")
    print(f"# {'kernel':58s} {'grid(threads)':>22s} {'wg':>5s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} "
          f"{'max_us':>10s} {'total_ms':>9s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>6s}")
    tot = 0.0
    for key, d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        short, gx, gy, gz, wx, vg, sg, lds = key
        tot += sum(d)
        print(f"  {short:58s} {f'{gx}x{gy}x{gz}':>22s} {wx:5d} {len(d):6d} {sum(d) / len(d) / 1e3:10.1f} "
              f"{min(d) / 1e3:10.1f} {max(d) / 1e3:10.1f} {sum(d) / 1e6:9.2f} {vg:5d} {sg:5d} {lds:6d}")
    print(f"# mrg:: kernels total {tot / 1e6:.2f} ms; other kernels (synthetic-frame generator, fills, copies): "
          f"{other[0]} calls, {other[1] / 1e6:.2f} ms")
    # the dominant kernel: median and quartiles next to the average above.  The first dozens of launches of a
    # process run slower (clock ramp, first touch of the scratch) and bench.py's host-fed leg at the end runs
    # its launches beside the uploads, so the average over ALL calls is not what the timed region sees.
    try:
        if groups:
            top = max(groups.items(), key=lambda kv: sum(kv[1]))
            d = sorted(top[1])
            if len(d) >= 8:
                q = lambda f: d[int(f * (len(d) - 1))] / 1e3
                print(f"# {top[0][0]}: {len(d)} launches, quartiles {q(0.25):.1f} / {q(0.5):.1f} / {q(0.75):.1f} us")
    except (ValueError, IndexError) as e:
        print(f"# (no quartile line: {e})")
    # idle time of the pixel stream between its kernels (pyramid -> small levels -> level 0 -> next pyramid)
    try:
        cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
        if "start" in cols and "end" in cols:
            pix = c.execute("select name, start, end from kernels where name like '%pyramid%' or name like '%chess_v1%' "
                            "order by start").fetchall()
            gaps = [(pix[i + 1][1] - pix[i][2]) / 1e3 for i in range(len(pix) - 1)]
            kinds = {}
            for i, g in enumerate(gaps):
                if 0 <= g < 200:
                    short = lambda n: ("chess+pyramid" if "chess_v1_pyr" in n else "pyramid" if "pyramid" in n
                                       else "multi" if "multi" in n else "chess")
                    kinds.setdefault(short(pix[i][0]) + " -> " + short(pix[i + 1][0]), []).append(g)
            for k, v in sorted(kinds.items()):
                v.sort()
                print(f"#   {k:20s} n={len(v):4d}  median {v[len(v) // 2]:6.1f} us  mean {sum(v) / len(v):6.1f} us")
            gaps = [g for g in gaps if 0 <= g < 200]          # drop the host-side pauses between phases of the run
            steps = sum(1 for n, _, _ in pix if "pyramid" in n or "chess_v1_pyr" in n)
            if gaps and steps:
                gaps.sort()
                print(f"# pixel stream: {len(gaps)} kernel boundaries, median gap {gaps[len(gaps) // 2]:.1f} us, mean "
                      f"{sum(gaps) / len(gaps):.1f} us, {sum(gaps) / steps:.1f} us per step ({steps} steps)")
    except sqlite3.Error as e:
        print(f"# (no gap analysis: {e})")


if __name__ == "__main__":
    main(sys.argv[1])
