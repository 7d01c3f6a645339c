#!/usr/bin/env python3
"""Average the rocprofv3 --pmc counters of this repo's kernels.

    python tools/pmc_summary.py gpurun_out/<run>/p*/p_counter_collection.csv

Prints, per kernel name and grid size (= pyramid level), the mean of every counter over its dispatches (the last
dispatch only with --last).  FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3
reports them; see MI355X_MICROARCH.md (HBM section) for the gfx950 correction
(FETCH_SIZE counts 64 B per 128-B request on wide streaming reads: double it).
"""
import csv
import sys
from collections import defaultdict


def main(paths):
    acc = defaultdict(lambda: defaultdict(list))
    for p in paths:
        for row in csv.DictReader(open(p)):
            name = row["Kernel_Name"]
            if "mrg::" not in name:
                continue
            short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "") + f"  grid={row['Grid_Size']} wg={row['Workgroup_Size']}"
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, ctrs in acc.items():
        print(k)
        for c, v in sorted(ctrs.items()):
            print(f"    {c:28s} mean {sum(v) / len(v):16.1f}   n={len(v)}")


if __name__ == "__main__":
    main(sys.argv[1:])
