#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 --pmc run: python scripts/pmc_summary.py <dir> [name-substring]
-> JSON {kernel: {dispatches, counter: mean per dispatch}}."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    root = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    meta = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                if sub and sub not in k:
                    continue
                k = k.split("(")[0][:90]
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[k].add(row["Dispatch_Id"])
                meta[k] = {"vgpr": row.get("VGPR_Count"), "sgpr": row.get("SGPR_Count"), "lds": row.get("LDS_Block_Size"),
                           "grid": row.get("Grid_Size"), "wg": row.get("Workgroup_Size")}
    out = {}
    for k, c in acc.items():
        n = max(len(disp[k]), 1)
        out[k] = {"dispatches": n, **meta[k], **{name: round(v / n, 1) for name, v in sorted(c.items())}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
