#!/usr/bin/env python3
"""Summarise a tools/pmc.sh output directory for one kernel:  python tools/pmc_summary.py gpurun_out/<dir> <kernel-substring> [--md out.md]"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    d, kern = sys.argv[1], sys.argv[2]
    res = {"kernel": kern}
    tr = glob.glob(os.path.join(d, "stats", "**", "*kernel_trace.csv"), recursive=True)
    if tr:
        rows = [r for r in csv.DictReader(open(tr[0])) if kern in r["Kernel_Name"]]
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
        if dur:
            res["launches"] = len(dur); res["avg_us"] = sum(dur) / len(dur) / 1e3; res["min_us"] = min(dur) / 1e3
            for k in ("VGPR_Count", "Accum_VGPR_Count", "Scratch_Size", "Private_Segment_Size", "LDS_Block_Size", "Workgroup_Size", "Grid_Size"):
                if k in rows[0]:
                    res[k] = rows[0][k]
    for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            res[k] = sum(v) / len(v)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
