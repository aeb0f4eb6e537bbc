#!/usr/bin/env python3
"""Summarise rocprofv3 output directories for the library's own kernels (names containing ``egs::``):
   pmc_summary.py counters DIR [DIR ...]  -> JSON {kernel: {counter: sum over dispatches, "launches": n}}
   pmc_summary.py trace DIR               -> CSV  kernel, calls, average ns, total ns (from *_kernel_trace.csv)
Several DIRs = several --pmc passes of the same command (counters are merged)."""
import csv
import glob
import json
import os
import re
import sys

csv.field_size_limit(1 << 30)


def short(name):
    m = re.search(r"egs::([A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    return "egs::" + m.group(1) if m else None


def counters(dirs):
    out = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            seen = {}
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    k = short(row["Kernel_Name"])
                    if k is None:
                        continue
                    ent = out.setdefault(k, {})
                    c = row["Counter_Name"]
                    ent[c] = ent.get(c, 0.0) + float(row["Counter_Value"])
                    seen.setdefault(k, set()).add(row["Dispatch_Id"])
            for k, s in seen.items():
                out[k]["launches"] = max(out[k].get("launches", 0), len(s))
    return out


def trace(d):
    acc = {}
    for path in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if k is None:
                    continue
                dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                a = acc.setdefault(k, [0, 0, 1 << 62, 0, row.get("VGPR_Count"), row.get("SGPR_Count"),
                                       row.get("Scratch_Size"), row.get("LDS_Block_Size")])
                a[0] += 1
                a[1] += dur
                a[2] = min(a[2], dur)
                a[3] = max(a[3], dur)
    print("Name,Calls,AverageNs,TotalDurationNs,MinNs,MaxNs,VGPR,SGPR,Scratch,LDS")
    for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f'"{k}",{a[0]},{a[1] / a[0]:.0f},{a[1]},{a[2]},{a[3]},{a[4]},{a[5]},{a[6]},{a[7]}')


if __name__ == "__main__":
    if sys.argv[1] == "counters":
        print(json.dumps(counters(sys.argv[2:]), indent=1, sort_keys=True))
    else:
        trace(sys.argv[2])
