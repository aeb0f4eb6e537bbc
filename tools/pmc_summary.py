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


def _groups(durs, ratio=1.5):
    """Split a kernel's launch durations where consecutive sorted values differ by more than `ratio`: bench.py launches
    the same kernel on different batch sizes (the timed C2 batch and, e.g., the quarter-size Gaussian set), and one
    average over both would describe neither."""
    ds = sorted(durs)
    out, cur = [], [ds[0]]
    for v in ds[1:]:
        if v > ratio * cur[-1]:
            out.append(cur)
            cur = []
        cur.append(v)
    out.append(cur)
    return out


def trace(d):
    acc, meta = {}, {}
    for path in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if k is None:
                    continue
                dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                acc.setdefault(k, []).append(dur)
                meta[k] = [row.get("VGPR_Count"), row.get("SGPR_Count"), row.get("Scratch_Size"), row.get("LDS_Block_Size")]
    print("Name,Calls,AverageNs,TotalDurationNs,MinNs,MaxNs,VGPR,SGPR,Scratch,LDS")
    rows = []
    for k, durs in acc.items():
        gs = _groups(durs)
        for gi, g in enumerate(sorted(gs, key=lambda g: -sum(g))):
            # one row per duration group; a kernel launched on ONE problem size has one group and keeps its plain name
            name = k if len(gs) == 1 else "%s [launch group %d of %d by duration]" % (k, gi + 1, len(gs))
            rows.append((sum(g), name, len(g), sum(g) / len(g), min(g), max(g), meta[k]))
    for tot, name, n, avg, lo, hi, m in sorted(rows, key=lambda r: -r[0]):
        print(f'"{name}",{n},{avg:.0f},{tot},{lo},{hi},{m[0]},{m[1]},{m[2]},{m[3]}')


if __name__ == "__main__":
    if sys.argv[1] == "counters":
        print(json.dumps(counters(sys.argv[2:]), indent=1, sort_keys=True))
    else:
        trace(sys.argv[2])
