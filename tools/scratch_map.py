#!/usr/bin/env python3
"""Where a kernel's scratch (spill) instructions come from: (source file, line, loop depth, load/store) -> count, from
`hipcc -S -gline-tables-only` (no GPU needed).
    python tools/scratch_map.py eig_real_lds.hip eig_real64_kernel [-DFLAG ...]"""
import os, re, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "task-level-insights-from-eigenvalues-across-sequence-models_amd")
src, kern, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(PKG, "csrc"), "-ffp-contract=on", "-gline-tables-only", "-S", "--cuda-device-only",
                    "-o", out, os.path.join(PKG, "csrc", src)] + extra, check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2))
starts = [i for i, l in enumerate(lines) if re.match(r"_Z\w*%s\w*:" % kern, l)]
for start in starts:
    end = next((i for i in range(start, len(lines)) if lines[i].strip().startswith(".end_amdhsa_kernel")), len(lines))
    cur, depth, cnt = None, 0, Counter()
    for i in range(start, end):
        l = lines[i]
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?").split("/")[-1], int(m.group(2)))
        d = re.search(r"Depth=(\d+)", l)
        if d and "Loop" in l:
            depth = int(d.group(1))
        elif l.startswith(".LBB") and "Depth" not in l:
            depth = 0
        if "\tscratch_" in l:
            cnt[(cur, depth, "store" if "store" in l else "load")] += 1
    print(lines[start].rstrip(":"), "scratch instructions:", sum(cnt.values()))
    for k, v in sorted(cnt.items(), key=lambda kv: (kv[0][0] or ("", 0))):
        print("   ", k, v)
