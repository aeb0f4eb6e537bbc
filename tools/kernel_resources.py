#!/usr/bin/env python3
"""Register / spill / scratch metadata of every eigenvalue kernel, from `hipcc -S` (no GPU needed), plus where the
scratch instructions sit: their count per loop depth and whether any lies in a loop that also contains a barrier or an
LDS instruction (the condition the round-2 review asked to be zero).  Writes profiles/<tag>_kernel_resources.json.
    python tools/kernel_resources.py r03"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "task-level-insights-from-eigenvalues-across-sequence-models_amd")
FILES = ["eig_real_lds.hip", "eig_real_reg.hip", "eig_cplx_lds.hip", "eig_cplx.hip", "eig_zlarge.hip"]


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        return name


def analyse(asm):
    out = {}
    # split into kernels by the .amdhsa_kernel marker ... end of function
    funcs = re.split(r"\n(?=_Z[\w]+:\s)", asm)
    meta = {}
    for m in re.finditer(r"- \.agpr_count:.*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+)"
                         r".*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", asm, re.S):
        meta[m.group(1)] = dict(scratch_bytes=int(m.group(2)), sgpr=int(m.group(3)), sgpr_spills=int(m.group(4)),
                                vgpr=int(m.group(5)), vgpr_spills=int(m.group(6)))
    for f in funcs:
        head = f.split(":", 1)[0].strip()
        if head not in meta:
            continue
        depth, by_depth = 0, {}
        loops = {}          # loop header label -> [has_barrier_or_lds, scratch_count]
        cur_loop = None
        for line in f.split("\n"):
            m = re.search(r"Depth=(\d+)", line)
            if line.startswith(".LBB") or "Loop Header" in line or "in Loop" in line:
                if m:
                    depth = int(m.group(1))
                elif line.startswith(".LBB"):
                    depth = 0
                h = re.search(r"Header=(\w+)", line)
                cur_loop = h.group(1) if h else (line.split(":")[0] if "Loop Header" in line else (cur_loop if depth else None))
            ins = line.strip().split(" ")[0] if line.startswith("\t") else ""
            if depth and cur_loop:
                ent = loops.setdefault(cur_loop, [False, 0])
                if ins.startswith("s_barrier") or ins.startswith("ds_"):
                    ent[0] = True
                if ins.startswith("scratch_"):
                    ent[1] += 1
            if ins.startswith("scratch_"):
                by_depth[depth] = by_depth.get(depth, 0) + 1
        d = meta[head]
        d["scratch_instructions_by_loop_depth"] = {str(k): v for k, v in sorted(by_depth.items())}
        # blocks whose innermost enclosing loop issues LDS instructions or barriers (this includes a kernel's outer
        # per-matrix / per-window loops; the hot chase / sub-pass loops are the ones at the largest depths)
        d["scratch_instructions_in_loops_that_touch_lds"] = sum(c for b, c in loops.values() if b)
        d["max_loop_depth"] = max([0] + [int(x) for x in re.findall(r"Depth=(\d+)", f)])
        m2 = re.search(r"; codeLenInByte = (\d+)", f)
        if m2:
            d["code_bytes"] = int(m2.group(1))
        out[demangle(head).split("(")[0].replace("void ", "")] = d
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for f in FILES:
            s = os.path.join(tmp, f + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                            "-I" + os.path.join(PKG, "csrc"), "-ffp-contract=on", "-S", "--cuda-device-only", "-o", s,
                            os.path.join(PKG, "csrc", f)], check=True, capture_output=True)
            res[f] = analyse(open(s).read())
    dst = os.path.join(ROOT, "profiles", "%s_kernel_resources.json" % tag)
    json.dump({"method": "hipcc -O3 --offload-arch=gfx950 -S metadata; scratch_* instructions located by the loop-depth "
                         "comments of the assembly", "kernels": res}, open(dst, "w"), indent=1)
    for f, ks in res.items():
        for k, d in ks.items():
            print("%-60s vgpr %3d spills v%3d s%3d scratch %4d B  by depth %s  in lds/barrier loops: %d" % (
                k[:60], d["vgpr"], d["vgpr_spills"], d["sgpr_spills"], d["scratch_bytes"],
                d["scratch_instructions_by_loop_depth"], d["scratch_instructions_in_loops_that_touch_lds"]))


if __name__ == "__main__":
    main()
