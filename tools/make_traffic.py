#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the PMC summary of tools/prof_bench.sh (gpurun_out/benchprof/pmc_counters.json).

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (counters are KB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B,
so the read side is doubled -- MI355X_MICROARCH.md, HBM section), summed over the launches of a kernel and divided by
their number.  For the eigenvalue kernels the file also records the matrices per launch, the bytes per matrix and the
sha256 of the kernel's source files (bench.kernel_source_sha): bench.py quotes the traffic only while that hash matches
the library it runs, so a stale measurement can never be reported for a changed kernel.
    python tools/make_traffic.py gpurun_out/benchprof/pmc_counters.json r03 [nmat_per_launch]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    src, tag = sys.argv[1], sys.argv[2]
    nmat = int(sys.argv[3]) if len(sys.argv) > 3 else 524288
    # launches of the n = 64 solver in one bench.py run, "count x matrices" each: the timed batch (1 warm-up + 3 steps + 4
    # event-timed launches) and, since round 5, the quarter-size Gaussian set (1 + 3).  Bytes per matrix = bytes of ALL
    # launches / matrices of ALL launches; the per-launch figure quoted for the timed batch is that x its matrix count.
    plan = sys.argv[4] if len(sys.argv) > 4 else "8x%d,4x131072" % nmat
    plan = [tuple(int(v) for v in part.split("x")) for part in plan.split(",")]
    plan_launches = sum(c for c, _ in plan)
    plan_matrices = sum(c * m for c, m in plan)
    import bench
    pmc = json.load(open(src))
    out = {"method": __doc__.split("\n\n")[1].replace("\n", " "), "source_pmc": src, "kernels": {}}
    algo = {"eig_real64_kernel": (64 * 64 * 8 + 16 * 64 + 4) * nmat}
    for name, c in sorted(pmc.items()):
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        n = max(1, int(c.get("launches", 1)))
        fetch = c["FETCH_SIZE"] * 1024.0 / n
        write = c["WRITE_SIZE"] * 1024.0 / n
        short = name.replace("egs::", "")
        ent = {"launches": n, "fetch_bytes_raw": fetch, "write_bytes": write, "fetch_correction": 2.0,
               "hbm_bytes_per_launch": 2.0 * fetch + write}
        base = short.split("<")[0]
        if base in bench.KERNEL_SOURCES:
            ent["source_sha"] = bench.kernel_source_sha(base)
        if base in ("eig_real64_kernel", "hqr_tail_kernel"):
            if n != plan_launches:
                raise SystemExit("%s: %d launches counted, the launch plan %s has %d" % (short, n, plan, plan_launches))
            per_matrix = ent["hbm_bytes_per_launch"] * n / plan_matrices
            ent["launch_plan"] = ["%d x %d matrices" % (c, m) for c, m in plan]
            ent["hbm_bytes_per_launch_all_sizes_averaged"] = ent["hbm_bytes_per_launch"]
            ent["matrices_per_launch"] = nmat
            ent["hbm_bytes_per_matrix"] = per_matrix
            ent["hbm_bytes_per_launch"] = per_matrix * nmat          # the timed batch's launches
        if base in algo:
            ent["algorithmic_bytes_per_launch"] = algo[base]
            ent["traffic_over_algorithmic"] = ent["hbm_bytes_per_launch"] / algo[base]
        out["kernels"][short] = ent
    # the n = 64 solver is two kernels per step (bench.N64_KERNELS): their sum is what bench.py's roofline object quotes
    parts = [p.strip() for p in bench.N64_KERNELS.split("+")]
    if all(p in out["kernels"] for p in parts):
        tot = sum(out["kernels"][p]["hbm_bytes_per_launch"] for p in parts)
        a = algo["eig_real64_kernel"]
        out["kernels"][bench.N64_KERNELS] = {
            "launches": out["kernels"][parts[0]]["launches"], "hbm_bytes_per_launch": tot, "matrices_per_launch": nmat,
            "hbm_bytes_per_matrix": tot / nmat, "algorithmic_bytes_per_launch": a, "traffic_over_algorithmic": tot / a,
            "source_sha": bench.kernel_source_sha(bench.N64_KERNELS),
            "note": "includes the hand-off records between the two phases (rows r <= c + 3 of a block of order <= 44, "
                    "written by the first kernel and read by the second): a deliberate trade of ~1 % of HBM time for "
                    "twice the occupancy in the second half of the iteration"}
    dst = os.path.join(ROOT, "profiles", "%s_traffic.json" % tag)
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst, "with", len(out["kernels"]), "kernels")


if __name__ == "__main__":
    main()
