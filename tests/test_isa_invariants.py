"""Static checks on the gfx950 assembly of the n = 64 eigenvalue kernels (hipcc cross-compiles without a GPU).

The steady-state loop of the two-bulge sweep issues its LDS reads by hand and leaves some of them in flight beside the
reflector arithmetic (csrc/eig_real_lds.hip, lds_issue3x2 / lds_issue3x3): the compiler does not know that those
registers are not valid yet, so nothing may touch them before the `s_waitcnt lgkmcnt(0)` that lands them -- in
particular no spill, copy or scheduling decision of a future compiler version.  This test re-derives that from the
emitted code, and checks that the inner loops of the sweep contain no scratch traffic."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "task-level-insights-from-eigenvalues-across-sequence-models_amd")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "eig_real_lds.s"
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(PKG, "csrc"), "-ffp-contract=on", "-S", "--cuda-device-only", "-o", str(out),
           os.path.join(PKG, "csrc", "eig_real_lds.hip")]
    subprocess.run(cmd, check=True, capture_output=True)
    return open(out).read()


def _regs(tok):
    """v12 -> {12};  v[12:13] -> {12, 13}"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def _vregs_of(line):
    regs = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", line):
        regs |= _regs(tok)
    return regs


def _kernel(asm, name):
    start = asm.index("\n" + name + ":")
    end = asm.index(".end_amdhsa_kernel", start)
    return asm[start:end].split("\n")


@pytest.mark.parametrize("kernel", ["_ZN3egs17eig_real64_kernelEPKdlilPdPiS2_iPyS2_", "_ZN3egs15hqr_real_kernelILb1EEEvPKdliPdPiiiPy",
                                    "_ZN3egs15hqr_tail_kernelEPKdlPdPiPy"])
def test_in_flight_lds_operands_are_not_touched(asm, kernel):
    lines = [ln.strip() for ln in _kernel(asm, kernel)]
    batches = 0
    i = 0
    while i < len(lines):
        if not lines[i].startswith("ds_read_b64"):
            i += 1
            continue
        dests = []
        while i < len(lines) and lines[i].startswith("ds_read_b64"):
            dests.append(_regs(lines[i].split()[1].rstrip(",")))
            i += 1
        m = re.match(r"s_waitcnt lgkmcnt\((\d+)\)", lines[i]) if i < len(lines) else None
        # hand-issued row-phase batch: the bulge column (first three reads) is waited for, the rest flies
        if len(dests) not in (6, 9) or m is None or int(m.group(1)) != len(dests) - 3:
            continue
        k = int(m.group(1))
        flying = set().union(*dests[-k:])
        i += 1
        landed = False
        while i < len(lines):
            ln = lines[i]
            if ln.startswith("s_waitcnt") and "lgkmcnt(0)" in ln:
                landed = True
                break
            if ln and not ln.startswith(";") and not ln.startswith("."):
                assert not (_vregs_of(ln) & flying), "in-flight LDS operand touched before it landed: " + ln
                assert not ln.startswith("scratch_"), ln          # (the one branch in between guards the negligible-column fix-up)
            i += 1
        assert landed
        batches += 1
    assert batches >= 8                                # 2 layouts x (4 steady segments + special iterations)


def test_sweep_loops_have_no_scratch_traffic(asm):
    """No spill or reload inside any loop nested deeper than the per-matrix and the per-eigenvalue loops."""
    for kernel in ("_ZN3egs17eig_real64_kernelEPKdlilPdPiS2_iPyS2_", "_ZN3egs15hqr_real_kernelILb1EEEvPKdliPdPiiiPy",
                   "_ZN3egs15hqr_tail_kernelEPKdlPdPiPy"):
        depth = 0
        for ln in _kernel(asm, kernel):
            m = re.search(r"Depth=(\d+)", ln)
            if ln.startswith(".LBB") or "Loop Header" in ln or "in Loop" in ln:
                depth = int(m.group(1)) if m else (0 if ln.startswith(".LBB") else depth)
            if ln.strip().startswith("scratch_"):
                assert depth <= 2, (kernel, depth, ln)
