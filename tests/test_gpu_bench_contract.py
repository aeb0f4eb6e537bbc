"""The driver's contract with bench.py, at a reduced batch: one JSON line on stdout with the metric, the whole-job value,
the `roofline` object of the dominant kernel (measured live with HIP events on the launch stream) and the
`cpu_baseline` object -- and the values in it are consistent with each other."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_schema_and_consistency():
    nmat = 16384
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--nmat", str(nmat), "--cpu-seconds", "2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and "workload" in d["config"]
    assert d["unconverged"] == 0 and d["max_abs_dlambda_vs_lapack"] < 1e-5           # north_star's tolerance
    # value = matrices per second of the timed steps
    assert abs(d["value"] - nmat / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["kernel"] == "eig_real64_kernel + hqr_tail_kernel" and r["bound"] in ("hbm", "mfma") and r["unit"] == "TFLOP/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = algorithmic flops per launch / measured launch time: (40/3) n^3 per matrix
    want = (40.0 / 3.0) * 64 ** 3 * nmat / (r["avg_launch_ms"] * 1e-3) / 1e12
    assert abs(r["achieved"] - want) < 1e-9 * want
    # the kernel's launch time cannot exceed the step it is part of (and is most of it)
    assert 0.5 * d["ms_per_step"] < r["avg_launch_ms"] <= 1.05 * d["ms_per_step"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


def test_bench_two_ranks_on_one_gpu():
    """`bench.py --gpus 2` started bare: two fresh ranks (both on this box's one GPU, collectives staged through gloo)
    run the hot path on their own matrices and all-gather the spectra; the line reports the whole job."""
    nmat = 8192
    env = dict(os.environ, EGS_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--nmat", str(nmat), "--no-cpu"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world"] == 2 and d["dist_backend"] == "gloo" and d["scaling"] == "weak"
    assert d["unconverged"] == 0 and d["max_abs_dlambda_vs_lapack"] < 1e-5
    assert abs(d["value"] - 2 * nmat / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
