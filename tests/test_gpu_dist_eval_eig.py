"""The sharded ``eval_eig`` path end to end on the GPU box: two fresh ranks (both on the one GPU, gloo with
host-staged collectives) started like ``torchrun launch.py`` would start them -- the caller never touches
torch.distributed (launch.py:176, :183), eval_eig bootstraps the group from the environment -- against the
single-process result of the same call.  Covers SURVEY.md 8(e): contiguous batch slabs, ONE all_gather of the
spectra (or the histogram-only all_reduce), rank-0-only artefacts, and the train-mode init pass (quirk Q8) with live
dropout, which must not depend on the world size."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, tmp_path, tag, attention_fn, dropout, exchange=None, backend="gloo"):
    sys.path.insert(0, HERE)
    import _dist_eval_worker as W
    torch.manual_seed(7)
    ckpt = str(tmp_path / "ckpt.pth")
    if not os.path.exists(ckpt):
        torch.save(W.ToyTransformer({"num_layers": W.NL, "dropout": dropout}).state_dict(), ckpt)
    port = _free_port()
    procs, dirs = [], []
    for r in range(world):
        cwd = tmp_path / f"{tag}_rank{r}"
        os.mkdir(cwd)
        dirs.append(cwd)
        # gloo: both ranks on device 0, collectives staged through the host; nccl (= RCCL): one device per rank
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if backend == "gloo" else str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), EGS_DIST_BACKEND=backend)
        spec = {"cwd": str(cwd), "ckpt": ckpt, "attention_fn": attention_fn, "dropout": dropout, "exchange": exchange}
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_eval_worker.py"), json.dumps(spec)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600)[0])
    except subprocess.TimeoutExpired:
        # a hang of the sharded path is a failure with evidence, never a skip: stop every rank and show what they printed
        for q in procs:
            q.kill()
        pytest.fail("ranks did not finish within 600 s:\n" + "\n".join(q.communicate()[0][-3000:] for q in procs))
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return dirs, [dict(np.load(d / "out.npz")) for d in dirs]


@pytest.mark.parametrize("attention_fn,dropout", [("lin-attention", 0.0), ("sm-attention", 0.25)])
def test_two_ranks_equal_single_process(tmp_path, attention_fn, dropout):
    _, (single,) = _run(1, tmp_path, "w1", attention_fn, dropout)
    dirs, ranks = _run(2, tmp_path, "w2", attention_fn, dropout)
    for r in range(2):
        for k, v in single.items():
            if k.startswith("pct"):
                np.testing.assert_array_equal(ranks[r][k], v, err_msg=f"rank {r} {k}")
            else:
                np.testing.assert_allclose(ranks[r][k], v, rtol=1e-12, atol=0, err_msg=f"rank {r} {k}")
    # one writer: the report and the artefact directory exist for rank 0 only
    assert os.path.exists(dirs[0] / "percentage_file.txt") and not os.path.exists(dirs[1] / "percentage_file.txt")
    assert any(e.is_dir() for e in os.scandir(dirs[0])) and not any(e.is_dir() for e in os.scandir(dirs[1]))
    if dropout > 0:       # the train-mode pass really had live dropout: eig_init differs from what eval mode would give
        assert single["eig_init"].shape == single["eig"].shape


def test_histogram_only_exchange(tmp_path):
    """exchange: histogram -- spectra stay on their GPU (each rank returns its batch rows), the (bins, B, H, layers)
    percentages are completed by one all_reduce and are identical to the gathered path's."""
    _, (single,) = _run(1, tmp_path, "w1", "lin-attention", 0.0)
    _, ranks = _run(2, tmp_path, "w2h", "lin-attention", 0.0, exchange="histogram")
    lo = 0
    for r in range(2):
        for k in ("pct", "pct_init", "pct_phase", "pct_phase_init"):
            np.testing.assert_array_equal(ranks[r][k], single[k])
        nloc = ranks[r]["eig"].shape[0]
        np.testing.assert_allclose(ranks[r]["eig"], single["eig"][lo:lo + nloc], rtol=1e-12)
        np.testing.assert_allclose(ranks[r]["eig_init"], single["eig_init"][lo:lo + nloc], rtol=1e-12)
        lo += nloc
    assert lo == single["eig"].shape[0]
    # the saved spectra are rank 0's rows only: their file names and used_config.yaml say so (the percentages are complete)
    art = [e.path for e in os.scandir(tmp_path / "w2h_rank0") if e.is_dir()]
    assert len(art) == 1
    names = sorted(os.listdir(art[0]))
    nloc0 = ranks[0]["eig"].shape[0]
    assert "eig_rank0_rows0-%d.npy" % nloc0 in names and "eig_init_rank0_rows0-%d.npy" % nloc0 in names
    assert "eig.npy" not in names and "percentage.npy" in names
    import yaml
    cfg = yaml.safe_load(open(os.path.join(art[0], "used_config.yaml")))
    assert cfg["eigspec_sharding"] == {"exchange": "histogram", "world_size": 2, "rank": 0, "batch_rows": [0, nloc0],
                                       "batch_size": single["eig"].shape[0]}


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="two RCCL ranks need two GPUs (RCCL rejects duplicate devices); "
                                                          "the same comparison runs over gloo above")
@pytest.mark.parametrize("exchange", [None, "histogram"])
def test_two_rccl_ranks_equal_single_process(tmp_path, exchange):
    """The same comparisons over RCCL (one device per rank) wherever the suite is given two GPUs: the gathered spectra,
    and the histogram-only exchange (one all_reduce of the percentages, spectra stay on their GPU)."""
    _, (single,) = _run(1, tmp_path, "w1", "lin-attention", 0.0)
    _, ranks = _run(2, tmp_path, "w2r", "lin-attention", 0.0, exchange=exchange, backend="nccl")
    lo = 0
    for r in range(2):
        for k in ("pct", "pct_init", "pct_phase", "pct_phase_init"):
            np.testing.assert_array_equal(ranks[r][k], single[k])
        if exchange == "histogram":
            nloc = ranks[r]["eig"].shape[0]
            np.testing.assert_allclose(ranks[r]["eig"], single["eig"][lo:lo + nloc], rtol=1e-12)
            lo += nloc
        else:
            np.testing.assert_allclose(ranks[r]["eig"], single["eig"], rtol=1e-12)
            np.testing.assert_allclose(ranks[r]["eig_init"], single["eig_init"], rtol=1e-12)
