#!/usr/bin/env python3
"""Benchmark of the eigenvalue-analysis hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): S5-sized dense spectra -- d_state = 64, seq_len = 1024, batch = 512, i.e.
524,288 real fp64 64x64 state-transition matrices PER GPU (weak scaling: every rank owns that many; the matrices
are independent, so there is no data-path collective, only the single all-gather of the spectra at N > 1).
One "step" = one pass of the hot path over the whole resident batch: Householder-Hessenberg + Francis QR
(egs_eig_general_batched through the C ABI) [+ all_gather of the spectra when N > 1].

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (metric = eigen-spectra/sec, plus max |dlambda| vs the oracle after matching).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "task-level-insights-from-eigenvalues-across-sequence-models_amd"

FP64_PEAK_TFLOPS = 78.6        # MI355X fp64 vector = matrix peak (datasheet)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
CPU_POOL_PROCS = 16            # process pool of the multi-core CPU figure: the host-core share of ONE GPU on the bench box


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run (127.0.0.1 rendezvous) as
    child processes and relay their output.  This parent never initialises the GPU."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, env=env)
    raise SystemExit(proc.returncode)


def gen_known_spectrum(cnt, n, gen, device):
    """Real matrices Q (D + N) Q^T with S5-style eigenvalues exp((-0.5 + i w) dt) in conjugate pairs (SURVEY.md 8(d)
    C2): D = 2x2 rotation-scaling blocks, N strictly block-upper (non-normal part), Q = three Householder reflectors.
    Returns (a [cnt,n,n] float64, lam [cnt,n] complex128)."""
    npair = n // 2
    om = torch.rand((cnt, npair), generator=gen, device=device, dtype=torch.float64) * 40.0
    lo, hi = np.log(1e-3), np.log(1e-1)
    dt = torch.exp(torch.rand((cnt, npair), generator=gen, device=device, dtype=torch.float64) * (hi - lo) + lo)
    lam = torch.exp(torch.complex(-0.5 * dt, om * dt))
    a = torch.randn((cnt, n, n), generator=gen, device=device, dtype=torch.float64).triu_(2) * (0.1 / np.sqrt(n))
    idx = torch.arange(npair, device=device) * 2
    a[:, idx, idx] = lam.real
    a[:, idx + 1, idx + 1] = lam.real
    a[:, idx, idx + 1] = lam.imag
    a[:, idx + 1, idx] = -lam.imag
    for _ in range(3):
        u = torch.randn((cnt, n, 1), generator=gen, device=device, dtype=torch.float64)
        u = u / torch.linalg.vector_norm(u, dim=1, keepdim=True)
        a = a - 2.0 * u @ (u.transpose(1, 2) @ a)
        a = a - 2.0 * (a @ u) @ u.transpose(1, 2)
    return a, torch.cat([lam, lam.conj()], dim=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nmat", type=int, default=512 * 1024, help="matrices per GPU (batch 512 x seq_len 1024)")
    ap.add_argument("--n", type=int, default=64, help="d_state")
    ap.add_argument("--matrices", choices=["s5", "randn"], default="s5")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU-baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-procs", default=str(CPU_POOL_PROCS),
                    help="processes of the pooled CPU figure: a number, or 'all' for every host core (default: one GPU's "
                         "share of the bench box's cores)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --nmat matrices PER rank (default, the driver's run); strong: --nmat matrices in TOTAL, cut "
                         "into contiguous per-rank slabs (dist.shard_range), plus config C4's 4096-sequence batch cut the "
                         "same way as a secondary line")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="no hot path and no GPU: only the launcher + process-group plumbing of --gpus N (gloo), for the "
                         "CPU test suite; prints a line with value = null")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Started bare (`python bench.py --gpus N`): become the launcher.  The N ranks are FRESH child processes started
        # before this process has made any GPU call (it never makes one: it only relays rank 0's JSON line and the exit
        # code), one rank per GPU over RCCL -- exactly the command the driver uses for N > 1.
        return launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    # multi-core CPU figure: a separate process with a forked pool of CPU_POOL_PROCS workers (NOT all of
    # the host's cores: `cores` and `host_cpu_count` are both in the object), started BEFORE this process initialises
    # the GPU
    cpu_multi = None
    if world == 1 and not args.no_cpu and args.n == 64:
        import subprocess
        try:
            ncpu = os.cpu_count() or 1
            procs = ncpu if args.cpu_procs == "all" else max(1, min(int(args.cpu_procs), ncpu))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--procs", str(procs),
                                  "--seconds", str(min(args.cpu_seconds, 10.0))], capture_output=True, text=True,
                                 timeout=120)
            cpu_multi = json.loads(out.stdout.strip().splitlines()[-1])
            # say what share of the host this is, and what a linear extrapolation to every core would be (NOT measured:
            # the bench box gives one GPU's job 16 of its cores; `--cpu-procs all` measures it where that is allowed)
            cpu_multi["cores_of_host"] = "%d of %d" % (procs, ncpu)
            cpu_multi["sample"] = "%s of the host's %d cores; %s" % (procs, ncpu, cpu_multi.get("sample", ""))
            cpu_multi["all_cores_extrapolated"] = {"value": cpu_multi["value"] * ncpu / procs, "unit": "spectra/s",
                                                   "cores": ncpu, "how": "linear in cores from the measured pool; not a measurement"}
        except Exception as exc:            # the baseline is a courtesy figure; never fail the bench on it
            cpu_multi = {"error": repr(exc)}
    if args.rehearse_launch:
        return rehearse_launch(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the eigenvalue-analysis path has no CPU fallback")
    # one rank per GPU; EGS_DIST_BACKEND=gloo (host-staged collectives, several ranks may share a GPU) exists so that the
    # N > 1 path can be rehearsed on a one-GPU box -- the driver's runs use the default, "nccl" (= RCCL over xGMI)
    backend = os.environ.get("EGS_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    pkg = importlib.import_module(PKG)
    ops, egs_dist = pkg.ops, pkg.dist
    import oracle as O

    n = args.n
    # weak (default, the driver's run): every rank owns --nmat matrices.  strong: --nmat is the WHOLE batch, cut into
    # contiguous slabs exactly as eval_eig cuts an analysis batch (dist.shard_range); the gather pads to the common slab.
    strong = args.scaling == "strong"
    nmat_total = args.nmat if strong else args.nmat * world
    if strong:
        lo, hi = egs_dist.shard_range(args.nmat, rank, world)
        nmat = hi - lo
    else:
        nmat = args.nmat
    gen = torch.Generator(device=device)
    gen.manual_seed(64 + rank)
    a = torch.empty((nmat, n, n), dtype=torch.float64, device=device)
    lam_sample = None
    gchunk = 16384
    for m0 in range(0, nmat, gchunk):
        cnt = min(gchunk, nmat - m0)
        if args.matrices == "s5":
            blk, lam = gen_known_spectrum(cnt, n, gen, device)
            if m0 == 0:
                lam_sample = lam[:512].cpu().numpy()
        else:
            blk = torch.randn((cnt, n, n), generator=gen, device=device, dtype=torch.float64) / np.sqrt(n)
        a[m0:m0 + cnt] = blk
        del blk
    ws_bytes = pkg._capi.lib().egs_eig_workspace_bytes(pkg._capi.EGS_F64, nmat, n)
    workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    torch.cuda.synchronize()

    # One step = the solver on this rank's matrices + the ONE collective of the path (RCCL all-gather of the spectra).
    # The collective of step k runs on RCCL's stream while step k + 1's kernels run on the launch stream (what eval_eig
    # does from layer to layer); every collective of the K timed steps completes inside the timed region.
    def run_steps(k):
        pending, w, info = None, None, None
        for _ in range(k):
            w, info = ops.eig_general_batched(a, workspace=workspace)
            if world > 1:
                h = egs_dist.all_gather_slabs_async(w, nmat_total)
                if pending is not None:
                    pending.wait()
                pending = h
        if pending is not None:
            w = pending.wait()
        return w, info

    w, info = run_steps(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w_step, info_step = run_steps(args.steps)
    if w_step is not None:
        w, info = w_step, info_step
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = nmat_total / (elapsed / args.steps)
    devices = device_identities(device, rank, local_rank, world, backend)

    # ---- per-kernel timing (HIP events on the launch stream) for the roofline object ----
    # n = 64: the step is one launch of the fused kernel (eig_real64_kernel: Hessenberg reduction in registers + QR in
    # LDS per wave, whole batch behind a work queue) followed by one of hqr_tail_kernel, which finishes every matrix
    # from the point where its active block fits a 44 x 44 array (twice the occupancy); the events bracket the pair.
    # ops launches on torch's current stream, which is the stream the events are recorded on.  Other orders: the
    # two-kernel path, timed per stage below.
    import ctypes
    lib = pkg._capi.lib()
    fused = (n == 64) and not os.environ.get("EGS_EIG_UNFUSED")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    fused_ms = []
    if fused:
        for rep in range(4):
            ev[0].record()
            ops.eig_general_batched(a, workspace=workspace)
            ev[1].record()
            ev[1].synchronize()
            if rep > 0:
                fused_ms.append(ev[0].elapsed_time(ev[1]))
    # the two stages as kernels of their own (egs_hessenberg_batched / egs_hseqr_batched), 64 Ki matrices per launch
    hq_ms, hs_ms = [], []
    chunk = min(nmat, 65536)
    hbuf = torch.empty((chunk, n, n), dtype=torch.float64, device=device)
    wtmp = torch.empty((chunk, n), dtype=torch.complex128, device=device)
    itmp = torch.empty((chunk,), dtype=torch.int32, device=device)
    qbuf = torch.zeros((32,), dtype=torch.int64, device=device)       # work-queue counter of the QR kernel
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nlaunch = 0
    for rep in range(2):
        for m0 in range(0, nmat, chunk):
            cnt = min(chunk, nmat - m0)
            ev[0].record()
            lib.egs_hessenberg_batched(ctypes.c_void_p(a[m0].data_ptr()), cnt, n, n, n * n,
                                       ctypes.c_void_p(hbuf.data_ptr()), stream)
            ev[1].record()
            lib.egs_hseqr_batched(ctypes.c_void_p(hbuf.data_ptr()), cnt, n, ctypes.c_void_p(wtmp.data_ptr()),
                                  ctypes.c_void_p(itmp.data_ptr()), ctypes.c_void_p(qbuf.data_ptr()), stream)
            ev[2].record()
            ev[2].synchronize()
            if rep == 1 and cnt == chunk:
                hs_ms.append(ev[0].elapsed_time(ev[1]))
                hq_ms.append(ev[1].elapsed_time(ev[2]))
                nlaunch += 1
    hq_avg = float(np.mean(hq_ms))
    hs_avg = float(np.mean(hs_ms))
    # algorithmic work per matrix: (10/3) n^3 Hessenberg + 10 n^3 QR flops; n^2*8 + 16 n bytes
    qr_flops = 10.0 * n ** 3 * chunk
    hs_flops = (10.0 / 3.0) * n ** 3 * chunk
    stages = {"hess_kernel": {"kernel": "hess_reg_kernel" if n == 64 else "hess_real_kernel", "avg_launch_ms": hs_avg,
                              "achieved_tflops": hs_flops / (hs_avg * 1e-3) / 1e12},
              "hqr_real_kernel": {"avg_launch_ms": hq_avg, "achieved_tflops": qr_flops / (hq_avg * 1e-3) / 1e12,
                                  "frac": qr_flops / (hq_avg * 1e-3) / 1e12 / FP64_PEAK_TFLOPS},
              "matrices_per_launch": chunk, "launches_timed": nlaunch}
    if fused:
        kname, per_launch, k_ms = N64_KERNELS, nmat, float(np.mean(fused_ms))
        k_flops = (40.0 / 3.0) * n ** 3 * per_launch
        note = ("fp64 (vector = matrix peak); algorithmic (40/3) n^3 flops per matrix (Hessenberg (10/3) n^3 + QR 10 n^3), "
                "%d matrices per launch; avg_launch_ms = the two phases of the solver back to back (rocprofv3: the sum of "
                "the two kernels' average durations); `two_kernel_stages` times Hessenberg reduction and QR iteration "
                "as separate kernels" % per_launch)
    else:
        kname, per_launch, k_ms = "hqr_real_kernel", chunk, hq_avg
        k_flops = qr_flops
        note = "fp64 (vector = matrix peak); algorithmic 10 n^3 flops per matrix, %d matrices per launch" % chunk
    achieved_tf = k_flops / (k_ms * 1e-3) / 1e12
    # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE / WRITE_SIZE cannot be read from
    # inside the process; collected with rocprofv3 --pmc on this same command, tools/prof_bench.sh + tools/make_traffic.py).
    # Only the NEWEST traffic file counts, it must list the timed kernel, and it must have been measured on the kernel
    # source this library was built from (sha256 of the kernel's source files, recorded by make_traffic.py): anything
    # else is reported as traffic = null with the reason -- never a silent fall-back to an older round's number.
    traffic, traffic_src, traffic_note = kernel_traffic(kname, per_launch)
    roofline = {
        # the n = 64 solver is fp64 VECTOR arithmetic on an LDS-resident matrix ("bound by fp64 VALU + LDS
        # bandwidth + serial bulge-chase latency"); the matrix cores only add up the register-resident reduction's partial
        # sums.  The peak it is priced against is the fp64 vector peak (= the fp64 matrix peak on MI355X).
        "bound": "valu+lds", "kernel": kname, "achieved": achieved_tf, "peak": FP64_PEAK_TFLOPS,
        "unit": "TFLOP/s", "frac": achieved_tf / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src, "traffic_note": traffic_note,
        "note": note, "avg_launch_ms": k_ms, "launches_timed": len(fused_ms) if fused else nlaunch,
        "two_kernel_stages": stages,
        "hbm_gbs_algorithmic": (n * n * 8 + 16 * n) * per_launch / (k_ms * 1e-3) / 1e9 if fused
        else (n * n * 8 + 16 * n) * chunk / ((hq_avg + hs_avg) * 1e-3) / 1e9,
    }

    # ---- HBM-bound kernels of the same path (north_star: >= 40 % of HBM peak on the scan kernels), timed live ----
    secondary, headline = None, None
    if rank == 0 and world == 1:
        secondary = stream_rooflines(ops, device)
        if not args.no_cpu:
            headline = mqar_headline(ops, device)

    # ---- SURVEY 8(d)'s second C2 set: plain N(0,1)/sqrt(n) matrices (no structure to converge on) at a quarter batch ----
    gaussian = None
    if rank == 0 and world == 1 and n == 64 and args.matrices == "s5":
        gaussian = gaussian_c2(ops, device, min(nmat, 131072), n, workspace)
    # ---- strong scaling's secondary line: config C4 (4096 sequences x 4096 tokens, D = 128) cut over the ranks ----
    c4_strong, c5_strong = None, None
    if strong:
        c4_strong = c4_strong_scaling(ops, egs_dist, device, rank, world, backend)
        c5_strong = c5_strong_scaling(ops, egs_dist, device, rank, world, backend)

    # ---- parity: GPU spectra vs the known spectrum and vs LAPACK on a sample ----
    info_bad = int((info != 0).sum().item())
    w_host = w[:512].cpu().numpy() if rank == 0 else None
    result = None
    if rank == 0:
        a_sample = a[:512].cpu().numpy()
        d_lapack = O.max_abs_dlambda(w_host, O.eigvals_dense(a_sample))
        d_known = O.max_abs_dlambda(w_host, lam_sample) if lam_sample is not None else None
        cpu = None
        if world == 1 and not args.no_cpu:
            cpu = cpu_baseline(a, args.cpu_seconds, O)
        result = {
            "metric": "eigen-spectra/sec (batch x seq_len x d_state matrices) + max |dlambda| vs reference",
            "value": value, "unit": "spectra/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: S5 d_state=64, seq_len=1024, batch=512 -> 524288 dense 64x64 fp64 matrices per GPU "
                                   "(known S5-style spectrum under a non-normal orthogonal similarity)",
                       "d_state": n, "matrices_per_gpu": nmat, "matrices_total": nmat_total, "matrix_kind": args.matrices,
                       "parallelism": "matrix-sharded x%d, one all_gather of spectra" % world},
            "max_abs_dlambda_vs_lapack": d_lapack, "max_abs_dlambda_vs_known": d_known, "unconverged": info_bad,
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_pool": cpu_multi,
            "secondary_rooflines": secondary, "north_star_mqar_batch": headline,
            # what the process group actually looked like (so a SCALE record shows that RCCL saw N ranks on N devices)
            "world": world, "dist_backend": backend if world > 1 else None,
            "visible_devices": torch.cuda.device_count(), "device_name": torch.cuda.get_device_name(device),
            # one entry per rank (gathered over the process group): which physical device each rank drove, and the size
            # of the communicator the collective ran on -- a SCALE record with N distinct uuids proves N GPUs took part
            "ranks": devices, "comm_size": (dist.get_world_size() if world > 1 else 1),
            "distinct_devices": len({d.get("uuid") or (d.get("host"), d.get("index")) for d in devices}),
            "secondary_c2_gaussian": gaussian, "secondary_c4_strong": c4_strong, "secondary_c5_strong": c5_strong,
        }
        if cpu is not None:
            result["speedup_vs_cpu_1core"] = value / cpu["value"]
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def device_identities(device, rank, local_rank, world, backend):
    """[{rank, local_rank, index, uuid, name, host}] for every rank, gathered over the process group (all_gather_object)."""
    import socket
    props = torch.cuda.get_device_properties(device)
    me = {"rank": rank, "local_rank": local_rank, "index": device.index, "name": props.name, "host": socket.gethostname(),
          "uuid": str(getattr(props, "uuid", "")) or None,
          "pci_bus_id": getattr(props, "pci_bus_id", None)}
    if world == 1:
        return [me]
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, me)
    return out


def gaussian_c2(ops, device, cnt, n, workspace):
    """The solver on N(0,1)/sqrt(n) matrices (second set), HIP events on the launch stream, median of 3."""
    import oracle as O
    g = torch.Generator(device=device)
    g.manual_seed(65)
    a = torch.randn((cnt, n, n), generator=g, device=device, dtype=torch.float64) / np.sqrt(n)
    w, info = ops.eig_general_batched(a, workspace=workspace)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    for r in range(3):
        w, info = ops.eig_general_batched(a, workspace=workspace)
        ev[r + 1].record()
    torch.cuda.synchronize()
    ms = float(np.median([ev[r].elapsed_time(ev[r + 1]) for r in range(3)]))
    tf = (40.0 / 3.0) * n ** 3 * cnt / (ms * 1e-3) / 1e12
    d = float(O.max_abs_dlambda(w[:256].cpu().numpy(), O.eigvals_dense(a[:256].cpu().numpy())))
    return {"workload": "C2, second set: %d dense N(0,1)/sqrt(n) 64x64 fp64 matrices" % cnt, "kernel": N64_KERNELS,
            "bound": "valu+lds", "ms": ms, "spectra_per_s": cnt / (ms * 1e-3), "achieved": tf, "peak": FP64_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS, "unconverged": int((info != 0).sum().item()),
            "max_abs_dlambda_vs_lapack": d}


def c4_strong_scaling(ops, egs_dist, device, rank, world, backend):
    """Config C4 as a strong-scaling line: linear-attention eta of 4096 sequences x 4096 tokens (D = 128, d_qk = 128,
    H = 1), the batch cut over the ranks, ONE all-gather of eta; max over ranks of the step time."""
    B, L, D, dqk = 4096, 4096, 128, 128
    lo, hi = egs_dist.shard_range(B, rank, world)
    g = torch.Generator(device=device)
    g.manual_seed(4096 + rank)
    x = torch.randn((hi - lo, L, D), generator=g, device=device)
    g.manual_seed(4096)
    w = (torch.rand((2 * dqk, D), generator=g, device=device) * 2 - 1) / np.sqrt(D)
    b = (torch.rand((2 * dqk,), generator=g, device=device) * 2 - 1) / np.sqrt(D)

    def step():
        eta = ops.linattn_eta(x, w, b, dqk, 1)
        return egs_dist.all_gather_slabs(eta, B) if world > 1 else eta
    step()
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(3):
        full = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = (time.perf_counter() - t0) / 3
    if world > 1:
        tmax = torch.tensor([el], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())
    return {"workload": "C4: linear attention eta, B=4096, L=4096, D=128, d_qk=128, H=1, batch cut over %d rank(s)" % world,
            "scaling": "strong", "sequences_per_rank": hi - lo, "ms_per_step": el * 1e3, "eta_values_per_s": B * (L - 1) / el,
            "sequences_per_s": B / el, "gathered_shape": list(full.shape)}


def c5_strong_scaling(ops, egs_dist, device, rank, world, backend):
    """Config C5 as a strong-scaling line ("S4 HiPPO-init d_state=256 ... roofline fraction at 1/2/4/8 GPUs"): the 768
    feature columns (6 layers x 128) cut over the ranks, fused DPLR build + dense complex solver on each slab, ONE
    all-gather of the spectra; max over ranks.  One workgroup solves one matrix, so with fewer matrices than compute units
    per rank (8 ranks: 96) the step is one matrix's latency: the fraction falls with N by construction."""
    import oracle as O
    n, h = 256, 768
    lam, p, _, _, _ = O.make_dplr_hippo(n)
    ls = np.random.default_rng(256).uniform(np.log(1e-3), np.log(1e-1), h)
    lo, hi = egs_dist.shard_range(h, rank, world)
    dd = lambda z: torch.from_numpy(np.ascontiguousarray(z)).to(device)
    cnt = hi - lo
    pr = (dd(np.tile(lam.real, (cnt, 1))), dd(np.tile(lam.imag, (cnt, 1))), dd(np.tile(p, (cnt, 1))), dd(ls[lo:hi]))

    def step():
        w, info = ops.eig_general_batched(ops.s4_dplr_build(*pr))
        return (egs_dist.all_gather_slabs(w, h) if world > 1 else w), info
    step()
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(3):
        full, info = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = (time.perf_counter() - t0) / 3
    if world > 1:
        tmax = torch.tensor([el], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())
    tf = 4.0 * (40.0 / 3.0) * n ** 3 * h / el / 1e12
    return {"workload": "C5: 768 HiPPO-DPLR complex128 256x256 matrices cut over %d rank(s), build + dense solve + gather" % world,
            "scaling": "strong", "matrices_per_rank": cnt, "ms_per_step": el * 1e3, "spectra_per_s": h / el,
            "achieved": tf, "peak": FP64_PEAK_TFLOPS * world, "unit": "TFLOP/s", "frac": tf / (FP64_PEAK_TFLOPS * world),
            "unconverged_local": int((info != 0).sum().item()), "gathered_shape": list(full.shape)}


def rehearse_launch(args, world, rank):
    """The N-rank plumbing without the hot path: rendezvous, barrier, the path's ONE collective on a small host tensor."""
    pkg = importlib.import_module(PKG)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    slab = torch.full((3, 4), float(rank), dtype=torch.complex128)
    full = pkg.dist.all_gather_slabs(slab, 3 * world)
    ok = bool(all(float(full[3 * r, 0].real) == r for r in range(world)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "eigen-spectra/sec", "value": None, "n_gpus": world, "world": world,
                          "dist_backend": "gloo" if world > 1 else None, "rehearsal": True, "gather_ok": ok}))


N64_KERNELS = "eig_real64_kernel + hqr_tail_kernel"      # the two phases of the n = 64 solver (one step = one launch of each)
KERNEL_SOURCES = {      # the files a kernel is compiled from; their hash ties a PMC measurement to a build
    N64_KERNELS: ("eig_real_lds.hip", "hess_reg.h", "egs_common.h"),
    "eig_real64_kernel": ("eig_real_lds.hip", "hess_reg.h", "egs_common.h"),
    "hqr_tail_kernel": ("eig_real_lds.hip", "egs_common.h"),
    "hqr_real_kernel": ("eig_real_lds.hip", "egs_common.h"),
    "zhess_blocked_kernel": ("eig_zlarge.hip", "zq_common.h", "egs_common.h"),
    "zmsqr_kernel": ("eig_zlarge.hip", "zq_common.h", "egs_common.h"),
}


def kernel_source_sha(kname):
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES.get(kname, ()):
        with open(os.path.join(ROOT, PKG, "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def kernel_traffic(kname, per_launch):
    """(bytes per launch, source file, note) from the newest profiles/rNN_traffic.json."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True)
    if not files:
        return None, None, "no profiles/r*_traffic.json"
    src = "profiles/" + os.path.basename(files[0])
    try:
        tr = json.load(open(files[0]))
        ent = tr["kernels"][kname]
    except Exception:
        return None, src, "newest traffic file does not list %s" % kname
    want, have = kernel_source_sha(kname), ent.get("source_sha")
    if have != want:
        return None, src, "stale: measured on kernel source %s, this build is %s" % (have, want)
    return ent["hbm_bytes_per_matrix"] * per_launch, src, "PMC FETCH_SIZE x2 + WRITE_SIZE, kernel source %s" % want


def kernel_traffic_launch(entry, base):
    """(HBM-side bytes per launch, source, note) of a kernel listed by its full name in the newest traffic file."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True)
    if not files:
        return None, None, "no profiles/r*_traffic.json"
    src = "profiles/" + os.path.basename(files[0])
    try:
        ent = json.load(open(files[0]))["kernels"][entry]
    except Exception:
        return None, src, "newest traffic file does not list %s" % entry
    want, have = kernel_source_sha(base), ent.get("source_sha")
    if have != want:
        return None, src, "stale: measured on kernel source %s, this build is %s" % (have, want)
    return ent["hbm_bytes_per_launch"], src, "PMC FETCH_SIZE x2 + WRITE_SIZE (HBM + Infinity Cache side), kernel source %s" % want


def stream_rooflines(ops, device):
    """Achieved algorithmic HBM GB/s of K2 (Mamba-2 lambda_t, config C3) and achieved fp32 TFLOP/s of K3's MFMA
    projection (linear attention, one GPU's shard of config C4), HIP events on the launch stream, median of 5."""
    def med(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for r in range(reps):
            fn()
            ev[r + 1].record()
        torch.cuda.synchronize()
        return float(np.median([ev[r].elapsed_time(ev[r + 1]) for r in range(reps)]))

    g = torch.Generator(device=device)
    g.manual_seed(16)
    out = []
    B, L, D, nh = 1024, 2048, 128, 4
    x = torch.randn((B, L, D), generator=g, device=device)
    w = (torch.rand((nh, D), generator=g, device=device) * 2 - 1) / np.sqrt(D)
    dtb = torch.randn(nh, generator=g, device=device)
    alog = torch.log(torch.rand(nh, generator=g, device=device) * 15 + 1)
    ms = med(lambda: ops.mamba2_lambda(x, w, None, dtb, alog))
    gbs = B * L * (4 * D + 4 * nh) / ms / 1e6
    out.append({"kernel": "mamba2_lambda_fast_kernel", "workload": "C3: Mamba-2 lambda_t, B=1024, L=2048, D=128, 4 heads",
                "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "ms": ms, "algorithmic_bytes": B * L * (4 * D + 4 * nh)})
    del x
    B, L, D, dqk = 512, 4096, 128, 128
    x = torch.randn((B, L, D), generator=g, device=device)
    w = (torch.rand((2 * dqk, D), generator=g, device=device) * 2 - 1) / np.sqrt(D)
    b = (torch.rand((2 * dqk,), generator=g, device=device) * 2 - 1) / np.sqrt(D)
    ms = med(lambda: ops.linattn_eta(x, w, b, dqk, 1), reps=3)
    tf = B * L * 2.0 * D * 2 * dqk / ms / 1e9
    out.append({"kernel": "linattn_fused_kernel<1> + eta_finalize_kernel",
                "workload": "C4 / 8 GPUs: linear attention eta, B=512, L=4096, D=128, d_qk=128, H=1",
                "bound": "mfma", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3, "ms": ms,
                "note": "fp32 MFMA (= fp32 vector peak); projection, elu+1, fp64 prefix and normaliser fused: q, k never reach HBM "
                        "(PMC fetch + write = 1.00x the algorithmic bytes, profiles/r02_traffic.json)",
                "hbm_gbs_algorithmic": B * L * (4 * D + 8) / ms / 1e6})
    del x
    # K4: softmax eta on the 4096-sequence MQAR batch (L = 512, hidden 128, d_qk = 64): ONE pass of causal score tiles
    # (online maximum; rounds 1-3 made two passes and counted two)
    B, L, D, dqk = 4096, 512, 128, 64
    x = torch.randn((B, L, D), generator=g, device=device)
    w = (torch.rand((2 * dqk, D), generator=g, device=device) * 2 - 1) / np.sqrt(D)
    b = (torch.rand((2 * dqk,), generator=g, device=device) * 2 - 1) / np.sqrt(D)
    ms = med(lambda: ops.softmax_eta(x, w, b, dqk, 1), reps=3)
    tf = (B * L * (L + 1) / 2 * dqk * 2 + B * L * 2.0 * D * 2 * dqk) / ms / 1e9
    out.append({"kernel": "qk_proj_mfma2_kernel + softmax_nu_mfma_kernel + eta_finalize_kernel",
                "workload": "MQAR analysis batch: softmax attention eta, B=4096, L=512, D=128, d_qk=64, H=1",
                "bound": "mfma", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3, "ms": ms,
                "note": "fp32 MFMA; algorithmic flops = projection + ONE pass of the causal q.k tiles (round 4: online row "
                        "maximum, fp32 exp with exact range reduction, fp64 row sums; the two-pass form of rounds 1-3 "
                        "executed and counted the tiles twice: 5.3 ms, 24.6 % on that count)"})
    del x
    # block forward: causal linear attention mixer on one GPU's shard of C4
    B, L, D, dqk = 512, 4096, 128, 128
    qk = torch.randn((B, L, 2, 1, dqk), generator=g, device=device) / np.sqrt(dqk)
    v = torch.randn((B, L, D), generator=g, device=device)
    ms = med(lambda: ops.linear_recurrence_fwd(qk[:, :, 0], qk[:, :, 1], v, phi=True, normalise=True), reps=3)
    tf = B * L * 4.0 * dqk * D / ms / 1e9
    out.append({"kernel": "linattn_chunk_mfma_kernel",
                "workload": "C4 / 8 GPUs block forward: causal linear attention, B=512, L=4096, d_qk=128, 128 channels",
                "bound": "mfma", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3, "ms": ms,
                "note": "fp32 MFMA, chunked recurrence; algorithmic flops = 4 d C per token (state update + readout); "
                        "the kernel executes 1.6x that (intra-chunk tiles; every wave recomputes the 16 x 16 score tile), matrix pipe 67 % busy at two waves per SIMD (round 4: __launch_bounds__(256, 2); profiles/r04_blocks_pmc_counters.json)"})
    del qk, v
    # K5 at config C5: 768 HiPPO-DPLR 256 x 256 complex128 matrices (6 layers x 128 feature columns, per-column steps):
    # fused DPLR build, blocked Hessenberg on the fp64 matrix cores, windowed multishift QR -- the two stages timed apart
    import oracle as O
    n, h = 256, 768
    lam, p, _, _, _ = O.make_dplr_hippo(n)
    rng = np.random.default_rng(256)
    ls = rng.uniform(np.log(1e-3), np.log(1e-1), h)
    dd = lambda z: torch.from_numpy(np.ascontiguousarray(z)).to(device)
    ab = ops.s4_dplr_build(dd(np.tile(lam.real, (h, 1))), dd(np.tile(lam.imag, (h, 1))), dd(np.tile(p, (h, 1))), dd(ls))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    hs, qr = [], []
    unconv = 0
    for rep in range(4):
        ev[0].record()
        hh, ww, ii = ops.hessenberg_general_batched(ab)
        ev[1].record()
        ops.hseqr_general_batched(hh, ww, ii)
        ev[2].record()
        ev[2].synchronize()
        if rep:
            hs.append(ev[0].elapsed_time(ev[1]))
            qr.append(ev[1].elapsed_time(ev[2]))
        unconv = int((ii != 0).sum().item())
    hs_ms, qr_ms = float(np.median(hs)), float(np.median(qr))
    flops = 4.0 * (40.0 / 3.0) * n ** 3 * h            # (40/3) n^3 real-matrix flops, x4 for complex
    tf = flops / ((hs_ms + qr_ms) * 1e-3) / 1e12
    tf_h = 4.0 * (10.0 / 3.0) * n ** 3 * h / (hs_ms * 1e-3) / 1e12
    tf_q = 4.0 * 10.0 * n ** 3 * h / (qr_ms * 1e-3) / 1e12
    tr = torch.diagonal(ab, dim1=1, dim2=2).sum(dim=1)
    trerr = float(((ww.sum(dim=1) - tr).abs() / torch.linalg.matrix_norm(ab)).max().item())
    # the optional structured path (`s4_solver: dplr`): same 768 columns from the secular equation of diag(Lambda) - P P^H,
    # no matrix formed.  At initialisation every column shares (Lambda, P): ONE solve; "all_columns" disables that sharing
    # (what 768 differently TRAINED columns would cost).  The line's C5 figure stays the dense solver's.
    pr = (dd(np.tile(lam.real, (h, 1))), dd(np.tile(lam.imag, (h, 1))), dd(np.tile(p, (h, 1))), dd(ls))
    ms_shared = med(lambda: ops.s4_dplr_eigs(*pr), reps=3)
    ms_all = med(lambda: ops.s4_dplr_eigs(*pr, dedupe=False), reps=3)
    wf, inf_f = ops.s4_dplr_eigs(*pr, dedupe=False)
    dplr = {"kernel": "dplr_secular_kernel + s4_cayley_map_kernel", "ms_shared_solve": ms_shared, "ms_all_columns": ms_all,
            "unconverged": int((inf_f != 0).sum().item()),
            "note": "eig(Ab) = (g + a)/(g - a), a = roots of 1 - sum |P_k|^2/(Lambda_k - a) (Ehrlich-Aberth, O(n^2) per "
                    "iteration); agrees with the dense path by backward error on these raw-HiPPO columns "
                    "(tests/test_gpu_dplr_secular.py); optional analysis key s4_solver: dplr"}
    # counted HBM-side traffic of the Hessenberg stage (PMC, profiles/rNN_traffic.json; null when stale): its level-2 half
    # re-reads the trailing matrix once per column, so the stage is MEMORY-bound -- priced against the HBM peak
    hz_traffic, hz_src, hz_note = kernel_traffic_launch("zhess_blocked_kernel<16, 2>", "zhess_blocked_kernel")
    out.append({"kernel": "zhess_blocked_kernel<16,2> + zmsqr_kernel",
                "workload": "C5: S4 HiPPO-DPLR d_state=256, 6 layers x 128 columns = 768 complex128 256x256 matrices",
                "bound": "hbm (Hessenberg stage) + lds/valu behind barriers, mfma in the deferred updates (QR stage)",
                "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS,
                "ms": hs_ms + qr_ms, "spectra_per_s": h / ((hs_ms + qr_ms) * 1e-3), "unconverged": unconv,
                "max_rel_trace_error": trerr, "structured_fast_path": dplr,
                "stages": {"zhess_blocked_kernel": {"bound": "hbm", "ms": hs_ms, "achieved_tflops": tf_h,
                                                    "frac_of_fp64_peak": tf_h / FP64_PEAK_TFLOPS,
                                                    "traffic": hz_traffic, "traffic_source": hz_src, "traffic_note": hz_note,
                                                    "counted_traffic_gbs": (hz_traffic / (hs_ms * 1e-3) / 1e9) if hz_traffic else None,
                                                    "frac": (hz_traffic / (hs_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if hz_traffic
                                                    else 16.0 * n ** 3 / 3.0 * h / (hs_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                    # its bound: one matrix-vector product with the trailing matrix per
                                                    # column = 16 n^3 / 3 bytes per matrix that no blocking removes
                                                    "hbm_gbs_algorithmic": 16.0 * n ** 3 / 3.0 * h / (hs_ms * 1e-3) / 1e9,
                                                    "hbm_frac": 16.0 * n ** 3 / 3.0 * h / (hs_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                           "zmsqr_kernel": {"bound": "lds+valu (barrier-synchronous chase), mfma (deferred updates)",
                                            "ms": qr_ms, "achieved_tflops": tf_q, "frac": tf_q / FP64_PEAK_TFLOPS}},
                "note": "fp64 (matrix = vector peak); algorithmic flops 4 x (40/3) n^3 per matrix (Hessenberg 4 x (10/3) "
                        "n^3, QR 4 x 10 n^3); HIP events on the launch stream, median of 3"})
    return out


def mqar_headline(ops, device):
    """north_star's own headline: a 4096-sequence MQAR analysis batch (L = 512, hidden 128, d_state = d_qk = 64, 1 head;
    configs/iclr2026/mqar/mqar-{lin,sm}-attention.yaml:6, analysis_configs/mqar_analysis_config.yaml:1) through the GPU
    path -- eta kernels + device histogram + the D2H copy the artefacts need, wall clock -- next to the reference-style
    CPU path: the oracle's quadratic (B, L, L, H) restatement of analysis/eval_eig.py:43-135 timed on 64 sequences (the
    reference's own analysis batch) on ONE host core and scaled linearly in B (labelled as extrapolated)."""
    import oracle as O
    B, L, D, dqk, H, nb = 4096, 512, 128, 64, 1, 64
    g = torch.Generator(device=device)
    g.manual_seed(1919)
    x = torch.randn((B, L, D), generator=g, device=device)
    w = (torch.rand((2 * dqk + D, D), generator=g, device=device) * 2 - 1) / np.sqrt(D)
    b = (torch.rand((2 * dqk + D,), generator=g, device=device) * 2 - 1) / np.sqrt(D)
    wq, bq = w[:2 * dqk].contiguous(), b[:2 * dqk].contiguous()
    xs, ws, bs = x[:nb].cpu().numpy(), w.cpu().numpy(), b.cpu().numpy()
    out = {"workload": "MQAR analysis batch: B=4096 sequences, L=512, hidden 128, d_state=d_qk=64, 1 head",
           "gpu_timed": "eta kernels + threshold histogram on the device + D2H copy of eta and percentages (median of 3)",
           "cpu_timed": "oracle eta_linear / eta_softmax (quadratic NumPy form) + threshold_analysis on %d sequences, "
                        "1 process; extrapolated linearly in B to 4096" % nb}
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:
        limiter = None
    for name, gpu_fn, cpu_fn in (("lin-attention", ops.linattn_eta, O.eta_linear),
                                 ("sm-attention", ops.softmax_eta, O.eta_softmax)):
        gpu_fn(x, wq, bq, dqk, H)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            eta = gpu_fn(x, wq, bq, dqk, H)
            pct = ops.threshold_hist(eta.unsqueeze(-1), O.THRESHOLDS_RADIUS, "value")
            eta_h, pct_h = eta.cpu().numpy(), pct.cpu().numpy()
            ts.append(time.perf_counter() - t0)
        t_gpu = float(np.median(ts))
        t0 = time.perf_counter()
        ref = cpu_fn(xs, ws, bs, dqk, H, D)
        refp = O.threshold_analysis(ref, O.THRESHOLDS_RADIUS, 1, H, nb)
        t_cpu = time.perf_counter() - t0
        out[name] = {
            "gpu_s": t_gpu, "gpu_sequences_per_s": B / t_gpu,
            "cpu_baseline": {"value": nb / t_cpu, "unit": "sequences/s", "cores": 1, "kind": "port",
                             "sample": "%d of the batch's own sequences, %.2f s" % (nb, t_cpu),
                             "extrapolated_full_batch_s": t_cpu * B / nb},
            "speedup_vs_cpu_extrapolated": (t_cpu * B / nb) / t_gpu,
            "max_rel_err_vs_oracle_on_sample": float(np.max(np.abs(eta_h[:nb] / ref[..., 0] - 1))),
            "hist_max_abs_diff_pct_on_sample": float(np.max(np.abs(pct_h[:, :nb].reshape(refp[..., 0].shape) - refp[..., 0]))),
        }
    if limiter is not None and hasattr(limiter, "unregister"):
        limiter.unregister()
    return out


def cpu_baseline(a, budget_s, O):
    """The oracle's dense path (np.linalg.eigvals -> LAPACK dgeev, the same third-party arithmetic the reference
    calls at analysis/eval_eig.py:296) timed on this node's host cores on a bounded sample of the same matrices."""
    # one BLAS thread = the reference's effective behaviour (stacked eigvals is a serial gufunc loop).  If the thread
    # pool cannot be pinned, say so: `cores` then reports what the BLAS pool actually has, never a silent "1".
    cores, pinned = 1, True
    try:
        from threadpoolctl import threadpool_limits, threadpool_info
        limiter = threadpool_limits(limits=1)
    except Exception:
        limiter, pinned = None, False
        try:
            from threadpoolctl import threadpool_info
            cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
        except Exception:
            cores = int(os.environ.get("OPENBLAS_NUM_THREADS", os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)))
    done, spent = 0, 0.0
    blk = 1024
    while spent < budget_s and done + blk <= min(a.shape[0], 65536):
        host = a[done:done + blk].cpu().numpy()
        t0 = time.perf_counter()
        O.eigvals_dense(host)
        spent += time.perf_counter() - t0
        done += blk
    if limiter is not None:
        limiter.unregister() if hasattr(limiter, "unregister") else None
    return {"value": done / spent, "unit": "spectra/s", "cores": cores, "kind": "port", "blas_threads_pinned": pinned,
            "sample": "first %d of the benchmark's own matrices, stacked np.linalg.eigvals, %d thread(s), %.1f s"
                      % (done, cores, spent),
            "host_cpu_count": os.cpu_count()}


if __name__ == "__main__":
    main()
