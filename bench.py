#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path (BASELINE.json metric).

One "step" = one full `symeig(A, neig=6, mode="lowest", method="davidson", min_eps=1e-8)` call
(native HIP block Davidson) on BASELINE.json configs[1]: a batch of 64 dense symmetric fp64
operators of size N=16384 per GPU (137.4 GB resident in HBM, synthetic closed-form spectrum S1 —
SURVEY.md §8d), start block drawn on the device inside the step (seed 12421, like the reference).  Nothing is skipped or
cached between steps.

  value      = eigenpairs per second, whole job (all ranks), inputs already resident in HBM
  roofline   = the K1 operator-panel-product kernel (the "Lanczos matvec" of the metric):
               algorithmic bytes per launch (B*N^2*s + 2*B*N*p*s) / its average duration, measured
               live with HIP events on the launch stream inside the timed region, vs 8 TB/s
  cpu_baseline = the oracle (CPU restatement of the reference, bit-identical to it) timed on this
               box's host cores on a bounded sample of the same workload (rank 0, N=1 only)

Multi-GPU (launched by torch.distributed.run): the batched-operator dimension is sharded, one
process per GPU; each rank owns 64 operators (weak scaling, default) or 64/N (--scaling strong).
The only exchange is the per-iteration all-reduce(MAX) of the residual (RCCL), which keeps the
reference's global stopping rule.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="operators per GPU (weak) / in total (strong)")
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--neig", type=int, default=6)
    ap.add_argument("--spectrum", default="S1")
    ap.add_argument("--min-eps", type=float, default=1e-8)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--max-niter", type=int, default=200, help="guard only; ~19 iterations are needed")
    ap.add_argument("--no-overlap", action="store_true",
                    help="one batch group on one stream (default: two groups, panel products on a CU-masked stream)")
    ap.add_argument("--reserve-cus", type=int, default=64,
                    help="compute units the panel-product stream leaves to the small kernels of the other batch half")
    ap.add_argument("--k1", default="auto", choices=["auto", "general"],
                    help="auto: upper-triangle kernel when the storage is exactly symmetric; general: full matrix")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="4x4096", help="BxN of the CPU-baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=32)
    return ap.parse_args()


def cpu_baseline(args):
    """Oracle Davidson on the host cores, bounded sample of the same workload."""
    from oracle import ops as oops, symeig as osym
    from xitorch_amd import synthetic
    b, n = [int(v) for v in args.cpu_sample.split("x")]
    # torch-CPU collapses when oversubscribed on these skinny products (256 threads: 300 s for what
    # 32 threads do in seconds), so the baseline uses at most 32 threads and says so in `cores`.
    cores = min(os.cpu_count() or 1, args.cpu_threads)
    torch.set_num_threads(cores)
    mat = synthetic.dense_symmetric(b, n, args.spectrum)
    op = oops.DenseOp(mat, True)
    V0 = None      # the oracle draws its start block exactly like the reference (seed 12421, randn)
    times, tr = [], {}
    t_all = time.time()
    while True:
        t0 = time.time()
        osym.davidson(op, args.neig, "lowest", min_eps=args.min_eps, V0=V0, trace=tr)
        times.append(time.time() - t0)
        if (len(times) >= 3 and time.time() - t_all > 10.0) or time.time() - t_all > 25.0:
            break
    t = sorted(times)[len(times) // 2]
    s = 8
    k1_bytes = tr["napply"] * (b * n * n * s + 2 * b * n * args.neig * s)
    return {"value": b * args.neig / t, "unit": "eigpairs/s", "cores": cores, "kind": "port",
            "sample": "oracle davidson (torch-CPU restatement of the reference), dense symmetric %s, batch=%d "
                      "N=%d fp64 neig=%d min_eps=%g, median of %d runs (%.2f s each, %d iterations); "
                      "full config does not fit host RAM" % (args.spectrum, b, n, args.neig, args.min_eps,
                                                             len(times), t, tr["niter"]),
            "seconds": t, "matvec_GBps": k1_bytes / t / 1e9, "threads": torch.get_num_threads()}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the hot path has no CPU fallback)")
    if args.gpus != world:
        # one process per GPU: N > 1 means `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d; launch it with torch.distributed.run "
                         "(one rank per GPU)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)
        group = dist.group.WORLD

    from xitorch_amd import MatrixLinearOperator, LinearOperator, synthetic, kernels as XK
    from xitorch_amd.linalg import symeig

    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    esize = 8 if args.dtype == "f64" else 4
    if args.scaling == "weak":
        b_local, b_total, offset = args.batch, args.batch * world, rank * args.batch
    else:
        assert args.batch % world == 0, "strong scaling needs batch % gpus == 0"
        b_local, b_total, offset = args.batch // world, args.batch, rank * (args.batch // world)
    N, p = args.n, args.neig

    # ---- resident input: the operator batch in HBM (generated on the device, closed form) ----
    mat = torch.empty((b_local, N, N), dtype=dtype, device=dev)
    synthetic.dense_symmetric(b_local, N, args.spectrum, dtype=dtype, device=dev, out=mat, batch_offset=offset)
    # LinearOperator.m scans the matrix (like the reference's symmetry check, linop.py:97-105) and also learns
    # that the storage is EXACTLY symmetric, which lets the panel product stream only the upper triangle
    # (K1s).  --k1 general forces the full-matrix kernel.
    if args.k1 == "general":
        A = MatrixLinearOperator(mat, is_hermitian=True, symmetric_storage=False)
    else:
        A = LinearOperator.m(mat, is_hermitian=True)
    symm = bool(A.symmetric_storage)
    exact = synthetic.spectrum(args.spectrum, N, torch.float64, dev)[:p]

    k1_events = []
    traces = []

    def step(timed):
        tr = {"k1_events": k1_events if timed else None}
        with torch.no_grad():
            evals, evecs = symeig(A, neig=p, mode="lowest", method="davidson", min_eps=args.min_eps,
                                  v_init="randn", rng_device="device", max_niter=args.max_niter,
                                  overlap=(False if args.no_overlap else "auto"), reserve_cus=args.reserve_cus,
                                  process_group=group, trace=tr)
        if timed:
            traces.append(tr)
        return evals, evecs

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    step_marks = []
    for _ in range(args.steps):
        evals, evecs = step(True)
        step_marks.append(time.perf_counter())      # host clock only (davidson returns after its last status read)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = tt.item()

    # ---- result checks (outside the timed region) ----
    tol = 1e-10 if dtype == torch.float64 else 1e-3
    eval_err = (evals.double() - exact).abs().max().item()
    resid = traces[-1]["best_resid"]
    ok = eval_err <= tol * 100.0 and resid < args.min_eps

    # ---- K1 roofline from the live HIP events ----
    # every timed launch of the panel product: (duration, batch members it covered); with the two-group
    # pipeline a launch covers half of the rank's batch, and launches of the two groups never overlap
    launches = [(e0.elapsed_time(e1) * 1e-3, nb) for (e0, e1, pc, nb) in k1_events if pc == p]
    durs = [d for d, _ in launches]
    nb_launch = launches[0][1] if launches else b_local
    k1_avg = sum(durs) / max(len(durs), 1)
    k1_bytes = nb_launch * N * N * esize + 2 * nb_launch * N * p * esize       # A counted in full
    achieved = k1_bytes / k1_avg / 1e9 if k1_avg > 0 else 0.0
    kernel_name = ("K1s xk::dense_symm_tiles + symm_fold (upper-triangle panel product, exactly symmetric storage)"
                   if symm else "K1 xk::dense_rmm_cols + fold_slabs (column-oriented panel product, full matrix)")
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "k1s_pmc_traffic.json" if symm else "k1_pmc_traffic.json")
    if os.path.exists(pmc_file):
        try:
            rec = json.load(open(pmc_file))
            if rec.get("N") == N and rec.get("P") == p and rec.get("dtype") == args.dtype and rec.get("B"):
                # PMC record was taken on a launch over rec["B"] members; traffic scales linearly with the batch
                traffic = rec.get("hbm_bytes_per_launch") * nb_launch / rec["B"]
        except Exception:
            traffic = None
    if symm and nb_launch < b_local:
        # two-group pipeline: the events bracket the tile kernel on the panel-product stream; its 0.11 ms fold runs
        # on the group's own stream
        kernel_name = "K1s xk::dense_symm_tiles (upper-triangle panel product, half-batch launch; symm_fold runs beside it)"
    roofline = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "traffic": traffic, "kernel": kernel_name, "launches_timed": len(durs),
                "avg_launch_ms": k1_avg * 1e3, "algorithmic_bytes_per_launch": k1_bytes,
                "batch_members_per_launch": nb_launch}
    if symm:
        # what the upper-triangle kernel must move: the triangle incl. diagonal + the panels in and out
        tri_bytes = nb_launch * N * (N + 1) // 2 * esize + 2 * nb_launch * N * p * esize
        roofline["note"] = ("achieved uses SURVEY 8d's algorithmic bytes (A counted in full); K1s reads only the upper "
                            "triangle of the exactly symmetric operator, so it can exceed the HBM peak; "
                            "triangle_* fields price the same launch against the bytes it really has to move")
        roofline["triangle_bytes_per_launch"] = tri_bytes
        roofline["triangle_achieved"] = tri_bytes / k1_avg / 1e9 if k1_avg > 0 else 0.0
        roofline["triangle_frac"] = roofline["triangle_achieved"] / 8000.0
    # the general (full-matrix) K1 on the same resident operator, measured live for reference (untimed region)
    Xg = torch.randn((b_local, p, N), dtype=dtype, device=dev)
    Yg = torch.empty_like(Xg)
    XK.dense_mm(mat, Xg, out=Yg, trans=True)
    ge = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        XK.dense_mm(mat, Xg, out=Yg, trans=True)
        e1.record()
        ge.append((e0, e1))
    torch.cuda.synchronize()
    g_avg = sum(a.elapsed_time(b) for a, b in ge) / len(ge) * 1e-3
    roofline_general = {"bound": "hbm", "achieved": k1_bytes / g_avg / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": k1_bytes / g_avg / 1e9 / 8000.0, "avg_launch_ms": g_avg * 1e3,
                        "kernel": "K1 xk::dense_rmm_cols + fold_slabs (full matrix; used for operators whose storage "
                                  "is not exactly symmetric)", "launches_timed": len(ge)}

    if rank == 0:
        out = {
            "metric": "eigpairs/sec of symeig(davidson) + Lanczos-role matvec GB/s (roofline.achieved), batch=64 N=16384",
            "value": b_total * p * args.steps / elapsed,
            "unit": "eigpairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: linalg.symeig davidson lowest-%d, dense symmetric "
                                   "LinearOperator N=%d batch=%d/GPU %s (%s spectrum), min_eps=%g"
                                   % (p, N, b_local, args.dtype, args.spectrum, args.min_eps),
                       "global_batch": b_total, "parallelism": "batch-sharded x%d (%s)" % (world, args.scaling),
                       "iterations_per_step": traces[-1]["niter"], "panel_products_per_step": traces[-1]["napply"],
                       "basis_size": traces[-1]["basis_size"]},
            "roofline": roofline,
            "roofline_general_k1": roofline_general,
            "matvec_fraction_of_step": sum(durs) / elapsed if elapsed > 0 else None,
            "check": {"ok": bool(ok), "max_eval_err_vs_exact": eval_err, "max_resid": resid},
            "step_ms": [round((b - a) * 1e3, 2) for a, b in zip([t0] + step_marks[:-1], step_marks)],
            "k1_ms_first_last": [round(durs[0] * 1e3, 3), round(durs[-1] * 1e3, 3)] if durs else None,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
