#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path (BASELINE.json metric).

One "step" = one full `symeig(A, neig=6, mode="lowest", method="davidson", min_eps=1e-8)` call
(native HIP block Davidson) on BASELINE.json configs[1]: a batch of 64 dense symmetric fp64
operators of size N=16384 (137.4 GB, synthetic closed-form spectrum S1 — SURVEY.md §8d) resident in
HBM, start block drawn on the device inside the step (seed 12421, like the reference).  Nothing is
skipped or cached between steps.

  value      = eigenpairs per second, whole job (all ranks), inputs already resident in HBM
  roofline   = the dominant kernel = the operator-panel product (the "Lanczos matvec" of the metric),
               priced on the bytes THAT kernel has to move (K1s reads the upper triangle of the exactly
               symmetric storage: B*N(N+1)/2*s + 2*B*N*p*s; the general K1: B*N^2*s + 2*B*N*p*s)
               / its average duration, measured live with HIP events on the launch stream inside the
               timed region, vs 8 TB/s.  `full_matrix_equivalent_*` restates the same launch with
               SURVEY 8d's bytes (A counted in full) — a rate, not a roofline fraction.
  general_k1 = the same workload with the full-matrix panel kernel forced (what an operator that is
               only allclose-symmetric gets): its own eigpairs/s and roofline, measured after the
               timed region.
  cpu_baseline = the oracle (CPU restatement of the reference, bit-identical to it) timed on this
               box's host cores on a bounded sample of the same workload (rank 0, N=1 only)

Multi-GPU: `python bench.py --gpus N` re-executes itself under torch.distributed.run (one rank per
GPU, RCCL); launched by torch.distributed.run directly it just runs.  The batched-operator dimension
is sharded.  Default = STRONG scaling of the metric's batch of 64 (64/N operators per GPU); a short
weak-scaled run (64 operators per GPU) follows outside the timed region and is reported as
`weak_extra`.  The only exchange is the per-iteration all-reduce(MAX) of {residual, flag} (RCCL),
which keeps the reference's global stopping rule.
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
    ap.add_argument("--batch", type=int, default=64, help="operators in total (strong, default) / per GPU (weak)")
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--neig", type=int, default=6)
    ap.add_argument("--spectrum", default="S1")
    ap.add_argument("--min-eps", type=float, default=1e-8)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="strong (default): --batch operators in total, split over the GPUs; weak: --batch per GPU")
    ap.add_argument("--no-weak-extra", action="store_true", help="N > 1: skip the short weak-scaled extra run")
    ap.add_argument("--no-general-extra", action="store_true", help="skip the extra run with the full-matrix kernel")
    ap.add_argument("--max-niter", type=int, default=200, help="guard only; ~19 iterations are needed")
    ap.add_argument("--no-overlap", action="store_true",
                    help="one batch group on one stream (default: two groups, panel products on a CU-masked stream)")
    ap.add_argument("--reserve-cus", type=lambda v: v if v == "auto" else int(v), default="auto",
                    help="compute units the panel-product stream leaves to the small kernels of the other batch half")
    ap.add_argument("--k1", default="auto", choices=["auto", "general"],
                    help="auto: upper-triangle kernel when the storage is exactly symmetric; general: full matrix")
    ap.add_argument("--k1s-run", type=int, default=0,
                    help="measurement: column slabs per workgroup run of the upper-triangle kernel (0 = library default)")
    ap.add_argument("--k1s-opts", type=int, default=-1,
                    help="measurement: low bits of the `opts` argument of the upper-triangle kernel (include/xitorch_amd.h), "
                         "-1 = what the package ships")
    ap.add_argument("--k1-streams", default="auto", choices=["auto", "1", "2"],
                    help="measurement: panel products of the two batch groups on one CU-masked stream (1) or one each (2); "
                         "auto = one each when the resident K1s launch is on")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: run the launcher / process-group / timing / JSON plumbing of the N > 1 path on CPU "
                         "ranks over gloo with a tiny sharded stand-in step (no performance numbers)")
    ap.add_argument("--config", default=None, choices=["c3", "c3g", "c4", "c5", "c5w"],
                    help="one of the OTHER BASELINE.json configs as a bench line of the same shape (bench_secondary.py): "
                         "c3 BiCGStab banded + implicit backward, c3g native GMRES on the same systems, c4 Broyden shard + "
                         "implicit backward, c5 / c5w the fp32 N=32768 shard with a 6- / 16-column block")
    ap.add_argument("--cfg-batch", type=int, default=0, help="--config: operators / systems per GPU (0 = the config's own)")
    ap.add_argument("--gmres-restart", type=int, default=0, help="--config c3g: restarted GMRES(m) (0 = un-restarted)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` block (the other BASELINE.json configs, measured after the headline: c0 N=512 "
                         "symeig, c3 BiCGStab fwd+bwd, c4 Broyden shard, c5w fp32 16-column shard; each with its roofline and "
                         "a CPU baseline)")
    ap.add_argument("--cfg-steps", type=int, default=5, help="timed steps of every entry of the `configs` block")
    ap.add_argument("--no-standalone", action="store_true",
                    help="skip the stream-read / standalone whole-batch launches after the timed region (profiled runs: the "
                         "trace then holds the timed region's panel launches only)")
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
    logical = os.cpu_count() or 1
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or logical
    except Exception:
        physical = logical
    cores = min(logical, args.cpu_threads)
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
    out = {"value": b * args.neig / t, "unit": "eigpairs/s", "cores": cores, "kind": "port",
           "sample": "oracle davidson (torch-CPU restatement of the reference), dense symmetric %s, batch=%d "
                     "N=%d fp64 neig=%d min_eps=%g, median of %d runs (%.2f s each, %d iterations); "
                     "full config does not fit host RAM" % (args.spectrum, b, n, args.neig, args.min_eps,
                                                            len(times), t, tr["niter"]),
           "seconds": t, "matvec_GBps": k1_bytes / t / 1e9, "threads": torch.get_num_threads(),
           "full_config_seconds_extrapolated": t * (args.n / float(n)) ** 2 * args.batch / float(b),
           "extrapolation": "x %.0f to the full config (%d x %d^2: the panel product is O(B N^2) per iteration and "
                            "dominates; the closed-form spectrum converges in the same number of iterations at both "
                            "orders)" % ((args.n / float(n)) ** 2 * args.batch / float(b), args.batch, args.n),
           "physical_cores": physical, "logical_cpus": logical,
           "note": "cores = threads actually used (torch-CPU collapses when oversubscribed on these skinny "
                   "products); the box has physical_cores / logical_cpus"}
    del mat, op
    # ---- BASELINE configs[0] exactly (N=512, batch=1, lowest 6, fp64): bench_secondary.cpu_baseline_c0
    try:
        import bench_secondary as _bs
        out["config1_n512_b1"] = _bs.cpu_baseline_c0(cores)
    except Exception as err:                    # a baseline extra never costs the headline line
        out["config1_n512_b1"] = {"error": repr(err)}
    # ---- the K1 panel product alone on the CPU at the metric's N, largest batch that fits host RAM comfortably
    try:
        import psutil
        avail = psutil.virtual_memory().available
        n1 = args.n
        bk = int(max(1, min(4, (avail * 0.25) // (n1 * n1 * 8))))
        matk = synthetic.dense_symmetric(bk, n1, args.spectrum)
        xk_ = torch.randn(bk, n1, args.neig, dtype=torch.float64)
        torch.matmul(matk, xk_)
        ts = []
        t_all = time.time()
        while len(ts) < 3 or (time.time() - t_all < 5.0 and len(ts) < 20):
            t0 = time.time()
            torch.matmul(matk, xk_)
            ts.append(time.time() - t0)
        tk = sorted(ts)[len(ts) // 2]
        kb = bk * n1 * n1 * 8 + 2 * bk * n1 * args.neig * 8
        out["k1_product_cpu"] = {"workload": "torch.matmul(mat (%d, %d, %d), x (.., %d)) fp64 = the reference's "
                                             "MatrixLinearOperator._mm (linop.py:695-696)" % (bk, n1, n1, args.neig),
                                 "ms": tk * 1e3, "GBps": kb / tk / 1e9, "threads": torch.get_num_threads(),
                                 "bytes": kb, "note": "batch bounded by host RAM (a quarter of what is available)"}
        del matk
    except Exception as err:
        out["k1_product_cpu"] = {"error": repr(err)}
    return out


def _multi_gpu_diag(group, dev, elapsed_local, k1_avg_local, steps):
    """What makes the first real N-GPU run diagnosable from its one line: every rank's own wall time of the timed region
    and its own panel-launch average (all-gathered), the spread between the ranks, and the latency of the solver's
    per-iteration status all-reduce — the same call the eigensolver makes (`xitorch_amd.dist.allreduce_max_`: in-stream
    RCCL through the C ABI on HIP tensors, c10d otherwise), timed back to back on the current stream after the run."""
    import torch.distributed as dist
    from xitorch_amd import dist as xd
    world = dist.get_world_size(group)
    mine = torch.tensor([elapsed_local * 1e3 / max(steps, 1), (k1_avg_local or 0.0) * 1e3], dtype=torch.float64, device=dev)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine, group=group)
    per_step = [round(v[0].item(), 3) for v in allv]
    per_k1 = [round(v[1].item(), 4) for v in allv]
    st = torch.zeros(5, dtype=torch.float64, device=dev)
    reps = 50
    for _ in range(5):
        xd.allreduce_max_(st, group)
    if dev.type == "cuda":
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            xd.allreduce_max_(st, group)
        e1.record()
        torch.cuda.synchronize()
        lat_us = e0.elapsed_time(e1) * 1e3 / reps
        how = "HIP events around %d back-to-back in-stream all-reduces of the 5-double status vector" % reps
    else:
        t0 = time.perf_counter()
        for _ in range(reps):
            xd.allreduce_max_(st, group)
        lat_us = (time.perf_counter() - t0) * 1e6 / reps
        how = "host clock around %d back-to-back all-reduces of the 5-double status vector (CPU ranks)" % reps
    lat = torch.tensor([lat_us], dtype=torch.float64, device=dev)
    dist.all_reduce(lat, op=dist.ReduceOp.MAX, group=group)
    return {"per_rank_ms_per_step": per_step, "per_rank_k1_avg_launch_ms": per_k1,
            "rank_skew_ms_per_step": round(max(per_step) - min(per_step), 3),
            "status_allreduce_latency_us": round(lat.item(), 2), "status_allreduce_timing": how,
            "status_allreduce_backend": ("device RCCL communicator (C ABI, in stream)"
                                         if (dev.type == "cuda" and xd.device_comm(group, dev) is not None)
                                         else "c10d " + str(dist.get_backend(group)))}


def _respawn(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def _panel_bytes(nb, N, p, esize, symm):
    """bytes one panel-product launch over nb operators HAS to move (what roofline.achieved is priced on)"""
    mat = nb * N * (N + 1) // 2 * esize if symm else nb * N * N * esize
    return mat + 2 * nb * N * p * esize


def _k1s_source_hash(names=("xk_symm.hip", "xk_common.h")):
    import hashlib
    h = hashlib.sha256()
    for name in names:
        h.update(open(os.path.join(ROOT, "xitorch_amd", "csrc", name), "rb").read())
    return h.hexdigest()


def _k1_periods(k1_events, p):
    """Per-launch busy time of the panel-product launches of the timed region.

    Every launch carries HIP events (e0 before, e1 after) on the stream it ran on.  With every launch on ONE stream the
    launches are disjoint and a launch's time is e1 - e0.  With the resident launches of the two batch groups on two
    streams, a launch is enqueued while the previous one still holds the machine and moves into the workgroup slots its
    tail frees, so the [e0, e1] intervals overlap: what one launch costs the step is its COMPLETION PERIOD
    e1_i - max(e0_i, e1_{i-1}) (launches ordered by completion), i.e. the union of the busy intervals split at the
    completions.  Both forms sum to the time during which a panel product was running or waiting for slots held by
    one; for disjoint launches the two definitions coincide."""
    sel = [(e0, e1, nb) for (e0, e1, pc, nb) in k1_events if pc == p]
    if not sel:
        return [], [], None
    base = sel[0][0]
    iv = sorted(((base.elapsed_time(e0) * 1e-3, base.elapsed_time(e1) * 1e-3) for (e0, e1, nb) in sel),
                key=lambda t: t[1])
    raw = [e - s for s, e in iv]
    periods, prev_end = [], None
    for s0, e0_ in iv:
        periods.append(e0_ - (s0 if prev_end is None or s0 > prev_end else prev_end))
        prev_end = e0_
    return periods, raw, sel[0][2]


def _pct(vals, q):
    v = sorted(vals)
    return v[min(len(v) - 1, max(0, int(round(q * (len(v) - 1)))))] if v else None


def _k1_roofline(k1_events, N, p, esize, symm, b_local):
    durs, raw, nb0 = _k1_periods(k1_events, p)
    nb_launch = nb0 if nb0 is not None else b_local
    k1_avg = sum(durs) / max(len(durs), 1)
    need = _panel_bytes(nb_launch, N, p, esize, symm)
    full = _panel_bytes(nb_launch, N, p, esize, False)
    achieved = need / k1_avg / 1e9 if k1_avg > 0 else 0.0
    if symm:
        kernel = "K1s xk::dense_symm_tiles (upper-triangle panel product of exactly symmetric storage%s)" % (
            "; half-batch launch, symm_fold runs beside it on the group's stream" if nb_launch < b_local
            else " + symm_fold")
    else:
        kernel = "K1 xk::dense_rmm_cols + fold_slabs (column-oriented panel product, full matrix)"
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "k1s_pmc_traffic.json" if symm else "k1_pmc_traffic.json")
    traffic_note = None
    if os.path.exists(pmc_file):
        try:
            rec = json.load(open(pmc_file))
            if rec.get("N") == N and rec.get("P") == p and rec.get("B"):
                # PMC record of a launch over rec["B"] members; traffic scales linearly with the batch.  The record
                # carries the hash of the kernel source it was measured on (scripts/pmc_traffic.sh): a record of
                # another kernel is not reported
                stamp = rec.get("kernel_source_sha256")
                if stamp != (_k1s_source_hash() if symm else _k1s_source_hash(("xk_dense.hip", "xk_common.h"))):
                    traffic_note = "PMC record is stale (kernel source changed since scripts/pmc_traffic.sh ran): not reported"
                else:
                    traffic = rec.get("hbm_bytes_per_launch") * nb_launch / rec["B"]
                    traffic_note = "PMC (FETCH_SIZE x2 + WRITE_SIZE, separate passes) of a whole-batch launch of the same " \
                                   "kernel source, scaled to the launch's batch members (profiles/%s)" % os.path.basename(pmc_file)
        except Exception:
            traffic = None
    roof = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
            "traffic": traffic, "traffic_note": traffic_note, "kernel": kernel, "launches_timed": len(durs), "avg_launch_ms": k1_avg * 1e3,
            "launch_ms_p10_p50_p90": [round(_pct(durs, q) * 1e3, 3) for q in (0.1, 0.5, 0.9)] if durs else None,
            "launch_ms_own_interval_avg": (sum(raw) / len(raw) * 1e3) if raw else None,
            "launch_time_definition": "completion period e1_i - max(e0_i, e1_{i-1}) over the HIP events of the launches "
                                      "(== e1 - e0 when the launches do not overlap); own_interval = plain e1 - e0, which "
                                      "includes waiting for slots when two resident launches overlap",
            "algorithmic_bytes_per_launch": need, "batch_members_per_launch": nb_launch,
            "bytes_formula": ("B*N*(N+1)/2*s + 2*B*N*p*s (triangle incl. diagonal + panel in + panel out)" if symm
                              else "B*N^2*s + 2*B*N*p*s (SURVEY 8d)")}
    if symm:
        roof["full_matrix_equivalent_bytes_per_launch"] = full
        roof["full_matrix_equivalent_GBps"] = full / k1_avg / 1e9 if k1_avg > 0 else 0.0
        roof["note"] = ("frac prices the launch on the bytes the upper-triangle kernel must move; "
                        "full_matrix_equivalent_GBps = SURVEY 8d's bytes (A counted in full) / the same time: the rate a "
                        "full-matrix kernel would need to match it, not a roofline fraction")
    return roof, durs


def _flat_scalars(out):
    """The driver's record keeps scalars of `roofline` only (nested dicts, lists and extra top-level keys are dropped):
    every figure of the nested blocks that a reader of the record needs is repeated here as a flat scalar of
    `roofline` — the SURVEY 8d-conformant full-matrix figure, the bare stream, the launch-time percentiles, and one
    scalar set per secondary BASELINE config.  The nested blocks stay for humans."""
    roof = out.get("roofline")
    if not isinstance(roof, dict):
        return

    def num(v):
        return v if isinstance(v, (int, float)) and not isinstance(v, bool) else None

    def put(key, v):
        v = num(v)
        if v is not None:
            roof[key] = v
    s8 = roof.get("survey_8d") or {}
    put("frac_survey_8d", s8.get("frac"))
    put("survey_8d_frac", s8.get("frac"))
    put("survey_8d_GBps", s8.get("achieved"))
    put("survey_8d_ms", s8.get("avg_launch_ms"))
    put("survey_8d_bytes_per_launch", s8.get("algorithmic_bytes_per_launch"))
    put("survey_8d_traffic", s8.get("traffic"))
    put("survey_8d_eigpairs_per_s", s8.get("eigpairs_per_s"))
    put("survey_8d_ms_per_step", s8.get("ms_per_step"))
    put("stream_read_GBps", (roof.get("stream_read") or {}).get("GBps"))
    pct = roof.get("launch_ms_p10_p50_p90") or [None] * 3
    for k, v in zip(("p10_ms", "p50_ms", "p90_ms"), pct):
        put(k, v)
    sa = roof.get("standalone_whole_batch_launch") or {}
    put("standalone_ms", sa.get("avg_launch_ms"))
    put("standalone_frac", sa.get("frac"))
    if s8:
        roof["note"] = ("`frac` prices the TIMED kernel (K1s, reads the upper triangle only) on the bytes it must move; "
                        "SURVEY 8d's figure (A read once IN FULL, no credit for symmetry) is `frac_survey_8d`: the same "
                        "workload on the full-matrix kernel K1, timed the same way after the timed region "
                        "(survey_8d_ms / survey_8d_eigpairs_per_s)")
    cfgs = out.get("configs") or {}
    for name in ("c0", "c3", "c4", "c5w"):
        rec = cfgs.get(name)
        if not isinstance(rec, dict) or "error" in rec:
            continue
        r = rec.get("roofline") or {}
        put(name + "_ms", rec.get("ms_per_step"))
        put(name + "_value", rec.get("value"))
        put(name + "_frac", r.get("frac"))
        put(name + "_launch_ms", r.get("avg_launch_ms"))
        put(name + "_traffic", r.get("traffic"))
        put(name + "_bytes_per_launch", r.get("algorithmic_bytes_per_launch"))
        put(name + "_mfma_frac", r.get("frac_of_fp32_matrix_peak"))
        cb = rec.get("cpu_baseline") or {}
        put(name + "_cpu", cb.get("value"))
        put(name + "_cpu_full_config_s", cb.get("full_config_seconds_extrapolated"))
        put(name + "_check_ok", int(bool((rec.get("check") or {}).get("ok"))) if rec.get("check") else None)
        if name == "c0":
            put("c0_exacteig_ms", (rec.get("exacteig") or {}).get("ms_per_step"))
            put("c0_cpu_exacteig", cb.get("exacteig_value"))


def main_dry(args):
    """`--dry-run`: everything of the N > 1 path that does not need a GPU, end to end on CPU ranks over gloo — the
    self-respawn under torch.distributed.run, the rendezvous, the strong / weak shard arithmetic, the fenced timing
    with its MAX all-reduce, the weak-scaling extra and rank 0's single JSON line.  The step is a stand-in: the
    product's batch-sharded quasi-Newton driver (linear mixing: global SUM all-reduces, needs no device kernel) on a
    tiny tanh system.  No number in the line is a measurement (`value` is null)."""
    import torch.distributed as dist
    from xitorch_amd import dist as xd, synthetic
    from xitorch_amd.optimize import native_root as nr
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.set_num_threads(1)
    group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (gloo reports its connections on the C-level stdout: same fd 1 -> fd 2 detour as for RCCL's banner below)
        import ctypes
        libc = ctypes.CDLL(None)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo")
            dist.barrier()
        finally:
            libc.fflush(None)
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        group = dist.group.WORLD

    def fcn(y, A):
        return torch.tanh(torch.einsum("bij,bj->bi", A, y) + 0.1) + y / 2.0

    def run(b_local, offset, steps, warmup):
        A = synthetic.root_matrix(offset + b_local, 24)[offset:offset + b_local] * 2.0
        y0 = torch.zeros((b_local, 24), dtype=torch.float64)
        tr = {}

        def step():
            return nr.linearmixing(fcn, y0, (A,), alpha=-1.0, f_tol=1e-10, x_tol=1e-10, maxiter=400,
                                   process_group=group, trace=tr)
        for _ in range(warmup):
            step()
        if group is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = step()
        if group is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        run.local_elapsed = elapsed
        if group is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = tt.item()
        return elapsed, fcn(y, A).abs().max().item(), tr

    def shard(scaling):
        if scaling == "weak":
            return args.batch, args.batch * world, rank * args.batch
        lo, hi = xd.shard_range(args.batch, world, rank)          # uneven batches allowed here (the GPU job asserts)
        return hi - lo, args.batch, lo

    b_local, b_total, offset = shard(args.scaling)
    elapsed, fmax, tr = run(b_local, offset, args.steps, args.warmup)
    multi_gpu = _multi_gpu_diag(group, torch.device("cpu"), run.local_elapsed, None, args.steps) if group is not None else None
    weak_extra = None
    if world > 1 and args.scaling == "strong" and not args.no_weak_extra:
        wl, wt, woff = shard("weak")
        w_el, _, _ = run(wl, woff, 2, 1)
        weak_extra = {"scaling": "weak", "value": None, "ms_per_step": w_el / 2 * 1e3, "global_batch": wt,
                      "batch_per_gpu": wl, "steps": 2, "warmup": 1}
    # every rank ran the same number of iterations (the whole batch is one flat system)
    nit = torch.tensor([float(tr["niter"]), -float(tr["niter"])], dtype=torch.float64)
    if group is not None:
        dist.all_reduce(nit, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({
            "metric": "eigpairs/sec of symeig(davidson) + Lanczos-role matvec GB/s (roofline.achieved), batch=64 N=16384",
            "dry_run": True, "value": None, "unit": "eigpairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "DRY RUN on CPU ranks (gloo): sharded linear-mixing root solve, tanh system 24 "
                                   "unknowns x %d members; plumbing only, no measurement" % b_total,
                       "global_batch": b_total, "batch_per_gpu": b_local,
                       "parallelism": "batch-sharded x%d (%s)" % (world, args.scaling),
                       "comm_backend": dist.get_backend(group) if group is not None else None,
                       "comm_world_size": dist.get_world_size(group) if group is not None else 1},
            "roofline": None, "weak_extra": weak_extra, "multi_gpu": multi_gpu,
            "check": {"ok": bool(fmax < 1e-8 and nit[0].item() == -nit[1].item()), "max_abs_f": fmax,
                      "iterations": int(nit[0].item())}}), flush=True)
    if group is not None:
        dist.destroy_process_group()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _respawn(args)
    if args.dry_run:
        return main_dry(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the hot path has no CPU fallback)")
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.k1s_run > 0 or args.k1s_opts >= 0:
        from xitorch_amd import kernels as _XK
        low = ((_XK.K1S_OPTS or 0) & 0xff) if args.k1s_opts < 0 else args.k1s_opts & 0xff
        _XK.K1S_OPTS = (int(args.k1s_run) << 8) | low   # column slabs per workgroup run | flag bits (arguments of K1s)
    group, backend, rccl_world = None, None, 1
    # XITORCH_BENCH_FORCE_PG=1: create the RCCL process group even for one rank (smoke test of the N > 1 plumbing —
    # init, barrier, all-reduce of the timing — on a single-GPU box; the solver's own all-reduces need >= 2 ranks)
    if world > 1 or os.environ.get("XITORCH_BENCH_FORCE_PG") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                      # the forced single-rank group: no launcher has set the rendezvous up
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("LOCAL_RANK", "0")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        # RCCL prints a version banner on the C-level stdout when its first communicator comes up; rank 0's
        # stdout carries exactly one JSON line, so the banner is sent to stderr (fd 1 -> fd 2 while the group and its
        # first collective are set up, C stdio flushed before fd 1 is restored)
        import ctypes
        libc = ctypes.CDLL(None)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="nccl", device_id=dev)         # "nccl" == RCCL on ROCm
            dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize()
        finally:
            libc.fflush(None)
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        group = dist.group.WORLD
        backend, rccl_world = dist.get_backend(group), dist.get_world_size(group)

    from xitorch_amd import LinearOperator, synthetic, kernels as XK
    from xitorch_amd.linalg import symeig

    if args.config:
        # one of the other BASELINE configs: same fences, same JSON shape, its own roofline (bench_secondary.py)
        import bench_secondary

        def cfg_fence():
            if group is not None:
                torch.distributed.barrier()
            torch.cuda.synchronize()
        line = bench_secondary.run(args, dev, group, world, rank, cfg_fence, backend)
        if rank == 0:
            import ctypes
            ctypes.CDLL(None).fflush(None)
            print(json.dumps(line), flush=True)
        if group is not None:
            torch.distributed.destroy_process_group()
        return

    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    esize = 8 if args.dtype == "f64" else 4
    N, p = args.n, args.neig
    exact = synthetic.spectrum(args.spectrum, N, torch.float64, dev)[:p]

    def fence():
        if group is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def make_operator(b_local, offset, k1):
        # resident input: the operator batch in HBM (generated on the device, closed form).
        # LinearOperator.m scans the matrix (like the reference's symmetry check, linop.py:97-105) and also learns
        # that the storage is EXACTLY symmetric, which lets the panel product stream only the upper triangle (K1s).
        mat = torch.empty((b_local, N, N), dtype=dtype, device=dev)
        synthetic.dense_symmetric(b_local, N, args.spectrum, dtype=dtype, device=dev, out=mat, batch_offset=offset)
        A = LinearOperator.m(mat, is_hermitian=True)
        if k1 == "general":
            A.symmetric_storage = False          # what a merely allclose-symmetric operator gets: the full-matrix kernel
        return mat, A

    def run(A, steps, warmup, events):
        """`warmup` untimed + exactly `steps` timed symeig calls between two barrier+synchronize fences; MAX over ranks."""
        traces = []

        if events is not None:
            # the timing events of the panel launches are created (and instantiated by one record each) BEFORE the
            # timed region: creating them costs host time — visibly so under rocprofv3, where the GPU then idles
            # between an event and its kernel — and now and then a ~30 ms stall when the runtime grows its pool
            XK.prefill_timing_events(2 * 48 * (steps + 1))

        def step(timed):
            tr = {"k1_events": events if timed else None}
            with torch.no_grad():
                evals, evecs = symeig(A, neig=p, mode="lowest", method="davidson", min_eps=args.min_eps,
                                      v_init="randn", rng_device="device", max_niter=args.max_niter,
                                      overlap=(False if args.no_overlap else "auto"), reserve_cus=args.reserve_cus,
                                      k1_streams=({"auto": "auto", "1": False, "2": True}[args.k1_streams]),
                                      process_group=group, trace=tr)
            if timed:
                traces.append(tr)
            return evals, evecs
        for _ in range(warmup):
            step(False)
        fence()
        t0 = time.perf_counter()
        marks = []
        for _ in range(steps):
            evals, evecs = step(True)
            marks.append(time.perf_counter())     # host clock only (davidson returns after its last status read)
        t_own = (marks[-1] - t0) if marks else 0.0      # this rank's own steps, before the closing fence
        fence()
        elapsed = time.perf_counter() - t0
        run.local_elapsed = t_own
        if group is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            elapsed = tt.item()
        step_ms = [round((b - a) * 1e3, 2) for a, b in zip([t0] + marks[:-1], marks)]
        return elapsed, evals, traces, step_ms

    def shard(scaling):
        if scaling == "weak":
            return args.batch, args.batch * world, rank * args.batch
        assert args.batch % world == 0, "strong scaling needs batch % gpus == 0"
        return args.batch // world, args.batch, rank * (args.batch // world)

    # ---------------- the timed job ----------------
    b_local, b_total, offset = shard(args.scaling)
    mat, A = make_operator(b_local, offset, args.k1)
    symm = bool(A.symmetric_storage)
    k1_events = []
    elapsed, evals, traces, step_ms = run(A, args.steps, args.warmup, k1_events)

    # result checks (outside the timed region)
    tol = 1e-10 if dtype == torch.float64 else 1e-3
    eval_err = (evals.double() - exact).abs().max().item()
    resid = traces[-1]["best_resid"]
    ok = eval_err <= tol * 100.0 and resid < args.min_eps
    roofline, durs = _k1_roofline(k1_events, N, p, esize, symm, b_local)
    multi_gpu = None
    if group is not None:
        multi_gpu = _multi_gpu_diag(group, dev, run.local_elapsed, (sum(durs) / len(durs)) if durs else None, args.steps)

    # ---------------- extra: the timed kernel alone on the GPU (no other batch group beside it) ----------------
    # inside the two-group pipeline the panel product shares HBM with the other group's small kernels (that is the
    # point of the pipeline); the same kernel on the whole batch with nothing else running shows what the sharing costs
    # what the operator batch streams at with no arithmetic at all (3 read-only passes, idle GPU): the practical
    # ceiling of any panel kernel on this box and this placement of the batch
    try:
        if args.no_standalone:
            raise RuntimeError("skipped (--no-standalone)")
        XK.stream_read(mat)
        re_ = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            nread = XK.stream_read(mat)
            e1.record()
            re_.append((e0, e1))
        torch.cuda.synchronize()
        r_avg = sum(a.elapsed_time(b) for a, b in re_) / len(re_) * 1e-3
        roofline["stream_read"] = {"GBps": nread / r_avg / 1e9, "frac_of_peak": nread / r_avg / 1e9 / 8000.0,
                                   "bytes": nread, "avg_ms": r_avg * 1e3,
                                   "note": "xk_stream_read over the whole operator batch: 16 B/lane nt loads, no arithmetic"}
        roofline["frac_of_stream_read"] = roofline["achieved"] / roofline["stream_read"]["GBps"]
    except Exception as err:        # a measurement extra never costs the headline line
        roofline["stream_read"] = {"error": repr(err)}
    if symm and not args.no_standalone:
        Xs = torch.randn((b_local, p, N), dtype=dtype, device=dev)
        Ys = torch.empty_like(Xs)
        XK.dense_symm(mat, Xs, out=Ys)
        se = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            XK.dense_symm(mat, Xs, out=Ys)
            e1.record()
            se.append((e0, e1))
        torch.cuda.synchronize()
        s_avg = sum(a.elapsed_time(b) for a, b in se) / len(se) * 1e-3
        sb = _panel_bytes(b_local, N, p, esize, True)
        roofline["standalone_whole_batch_launch"] = {
            "avg_launch_ms": s_avg * 1e3, "achieved": sb / s_avg / 1e9, "frac": sb / s_avg / 1e9 / 8000.0,
            "bytes": sb, "note": "tile kernel + fold, %d operators, idle GPU, outside the timed region" % b_local}
        if "GBps" in roofline.get("stream_read", {}):
            # the triangle kernel against the bare stream: the same bytes per second scale, no data-sheet number
            roofline["standalone_whole_batch_launch"]["frac_of_stream_read"] = \
                (sb / s_avg / 1e9) / roofline["stream_read"]["GBps"]
        del Xs, Ys

    # ---------------- extra: the full-matrix panel kernel on the same resident operator ----------------
    general = None
    if symm and not args.no_general_extra:
        A.symmetric_storage = False
        gev = []
        g_el, g_evals, g_tr, g_ms = run(A, max(1, min(args.steps, 10)), 1, gev)     # (10 x 0.4 s: VERDICT r03 #6)
        A.symmetric_storage = True
        g_roof, _ = _k1_roofline(gev, N, p, esize, False, b_local)
        general = {"value": b_total * p * len(g_ms) / g_el, "unit": "eigpairs/s", "ms_per_step": g_el / len(g_ms) * 1e3,
                   "steps": len(g_ms), "iterations_per_step": g_tr[-1]["niter"], "roofline": g_roof,
                   "max_eval_err_vs_exact": (g_evals.double() - exact).abs().max().item(),
                   "note": "same workload, full-matrix panel kernel (operators whose storage is not exactly symmetric)"}
        gr = g_roof
        roofline["survey_8d"] = {
            "kernel": gr["kernel"], "frac": gr["frac"], "achieved": gr["achieved"], "avg_launch_ms": gr["avg_launch_ms"],
            "algorithmic_bytes_per_launch": gr["algorithmic_bytes_per_launch"], "traffic": gr["traffic"],
            "launches_timed": gr["launches_timed"], "eigpairs_per_s": general["value"], "ms_per_step": general["ms_per_step"],
            "note": "SURVEY 8d prices the panel product on A read once IN FULL (no credit for symmetry): this is the same "
                    "workload on the full-matrix kernel K1 (what an operator whose storage is not exactly symmetric gets), "
                    "timed the same way after the headline's timed region; `frac` above prices the upper-triangle kernel "
                    "on the bytes IT must move"}
    elif not symm:
        # one whole-batch launch of the general kernel, timed alone (untimed region), bytes of the WHOLE batch
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
        gb = _panel_bytes(b_local, N, p, esize, False)
        general = {"standalone_whole_batch_launch": {"achieved": gb / g_avg / 1e9, "frac": gb / g_avg / 1e9 / 8000.0,
                                                     "avg_launch_ms": g_avg * 1e3, "bytes": gb}}

    # ---------------- extra: weak scaling (64 operators per GPU), N > 1 only ----------------
    weak_extra = None
    if world > 1 and args.scaling == "strong" and not args.no_weak_extra:
        del A, mat
        torch.cuda.empty_cache()
        try:
            wl, wt, woff = shard("weak")
            wmat, wA = make_operator(wl, woff, args.k1)
            w_el, _, w_tr, w_ms = run(wA, 2, 1, None)
            weak_extra = {"scaling": "weak", "value": wt * p * 2 / w_el, "unit": "eigpairs/s", "ms_per_step": w_el / 2 * 1e3,
                          "global_batch": wt, "batch_per_gpu": wl, "steps": 2, "warmup": 1}
            del wA, wmat
        except Exception as err:            # never lose the headline line to the extra
            weak_extra = {"error": repr(err)}

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
                                   "LinearOperator N=%d batch=%d (%d per GPU) %s (%s spectrum), min_eps=%g"
                                   % (p, N, b_total, b_local, args.dtype, args.spectrum, args.min_eps),
                       "global_batch": b_total, "batch_per_gpu": b_local,
                       "parallelism": "batch-sharded x%d (%s)" % (world, args.scaling),
                       "comm_backend": backend, "comm_world_size": rccl_world,
                       "iterations_per_step": traces[-1]["niter"], "panel_products_per_step": traces[-1]["napply"],
                       "basis_size": traces[-1]["basis_size"], "batch_groups": traces[-1].get("groups")},
            "roofline": roofline,
            "general_k1": general,
            "weak_extra": weak_extra,
            "multi_gpu": multi_gpu,
            "matvec_fraction_of_step": sum(durs) / elapsed if elapsed > 0 else None,
            "check": {"ok": bool(ok), "max_eval_err_vs_exact": eval_err, "max_resid": resid},
            "step_ms": step_ms,
            "k1_ms_first_last": [round(durs[0] * 1e3, 3), round(durs[-1] * 1e3, 3)] if durs else None,
            "k1_ms_last_step": [round(d * 1e3, 2) for d in durs[-2 * traces[-1]["niter"]:]] if durs else None,
            "k1_ms_last_step_note": "completion periods of the last step's panel launches in completion order (the two "
                                    "batch groups alternate; the basis grows by 6 vectors per pair, and with it the bytes "
                                    "the other group's chain moves beside the launch)",
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        if world == 1 and not args.no_configs:
            try:
                del A, mat
            except NameError:
                pass
            torch.cuda.empty_cache()
            import bench_secondary as _bs
            out["configs"] = _bs.configs_block(args, dev, out.get("cpu_baseline"))
        _flat_scalars(out)
        import ctypes
        ctypes.CDLL(None).fflush(None)          # nothing buffered by native libraries may follow the line
        print(json.dumps(out), flush=True)
    if group is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
