"""`python bench.py --config c3|c3g|c4|c5|c5w`: the other BASELINE.json configs as bench lines of the same JSON shape as
the headline (`metric / value / unit / ms_per_step / roofline / check`), measured the same way: W untimed warm-up steps,
exactly K timed steps between barrier + synchronize fences, the dominant kernel's launch duration from HIP events inside
the timed region, `roofline.achieved` = SURVEY 8d's ALGORITHMIC bytes per launch / that duration.

  c3   BASELINE configs[2]: linalg.solve, BiCGStab on the banded (bw = 127) operator, N = 65536, batch = 256, fp64, WITH
       its implicit backward (adjoint BiCGStab + band gradient); dominant kernel: the banded apply,
       bytes = B (2 hb + 1) N s + 2 B N c s per launch (solve.py:192-324, linalg/solve.py:119-222)
  c3g  the same systems through the native GMRES (solve.py:326-433), forward only
  c4   BASELINE configs[3], per-GPU shard: optimize.rootfinder Broyden on tanh(A y + 0.1) + y / 2, N = 8192, batch = 64,
       fp64, implicit backward through linalg.solve; dominant kernel: the function evaluation's batched matvec,
       bytes = B N^2 s per evaluation; Gm.mv of rank r moves 2 r L s + 2 L s (rootsolver.py:15-206, _jacobian.py:51-222)
  c5   BASELINE configs[4], per-GPU shard: symeig davidson on 16 x 32768^2 fp32, 6-column block (K1s, triangle)
  c5w  the same with the 16-column block BASELINE states ("MFMA A@V panel"): K1w on the matrix cores,
       bytes = B N^2 s + 2 B N p s per launch

With --gpus N each rank holds one shard (weak scaling) and the solvers exchange their few-byte global decisions through
`process_group`; `value` counts the units of all ranks over the MAX time.
"""
import json
import os
import sys
import time
import warnings

import torch

PEAK_HBM = 8000.0        # GB/s (MI355X_MICROARCH.md)
PEAK_F32 = 157.3         # TFLOP/s dense fp32 MFMA


def _avg_ms(events):
    """average COMPLETION PERIOD of the launches (bench.py::_k1_periods: e1_i - max(e0_i, e1_{i-1}) in completion order;
    the plain e1 - e0 when the launches do not overlap — resident panel launches of two batch groups do)"""
    if not events:
        return float("nan"), 0
    base = events[0][0]
    iv = sorted(((base.elapsed_time(a), base.elapsed_time(b)) for (a, b) in events), key=lambda t: t[1])
    per, prev = [], None
    for s0, e0 in iv:
        per.append(e0 - (s0 if prev is None or s0 > prev else prev))
        prev = e0
    return sum(per) / len(per), len(per)


def _timed(step, steps, warmup, fence, group, dev):
    for _ in range(warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    marks, last = [], None
    for _ in range(steps):
        last = step(True)
        marks.append(time.perf_counter())
    fence()
    elapsed = time.perf_counter() - t0
    if group is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = tt.item()
    step_ms = [round((b - a) * 1e3, 2) for a, b in zip([t0] + marks[:-1], marks)]
    return elapsed, step_ms, last


def _roofline(bytes_per_launch, avg_ms, nlaunch, kernel, extra=None):
    ach = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms == avg_ms and avg_ms > 0 else None
    r = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": PEAK_HBM, "unit": "GB/s",
         "frac": (ach / PEAK_HBM) if ach else None, "traffic": None, "algorithmic_bytes_per_launch": bytes_per_launch,
         "avg_launch_ms": avg_ms, "launches_timed": nlaunch,
         "timing": "HIP events around every launch of the kernel inside the timed region (the stream it runs on)"}
    if extra:
        r.update(extra)
    return r


def _pmc_traffic(name, units_per_launch):
    """HBM bytes per launch from a committed PMC record (profiles/<name>_pmc_traffic.json, scripts/pmc_collect.py:
    FETCH_SIZE x2 + WRITE_SIZE from separate rocprofv3 passes), scaled to this launch's batch members; only when the
    record was measured on the kernel source in the tree."""
    try:
        import hashlib
        root = os.path.dirname(os.path.abspath(__file__))
        rec = json.load(open(os.path.join(root, "profiles", name + "_pmc_traffic.json")))
        h = hashlib.sha256()
        for src in rec["kernel_source_files"]:
            h.update(open(os.path.join(root, "xitorch_amd", "csrc", src), "rb").read())
        if h.hexdigest() != rec["kernel_source_sha256"] or not rec.get("B"):
            return {"traffic": None, "traffic_note": "PMC record is stale (kernel source changed since it was measured): "
                                                     "not reported"}
        return {"traffic": rec["hbm_bytes_per_launch"] * units_per_launch / rec["B"],
                "traffic_note": "PMC (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes) of a standalone launch of the "
                                "same kernel source over %d batch members, scaled to this launch's (profiles/%s_pmc_traffic.json)"
                                % (rec["B"], name)}
    except Exception:                                   # noqa: no record -> traffic stays null
        return {}


# ------------------------------------------------------------------------------------------------- c3 / c3g
def _banded_problem(dev, B, N, hb, offset):
    from xitorch_amd import synthetic as syn
    import xitorch_amd as xa
    band = syn.banded(B, N, hb=hb, device=dev, batch_offset=offset)
    xs = syn.banded_rhs_solution(B, N, device=dev, batch_offset=offset)
    with torch.no_grad():
        rhs = xa.BandedLinearOperator(band).mm(xs)
    return band, xs, rhs


def config_c3(args, dev, group, world, rank, fence):
    import xitorch_amd as xa
    from xitorch_amd.linalg import solve
    B, N, hb = args.cfg_batch or 256, 65536, 63
    band, xs, rhs = _banded_problem(dev, B, N, hb, rank * B)
    band.requires_grad_()
    opts = dict(rtol=1e-10, atol=1e-12, posdef=True)
    traces = []

    def step(timed):
        ev = []
        tr = {"k1_events": ev if timed else None}
        btr = {"k1_events": ev if timed else None}
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("error")                 # a convergence warning would be a failed step
            x = solve(xa.BandedLinearOperator(band), rhs, method="bicgstab", process_group=group, trace=tr,
                      bck_options=dict(method="bicgstab", process_group=group, trace=btr, **opts), **opts)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            gband, = torch.autograd.grad(x.sum(), (band,))
            torch.cuda.synchronize()
        t2 = time.perf_counter()
        if timed:
            traces.append((tr, btr, ev, (t1 - t0) * 1e3, (t2 - t1) * 1e3, x.detach(), float(gband[0, hb, 0])))
        del gband
        return x.detach()
    elapsed, step_ms, x = _timed(step, args.steps, args.warmup, fence, group, dev)
    tr, btr, _, fwd_ms, bwd_ms, _, _ = traces[-1]
    events = [(a, b) for t in traces for (a, b, pc, nb) in t[2]]
    avg, nl = _avg_ms(events)
    s, c = 8, 1
    apply_bytes = B * (2 * hb + 1) * N * s + 2 * B * N * c * s
    it_bytes = 2 * apply_bytes + 16 * B * N * c * s            # 2 applies + (10 reads + 6 writes) vectors
    niter_f, niter_b = tr["niter"], btr.get("niter")
    err = (x - xs).abs().max().item()
    return {
        "metric": "linear systems/s of linalg.solve(bicgstab) with implicit backward, banded bw=127 N=65536 batch=256 "
                  "+ banded-apply GB/s (roofline.achieved)",
        "value": B * world * args.steps / elapsed, "unit": "systems/s (forward + backward)",
        "ms_per_step": elapsed / args.steps * 1e3, "dtype": "f64",
        "config": {"workload": "BASELINE configs[2]: linalg.solve BiCGStab on banded (bw=127) N=65536 batch=%d (%d per "
                               "GPU) fp64 with implicit backward (adjoint BiCGStab + band gradient), rtol=1e-10"
                               % (B * world, B), "global_batch": B * world, "batch_per_gpu": B,
                   "forward_ms": fwd_ms, "backward_ms": bwd_ms, "niter_forward": niter_f, "niter_backward": niter_b,
                   "applies_forward": tr["napply"], "applies_backward": btr.get("napply"),
                   "forward_only_systems_per_s": B * world / (fwd_ms * 1e-3)},
        "roofline": _roofline(apply_bytes, avg, nl, "banded_mm_kernel (xk_banded_mm: A x and A^T x)", {
            **_pmc_traffic("c3", B),
            "bicgstab_iteration": {"algorithmic_bytes": it_bytes, "achieved_GBps_forward":
                                   it_bytes * niter_f / (fwd_ms * 1e-3) / 1e9,
                                   "frac_forward": it_bytes * niter_f / (fwd_ms * 1e-3) / 1e9 / PEAK_HBM,
                                   "note": "2 applies + (10 reads + 6 writes) of B x N vectors per iteration (SURVEY 8d) "
                                           "over the whole forward solve, host syncs and set-up included"}}),
        "check": {"ok": bool(err < 1e-7), "max_err_vs_manufactured_solution": err,
                  "tolerance": "1e-7 absolute on |x*| <= 1 (rtol 1e-10 times the condition number)"},
        "step_ms": step_ms,
    }


def config_c3g(args, dev, group, world, rank, fence):
    import xitorch_amd as xa
    from xitorch_amd.linalg import native_krylov as nk
    B, N, hb = args.cfg_batch or 256, 65536, 63
    m = 30 if not args.gmres_restart else max(args.max_niter, args.gmres_restart + 1)   # total Arnoldi steps allowed
    band, xs, rhs = _banded_problem(dev, B, N, hb, rank * B)
    A = xa.BandedLinearOperator(band)
    traces = []

    def step(timed):
        ev = []
        tr = {"k1_events": ev if timed else None}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                # (30 un-restarted steps stop at ~7e-7: reported, not hidden)
            x = nk.gmres(A, rhs, rtol=1e-10, atol=1e-12, posdef=True, max_niter=m, process_group=group, trace=tr,
                         **({"restart": args.gmres_restart} if args.gmres_restart else {}))
        if timed:
            traces.append((tr, ev))
        return x
    elapsed, step_ms, x = _timed(step, args.steps, args.warmup, fence, group, dev)
    tr = traces[-1][0]
    events = [(a, b) for t in traces for (a, b, pc, nb) in t[1]]
    avg, nl = _avg_ms(events)
    apply_bytes = B * (2 * hb + 1) * N * 8 + 2 * B * N * 8
    err = (x - xs).abs().max().item()
    return {
        "metric": "linear systems/s of native gmres (max_niter=%d), banded bw=127 N=65536 batch=256 + banded-apply GB/s" % m,
        "value": B * world * args.steps / elapsed, "unit": "systems/s", "ms_per_step": elapsed / args.steps * 1e3,
        "dtype": "f64",
        "config": {"workload": "BASELINE configs[2]'s systems through gmres (solve.py:326-433), reference semantics "
                               "(true residual every iteration), max_niter=%d%s" %
                               (m, (", restart=%d" % args.gmres_restart) if args.gmres_restart else ""),
                   "global_batch": B * world, "batch_per_gpu": B, "arnoldi_steps": tr.get("arnoldi_steps"),
                   "applies": tr["napply"], "host_syncs": tr.get("host_syncs"), "converged": tr["converged"],
                   "best_resid": tr["best_resid"]},
        "roofline": _roofline(apply_bytes, avg, nl, "banded_mm_kernel (xk_banded_mm)"),
        "check": {"ok": bool(err < 1e-4), "max_err_vs_manufactured_solution": err, "converged": tr["converged"]},
        "step_ms": step_ms,
    }


# ------------------------------------------------------------------------------------------------- c4
def config_c4(args, dev, group, world, rank, fence):
    import xitorch_amd as xa
    from xitorch_amd import synthetic as syn
    from xitorch_amd.optimize import rootfinder
    B, N = args.cfg_batch or 64, 8192
    A = syn.root_matrix(B, N, device=dev, batch_offset=rank * B) * 2.0
    y0 = torch.zeros(B, N, dtype=torch.float64, device=dev)
    mv_events = []
    record = [False]

    def fcn(y, A_):
        op = xa.LinearOperator.m(A_, is_hermitian=False)
        if record[0] and not torch.is_grad_enabled():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            z = op.mv(y)
            e1.record()
            mv_events.append((e0, e1))
        else:
            z = op.mv(y)
        return torch.tanh(z + 0.1) + y / 2.0
    Ad = A.clone().requires_grad_()
    traces = []

    def step(timed):
        record[0] = timed
        tr = {}
        t0 = time.perf_counter()
        y = rootfinder(fcn, y0, params=(Ad,), method="broyden1", alpha=-1.0, max_rank=32, f_tol=1e-8,
                       process_group=group, trace=tr,
                       bck_options=dict(method="bicgstab", posdef=True, rtol=1e-10, process_group=group))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        g, = torch.autograd.grad(y.sum(), (Ad,))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        record[0] = False
        if timed:
            traces.append((tr, (t1 - t0) * 1e3, (t2 - t1) * 1e3, float(g[0, 0, 0])))
        del g
        return y.detach()
    elapsed, step_ms, y = _timed(step, args.steps, args.warmup, fence, group, dev)
    tr, fwd_ms, bwd_ms, _ = traces[-1]
    avg, nl = _avg_ms(mv_events)
    s = 8
    fcn_bytes = B * N * N * s + 2 * B * N * s
    r = tr.get("rank") or 0
    L = B * N
    with torch.no_grad():
        fn = fcn(y, A).norm().item()
    return {
        "metric": "batch members/s of optimize.rootfinder(broyden1) + implicit backward, tanh(A y) N=8192 "
                  "+ function-evaluation matvec GB/s (roofline.achieved)",
        "value": B * world * args.steps / elapsed, "unit": "members/s (forward + backward)",
        "ms_per_step": elapsed / args.steps * 1e3, "dtype": "f64",
        "config": {"workload": "BASELINE configs[3] per-GPU shard: rootfinder broyden1 on tanh(A y + 0.1) + y/2, N=8192 "
                               "batch=%d (%d per GPU) fp64, alpha=-1, max_rank=32, f_tol=1e-8; backward: implicit, "
                               "linalg.solve bicgstab rtol=1e-10" % (B * world, B),
                   "global_batch": B * world, "batch_per_gpu": B, "forward_ms": fwd_ms, "backward_ms": bwd_ms,
                   "nfev": tr.get("nfev"), "niter": tr.get("niter"), "rank": r,
                   "host_syncs_forward": tr.get("host_syncs", tr.get("nfev")),
                   "Gm_mv_algorithmic_bytes_at_final_rank": 2 * r * L * s + 2 * L * s},
        "roofline": _roofline(fcn_bytes, avg, nl, "dense_mm_rows<double,1> (xk_dense_mm: the evaluation's A_b y_b)",
                              _pmc_traffic("c4", B)),
        "check": {"ok": bool(fn < 1e-6 * (B ** 0.5)), "fnorm_at_returned_root": fn,
                  "note": "the returned iterate is the one BEFORE the converged one (quirk Q1): |f| slightly above f_tol"},
        "step_ms": step_ms,
    }


# ------------------------------------------------------------------------------------------------- c5 / c5w
def _config_c5(args, dev, group, world, rank, fence, p, label):
    import xitorch_amd as xa
    from xitorch_amd import synthetic as syn, kernels as XK
    from xitorch_amd.linalg import symeig
    B, N = args.cfg_batch or 16, 32768
    kind = "S1" if p <= 6 else "S1:%d" % p
    mat = torch.empty((B, N, N), dtype=torch.float32, device=dev)
    syn.dense_symmetric(B, N, kind, dtype=torch.float32, device=dev, out=mat, batch_offset=rank * B)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    exact = syn.spectrum(kind, N, device=dev)[:p]
    traces = []
    XK.prefill_timing_events(2 * 40 * (args.steps + 1))

    def step(timed):
        ev = []
        tr = {"k1_events": ev if timed else None}
        with torch.no_grad():
            evals, X = symeig(A, neig=p, mode="lowest", method="davidson", min_eps=2e-3, rng_device="device",
                              max_niter=60, process_group=group, reserve_cus=args.reserve_cus, trace=tr)
        if timed:
            traces.append((tr, ev))
        return evals, X
    elapsed, step_ms, (evals, X) = _timed(step, args.steps, args.warmup, fence, group, dev)
    ntimed = len(traces)
    # extra (outside the timed region), wide block on symmetric storage only: the same workload on the FULL-MATRIX
    # matrix-core kernel K1w — what an operator whose storage is only allclose-symmetric gets — with its own roofline
    general = None
    if traces[-1][0].get("panel_kernel") == "K1sw" and not args.no_general_extra:
        A.symmetric_storage = False
        g_el, g_ms, _ = _timed(step, 2, 1, fence, group, dev)
        A.symmetric_storage = True
        gtr = traces[-1][0]
        gev = [(a, b) for t in traces[ntimed:] for (a, b, pc, nb) in t[1] if pc == p]
        gnb = [nb for t in traces[ntimed:] for (a, b, pc, nb) in t[1] if pc == p][0]
        gavg, gnl = _avg_ms(gev)
        gbytes = gnb * N * N * 4 + 2 * gnb * N * p * 4
        general = {"value": B * world * p * 2 / g_el, "unit": "eigpairs/s", "ms_per_step": g_el / 2 * 1e3, "steps": 2,
                   "panel_kernel": gtr.get("panel_kernel"),
                   "roofline": _roofline(gbytes, gavg, gnl, "dense_wide_cols<float,1,Mfma16f> (K1w, full matrix)"),
                   "note": "same workload, full-matrix kernel (storage not exactly symmetric): twice the bytes at a higher "
                           "fraction of the HBM roofline, slower per call"}
        del traces[ntimed:]
    tr = traces[-1][0]
    events = [(a, b) for t in traces for (a, b, pc, nb) in t[1] if pc == p]
    nbl = [nb for t in traces for (a, b, pc, nb) in t[1] if pc == p][0]
    avg, nl = _avg_ms(events)
    s = 4
    symm = tr.get("panel_kernel") in ("K1s", "K1sw")
    full_bytes = nbl * N * N * s + 2 * nbl * N * p * s
    tri_bytes = nbl * N * (N + 1) // 2 * s + 2 * nbl * N * p * s
    flops = 2.0 * nbl * N * N * p
    err = (evals.double() - exact).abs().max().item()
    extra = {"operators_per_launch": nbl, "TFLOPs": flops / (avg * 1e-3) / 1e12,
             "frac_of_fp32_matrix_peak": flops / (avg * 1e-3) / 1e12 / PEAK_F32,
             "full_matrix_equivalent_GBps": full_bytes / (avg * 1e-3) / 1e9}
    if symm:
        extra["priced_on"] = "the bytes the kernel must move: upper triangle + panels (the operator is exactly symmetric)"
    if tr.get("panel_kernel") == "K1sw":
        extra["note"] = ("K1sw streams half the bytes of K1w for the same flops (16 flop / B at P = 16 fp32, just under the "
                         "ridge): matrix pipe 0.54 busy at the clock the chip holds, the traffic half alone 5.9 TB/s, the MFMA "
                         "half alone 75 % of the pipe (profiles/r04_k1sw_coop_pmc_probe.json); `frac` stays priced on HBM as "
                         "SURVEY 8d defines it, `frac_of_fp32_matrix_peak` is the other roofline; the full-matrix kernel K1w "
                         "reaches 0.76 of HBM peak on twice the bytes and is 25-30 % slower per call")
        # HBM bytes from the PMC counters (separate rocprofv3 --pmc passes over scripts/k1sw_bench.py), only from a record
        # measured on the kernel source in the tree
        try:
            import hashlib
            import json as _json
            root = os.path.dirname(os.path.abspath(__file__))
            rec = _json.load(open(os.path.join(root, "profiles", "k1sw_pmc_traffic.json")))
            h = hashlib.sha256()
            for name in rec["kernel_source_files"]:
                h.update(open(os.path.join(root, "xitorch_amd", "csrc", name), "rb").read())
            if h.hexdigest() == rec["kernel_source_sha256"] and rec.get("B"):
                extra["traffic"] = rec["hbm_bytes_per_launch"] * nbl / rec["B"]
                extra["traffic_note"] = ("PMC (FETCH_SIZE x2 + WRITE_SIZE, separate passes) of a standalone launch of the same "
                                         "kernel source over %d operators, scaled to this launch's (profiles/"
                                         "k1sw_pmc_traffic.json)" % rec["B"])
            else:
                extra["traffic_note"] = "PMC record is stale (kernel source changed since it was measured): not reported"
        except Exception:                                   # noqa: no record -> traffic stays null
            pass
    return {
        "metric": "eigpairs/s of symeig(davidson) fp32 N=32768 (per-GPU shard of batch 128) + panel-product GB/s",
        "value": B * world * p * args.steps / elapsed, "unit": "eigpairs/s", "ms_per_step": elapsed / args.steps * 1e3,
        "dtype": "f32",
        "config": {"workload": "BASELINE configs[4] per-GPU shard: linalg.symeig davidson lowest-%d on "
                               "MatrixLinearOperator N=32768 batch=%d (%d per GPU) fp32, %s, min_eps=2e-3"
                               % (p, B * world, B, label), "global_batch": B * world, "batch_per_gpu": B,
                   "iterations_per_step": tr["niter"], "panel_products_per_step": tr["napply"],
                   "panel_kernel": tr.get("panel_kernel"), "batch_groups": tr.get("groups"),
                   "orth_redo": tr.get("orth_redo")},
        "roofline": _roofline(tri_bytes if symm else full_bytes, avg, nl,
                              {"K1s": "dense_symm_tiles<float,6> (K1s)",
                               "K1sw": "dense_symm_wide7_kernel (K1sw: triangle once, v_mfma_f32_16x16x4_f32 for both "
                                       "products); its fold runs beside it on the group's stream",
                               "K1w": "dense_wide_cols<float,1,Mfma16f> (K1w, v_mfma_f32_16x16x4_f32)"}.get(
                                   tr.get("panel_kernel"), str(tr.get("panel_kernel"))), extra),
        "general_k1w": general,
        "check": {"ok": bool(err < 5e-4), "max_eval_err_vs_closed_form": err,
                  "tolerance": "5e-4 absolute on a spectrum of scale 100 in fp32 (eps32 |A| ~ 1e-5, resid^2 / gap)"},
        "step_ms": step_ms,
    }


def config_c5(args, dev, group, world, rank, fence):
    return _config_c5(args, dev, group, world, rank, fence, 6, "6-column eigen-block (upper-triangle kernel K1s)")


def config_c5w(args, dev, group, world, rank, fence):
    return _config_c5(args, dev, group, world, rank, fence, 16,
                      "16-column eigen-block on the matrix cores, as BASELINE states the config")


CONFIGS = {"c3": config_c3, "c3g": config_c3g, "c4": config_c4, "c5": config_c5, "c5w": config_c5w}


def run(args, dev, group, world, rank, fence, backend):
    out = CONFIGS[args.config](args, dev, group, world, rank, fence)
    out.update({"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "data": "synthetic"})
    out["config"]["parallelism"] = "batch-sharded x%d (weak: one shard per GPU)" % world
    out["config"]["comm_backend"] = backend
    order = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "config", "roofline", "general_k1w", "check", "step_ms"]
    return {k: out[k] for k in order if k in out}


if __name__ == "__main__":
    sys.exit("run through bench.py: python bench.py --config c3|c3g|c4|c5|c5w")


# ------------------------------------------------------------------------------------------------- c0 + the `configs` block
def config_c0(args, dev, cpu):
    """BASELINE configs[0] on the GPU: symeig lowest-6 of ONE dense symmetric 512 x 512 fp64 operator (the reference's
    benchmarks_solve.py shape), `davidson` and the reference's default `exacteig`, as a bench record."""
    from xitorch_amd import LinearOperator as _LO, synthetic as _syn
    from xitorch_amd.linalg import symeig as _symeig
    m1 = _syn.random_symmetric(512, -1.0, 1.0, 123)
    Ag = _LO.m(m1.to(dev), is_hermitian=True)
    ref = torch.linalg.eigvalsh(m1)[:6]

    def timed(method, steps):
        def call():
            tr = {}
            with torch.no_grad():
                kw = dict(method="davidson", min_eps=1e-8, trace=tr) if method == "davidson" else {}
                ev, _ = _symeig(Ag, neig=6, mode="lowest", **kw)
            torch.cuda.synchronize()
            return ev, tr
        call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ev, tr = call()
        return (time.perf_counter() - t0) / steps, ev, tr
    t_d, ev_d, tr_d = timed("davidson", max(args.cfg_steps, 5))
    t_x, ev_x, _ = timed("exacteig", max(args.cfg_steps, 5))
    napply = tr_d.get("napply") or tr_d.get("niter") or 1
    by = 512 * 512 * 8 + 2 * 512 * 6 * 8
    rec = {"metric": "eigpairs/s of symeig lowest-6, dense symmetric N=512 batch=1 fp64",
           "value": 6 / t_d, "unit": "eigpairs/s", "ms_per_step": t_d * 1e3, "steps": max(args.cfg_steps, 5), "dtype": "f64",
           "config": {"workload": "BASELINE configs[0]: linalg.symeig lowest-6 of one dense symmetric 512 x 512 fp64 operator "
                                  "(benchmarks_solve.py shape), method=davidson, min_eps=1e-8",
                      "iterations_per_step": tr_d.get("niter"), "basis_size": tr_d.get("basis_size")},
           "exacteig": {"value": 6 / t_x, "unit": "eigpairs/s", "ms_per_step": t_x * 1e3,
                        "note": "the reference's default method (symeig method=None) on the native dense eigensolver",
                        "max_abs_err_vs_lapack": (ev_x.cpu() - ref).abs().max().item()},
           "roofline": {"bound": "latency", "kernel": "panel product of ONE 2 MB operator (xk_dense_mm)", "peak": 8000.0,
                        "unit": "GB/s", "algorithmic_bytes_per_launch": by, "launches_per_step": napply,
                        "achieved": by * napply / t_d / 1e9, "frac": by * napply / t_d / 1e9 / 8000.0, "traffic": None,
                        "avg_launch_ms": None,
                        "note": "one 512 x 512 operator is 2 MB: the call is a chain of a few hundred small launches, "
                                "bound by launch latency, not by HBM; achieved = all panel bytes of the call / the call"},
           "check": {"ok": bool((ev_d.cpu() - ref).abs().max().item() < 1e-9),
                     "max_abs_err_vs_lapack": (ev_d.cpu() - ref).abs().max().item()}}
    if cpu and "config1_n512_b1" in cpu and "cpu_davidson_ms" in cpu["config1_n512_b1"]:
        c1 = cpu["config1_n512_b1"]
        rec["cpu_baseline"] = {"value": c1["cpu_davidson_eigpairs_per_s"], "unit": "eigpairs/s", "cores": cpu.get("cores"),
                               "kind": "port", "sample": "the config itself (oracle davidson, %d threads): %.1f ms per call; "
                               "oracle exacteig: %.1f ms per call" % (cpu.get("cores", 0), c1["cpu_davidson_ms"],
                                                                     c1["cpu_exacteig_ms"]),
                               "exacteig_value": c1["cpu_exacteig_eigpairs_per_s"]}
    return rec


def configs_block(args, dev, cpu):
    """The other BASELINE.json configs, measured in this process after the headline (rank 0 of a one-GPU run): same
    fences and event timing as their own bench lines (`--config c3|c4|c5w`, bench_secondary.py), `--cfg-steps` timed
    steps each, every record with its roofline and a CPU baseline (oracle on a bounded sample)."""
    import copy
    import sys
    bs = sys.modules[__name__]
    out = {}

    def fence():
        torch.cuda.synchronize()
    sub = copy.copy(args)
    sub.steps, sub.warmup, sub.no_general_extra, sub.cfg_batch = args.cfg_steps, 1, True, 0
    keep_roof = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_note",
                 "algorithmic_bytes_per_launch", "avg_launch_ms", "launches_timed", "TFLOPs", "frac_of_fp32_matrix_peak")
    try:
        out["c0"] = config_c0(args, dev, cpu)
    except Exception as err:
        out["c0"] = {"error": repr(err)}
    plan = [("c3", bs.config_c3, lambda: bs.cpu_baseline_c3(args.cpu_threads)),
            ("c4", bs.config_c4, lambda: bs.cpu_baseline_c4(args.cpu_threads)),
            ("c5w", bs.config_c5w, lambda: bs.cpu_baseline_c5(args.cpu_threads, 16))]
    for name, fnc, cpufn in plan:
        try:
            torch.cuda.empty_cache()
            line = fnc(sub, dev, None, 1, 0, fence)
            rec = {k: line[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "check") if k in line}
            rec["steps"], rec["warmup"] = sub.steps, sub.warmup
            rec["config"] = {k: v for k, v in line["config"].items()
                             if k in ("workload", "global_batch", "batch_per_gpu", "forward_ms", "backward_ms",
                                      "iterations_per_step", "niter_forward", "niter_backward", "nfev", "panel_kernel")}
            rec["roofline"] = {k: line["roofline"].get(k) for k in keep_roof if k in line["roofline"]}
            if not args.no_cpu_baseline:
                try:
                    rec["cpu_baseline"] = cpufn()
                except Exception as err:
                    rec["cpu_baseline"] = {"error": repr(err)}
            out[name] = rec
        except Exception as err:            # a secondary record never costs the headline line
            out[name] = {"error": repr(err)}
        torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------- CPU baselines
# The oracle (CPU restatement of the reference's loops, oracle/*.py — test infrastructure) timed on the host cores on a
# BOUNDED sample of each config's workload.  Only bench.py's cpu_baseline leg calls these (rank 0, one GPU).
def _cpu_median(fn, budget_s, nmin=2, nmax=20):
    ts, t_all, last = [], time.time(), None
    while len(ts) < nmin or (time.time() - t_all < budget_s and len(ts) < nmax):
        t0 = time.time()
        last = fn()
        ts.append(time.time() - t0)
    return sorted(ts)[len(ts) // 2], len(ts), last


def cpu_baseline_c3(threads, B=8, N=65536, hb=63, budget_s=8.0):
    """configs[2] on the CPU: oracle BiCGStab on the banded operator, forward solve + the adjoint solve of the implicit
    backward (xitorch/linalg/solve.py:186-195: one more solve with A^H), same tolerances as the GPU step."""
    from oracle import ops as oops, solve as osolve
    from xitorch_amd import synthetic as syn
    torch.set_num_threads(threads)
    band = syn.banded(B, N, hb)
    xs = syn.banded_rhs_solution(B, N)
    op = oops.BandedOp(band)
    rhs = op._mm(xs)
    opH = oops.FuncOp(op.shape, op.dtype, mm=op._rmm, rmm=op._mm)
    tr = {}

    def step():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x = osolve.bicgstab(op, rhs, rtol=1e-10, atol=1e-12, posdef=True, trace=tr)
            g = osolve.bicgstab(opH, torch.ones_like(x), rtol=1e-10, atol=1e-12, posdef=True)
        return x, g
    t, n, (x, _) = _cpu_median(step, budget_s)
    return {"value": B / t, "unit": "systems/s (forward + backward)", "cores": threads, "kind": "port",
            "sample": "oracle bicgstab (torch-CPU restatement of solve.py:192-324) on the banded operator bw=%d N=%d, "
                      "batch=%d of the config's 256: forward solve + adjoint solve of the implicit backward, rtol=1e-10; "
                      "median of %d runs (%.2f s each, %d iterations forward)" % (2 * hb + 1, N, B, n, t, tr.get("niter", -1)),
            "seconds": t, "max_err_vs_manufactured_solution": (x - xs).abs().max().item(),
            "full_config_seconds_extrapolated": t * 256.0 / B,
            "extrapolation": "x %.0f to the config's 256 systems (members are independent; the cost is linear in the "
                             "batch at fixed N and band width, the sample has the config's N and bw)" % (256.0 / B)}


def cpu_baseline_c4(threads, B=2, N=8192, budget_s=8.0):
    """configs[3] on the CPU: oracle Broyden (rootsolver.py:15-206) on tanh(A y + 0.1) + y/2, then the implicit
    backward's linear solve with J^H (optimize/rootfinder.py backward -> linalg.solve), J applied in closed form."""
    from oracle import ops as oops, solve as osolve, rootfinder as oroot
    from xitorch_amd import synthetic as syn
    torch.set_num_threads(threads)
    A = syn.root_matrix(B, N) * 2.0
    y0 = torch.zeros(B, N, dtype=torch.float64)

    def fcn(y, A_):
        return torch.tanh(torch.matmul(A_, y.unsqueeze(-1)).squeeze(-1) + 0.1) + y / 2.0
    tr = {}

    def step():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            y = oroot.broyden1(fcn, y0, (A,), alpha=-1.0, max_rank=32, f_tol=1e-8, trace=tr)
            t_ = torch.tanh(torch.matmul(A, y.unsqueeze(-1)) + 0.1)                  # (B, N, 1)
            d = 1.0 - t_ * t_
            JH = oops.FuncOp((B, N, N), torch.float64, mm=lambda v: torch.matmul(A.transpose(-2, -1), d * v) + v / 2.0,
                             rmm=lambda v: d * torch.matmul(A, v) + v / 2.0)
            g = osolve.bicgstab(JH, torch.ones(B, N, 1, dtype=torch.float64), rtol=1e-10, atol=1e-12, posdef=True)
        return y, g
    t, n, (y, _) = _cpu_median(step, budget_s)
    return {"value": B / t, "unit": "members/s (forward + backward)", "cores": threads, "kind": "port",
            "sample": "oracle broyden1 (alpha=-1, max_rank=32, f_tol=1e-8) on tanh(A y + 0.1) + y/2, N=%d batch=%d of the "
                      "shard's 64, + the implicit backward's J^H solve (oracle bicgstab rtol=1e-10, J in closed form); "
                      "median of %d runs (%.2f s each, nfev=%s)" % (N, B, n, t, tr.get("nfev")),
            "seconds": t, "fnorm_at_returned_root": fcn(y, A).norm().item(),
            "full_config_seconds_extrapolated": t * 512.0 / B,
            "extrapolation": "x %.0f to the per-GPU shard of 64 members, x %.0f to the config's 512 (the batch is ONE flat "
                             "system in the reference, so its iteration count is that of the slowest member; the dense "
                             "products are linear in the batch; the sample has the config's N)" % (64.0 / B, 512.0 / B)}


def cpu_baseline_c5(threads, p, B=1, N=8192, budget_s=8.0):
    """configs[4] on the CPU: oracle davidson in fp32 with the config's block width on a smaller operator."""
    from oracle import ops as oops, symeig as osym
    from xitorch_amd import synthetic as syn
    torch.set_num_threads(threads)
    kind = "S1" if p <= 6 else "S1:%d" % p
    mat = syn.dense_symmetric(B, N, kind, dtype=torch.float32)
    op = oops.DenseOp(mat, True)
    tr = {}

    def step():
        return osym.davidson(op, p, "lowest", min_eps=2e-3, max_niter=60, trace=tr)
    t, n, (ev, _) = _cpu_median(step, budget_s)
    exact = syn.spectrum(kind, N)[:p]
    return {"value": B * p / t, "unit": "eigpairs/s", "cores": threads, "kind": "port",
            "sample": "oracle davidson (torch-CPU restatement of symeig.py:100-227) fp32 lowest-%d, dense symmetric %s "
                      "N=%d batch=%d (the shard is 16 x 32768^2: 68.7 GB), min_eps=2e-3; median of %d runs (%.2f s each, "
                      "%s iterations)" % (p, kind, N, B, n, t, tr.get("niter")),
            "seconds": t, "max_eval_err_vs_closed_form": (ev.double() - exact).abs().max().item(),
            "full_config_seconds_extrapolated": t * (32768.0 / N) ** 2 * 16.0 / B,
            "extrapolation": "x %.0f to the per-GPU shard (16 operators of 32768^2: the panel product is O(B N^2) per "
                             "iteration; the iteration count of the closed-form spectrum is the same at both orders), x 8 "
                             "more to the config's 128" % ((32768.0 / N) ** 2 * 16.0 / B)}


def cpu_baseline_c0(cores):
    """BASELINE configs[0] exactly: N=512, batch=1, lowest 6, fp64 — the reference's CPU case,
    oracle `davidson` and `exacteig` on `cores` threads, with the native calls on the GPU beside them when there is one."""
    from oracle import ops as oops, symeig as osym
    from oracle.symeig import exacteig as _oexact
    from xitorch_amd import synthetic
    m1 = synthetic.random_symmetric(512, -1.0, 1.0, 123)
    op1 = oops.DenseOp(m1, True)

    def med(f, nmin=3, budget=4.0):
        ts, t_all = [], time.time()
        while len(ts) < nmin or (time.time() - t_all < budget and len(ts) < 50):
            t0 = time.time()
            r = f()
            ts.append(time.time() - t0)
        return sorted(ts)[len(ts) // 2], r
    t_dav, (ev_d, _) = med(lambda: osym.davidson(op1, 6, "lowest", min_eps=1e-8))
    t_ex, (ev_x, _) = med(lambda: _oexact(op1, 6, "lowest", None))
    c1 = {"workload": "BASELINE configs[0]: symeig lowest-6, dense symmetric N=512 batch=1 fp64 "
                      "(benchmarks_solve.py shape), oracle on %d CPU threads" % cores,
          "cpu_davidson_ms": t_dav * 1e3, "cpu_exacteig_ms": t_ex * 1e3,
          "cpu_davidson_eigpairs_per_s": 6 / t_dav, "cpu_exacteig_eigpairs_per_s": 6 / t_ex,
          "max_abs_diff_davidson_vs_exacteig": (ev_d - ev_x).abs().max().item()}
    if torch.cuda.is_available():
        from xitorch_amd import LinearOperator as _LO
        from xitorch_amd.linalg import symeig as _symeig
        Ag = _LO.m(m1.cuda(), is_hermitian=True)

        def gpu_call():
            with torch.no_grad():
                r = _symeig(Ag, neig=6, mode="lowest", method="davidson", min_eps=1e-8)
            torch.cuda.synchronize()
            return r
        gpu_call()
        t_g, (ev_g, _) = med(gpu_call, nmin=5, budget=2.0)

        def gpu_exact():
            with torch.no_grad():
                r = _symeig(Ag, neig=6, mode="lowest")           # method=None -> exacteig, like the reference
            torch.cuda.synchronize()
            return r
        gpu_exact()
        t_gx, (ev_gx, _) = med(gpu_exact, nmin=5, budget=2.0)
        c1.update(gpu_exacteig_ms=t_gx * 1e3, gpu_exacteig_eigpairs_per_s=6 / t_gx,
                  max_abs_diff_gpu_exacteig_vs_cpu_exacteig=(ev_gx.cpu() - ev_x).abs().max().item(),
                  gpu_exacteig_note="the reference's default method (symeig method=None, benchmarks_solve.py) on "
                                    "the native dense eigensolver: only the wanted pairs are computed")
        c1.update(gpu_davidson_ms=t_g * 1e3, gpu_davidson_eigpairs_per_s=6 / t_g,
                  max_abs_diff_gpu_vs_cpu_exacteig=(ev_g.cpu() - ev_x).abs().max().item(),
                  gpu_note="one 512 x 512 operator: latency-bound (a few dozen small launches per iteration), "
                           "not what the GPU path is built for")
    return c1
