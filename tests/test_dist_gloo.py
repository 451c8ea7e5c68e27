"""not-gpu: the N>1 path on CPU — 2 processes, gloo backend.

What is exercised: the batch partition, the all-reduce helpers that complete the global decisions
(MAX for the eigen/linear solvers' stopping rule, SUM for Broyden's inner products), the sharded
quasi-Newton driver end to end (linear-mixing model, which needs no device kernel), and that the
all-reduced residual reproduces the reference's *global* stopping rule iteration by iteration.
"""
import os
import socket
import pytest
import torch
import torch.multiprocessing as mp

from tests import cases


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, results):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from xitorch_amd import dist as xd, synthetic
        from xitorch_amd.optimize import native_root as nr
        from oracle import ops as oops, symeig as osym
        grp = dist.group.WORLD
        out = {}

        # 1. partition
        total = 5
        spans = [xd.shard_range(total, world, r) for r in range(world)]
        out["spans"] = spans

        # 2. helpers
        t = torch.tensor([float(rank + 1), 10.0 - rank], dtype=torch.float64)
        out["max"] = xd.allreduce_max_(t.clone(), grp).tolist()
        out["sum"] = xd.allreduce_sum_(t.clone(), grp).tolist()
        assert xd.allreduce_max_(t.clone(), None).tolist() == t.tolist()

        out["agree"] = (xd.all_ranks_agree_true(True, "cpu", grp), xd.all_ranks_agree_true(rank == 0, "cpu", grp),
                        xd.all_ranks_agree_true(False, "cpu", grp), xd.all_ranks_agree_true(rank == 0, "cpu", None))

        # 3. sharded quasi-Newton driver == unsharded driver (the whole batch is ONE flat system, Q4)
        nb, n = 4, 12
        fcn, y0, (A,) = cases.root_inputs(dict(kind="tanh", nbatch=nb, n=n))
        lo, hi = xd.shard_range(nb, world, rank)
        tr_s, tr_f = {}, {}
        y_shard = nr.linearmixing(fcn, y0[lo:hi], (A[lo:hi],), alpha=-1.0, f_tol=1e-10, x_tol=1e-10, maxiter=300,
                                  process_group=grp, trace=tr_s)
        y_full = nr.linearmixing(fcn, y0, (A,), alpha=-1.0, f_tol=1e-10, x_tol=1e-10, maxiter=300, trace=tr_f)
        out["root_err"] = (y_shard - y_full[lo:hi]).abs().max().item()
        out["root_iters"] = (tr_s["niter"], tr_f["niter"], tr_s["nfev"], tr_f["nfev"])
        red = nr._Reduce(grp)
        v = torch.arange(1.0, 7.0, dtype=torch.float64)
        mine = v[rank * 3:(rank + 1) * 3]
        out["dot"] = (float(red.dot(mine, mine)), float(torch.dot(v, v)), float(red.norm(mine)), float(v.norm()),
                      red.total_numel(mine))

        # 4. global stopping rule: MAX over ranks of the shard residuals == the unsharded residual history
        B, N = 4, 96
        mat = synthetic.dense_symmetric(B, N, "S1")
        lo, hi = xd.shard_range(B, world, rank)
        trf, trs = {}, {}
        osym.davidson(oops.DenseOp(mat, True), 3, "lowest", min_eps=1e-8, trace=trf)
        # run the shard for as many iterations as the global rule demands: disable its local stop
        osym.davidson(oops.DenseOp(mat[lo:hi].contiguous(), True), 3, "lowest", min_eps=0.0,
                      max_niter=trf["niter"], trace=trs)
        stops = []
        for it in range(trf["niter"]):
            st = torch.tensor([trs["resid_history"][it], 0.0], dtype=torch.float64)
            xd.allreduce_max_(st, grp)
            stops.append(st[0].item())
        out["stops"] = stops
        out["own_hist"] = list(trs["resid_history"][:trf["niter"]])
        out["full_niter"] = trf["niter"]
        results[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world_size_2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, port, results), nprocs=world, join=True)
    assert len(results) == world
    for rank in range(world):
        r = results[rank]
        assert r["spans"] == [(0, 3), (3, 5)]
        assert r["max"] == [2.0, 10.0] and r["sum"] == [3.0, 19.0]
        # a shortcut around a loop with collectives is taken only when EVERY rank wants it
        assert r["agree"] == (True, False, False, rank == 0)
        assert r["root_err"] < 1e-12
        assert r["root_iters"][0] == r["root_iters"][1] and r["root_iters"][2] == r["root_iters"][3]
        d = r["dot"]
        assert abs(d[0] - d[1]) < 1e-12 and abs(d[2] - d[3]) < 1e-12 and d[4] == 6
        # identical start blocks are NOT guaranteed between the sharded and the unsharded draw (the RNG
        # stream is consumed batch-major), so only the rule itself is asserted: both ranks see the same
        # reduced value and stop at the same iteration
    # every rank sees the same reduced residual sequence, and it is the element-wise max of the shards'
    assert results[0]["stops"] == results[1]["stops"]
    expect = [max(a, b) for a, b in zip(results[0]["own_hist"], results[1]["own_hist"])]
    assert results[0]["stops"] == expect


def _worker8(rank, world, port, results):
    """world_size 8 on an UNEVEN batch (13 members: shards of 2,2,2,2,2,1,1,1): the sharded quasi-Newton driver against
    the unsharded one, and the Davidson stopping rule completed by the MAX all-reduce."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from xitorch_amd import dist as xd, synthetic
        from xitorch_amd.optimize import native_root as nr
        from oracle import ops as oops, symeig as osym
        grp = dist.group.WORLD
        out = {}
        nb, n = 13, 10
        out["span"] = xd.shard_range(nb, world, rank)
        lo, hi = out["span"]
        fcn, y0, (A,) = cases.root_inputs(dict(kind="tanh", nbatch=nb, n=n))
        tr_s, tr_f = {}, {}
        y_shard = nr.linearmixing(fcn, y0[lo:hi], (A[lo:hi],), alpha=-1.0, f_tol=1e-10, x_tol=1e-10, maxiter=400,
                                  process_group=grp, trace=tr_s)
        y_full = nr.linearmixing(fcn, y0, (A,), alpha=-1.0, f_tol=1e-10, x_tol=1e-10, maxiter=400, trace=tr_f)
        out["root_err"] = (y_shard - y_full[lo:hi]).abs().max().item()
        out["root_iters"] = (tr_s["niter"], tr_f["niter"], tr_s["nfev"], tr_f["nfev"])
        # Davidson: the unsharded run's iteration count must be what the all-reduced shard residuals dictate
        B, N = 13, 64
        mat = synthetic.dense_symmetric(B, N, "S1")
        trf, trs = {}, {}
        osym.davidson(oops.DenseOp(mat, True), 2, "lowest", min_eps=1e-8, trace=trf)
        osym.davidson(oops.DenseOp(mat[lo:hi].contiguous(), True), 2, "lowest", min_eps=0.0,
                      max_niter=trf["niter"], trace=trs)
        hist = list(trs["resid_history"][:trf["niter"]])
        hist += [0.0] * (trf["niter"] - len(hist))           # a shard whose basis became square stops early
        st = torch.tensor(hist, dtype=torch.float64)
        xd.allreduce_max_(st, grp)
        out["stops"] = st.tolist()
        out["own_hist"] = hist
        out["full_niter"] = trf["niter"]
        results[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_world_size_8_gloo_uneven_shards():
    world = 8
    port = _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker8, args=(world, port, results), nprocs=world, join=True)
    assert len(results) == world
    sizes = [results[r]["span"][1] - results[r]["span"][0] for r in range(world)]
    assert sizes == [2, 2, 2, 2, 2, 1, 1, 1] and results[7]["span"][1] == 13
    for r in range(world):
        assert results[r]["root_err"] < 1e-12
        it = results[r]["root_iters"]
        assert it[0] == it[1] and it[2] == it[3]
        assert results[r]["stops"] == results[0]["stops"]
    expect = [max(results[r]["own_hist"][i] for r in range(world)) for i in range(results[0]["full_niter"])]
    assert results[0]["stops"] == expect


def _worker_rows(rank, world, port, results):
    """Row-block sharding of ONE dense operator over the ranks (B < G: SURVEY 8e, last bullet): products, full matrix,
    gradient slice, and the eigensolver on the sharded operator == on the unsharded one."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import xitorch_amd as xa
        from xitorch_amd import dist as xd, synthetic
        from oracle import symeig as osym
        grp = dist.group.WORLD
        out = {}
        N = 101                                              # uneven row blocks
        g = torch.Generator().manual_seed(5)
        mat = torch.randn(2, N, N, dtype=torch.float64, generator=g)
        x = torch.randn(2, N, 3, dtype=torch.float64, generator=g)
        A = xa.RowShardedMatrixLinearOperator.from_full(mat, grp, is_hermitian=False)
        lo, hi = xd.shard_range(N, world, rank)
        out["rows"] = (lo, hi, tuple(A.local.shape))
        out["mm"] = (A.mm(x) - mat @ x).abs().max().item()
        out["rmm"] = (A.rmm(x) - mat.transpose(-2, -1) @ x).abs().max().item()
        out["mv"] = (A.mv(x[..., 0]) - (mat @ x[..., :1])[..., 0]).abs().max().item()
        out["full"] = (A.fullmatrix() - mat).abs().max().item()
        # the replicated loss differentiates into this rank's row block only
        loc = mat[..., lo:hi, :].clone().requires_grad_()
        Ag = xa.RowShardedMatrixLinearOperator(loc, N, grp)
        gl, = torch.autograd.grad((Ag.mm(x) ** 2).sum(), (loc,))
        full = mat.clone().requires_grad_()
        gf, = torch.autograd.grad(((full @ x) ** 2).sum(), (full,))
        out["grad"] = (gl - gf[..., lo:hi, :]).abs().max().item()
        # (r05) gradients w.r.t. the replicated INPUT: the dual collectives (all-reduce SUM for A x, all-gather
        # for A^H x) must make them complete and identical on every rank — also through a chain of both products, and
        # the row-block gradient must survive an x that itself depends on the block
        xg = x.clone().requires_grad_()
        xf = x.clone().requires_grad_()
        w = torch.linspace(0.5, 1.5, N, dtype=torch.float64).reshape(N, 1)
        gx_mm, = torch.autograd.grad((A.mm(xg) ** 2 * w).sum(), (xg,))
        gx_mm_f, = torch.autograd.grad(((mat @ xf) ** 2 * w).sum(), (xf,))
        gx_rmm, = torch.autograd.grad((A.rmm(xg) ** 2 * w).sum(), (xg,))
        gx_rmm_f, = torch.autograd.grad(((mat.transpose(-2, -1) @ xf) ** 2 * w).sum(), (xf,))
        gx_ch, gl_ch = torch.autograd.grad((Ag.rmm(torch.tanh(Ag.mm(xg))) * w).sum(), (xg, loc))
        gx_ch_f, gf_ch = torch.autograd.grad(((full.transpose(-2, -1) @ torch.tanh(full @ xf)) * w).sum(), (xf, full))
        out["grad_x"] = max((gx_mm - gx_mm_f).abs().max().item(), (gx_rmm - gx_rmm_f).abs().max().item(),
                            (gx_ch - gx_ch_f).abs().max().item(), (gl_ch - gf_ch[..., lo:hi, :]).abs().max().item())
        out["grad_x_bits"] = gx_ch.sum().item()
        # one symmetric operator (B = 1 < G), eigensolver replicated on every rank, operator stream split by rows
        S = synthetic.dense_symmetric(1, 96, "S1")
        As = xa.RowShardedMatrixLinearOperator.from_full(S, grp, is_hermitian=True)
        trs, trf = {}, {}
        ev_s, X_s = osym.davidson(As, 3, "lowest", min_eps=1e-8, trace=trs)
        ev_f, X_f = osym.davidson(xa.LinearOperator.m(S, is_hermitian=True), 3, "lowest", min_eps=1e-8, trace=trf)
        out["eig"] = ((ev_s - ev_f).abs().max().item(), trs["niter"], trf["niter"],
                      (S @ X_s - X_s * ev_s.unsqueeze(-2)).abs().max().item())
        out["evals"] = ev_s.tolist()
        with pytest.raises(RuntimeError):
            xa.RowShardedMatrixLinearOperator(mat[..., :5, :], N, grp)       # wrong block height
        results[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_row_block_sharded_operator_gloo(world):
    port = _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker_rows, args=(world, port, results), nprocs=world, join=True)
    assert len(results) == world
    covered = []
    for rank in range(world):
        r = results[rank]
        covered.append(r["rows"][:2])
        assert r["rows"][2] == (2, r["rows"][1] - r["rows"][0], 101)
        for key in ("mm", "rmm", "mv", "full", "grad"):
            assert r[key] < 1e-12, (key, r[key])
        assert r["grad_x"] < 1e-10, r["grad_x"]
        assert r["grad_x_bits"] == results[0]["grad_x_bits"]  # replicated parameters cannot drift apart
        err, it_s, it_f, resid = r["eig"]
        assert err < 1e-11 and abs(it_s - it_f) <= 1 and resid < 1e-7, r["eig"]
        assert r["evals"] == results[0]["evals"]             # every rank holds the same replicated answer
    assert covered[0][0] == 0 and covered[-1][1] == 101 and all(covered[i][1] == covered[i + 1][0] for i in range(world - 1))


def _worker_host_solvers(rank, world, port, results):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import xitorch_amd as xa
        from xitorch_amd import dist as xd, synthetic
        from xitorch_amd.linalg import native_krylov as nk
        from xitorch_amd.linalg.native_eig import davidson
        from xitorch_amd.optimize import native_root as nr
        grp = dist.group.WORLD
        out = {}
        # ---- Davidson, NO V0: every rank draws the start block of the whole batch and keeps its members
        B, N, neig = 3, 160, 3                                     # 3 members over 2 ranks: 2 + 1
        mat = synthetic.dense_symmetric(B, N, "S1") * torch.linspace(1.0, 1.5, B, dtype=torch.float64).reshape(B, 1, 1)
        lo, hi = xd.shard_range(B, world, rank)
        trf, trs = {}, {}
        ev_f, _ = davidson(xa.LinearOperator.m(mat, True), neig, "lowest", min_eps=1e-8, trace=trf)
        ev_s, X_s = davidson(xa.LinearOperator.m(mat[lo:hi].contiguous(), True), neig, "lowest", min_eps=1e-8, trace=trs,
                             process_group=grp)
        R = torch.matmul(mat[lo:hi], X_s) - X_s * ev_s.unsqueeze(-2)
        out["davidson"] = dict(niter=(trs["niter"], trf["niter"]), err=(ev_s - ev_f[lo:hi]).abs().max().item(),
                               resid=R.abs().max().item(),
                               hist=max(abs(a - b) for a, b in zip(trs["resid_history"], trf["resid_history"])))
        # ---- Krylov solves: the stopping test and the best-iterate rule are global decisions
        g = torch.Generator().manual_seed(21)
        nb, n = 3, 64
        Rm = torch.rand(nb, n, n, dtype=torch.float64, generator=g)
        Amat = (0.1 * Rm + torch.diag(torch.linspace(1.0, 4.0, n, dtype=torch.float64)))
        Amat = Amat * torch.linspace(1.0, 3.0, nb, dtype=torch.float64).reshape(nb, 1, 1)
        Bm = torch.rand(nb, n, 2, dtype=torch.float64, generator=g)
        l2, h2 = xd.shard_range(nb, world, rank)
        kry = {}
        for meth in ("bicgstab", "cg", "gmres"):
            herm = meth == "cg"
            Am = (Amat + Amat.transpose(-2, -1)) * 0.5 if herm else Amat
            kw = dict(rtol=1e-10, atol=1e-12, posdef=True)
            tf, ts = {}, {}
            Xf = getattr(nk, meth)(xa.LinearOperator.m(Am, herm), Bm, trace=tf, **kw)
            Xs = getattr(nk, meth)(xa.LinearOperator.m(Am[l2:h2].contiguous(), herm), Bm[l2:h2].contiguous(), trace=ts,
                                   process_group=grp, **kw)
            kry[meth] = dict(niter=(ts["niter"], tf["niter"]), err=(Xs - Xf[l2:h2]).abs().max().item())
        out["krylov"] = kry
        # a rank whose right-hand sides are all zero stays in the collectives
        Bz = Bm.clone()
        Bz[:2] = 0.0
        Xz = nk.bicgstab(xa.LinearOperator.m(Amat[l2:h2].contiguous(), False), Bz[l2:h2].contiguous(), process_group=grp,
                         rtol=1e-10, atol=1e-12, posdef=True)
        out["zero_shard"] = (Xz - torch.linalg.solve(Amat[l2:h2], Bz[l2:h2])).abs().max().item()
        # ---- Broyden: the whole batch is ONE flat system
        fcn, y0, (Ar,) = cases.root_inputs(dict(kind="tanh", nbatch=3, n=32))
        tf, ts = {}, {}
        yf = nr.broyden1(fcn, y0, (Ar,), alpha=-1.0, f_tol=1e-9, trace=tf)
        ys = nr.broyden1(fcn, y0[l2:h2], (Ar[l2:h2],), alpha=-1.0, f_tol=1e-9, trace=ts, process_group=grp)
        out["broyden"] = dict(niter=(ts["niter"], tf["niter"]), nfev=(ts["nfev"], tf["nfev"]),
                              err=(ys - yf[l2:h2]).abs().max().item())
        results[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_solvers_in_host_memory_two_ranks_gloo():
    """(r06) The batch-sharded PRODUCT drivers end to end on CPU ranks: with the host-memory drivers
    (xitorch_amd/linalg/host_*.py, the host branch of the Broyden model) davidson / cg / bicgstab / gmres / broyden1 run
    under gloo exactly as they run under RCCL on device tensors — same process-group plumbing (`dist.allreduce_*`), same
    global decisions.  3 members over 2 ranks (2 + 1): sharded == unsharded, same iteration counts, the Davidson start
    block WITHOUT `V0=` (slice of the global draw)."""
    world = 2
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    results = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_host_solvers, args=(r, world, port, results)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(280)
        assert p.exitcode == 0
    for rank in range(world):
        r = results[rank]["davidson"]
        assert r["niter"][0] == r["niter"][1] and r["err"] < 1e-10 * 160 and r["resid"] < 1e-7 and r["hist"] < 1e-6, r
        for meth, k in results[rank]["krylov"].items():
            assert k["niter"][0] == k["niter"][1] and k["err"] < 1e-9, (meth, k)
        assert results[rank]["zero_shard"] < 1e-8
        b = results[rank]["broyden"]
        assert b["niter"][0] == b["niter"][1] and b["nfev"][0] == b["nfev"][1] and b["err"] < 1e-9, b
