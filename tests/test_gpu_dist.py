"""-m gpu: the sharded solvers with a REAL process group — two processes on the one GPU of the test box
(gloo backend: RCCL refuses two ranks on one device; the code path in davidson is the same all-reduce MAX).
Each rank owns half of the batch; the iteration count, eigenvalues and residuals must equal the unsharded run,
with and without the two-group pipeline."""
import os
import socket
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


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
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import xitorch_amd as xa
        from xitorch_amd import dist as xd, synthetic
        from xitorch_amd.linalg.native_eig import davidson
        dev = torch.device("cuda:0")
        B, N, neig = 6, 768, 4
        mat = synthetic.dense_symmetric(B, N, "S1", device=dev)
        # make the members converge at different speeds: the slowest one (on rank 1) decides for everybody
        mat = mat * torch.linspace(1.0, 1.6, B, dtype=torch.float64, device=dev).reshape(B, 1, 1)
        lo, hi = xd.shard_range(B, world, rank)
        V0 = torch.randn(B, N, neig, dtype=torch.float64, generator=torch.Generator().manual_seed(7)).to(dev)
        out = {}
        for overlap in (False, True):
            tr_f, tr_s = {}, {}
            ev_f, _ = davidson(xa.LinearOperator.m(mat, True), neig, "lowest", min_eps=1e-8, trace=tr_f, overlap=False,
                               V0=V0)
            ev_s, X_s = davidson(xa.LinearOperator.m(mat[lo:hi].contiguous(), True), neig, "lowest", min_eps=1e-8,
                                 trace=tr_s, overlap=overlap, process_group=dist.group.WORLD, V0=V0[lo:hi])
            R = torch.matmul(mat[lo:hi], X_s) - X_s * ev_s.unsqueeze(-2)
            out[overlap] = dict(niter=(tr_s["niter"], tr_f["niter"]), groups=tr_s["groups"],
                                err=(ev_s - ev_f[lo:hi]).abs().max().item(), resid=R.abs().max().item(),
                                hist=max(abs(a - b) for a, b in zip(tr_s["resid_history"], tr_f["resid_history"])))
        # ---- (r06) NO `V0=`: every rank draws the start block of the WHOLE batch from the reference's
        # seed and keeps its own members, so member b of the sharded run is member b of the unsharded run — same
        # iterates, same iteration count — for the CPU draw (the reference's CPU path) and the device draw
        nov = {}
        for rdev in ("cpu", "device"):
            tr_f, tr_s = {}, {}
            ev_f, _ = davidson(xa.LinearOperator.m(mat, True), neig, "lowest", min_eps=1e-8, trace=tr_f, overlap=False,
                               rng_device=rdev)
            ev_s, _ = davidson(xa.LinearOperator.m(mat[lo:hi].contiguous(), True), neig, "lowest", min_eps=1e-8,
                               trace=tr_s, overlap=False, process_group=dist.group.WORLD, rng_device=rdev)
            nov[rdev] = dict(niter=(tr_s["niter"], tr_f["niter"]), err=(ev_s - ev_f[lo:hi]).abs().max().item(),
                             hist=max(abs(a - b) for a, b in zip(tr_s["resid_history"], tr_f["resid_history"])))
        out["no_v0"] = nov
        # ---- (r05) uneven shards: 3 operators over 2 ranks = 2 + 1.  With overlap=True rank 0 could run two
        # batch groups and rank 1 only one — one status all-reduce per group and iteration would pair up collectives of
        # different steps (or hang).  The ranks agree on the number of groups before any group exists.
        B3 = 3
        l3, h3 = xd.shard_range(B3, world, rank)
        tr_u, tr_uf = {}, {}
        ev_uf, _ = davidson(xa.LinearOperator.m(mat[:B3].contiguous(), True), neig, "lowest", min_eps=1e-8, trace=tr_uf,
                            overlap=False, V0=V0[:B3])
        ev_u, _ = davidson(xa.LinearOperator.m(mat[l3:h3].contiguous(), True), neig, "lowest", min_eps=1e-8, trace=tr_u,
                           overlap=True, process_group=dist.group.WORLD, V0=V0[l3:h3])
        out["uneven"] = dict(niter=(tr_u["niter"], tr_uf["niter"]), groups=tr_u["groups"], local=h3 - l3,
                             err=(ev_u - ev_uf[l3:h3]).abs().max().item())
        # ---- sharded Krylov solves and the sharded Broyden driver (same process group) ----------------
        from xitorch_amd.linalg import native_krylov as nk
        from xitorch_amd.optimize import native_root as nr
        from tests import cases
        g = torch.Generator().manual_seed(21)
        nb, n = 4, 256
        R = torch.rand(nb, n, n, dtype=torch.float64, generator=g)
        Amat = (0.1 * R + torch.diag(torch.linspace(1.0, 4.0, n, dtype=torch.float64))).to(dev)
        Amat = Amat * torch.linspace(1.0, 3.0, nb, dtype=torch.float64, device=dev).reshape(nb, 1, 1)
        Bm = torch.rand(nb, n, 2, dtype=torch.float64, generator=g).to(dev)
        l2, h2 = xd.shard_range(nb, world, rank)
        kry = {}
        for meth in ("bicgstab", "cg", "gmres"):
            herm = meth == "cg"
            Am = (Amat + Amat.transpose(-2, -1)) * 0.5 if herm else Amat
            kw = dict(rtol=1e-10, atol=1e-12, posdef=True)
            if meth == "gmres":
                kw["max_niter"] = 80
            tf, ts = {}, {}
            Xf = getattr(nk, meth)(xa.LinearOperator.m(Am, herm), Bm, trace=tf, **kw)
            Xs = getattr(nk, meth)(xa.LinearOperator.m(Am[l2:h2].contiguous(), herm), Bm[l2:h2].contiguous(), trace=ts,
                                   process_group=dist.group.WORLD, **kw)
            kry[meth] = dict(niter=(ts["niter"], tf["niter"]), err=(Xs - Xf[l2:h2]).abs().max().item())
        out["krylov"] = kry
        # a rank whose shard of the right-hand side is all zeros (an implicit backward where some batch members
        # receive no gradient) must stay in the collectives: the zero-rhs shortcut is a group decision
        Bz = Bm.clone()
        Bz[:2] = 0.0                                             # rank 0's shard
        zs = {}
        for meth in ("bicgstab", "cg"):
            herm = meth == "cg"
            Am = (Amat + Amat.transpose(-2, -1)) * 0.5 if herm else Amat
            Xs = getattr(nk, meth)(xa.LinearOperator.m(Am[l2:h2].contiguous(), herm), Bz[l2:h2].contiguous(),
                                   process_group=dist.group.WORLD, rtol=1e-10, atol=1e-12, posdef=True)
            ref = torch.linalg.solve(Am[l2:h2], Bz[l2:h2])
            zs[meth] = (Xs - ref).abs().max().item()
        from xitorch_amd.linalg import solve as xsolve
        Xs = xsolve(xa.LinearOperator.m(Amat[l2:h2].contiguous(), False), Bz[l2:h2].contiguous(), method="bicgstab",
                    process_group=dist.group.WORLD, rtol=1e-10, atol=1e-12, posdef=True)
        zs["frontend"] = (Xs - torch.linalg.solve(Amat[l2:h2], Bz[l2:h2])).abs().max().item()
        out["zero_shard"] = zs
        fcn, y0, (Ar,) = cases.root_inputs(dict(kind="tanh", nbatch=4, n=64))
        tf, ts = {}, {}
        yf = nr.broyden1(fcn, y0.to(dev), (Ar.to(dev),), alpha=-1.0, f_tol=1e-9, trace=tf)
        ys = nr.broyden1(fcn, y0[l2:h2].to(dev), (Ar[l2:h2].to(dev),), alpha=-1.0, f_tol=1e-9, trace=ts,
                         process_group=dist.group.WORLD)
        out["broyden"] = dict(niter=(ts["niter"], tf["niter"]), nfev=(ts["nfev"], tf["nfev"]),
                              err=(ys - yf[l2:h2]).abs().max().item())
        # ---- ONE operator, fewer members than ranks: row-block sharding (last bullet).  Each rank streams
        # its rows of A on the native K1 kernel, the p-column panel is all-gathered, the rest of the native Davidson runs
        # replicated and must be identical on both ranks and equal to the unsharded run
        Nr = 1030
        S1 = synthetic.dense_symmetric(1, Nr, "S1", device=dev)
        As = xa.RowShardedMatrixLinearOperator.from_full(S1, dist.group.WORLD, is_hermitian=True)
        trs, trf = {}, {}
        ev_s, X_s = davidson(As, 4, "lowest", min_eps=1e-8, trace=trs)
        ev_f, _ = davidson(xa.LinearOperator.m(S1, True), 4, "lowest", min_eps=1e-8, trace=trf)
        Rr = torch.matmul(S1, X_s) - X_s * ev_s.unsqueeze(-2)
        xg = torch.randn(1, Nr, 5, dtype=torch.float64, generator=torch.Generator().manual_seed(3)).to(dev)
        out["rows"] = dict(err=(ev_s - ev_f).abs().max().item(), niter=(trs["niter"], trf["niter"]),
                           resid=Rr.abs().max().item(), evals=ev_s.cpu().tolist(),
                           mm=(As.mm(xg) - S1 @ xg).abs().max().item(), rmm=(As.rmm(xg) - S1 @ xg).abs().max().item(),
                           local_rows=As.local.shape[-2])
        results[rank] = out
    finally:
        dist.destroy_process_group()


def test_sharded_davidson_two_ranks_one_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world = 2
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    results = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, results)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    for rank in range(world):
        for overlap in (False, True):
            r = results[rank][overlap]
            assert r["niter"][0] == r["niter"][1], r            # lock step with the global stopping rule
            assert r["groups"] == (2 if overlap else 1)
            assert r["err"] < 1e-10 * 160 and r["resid"] < 1e-7, r
            assert r["hist"] < 1e-6, r                           # the all-reduced residual IS the global one
        for rdev, r in results[rank]["no_v0"].items():
            assert r["niter"][0] == r["niter"][1] and r["err"] < 1e-12 * 160 and r["hist"] < 1e-9, (rdev, r)
        r = results[rank]["uneven"]
        assert r["groups"] == 1 and r["niter"][0] == r["niter"][1] and r["err"] < 1e-10 * 160, r
        for meth, r in results[rank]["krylov"].items():
            assert r["niter"][0] == r["niter"][1], (meth, r)      # global stopping / best-iterate decisions
            assert r["err"] < 1e-9, (meth, r)
        for meth, err in results[rank]["zero_shard"].items():
            assert err < 1e-8, (meth, err)
        r = results[rank]["broyden"]                              # the whole batch is ONE flat system (Q4)
        assert r["niter"][0] == r["niter"][1] and r["nfev"][0] == r["nfev"][1] and r["err"] < 1e-9, r
        r = results[rank]["rows"]                                 # one operator split by row blocks over the ranks
        assert r["local_rows"] == 515 and r["mm"] < 1e-11 and r["rmm"] < 1e-11, r
        assert r["err"] < 1e-10 and abs(r["niter"][0] - r["niter"][1]) <= 1 and r["resid"] < 1e-7, r
        assert r["evals"] == results[0]["rows"]["evals"]          # replicated: bit-identical on every rank


def test_device_comm_through_the_c_abi_single_rank(dev):
    """xk_comm_* / xk_allreduce_*: the RCCL communicator and the in-place all-reduce behind the C ABI (csrc/xk_comm.hip),
    on the one GPU of the test box: a communicator of one rank — library lookup, unique id, init, SUM / MAX / MIN of
    float64 and float32 on the default and on a side stream, size query, destroy.  (Two ranks need two devices: RCCL
    refuses a second rank on the same GPU; the multi-rank path verifies itself against known answers at creation,
    dist.device_comm.)"""
    from xitorch_amd import _capi
    from xitorch_amd.dist import DeviceComm
    assert _capi.fn("xk_comm_available")() == 1
    uid = DeviceComm.unique_id()
    assert len(uid) == 128 and any(b != 0 for b in uid)
    comm = DeviceComm.create(uid, 1, 0, dev)
    try:
        assert comm.size() == (1, 0)
        for dtype in (torch.float64, torch.float32):
            t = torch.tensor([3.0, -1.0, 7.5, 0.0, 1e-30], dtype=dtype, device=dev)
            for op in ("sum", "max", "min"):
                u = t.clone()
                comm.allreduce_(u, op)
                torch.cuda.synchronize()
                assert torch.equal(u, t), (dtype, op)
        side = torch.cuda.Stream(device=dev)
        big = torch.arange(100000, dtype=torch.float64, device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            v = big * 2.0
            comm.allreduce_(v, "sum")             # stream-ordered behind the kernel that produced v
            w = v + 1.0
        side.synchronize()
        assert torch.equal(w, big * 2.0 + 1.0)
        with pytest.raises(Exception):
            comm.allreduce_(torch.zeros(3, dtype=torch.int32, device=dev), "sum")
    finally:
        comm.close()
    # a process group of one rank never creates a communicator: the helpers are no-ops
    from xitorch_amd import dist as xd
    assert xd.device_comm(None, dev) is None
    t = torch.ones(2, dtype=torch.float64, device=dev)
    assert xd.allreduce_max_(t, None) is t and xd.allreduce_sum_(t, None) is t
