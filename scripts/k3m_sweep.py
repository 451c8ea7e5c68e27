"""K3g/K3m: the Rayleigh-Ritz eigensolver beyond order 128 with the tridiagonalisation spread over W workgroups per
matrix (one launch per Householder step) against the one-workgroup kernel and rocSOLVER, by order, batch and W.
    python scripts/k3m_sweep.py [quick]"""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xitorch_amd import kernels as K, _capi
dev = torch.device("cuda:0")
def tune(what, value):
    """launch shape of K3g through the Python layer's module attributes (arguments of the C entry points since r04);
    what 2 / 3 — leave the final kernel after a phase / skip parts of the step kernel: wrong results by construction —
    exist only in a library built with -DXK_DEBUG (xk_debug_small_eigh_big)"""
    if what == 0:
        K.K3G_WG = int(value)
    elif what == 1:
        K.K3G_THREADS = int(value)
    else:
        try:
            _capi.fn("xk_debug_small_eigh_big")(what, value)
        except Exception:                                   # noqa: the shipped library has no such symbol
            if value:
                raise SystemExit("phase timings need a -DXK_DEBUG build of xk_eigh_big.hip")
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"


def t_of(f, n=3):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[n // 2] * 1e3


for dtype in (torch.float64, torch.float32) if quick else (torch.float64, torch.float32):
    for B in (32, 4):
        for k in ((192, 384, 582) if quick else (130, 192, 256, 384, 512, 582, 640, 768)):
            g = torch.Generator().manual_seed(k)
            R = torch.randn(B, k, k, dtype=torch.float64, generator=g).to(dev)
            T = (R + R.transpose(-2, -1)).to(dtype).contiguous()
            ref = torch.linalg.eigvalsh(T.double())[:, :6]
            rec = {"dtype": str(dtype).split(".")[1], "B": B, "k": k, "p": 6}
            if dtype == torch.float64:
                rec["library_eigh_ms"] = round(t_of(lambda: torch.linalg.eigh(T)), 3)
            for threads in (512, 256):
                tune(1, threads)
                for W in ((0, 2, 4, 8, 16, 32) if threads == 512 else (8, 16, 32)):
                    if B * max(W, 1) > 1024:
                        continue
                    tune(0, W)
                    try:
                        ms = t_of(lambda: K.small_eigh_big(T, k, 6))
                        lam, Y, info = K.small_eigh_big(T, k, 6)
                        err = (lam.double() - ref).abs().max().item()
                        # residual + orthogonality of the vectors
                        Yt = Y.double()
                        Rm = torch.matmul(T.double(), Yt.transpose(1, 2)) - Yt.transpose(1, 2) * lam.double()[:, None, :]
                        G = torch.matmul(Yt, Yt.transpose(1, 2)) - torch.eye(6, dtype=torch.float64, device=dev)
                        rec["W%d_t%d" % (W, threads)] = {"ms": round(ms, 3), "eval_err": err, "resid": Rm.abs().max().item(),
                                                         "orth": G.abs().max().item(), "flag": int(info.max())}
                    except Exception as e:                                   # noqa
                        rec["W%d_t%d" % (W, threads)] = {"error": str(e)[:200]}
            tune(0, 0); tune(1, 512)
            print(json.dumps(rec), flush=True)

# where the time of the final (one workgroup per matrix) kernel goes: stop after bisection / inverse iteration / checks
for B, k in ((32, 582), (32, 384), (32, 192)):
    R = torch.randn(B, k, k, dtype=torch.float64, device=dev)
    T = (R + R.transpose(-2, -1)).contiguous()
    rec = {"phases_of_final_kernel": True, "B": B, "k": k}
    for stop, name in ((2, "tridiag+bisection"), (3, "+inverse_iteration"), (5, "+checks"), (0, "+back_transformation")):
        tune(2, stop)
        rec[name] = round(t_of(lambda: K.small_eigh_big(T, k, 6), 5), 3)
    tune(2, 0)
    print(json.dumps(rec), flush=True)
