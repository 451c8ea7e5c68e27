"""S2 un-restarted (64 x 16384^2 fp64, basis grows to 582 vectors) by how the Rayleigh-Ritz solver beyond order 128 is
run beside the panel products: one workgroup per matrix (W = -1) or the per-step kernels over W workgroups per matrix,
CUs left free by the panel stream, one batch group instead of two.  One process, operators generated once.
    python scripts/s2_k3_variants.py W:reserve_cus[:overlap] ...      e.g.  -1:64  2:64  8:64  8:128  4:64:False"""
import os, sys, json, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xitorch_amd as xa
from xitorch_amd import synthetic as syn, _capi, kernels as K
from xitorch_amd.linalg import symeig
dev = torch.device("cuda:0")
B, N, p = 64, 16384, 6
mat = torch.empty((B, N, N), dtype=torch.float64, device=dev)
syn.dense_symmetric(B, N, "S2", device=dev, out=mat)
A = xa.LinearOperator.m(mat, is_hermitian=True)
exact = syn.spectrum("S2", N, device=dev)[:p]
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
for spec in sys.argv[1:]:
    parts = spec.split(":")
    W, reserve = int(parts[0]), int(parts[1])
    overlap = "auto" if len(parts) < 3 else (parts[2] == "True")
    tune(0, W)
    ev = []
    tr = {"k1_events": ev}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        evals, X = symeig(A, neig=p, mode="lowest", method="davidson", min_eps=1e-8, rng_device="device", max_niter=3000,
                          reserve_cus=reserve, overlap=overlap, trace=tr)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    k1 = sum(a.elapsed_time(b) for (a, b, pc, nb) in ev) * 1e-3
    print(json.dumps({"W": W, "reserve_cus": reserve, "overlap": str(overlap), "ms": round(t * 1e3, 1), "niter": tr["niter"],
                      "basis": tr["basis_size"], "groups": tr.get("groups"), "panel_share": round(k1 / t, 4),
                      "k1_ms_total": round(k1 * 1e3, 1), "err": (evals - exact).abs().max().item(),
                      "fallbacks": tr["k3_fallbacks"]}), flush=True)
tune(0, 0)
