"""K1s A/B alone on the GPU, one process: the round-2 kernel (scripts/_ab/libsymm_r02.so, LDS float atomics, one tile
per workgroup) against the shipped one (phase-rotated deterministic accumulation, runs of L slabs per workgroup).
   python scripts/k1s_ab.py [B N P] [f32]  ->  one JSON line per variant (interleaved repetitions, median)"""
import os, sys, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xitorch_amd import kernels as K
from xitorch_amd._capi import fn, ptr, stream_ptr
dev = torch.device("cuda:0")
nums = [int(v) for v in sys.argv[1:] if v.isdigit()]
B, N, P = nums[:3] if len(nums) >= 3 else (32, 16384, 6)
dtype = torch.float32 if "f32" in sys.argv else torch.float64
sfx = "f32" if dtype == torch.float32 else "f64"
torch.manual_seed(0)
A = torch.empty(B, N, N, dtype=dtype, device=dev)
for b in range(B):
    R = torch.randn(N, N, dtype=dtype, device=dev)
    A[b] = torch.triu(R) + torch.triu(R, 1).transpose(-2, -1)
    del R
X = torch.randn(B, P, N, dtype=dtype, device=dev)
ref = torch.matmul(X[:2].double(), A[:2].double())
es = A.element_size()
tri_bytes = B * (N * (N + 1) // 2) * es + 2 * B * P * N * es

old = None
so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ab", "libsymm_r02.so")
if os.path.exists(so):
    old = ctypes.CDLL(so)
    f = getattr(old, "xk_dense_symm_" + sfx)
    Pv, I, Lg = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
    f.restype = I
    f.argtypes = [Pv, Pv, Pv, Pv, Lg, I, I, I, Lg, Lg, Lg, Lg, Lg, Lg, Pv]
    old.xk_dense_symm_workspace_elems.restype = Lg
    old.xk_dense_symm_workspace_elems.argtypes = [I, I, I, I]
    nws_old = old.xk_dense_symm_workspace_elems(B, N, P, es)
    ws_old = torch.empty(nws_old, dtype=dtype, device=dev)

    def run_old(Y):
        rc = f(ptr(A), ptr(X), ptr(Y), ptr(ws_old), nws_old, B, N, P, A.stride(1), A.stride(0), X.stride(1), X.stride(0),
               Y.stride(1), Y.stride(0), stream_ptr())
        assert rc == 0, rc


def run_new(Y):
    K.dense_symm(A, X, out=Y)


variants = []
if old is not None:
    variants.append(("r02", None))
for L in (1, 2, 4, 8):
    variants.append(("r03_L%d" % L, L))
Y = torch.empty_like(X)
times = {name: [] for name, _ in variants}
errs, repro = {}, {}
for name, L in variants:
    if L is not None:
        fn("xk_dense_symm_tune")(1, L)
    runner = run_old if L is None else run_new
    runner(Y); torch.cuda.synchronize()
    errs[name] = ((Y[:2].double() - ref).abs().max() / ref.abs().max()).item()
    Y1 = Y.clone()
    runner(Y); torch.cuda.synchronize()
    repro[name] = bool(torch.equal(Y, Y1))
for rep in range(5):
    for name, L in variants:
        if L is not None:
            fn("xk_dense_symm_tune")(1, L)
        runner = run_old if L is None else run_new
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            runner(Y)
        e1.record(); torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 3)
fn("xk_dense_symm_tune")(1, 1)
for name, L in variants:
    ts = sorted(times[name])
    ms = ts[len(ts) // 2]
    print(json.dumps({"variant": name, "B": B, "N": N, "P": P, "dtype": str(dtype), "ms_median": round(ms, 4),
                      "ms_all": [round(t, 3) for t in times[name]], "TBps_triangle": round(tri_bytes / ms / 1e9, 3),
                      "frac": round(tri_bytes / ms / 1e9 / 8.0, 4), "relerr": errs[name],
                      "bit_reproducible": repro[name]}), flush=True)
