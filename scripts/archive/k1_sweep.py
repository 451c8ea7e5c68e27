"""GPU micro-benchmark / tuning sweep for K1 (dense operator-panel product).
Run on the GPU box:  python scripts/k1_sweep.py [--full]
Prints one line per variant with achieved algorithmic GB/s.
"""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from xitorch_amd.kernels import dense_mm

dev = torch.device("cuda:0")
torch.manual_seed(0)


def check(B, M, N, P, dtype, trans):
    A = torch.randn(B, M, N, dtype=dtype, device=dev)
    X = torch.randn(B, P, M if trans else N, dtype=dtype, device=dev)
    Y = dense_mm(A, X, trans=trans)
    Aop = A.transpose(-2, -1) if trans else A
    ref = torch.matmul(Aop.double(), X.double().transpose(-2, -1)).transpose(-2, -1)
    err = (Y.double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-300)
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    ok = err < tol
    print("check B=%d M=%d N=%d P=%d %s trans=%d relerr=%.2e %s" % (B, M, N, P, str(dtype)[6:], trans, err, "OK" if ok else "FAIL"), flush=True)
    return ok


def timeit(f, reps):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    full = "--full" in sys.argv
    allok = True
    for (B, M, N, P, dt, tr) in [
        (2, 256, 256, 6, torch.float64, 0), (3, 100, 130, 1, torch.float64, 0), (2, 77, 201, 5, torch.float64, 0),
        (2, 1000, 1000, 2, torch.float64, 0), (1, 512, 512, 11, torch.float64, 0), (2, 256, 384, 6, torch.float32, 0),
        (2, 99, 131, 3, torch.float32, 0), (2, 256, 384, 6, torch.float64, 1), (3, 100, 131, 4, torch.float64, 1),
        (2, 300, 200, 7, torch.float32, 1), (2, 2048, 2048, 6, torch.float64, 1)]:
        allok &= check(B, M, N, P, dt, tr)
    print("ALL_CHECKS", "OK" if allok else "FAIL", flush=True)

    res = []
    N = 16384
    B = 64 if full else 16
    A = torch.empty(B, N, N, dtype=torch.float64, device=dev).uniform_(-1, 1)
    gb = lambda P, s=8: (B * N * N * s + 2 * B * N * P * s) / 1e9
    # streaming-read ceiling probes
    t = timeit(lambda: A.view(-1).sum(), 3)
    print(json.dumps({"probe": "torch.sum(A)", "GBps": B * N * N * 8 / 1e9 / t, "ms": t * 1e3}), flush=True)
    for P in (6, 1, 2, 4, 8):
        X = torch.randn(B, P, N, dtype=torch.float64, device=dev)
        Y = torch.empty(B, P, N, dtype=torch.float64, device=dev)
        for R in ((4, 8, 16) if P <= 4 else (4, 8)):
            for stg in (0, 1):
                t = timeit(lambda: dense_mm(A, X, out=Y, rows_hint=R, stagger=stg), 5)
                r = {"k": "mm_rows", "dtype": "f64", "B": B, "N": N, "P": P, "R": R, "stagger": stg, "ms": t * 1e3, "GBps": gb(P) / t}
                res.append(r); print(json.dumps(r), flush=True)
        if P in (1, 6):
            t = timeit(lambda: dense_mm(A, X, out=Y, trans=True), 5)
            r = {"k": "rmm_cols", "dtype": "f64", "B": B, "N": N, "P": P, "ms": t * 1e3, "GBps": gb(P) / t}
            res.append(r); print(json.dumps(r), flush=True)
            t = timeit(lambda: torch.matmul(A, X.transpose(-2, -1)), 3)
            r = {"k": "torch.matmul(rocBLAS)", "dtype": "f64", "B": B, "N": N, "P": P, "ms": t * 1e3, "GBps": gb(P) / t}
            res.append(r); print(json.dumps(r), flush=True)
    del A
    torch.cuda.empty_cache()
    # fp32 at the C5 per-GPU shape (16 x 32768^2 f32 = 68.7 GB) when --full, else 4
    B = 16 if full else 4
    N = 32768
    A = torch.empty(B, N, N, dtype=torch.float32, device=dev).uniform_(-1, 1)
    for P in (6, 8):
        X = torch.randn(B, P, N, dtype=torch.float32, device=dev)
        Y = torch.empty(B, P, N, dtype=torch.float32, device=dev)
        for R in (4, 8):
            t = timeit(lambda: dense_mm(A, X, out=Y, rows_hint=R, stagger=1), 5)
            r = {"k": "mm_rows", "dtype": "f32", "B": B, "N": N, "P": P, "R": R, "stagger": 1, "ms": t * 1e3,
                 "GBps": (B * N * N * 4 + 2 * B * N * P * 4) / 1e9 / t}
            res.append(r); print(json.dumps(r), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/k1_sweep.json", "w"), indent=1)


if __name__ == "__main__":
    main()
