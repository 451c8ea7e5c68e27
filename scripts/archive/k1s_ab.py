"""K1s A/B alone on the GPU, one process, any number of separately built kernels behind the same C entry points.
   python scripts/k1s_ab.py [B N P] [f32] name=path.so[:L] ...
Each `path.so` exports xk_dense_symm_{f64,f32} + xk_dense_symm_workspace_elems (the library itself, the round-2 kernel
scripts/_ab/libsymm_r02.so, or builds of csrc/xk_symm.hip with other -D knobs, see scripts/k1s_build_ab.sh); `:L` sets
the slabs-per-run knob through xk_dense_symm_tune when the library has it.  One JSON line per variant: median of
interleaved repetitions, error against torch, bit-reproducibility of two launches."""
import os, sys, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xitorch_amd._capi import ptr, stream_ptr
dev = torch.device("cuda:0")
nums = [int(v) for v in sys.argv[1:] if v.isdigit()]
B, N, P = nums[:3] if len(nums) >= 3 else (32, 16384, 6)
dtype = torch.float32 if "f32" in sys.argv else torch.float64
sfx = "f32" if dtype == torch.float32 else "f64"
specs = [v for v in sys.argv[1:] if "=" in v]
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
Pv, I, Lg = ctypes.c_void_p, ctypes.c_int, ctypes.c_long


class Variant:
    def __init__(self, spec):
        self.name, path = spec.split("=", 1)
        self.L = None
        if ":" in path:
            path, L = path.rsplit(":", 1)
            self.L = int(L)
        self.lib = ctypes.CDLL(os.path.abspath(path))
        self.f = getattr(self.lib, "xk_dense_symm_" + sfx)
        self.f.restype = I
        # the product library takes `opts` (r04: no process-wide tuning state); the archived kernels under micro/ keep
        # the r03 signature and their own xk_dense_symm_tune
        self.has_opts = not hasattr(self.lib, "xk_dense_symm_tune")
        self.f.argtypes = [Pv, Pv, Pv, Pv, Lg, I, I, I, Lg, Lg, Lg, Lg, Lg, Lg] + ([I] if self.has_opts else []) + [Pv]
        self.lib.xk_dense_symm_workspace_elems.restype = Lg
        self.lib.xk_dense_symm_workspace_elems.argtypes = [I, I, I, I]
        self.nws = self.lib.xk_dense_symm_workspace_elems(B, N, P, es)
        self.ws = torch.empty(self.nws, dtype=dtype, device=dev)

    def run(self, Y):
        if self.L is not None and not self.has_opts:
            self.lib.xk_dense_symm_tune(1, self.L)
        extra = ((self.L or 0) << 8,) if self.has_opts else ()
        rc = self.f(ptr(A), ptr(X), ptr(Y), ptr(self.ws), self.nws, B, N, P, A.stride(1), A.stride(0), X.stride(1),
                    X.stride(0), Y.stride(1), Y.stride(0), *extra, stream_ptr())
        assert rc == 0, (self.name, rc)


variants = [Variant(s) for s in specs]
Y = torch.empty_like(X)
times = {v.name: [] for v in variants}
errs, repro = {}, {}
for v in variants:
    v.run(Y); torch.cuda.synchronize()
    errs[v.name] = ((Y[:2].double() - ref).abs().max() / ref.abs().max()).item()
    Y1 = Y.clone()
    v.run(Y); torch.cuda.synchronize()
    repro[v.name] = bool(torch.equal(Y, Y1))
for rep in range(5):
    for v in variants:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            v.run(Y)
        e1.record(); torch.cuda.synchronize()
        times[v.name].append(e0.elapsed_time(e1) / 3)
for v in variants:
    ts = sorted(times[v.name])
    ms = ts[len(ts) // 2]
    print(json.dumps({"variant": v.name, "B": B, "N": N, "P": P, "dtype": str(dtype), "ms_median": round(ms, 4),
                      "ms_all": [round(t, 3) for t in times[v.name]], "TBps_triangle": round(tri_bytes / ms / 1e9, 3),
                      "frac": round(tri_bytes / ms / 1e9 / 8.0, 4), "relerr": errs[v.name],
                      "bit_reproducible": repro[v.name]}), flush=True)
