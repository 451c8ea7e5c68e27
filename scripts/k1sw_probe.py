"""Which half bounds K1sw (cooperative form)?  Two probe builds of xk_symmwide.hip — XK_SW_PROBE=1: the matrix traffic and
the LDS turn without the MFMA block; XK_SW_PROBE=2: the whole instruction stream without the matrix loads — timed beside
the product kernel on 8 x 32768^2 fp32, P = 16.  `python scripts/k1sw_probe.py build` (here, cross-compiled) then
`python scripts/k1sw_probe.py` on the GPU.  Probe results are NOT products; nothing in the package loads these libraries."""
import os, sys, json, subprocess, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "scripts", "_probe")
SRC = os.path.join(ROOT, "xitorch_amd", "csrc")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    for mode in (1, 2):
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", SRC,
               "-DXK_SW_PROBE=%d" % mode, os.path.join(SRC, "xk_symmwide.hip"), "-o", os.path.join(OUT, "k1sw_probe%d.so" % mode)]
        subprocess.check_call(cmd)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from xitorch_amd import kernels as K, synthetic
dev = torch.device("cuda:0")
B, N, P = 8, 32768, 16
A = torch.empty(B, N, N, dtype=torch.float32, device=dev)
synthetic.dense_symmetric(B, N, "S1:16", dtype=torch.float32, device=dev, out=A)
X = torch.randn(B, P, N, dtype=torch.float32, device=dev)
Y = torch.empty_like(X)
nws = K.fn("xk_dense_symm_wide_workspace_elems")(B, N)
ws = torch.empty(nws, dtype=torch.float32, device=dev)


def t_of(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


rec = {"B": B, "N": N, "P": P}
Lg, I, Pp = ctypes.c_long, ctypes.c_int, ctypes.c_void_p
for name, path in (("product", None), ("no_mfma", "k1sw_probe1.so"), ("no_matrix_loads", "k1sw_probe2.so")):
    if path is None:
        f = K.fn("xk_dense_symm_wide_tiles_f32")
    else:
        lib = ctypes.CDLL(os.path.join(OUT, path))
        f = lib.xk_dense_symm_wide_tiles_f32
        f.restype, f.argtypes = I, [Pp, Pp, Pp, Lg, I, I, I, Lg, Lg, Lg, Lg, I, Pp]
    for opts in (1, 3):
        def run():
            rc = f(A.data_ptr(), X.data_ptr(), ws.data_ptr(), nws, B, N, P, N, N * N, N, P * N, opts,
                   torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        rec["%s_opts%d_ms" % (name, opts)] = t_of(run)
print(json.dumps(rec))
