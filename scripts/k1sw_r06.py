"""K1sw round 6: the r05 cooperative form (opts 3) against the r06 form (opts 9: column part from the load registers, ring
of four blocks, two waves per SIMD) and the r06 form's two probe builds (XK_SW8_PROBE=1: no MFMA = traffic + LDS turn +
partials; 2: no matrix loads = MFMA + LDS), tile kernel only, on B x N^2 fp32, P = 16 (default: the configs[4] pipeline
group, 8 x 32768^2).  `python scripts/k1sw_r06.py build` (cross-compiles the probes into scripts/_probe/), then
`python scripts/k1sw_r06.py [B] [N]` on the GPU.  One JSON line.  Probe libraries are never loaded by the package."""
import os, sys, json, subprocess, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "scripts", "_probe")
SRC = os.path.join(ROOT, "xitorch_amd", "csrc")
# trial builds of the r06 form (tile kernel only; never loaded by the package)
VARIANTS = [("pitch272", ["-DXK_SW8_SWZ=0"]), ("prio", ["-DXK_SW8_PRIO=1"]), ("tr512", ["-DXK_SW_TR_BIG=512"]),
            ("tr2048", ["-DXK_SW_TR_BIG=2048"])]
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    procs = []
    builds = [("k1sw8_probe1.so", ["-DXK_SW8_PROBE=1"]), ("k1sw8_probe2.so", ["-DXK_SW8_PROBE=2"])]
    builds += [("k1sw8_var_%s.so" % name, flags) for name, flags in VARIANTS]
    for lib, flags in builds:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", SRC] + flags + \
              [os.path.join(SRC, "xk_symmwide.hip"), "-o", os.path.join(OUT, lib)]
        procs.append(subprocess.Popen(cmd))
    sys.exit(max(p.wait() for p in procs))
sys.path.insert(0, ROOT)
import torch
from xitorch_amd import kernels as K, synthetic
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
P = 16
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


tri = B * N * (N + 1) // 2 * 4 + 2 * B * N * P * 4
rec = {"B": B, "N": N, "P": P, "algorithmic_bytes": tri}
Lg, I, Pp = ctypes.c_long, ctypes.c_int, ctypes.c_void_p
outs = {}
for form in (3, 9):
    K.K1SW_OPTS = form
    rec["product_opts%d_with_fold_ms" % form] = t_of(lambda: K.dense_symm_wide(A, X, out=Y))
    outs[form] = Y.clone()
ref = torch.matmul(X[:1].double(), A[:1].double())
rec["rel_err_opts9_vs_fp64"] = ((outs[9][:1].double() - ref).abs().max() / ref.abs().max()).item()
rec["rel_err_opts3_vs_fp64"] = ((outs[3][:1].double() - ref).abs().max() / ref.abs().max()).item()
rec["rel_diff_opts9_vs_opts3"] = ((outs[9] - outs[3]).abs().max() / outs[3].abs().max()).item()
plan = [("tiles", None, (3, 9)), ("no_mfma", "k1sw8_probe1.so", (9,)), ("no_matrix_loads", "k1sw8_probe2.so", (9,))]
plan += [("variant_" + name, "k1sw8_var_%s.so" % name, (9,)) for name, _ in VARIANTS]
for name, path, forms in plan:
    if path is None:
        f = K.fn("xk_dense_symm_wide_tiles_f32")
    else:
        if not os.path.exists(os.path.join(OUT, path)):
            continue
        lib = ctypes.CDLL(os.path.join(OUT, path))
        f = lib.xk_dense_symm_wide_tiles_f32
        f.restype, f.argtypes = I, [Pp, Pp, Pp, Lg, I, I, I, Lg, Lg, Lg, Lg, I, Pp]
    for opts in forms:
        def run():
            rc = f(A.data_ptr(), X.data_ptr(), ws.data_ptr(), nws, B, N, P, N, N * N, N, P * N, opts,
                   torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        t = t_of(run)
        rec["%s_opts%d_ms" % (name, opts)] = t
        if name == "tiles":
            rec["tiles_opts%d_TBps" % opts] = tri / t / 1e9
            rec["tiles_opts%d_frac_hbm" % opts] = tri / t / 1e9 / 8.0
            rec["tiles_opts%d_TFLOPs" % opts] = 2.0 * B * N * N * P / t / 1e9
print(json.dumps(rec))
