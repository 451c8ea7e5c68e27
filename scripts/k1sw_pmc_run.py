"""PMC driver: ten launches of the K1sw TILE kernel of one form (argv[1] = opts: 3 = r05 cooperative, 9 = r06) on
8 x 32768^2 fp32, P = 16, nothing else — the process rocprofv3 --pmc passes are taken over (scripts/sessions/r06b.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xitorch_amd import kernels as K, synthetic
opts = int(sys.argv[1]) if len(sys.argv) > 1 else 9
B, N, P = 8, 32768, 16
dev = torch.device("cuda:0")
A = torch.empty(B, N, N, dtype=torch.float32, device=dev)
synthetic.dense_symmetric(B, N, "S1:16", dtype=torch.float32, device=dev, out=A)
X = torch.randn(B, P, N, dtype=torch.float32, device=dev)
nws = K.fn("xk_dense_symm_wide_workspace_elems")(B, N)
ws = torch.empty(nws, dtype=torch.float32, device=dev)
f = K.fn("xk_dense_symm_wide_tiles_f32")
for _ in range(10):
    rc = f(A.data_ptr(), X.data_ptr(), ws.data_ptr(), nws, B, N, P, N, N * N, N, P * N, opts,
           torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
torch.cuda.synchronize()
