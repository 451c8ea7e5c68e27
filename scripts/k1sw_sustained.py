import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xitorch_amd import kernels as K, synthetic
dev = torch.device("cuda:0")
B, N, P = 8, 32768, 16
A = torch.empty(B, N, N, dtype=torch.float32, device=dev)
synthetic.dense_symmetric(B, N, "S1:16", dtype=torch.float32, device=dev, out=A)
X = torch.randn(B, P, N, dtype=torch.float32, device=dev)
nws = K.fn("xk_dense_symm_wide_workspace_elems")(B, N)
ws = torch.empty(nws, dtype=torch.float32, device=dev)
f = K.fn("xk_dense_symm_wide_tiles_f32")
out = {}
for reserve in (0, 32, 64):
    st = K.masked_stream(dev, reserve) if reserve else torch.cuda.current_stream()
    with torch.cuda.stream(st):
        def run():
            rc = f(A.data_ptr(), X.data_ptr(), ws.data_ptr(), nws, B, N, P, N, N * N, N, P * N, 9, st.cuda_stream)
            assert rc == 0
        for n in (10, 100, 300):
            for _ in range(3): run()
            st.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(n): run()
            e1.record(st); st.synchronize()
            out["reserve%d_n%d_ms" % (reserve, n)] = round(e0.elapsed_time(e1) / n, 4)
print(json.dumps(out))
