"""cycle stamps of the phases of the K3t kernel (block 0)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xitorch_amd import kernels as K, synthetic
from xitorch_amd._capi import fn, ptr
dev = torch.device("cuda:0")
buf = torch.zeros(8, dtype=torch.int64, device=dev)
fn("xk_small_eigh_tri_set_profile")(ptr(buf))
for nthr in (256, 512, 1024):
    fn("xk_small_eigh_tri_set_threads")(nthr)
    for k in (24, 54, 108):
        D = synthetic.spectrum("S1", 4096, device=dev)
        Q, _ = torch.linalg.qr(torch.randn(4, 4096, k, dtype=torch.float64, device=dev))
        T = Q.transpose(-2, -1) @ (D[None, :, None] * Q)
        for _ in range(3):
            K.small_eigh(T, k, 6, method="tri")
        torch.cuda.synchronize()
        st = buf.cpu().tolist()
        d = [st[i + 1] - st[i] for i in range(5)]
        print(json.dumps({"threads": nthr, "k": k, "cycles": dict(zip(["tridiag", "bisect", "inviter", "check", "backtr"], d)), "total": st[5] - st[0]}))
