"""cycle stamps of the phases of the K3t kernel (block 0)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xitorch_amd import kernels as K, synthetic
dev = torch.device("cuda:0")
buf = torch.zeros(8, dtype=torch.int64, device=dev)
for nthr in (256, 512, 1024):
    for k in (24, 54, 108):
        D = synthetic.spectrum("S1", 4096, device=dev)
        Q, _ = torch.linalg.qr(torch.randn(4, 4096, k, dtype=torch.float64, device=dev))
        T = Q.transpose(-2, -1) @ (D[None, :, None] * Q)
        for _ in range(3):
            K.small_eigh(T, k, 6, method="tri", threads=nthr, profile=buf)
        torch.cuda.synchronize()
        st = buf.cpu().tolist()
        d = [st[i + 1] - st[i] for i in range(5)]
        print(json.dumps({"threads": nthr, "k": k, "cycles": dict(zip(["tridiag", "bisect", "inviter", "check", "backtr"], d)), "total": st[5] - st[0]}))
