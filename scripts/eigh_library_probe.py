"""What the library eigensolver (torch.linalg.eigh -> rocSOLVER) costs on the Rayleigh-Ritz matrices of an
un-restarted Davidson run beyond the native kernels' order 128:  python scripts/eigh_library_probe.py"""
import json, time, torch
dev = torch.device("cuda:0")
for B in (32, 4):
    for k in (128, 192, 256, 384, 512):
        g = torch.Generator(device="cpu").manual_seed(k)
        R = torch.randn(B, k, k, dtype=torch.float64, generator=g).to(dev)
        T = R + R.transpose(-2, -1)
        torch.linalg.eigh(T); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); torch.linalg.eigh(T); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(json.dumps({"B": B, "k": k, "eigh_ms": round(sorted(ts)[1] * 1e3, 3)}), flush=True)
