"""Which XCDs / compute units a CU-masked stream leaves to its kernels (xk_probe_xcc), by mask pattern and reserve.
One JSON line per (pattern, reserve): workgroups per XCD and compute units seen per XCD."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xitorch_amd import kernels as K
from xitorch_amd._capi import fn, ptr, check
dev = torch.device("cuda:0")
hist = torch.zeros(16, dtype=torch.int32, device=dev)
units = torch.zeros(128, dtype=torch.int32, device=dev)
for pattern in (0, 1):
    K.CU_MASK_PATTERN = pattern
    for reserve in (0, 16, 32, 64):
        st = K.masked_stream(dev, reserve, slot=200 + reserve)
        with torch.cuda.stream(st):
            rc = fn("xk_probe_xcc")(ptr(hist), ptr(units), 16384, 200000, K.stream_ptr())
            check(rc, "xk_probe_xcc")
        st.synchronize()
        h = hist.cpu().tolist()
        u = units.cpu().reshape(16, 8)
        ncu = [int(sum(bin(int(w) & 0xffffffff).count("1") for w in u[x])) for x in range(16)]
        print(json.dumps({"pattern": pattern, "reserve_cus": reserve, "workgroups_per_xcd": h[:8], "units_seen_per_xcd": ncu[:8]}))
