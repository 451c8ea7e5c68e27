"""iteration counts of the Davidson goldens by the order from which K3p replaces K3t (native_eig.K3P_MIN_K): the
mixed-convergence cases are chaotic in the rounding, the others do not move.  GPU."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import cases
import xitorch_amd as xa
from xitorch_amd.linalg import native_eig
from xitorch_amd.linalg.native_eig import davidson
dev = torch.device("cuda:0")
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for mk in (64, 72, 80, 96, 112, 129):
    native_eig.K3P_MIN_K = mk
    out = {}
    for case in cases.DAVIDSON_CASES:
        gold = np.load(os.path.join(GOLD, "davidson_%s.npz" % case["name"]))
        mat = cases.davidson_matrix(case); Mmat = cases.davidson_M(case)
        A = xa.LinearOperator.m(mat.to(dev), is_hermitian=True)
        Mop = xa.LinearOperator.m(Mmat.to(dev), is_hermitian=True) if Mmat is not None else None
        tr = {}
        ev, X = davidson(A, case["neig"], case["mode"], Mop, min_eps=case["min_eps"], v_init="randn", trace=tr)
        err = float(np.abs(ev.cpu().numpy() - gold["evals"]).max())
        out[case["name"]] = (tr["niter"], int(gold["niter"]), "%.1e" % err)
    print(mk, json.dumps(out))
