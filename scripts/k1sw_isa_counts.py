"""Instruction-level evidence for K1sw: compile xk_symmwide.hip with -save-temps (cross-compiles, no GPU)
and count, per kernel, what the band loop contains: MFMAs, DS operations by kind, vector-memory loads, and the waits —
how many `s_waitcnt lgkmcnt(0)` / `vmcnt(0)` there are per MFMA and which vmcnt values the load ring waits on.
    python scripts/k1sw_isa_counts.py [out.json]"""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "xitorch_amd", "csrc")
tmp = tempfile.mkdtemp()
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", SRC, "-c",
                       os.path.join(SRC, "xk_symmwide.hip"), "-save-temps", "-o", os.path.join(tmp, "x.o")], cwd=tmp)
asm = open(os.path.join(tmp, "xk_symmwide-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
kernels = {"r05 cooperative form (dense_symm_wide7_kernel<1,false>, opts 3)": "_ZN2xk23dense_symm_wide7_kernelILi1ELb0EE",
           "r06 form (dense_symm_wide8_kernel, opts 9)": "_ZN2xk23dense_symm_wide8_kernelE"}
out = {"source": "hipcc -O3 --offload-arch=gfx950 -save-temps xk_symmwide.hip (ROCm 7.2); counts over the whole kernel body; "
                 "per band and wave both forms issue 256 MFMAs (64 rows x 128 columns x 16 panel columns x 2 products)"}
for label, sym in kernels.items():
    start = next(i for i, l in enumerate(asm) if l.startswith(sym) and l.rstrip().endswith(sym + l[len(sym):].rstrip()) and ":" in l)
    end = next(i for i in range(start, len(asm)) if "s_endpgm" in asm[i])
    body = [l.strip() for l in asm[start:end]]
    c = collections.Counter()
    vm = collections.Counter()
    for l in body:
        op = l.split()[0] if l and not l.startswith((";", ".")) else None
        if not op:
            continue
        if op.startswith("v_mfma"):
            c["v_mfma_f32_16x16x4_f32"] += 1
        elif op.startswith("ds_"):
            c[op] += 1
        elif op.startswith(("buffer_load", "global_load")):
            c[op] += 1
        elif op.startswith(("buffer_store", "global_store")):
            c[op] += 1
        elif op == "s_barrier":
            c[op] += 1
        elif op == "s_cbranch_vccnz" or op.startswith("s_cbranch"):
            c["s_cbranch_*"] += 1
        elif op == "s_nop":
            c["s_nop"] += 1
        elif op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", l)
            if m:
                c["s_waitcnt lgkmcnt(%s)" % ("0" if m.group(1) == "0" else "n>0")] += 1
            m = re.search(r"vmcnt\((\d+)\)", l)
            if m:
                vm[int(m.group(1))] += 1
    meta = {}
    for l in asm:
        pass
    mf = c["v_mfma_f32_16x16x4_f32"]
    ds = sum(v for k, v in c.items() if k.startswith("ds_"))
    out[label] = {"counts": dict(sorted(c.items())), "vmcnt_waits_by_value": {str(k): v for k, v in sorted(vm.items())},
                  "ds_operations_per_mfma": round(ds / mf, 3),
                  "lgkmcnt0_waits_per_mfma": round(c["s_waitcnt lgkmcnt(0)"] / mf, 3),
                  "vmcnt0_waits": vm.get(0, 0)}
# register / LDS footprint from the metadata
txt = "\n".join(asm)
for label, sym in kernels.items():
    m = re.search(r"\.name:\s+%s\S*\n(?:.*\n){0,12}?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)" % re.escape(sym), txt)
    if m:
        out[label]["vgpr_count"], out[label]["vgpr_spill_count"] = int(m.group(1)), int(m.group(2))
dst = sys.argv[1] if len(sys.argv) > 1 else None
js = json.dumps(out, indent=1)
print(js)
if dst:
    open(dst, "w").write(js + "\n")
