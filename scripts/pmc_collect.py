"""HBM traffic and matrix-core occupancy of ONE kernel from separate rocprofv3 PMC passes (kernel-trace only; FETCH_SIZE and
WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md) -> a JSON record stamped with the hash of the kernel's source.
Corrections as the guide's HBM section prescribes: FETCH_SIZE x2 on gfx950 for 16 B/lane streaming reads, KB -> bytes
x1024, WRITE_SIZE as reported.
    python scripts/pmc_collect.py <kernel-name substring> <algorithmic bytes per launch> <source files, comma separated>
                                  <out.json> FETCH=<csv> WRITE=<csv> [MFMA=<csv>] [B=<operators per launch>] [note=...]"""
import csv, collections, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash(names):
    h = hashlib.sha256()
    for name in names:
        h.update(open(os.path.join(ROOT, "xitorch_amd", "csrc", name), "rb").read())
    return h.hexdigest()


def per_kernel(path, pat):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg["_dur_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


if __name__ == "__main__":
    pat, abytes, srcs, out = sys.argv[1], float(sys.argv[2]), sys.argv[3].split(","), sys.argv[4]
    kv = dict(a.split("=", 1) for a in sys.argv[5:])
    rec = {"kernel": pat, "B": int(kv["B"]) if "B" in kv else None, "algorithmic_bytes_per_launch": abytes,
           "kernel_source_files": srcs,
           "kernel_source_sha256": source_hash(srcs), "note": kv.get("note", "")}
    f = per_kernel(kv["FETCH"], pat)
    w = per_kernel(kv["WRITE"], pat)
    fetch, write = f["FETCH_SIZE"][0], w["WRITE_SIZE"][0]
    hbm = fetch * 2.0 * 1024.0 + write * 1024.0
    rec.update(launches_profiled=f["FETCH_SIZE"][1], FETCH_SIZE_KB_raw=fetch, WRITE_SIZE_KB_raw=write,
               correction="FETCH_SIZE x2 on gfx950 for 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM section); KB -> "
                          "bytes x1024; WRITE_SIZE as reported",
               hbm_bytes_per_launch=hbm, traffic_over_algorithmic_bytes=hbm / abytes,
               kernel_duration_ms_profiled={"fetch_pass": f["_dur_ms"][0], "write_pass": w["_dur_ms"][0]})
    if "MFMA" in kv:
        m = per_kernel(kv["MFMA"], pat)
        busy, gui = m["SQ_VALU_MFMA_BUSY_CYCLES"][0], m["GRBM_GUI_ACTIVE"][0]
        rec.update(SQ_VALU_MFMA_BUSY_CYCLES=busy, GRBM_GUI_ACTIVE=gui, SQ_BUSY_CYCLES=m.get("SQ_BUSY_CYCLES", (None, 0))[0],
                   mfma_busy_fraction=busy / (gui / 8.0 * 256 * 4),
                   shader_clock_GHz_during_kernel=gui / 8.0 / (m["_dur_ms"][0] * 1e6),
                   mfma_note="SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs), as in "
                             "r01_k1w_mfma_pmc.json / r03_c5w_mfma_pmc.json")
    json.dump(rec, open(out, "w"), indent=1)
    print("wrote", out, "traffic / algorithmic bytes = %.4f" % (hbm / abytes), "mfma busy", rec.get("mfma_busy_fraction"))
