"""Turn the two rocprofv3 PMC passes of scripts/pmc_k1s.py (FETCH_SIZE, WRITE_SIZE: separate passes, --kernel-trace only)
into profiles/k1s_pmc_traffic.json: HBM bytes per whole-batch K1s launch, corrected as MI355X_MICROARCH.md's HBM section
prescribes (FETCH_SIZE counts 64 B per 128 B request of a 16 B/lane streaming read on gfx950: x2; KB -> bytes x1024;
WRITE_SIZE as reported), plus the hash of the kernel source the numbers belong to (bench.py refuses a stale record).
    python scripts/pmc_k1s_collect.py FETCH_counter_collection.csv WRITE_counter_collection.csv [out.json]"""
import csv, collections, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    h = hashlib.sha256()
    for name in ("xk_symm.hip", "xk_common.h"):
        h.update(open(os.path.join(ROOT, "xitorch_amd", "csrc", name), "rb").read())
    return h.hexdigest()


def per_kernel(path, counter):
    agg, dur = collections.defaultdict(list), collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for key in ("dense_symm_tiles", "symm_fold"):
            if key in r["Kernel_Name"]:
                agg[key].append(float(r["Counter_Value"]))
                dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return ({k: sum(v) / len(v) for k, v in agg.items()}, {k: sum(v) / len(v) for k, v in dur.items()})


if __name__ == "__main__":
    fetch, dur = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "k1s_pmc_traffic.json")
    B, N, P, es = 64, 16384, 6, 8
    hbm = sum(fetch.values()) * 2.0 * 1024.0 + sum(write.values()) * 1024.0
    tri = B * (N * (N + 1) // 2) * es + 2 * B * N * P * es
    full = B * N * N * es + 2 * B * N * P * es
    rec = {"B": B, "N": N, "P": P, "dtype": "f64",
           "kernel": "xk::dense_symm_tiles<double,6> + xk::symm_fold<double> (one whole-batch K1s launch of 64 operators)",
           "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write,
           "correction": "FETCH_SIZE x2 on gfx950 for 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM section); "
                         "KB -> bytes x1024; WRITE_SIZE as reported",
           "hbm_bytes_per_launch": hbm, "algorithmic_bytes_full_matrix": full, "algorithmic_bytes_upper_triangle": tri,
           "traffic_over_full_matrix_bytes": hbm / full, "traffic_over_triangle_bytes": hbm / tri,
           "kernel_duration_ms_profiled": dur, "kernel_source_sha256": kernel_source_hash(),
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace --output-format csv "
                     "-- python scripts/pmc_k1s.py  (scripts/pmc_traffic.sh)"}
    json.dump(rec, open(out, "w"), indent=1)
    print("wrote", out, "traffic / triangle bytes = %.4f" % (hbm / tri))
