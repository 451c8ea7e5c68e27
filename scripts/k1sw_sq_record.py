"""Turn the counter listings of scripts/sessions/r06b.sh (scripts/pmc_parse.py output per pass) into one JSON record with the
derived ratios the verdict asks for: matrix-pipe busy, share of wave time waiting for issue / for counters, LDS work.
    python scripts/k1sw_sq_record.py gpurun_out/r06b profiles/r06_k1sw_sq_pmc.json"""
import json, re, sys
src, dst = sys.argv[1], sys.argv[2]
out = {"what": "SQ / TCC counters of the K1sw tile kernel, 8 x 32768^2 fp32, P = 16, standalone (scripts/k1sw_pmc_run.py <opts>), "
               "separate rocprofv3 --pmc passes with --kernel-trace only (scripts/sessions/r06b.sh); averages over 10 launches",
       "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES "
                "cycles summed over SIMDs; GRBM_GUI_ACTIVE cycles summed over the 8 XCDs; FETCH_SIZE / WRITE_SIZE in KB "
                "(FETCH x2 on gfx950: MI355X_MICROARCH.md, HBM section)"}
for form, label in ((3, "r05_form_opts3"), (9, "r06_form_opts9")):
    c, durs = {}, []
    for line in open("%s/sq_counters_form%d.txt" % (src, form)):
        m = re.match(r"(\w+)\s+n=(\d+) avg=([0-9.e+-]+)", line.strip())
        if not m:
            continue
        if m.group(1) == "_dur_us":
            durs.append(float(m.group(3)))
        elif not m.group(1).startswith("_"):
            c[m.group(1)] = float(m.group(3))
    d = {"counters": c, "kernel_ms_under_profiler_by_pass": [round(x / 1e3, 3) for x in durs]}
    gui = c.get("GRBM_GUI_ACTIVE")
    if gui:
        d["mfma_busy_fraction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8.0 * 256 * 4)
        d["shader_clock_GHz"] = gui / 8.0 / (durs[0] * 1e3)
    if "SQ_WAVE_CYCLES" in c:
        d["wait_inst_any_over_wave_cycles"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
        d["wait_any_over_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        d["active_inst_any_over_wave_cycles"] = c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_LDS" in c:
        d["lds_instructions_per_mfma"] = c["SQ_INSTS_LDS"] / c["SQ_INSTS_MFMA"]
        d["lds_bank_conflict_over_idx_active"] = c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0)
        d["vmem_reads_per_mfma"] = c["SQ_INSTS_VMEM_RD"] / c["SQ_INSTS_MFMA"]
    if "FETCH_SIZE" in c:
        d["hbm_bytes_per_launch"] = c["FETCH_SIZE"] * 2048.0 + c["WRITE_SIZE"] * 1024.0
        d["traffic_over_algorithmic_bytes"] = d["hbm_bytes_per_launch"] / 17213947904.0
    out[label] = d
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in out.items() if isinstance(v, dict)}, indent=1))
