import csv, sys, collections
path, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if pat in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg["_dur_us"].append(dur / 1e3)
        agg["_vgpr"].append(float(r["VGPR_Count"])); agg["_lds"].append(float(r["LDS_Block_Size"]))
for k, v in sorted(agg.items()):
    print("%-28s n=%d avg=%.4g" % (k, len(v), sum(v) / len(v)))
