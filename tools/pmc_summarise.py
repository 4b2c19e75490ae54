"""Per-kernel summary of rocprofv3 --pmc passes (tools/pmc.sh): for every kernel name, the counter values and durations of its
LAST `keep` dispatches in dispatch order (one eval step's launches), written to <dir>/summary.json."""
import csv, glob, json, os, re, sys

d = sys.argv[1]
keep = int(sys.argv[2]) if len(sys.argv) > 2 else 12
out = {}
for path in glob.glob(os.path.join(d, "*", "**", "*counter_collection.csv"), recursive=True):
    rows = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            name = re.sub(r"\(.*", "", name).strip()
            key = (name, int(r["Dispatch_Id"]))
            e = rows.setdefault(key, {"dur_ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "grid": int(r["Grid_Size"]),
                                      "vgpr": int(r["VGPR_Count"]), "lds": int(r["LDS_Block_Size"]), "scratch": int(r["Scratch_Size"])})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    by_kernel = {}
    for (name, disp), e in sorted(rows.items(), key=lambda kv: kv[0][1]):
        by_kernel.setdefault(name, []).append(e)
    for name, lst in by_kernel.items():
        k = out.setdefault(name, {})
        lst = lst[-keep:]
        for c in lst[0]:
            k.setdefault(c if c not in ("dur_ns",) else "dur_ns_" + os.path.basename(os.path.dirname(path))[:24], [e.get(c) for e in lst])
with open(os.path.join(d, "summary.json"), "w") as f:
    json.dump(out, f, indent=1)
for name, k in out.items():
    if any(x in name for x in ("layer_fwd", "dense", "combine", "hop_or", "tlayer")):
        print(name[:110])
        for c, v in k.items():
            print("   %-40s %s" % (c, v[-6:]))
