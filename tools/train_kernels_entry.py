"""profiles/train_kernels.json from a `rocprofv3 --kernel-trace --stats` run of `bench.py --train` (tools/r3_profiles.sh): per-kernel time
per training step for the kernels of the step, attached by bench.py --train to its `roofline.per_kernel` when config, batch and
rg_version match.    python tools/train_kernels_entry.py <kernel_stats.csv> <config> <batch> <rg_version> <steps incl. warm-up> <committed path>"""
import csv, json, os, re, sys

path, config, batch, version, steps, committed = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = list(csv.DictReader(open(path)))
total = sum(float(r["TotalDurationNs"]) for r in rows)
out, tensile = {}, 0.0
for r in rows:
    name = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")).strip()
    ms = float(r["TotalDurationNs"]) / steps / 1e6
    if name.startswith("Cijk_"):
        tensile += ms
        continue
    if any(k in name for k in ("layer_bwd", "drel", "bwd_combine", "dense_bwd", "dense_kernel", "gram_", "layer_fwd", "combine_kernel", "aq_sum")):
        short = name.split("<")[0]
        e = out.setdefault(short, dict(ms_per_step=0.0, launches_per_step=0.0))
        e["ms_per_step"] += ms
        e["launches_per_step"] += int(r["Calls"]) / steps
doc = dict(config=config, batch=batch, rg_version=version, kernels=out, tensile_gemm_ms_per_step=tensile, all_kernels_ms_per_step=total / steps / 1e6,
           source="rocprofv3 --kernel-trace --stats -- python3 bench.py --train --steps %d --warmup 2 --no-kernel-events: %s" % (steps - 2, committed))
with open(os.path.join(ROOT, "profiles", "train_kernels.json"), "w") as f:
    json.dump(doc, f, indent=1)
print(json.dumps(doc, indent=1))
