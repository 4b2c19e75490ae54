"""Build the profiles/traffic_layer_fwd.json entry of one workload from the PMC summary that tools/pmc_traffic.sh leaves
(gpurun_out/r3_traffic_<config>_<batch>/summary.json): HBM bytes of the last step's rg_layer_fwd launches (and of its dense launches).
    python tools/traffic_entry.py <summary.json> <config> <batch> <rg_version> <n_layer> <committed summary path> [steps_layer steps_dense]
steps_*: how many eval steps the profiled command ran with the layer kernels / with each dense kernel (default 3 = --steps 2 --warmup 1;
a bench run that also times the exact-fp32 dense step afterwards runs the layer kernels for 3 more steps).
Corrections as MI355X_MICROARCH.md (HBM) prescribes: on gfx950 FETCH_SIZE counts the 128-B requests of 16-B-per-lane reads at 64 B, so
read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact; the counters' KB are 1024 B."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYER_KERNELS = ("layer_fwd_wp_kernel", "layer_fwd_kernel", "combine_kernel")
DENSE_KERNELS = ("dense_split3_kernel", "dense_split_kernel", "dense_kernel", "dense128")


def last_step(vals, per_step):
    return vals[-per_step:] if per_step and len(vals) >= per_step else vals


def main():
    summary, config, batch, version, n_layer, committed = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    with open(summary) as f:
        s = json.load(f)
    steps_layer = int(sys.argv[7]) if len(sys.argv) > 7 else 3          # pmc_traffic.sh: --steps 2 --warmup 1
    steps_dense = int(sys.argv[8]) if len(sys.argv) > 8 else 3
    per_kernel, step_bytes, dense = {}, 0.0, {}
    for name, c in s.items():
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        short = name.split("<")[0].split("(")[0].strip()
        is_dense = any(k in short for k in DENSE_KERNELS)
        n_launch = len(c["FETCH_SIZE"]) // (steps_dense if is_dense else steps_layer)
        rec = dict(FETCH_SIZE_KB=last_step(c["FETCH_SIZE"], n_launch), WRITE_SIZE_KB=last_step(c["WRITE_SIZE"], n_launch),
                   TCC_HIT_sum=last_step(c.get("TCC_HIT_sum", []), n_launch), TCC_MISS_sum=last_step(c.get("TCC_MISS_sum", []), n_launch),
                   dur_ns=last_step(c.get("dur_ns_FETCH_SIZE", []), n_launch))
        nbytes = sum(2 * 1024.0 * x for x in rec["FETCH_SIZE_KB"]) + sum(1024.0 * x for x in rec["WRITE_SIZE_KB"])
        if any(k in short for k in LAYER_KERNELS) and "bwd" not in short:
            per_kernel[short] = rec
            step_bytes += nbytes
        elif any(k in short for k in DENSE_KERNELS):
            dense[short] = dict(rec, hbm_bytes_per_launch=nbytes / max(1, len(rec["FETCH_SIZE_KB"])))
    entry = dict(config=config, batch=batch, rg_version=version,
                 kernel="rg_layer_fwd launches of one eval step (word-parallel kernel / per-query walk, + combine_kernel each)",
                 hbm_bytes_per_launch=step_bytes / n_layer, hbm_bytes_per_step=step_bytes, per_kernel=per_kernel, dense=dense,
                 correction="gfx950: FETCH_SIZE counts 128-B requests at 64 B for 16-B-per-lane reads -> read bytes = 2 x FETCH_SIZE "
                            "(MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; KB = 1024 B",
                 source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum, separate passes (tools/pmc_traffic.sh %s %d): %s"
                        % (config, batch, committed))
    path = os.path.join(ROOT, "profiles", "traffic_layer_fwd.json")
    with open(path) as f:
        doc = json.load(f)
    doc["entries"] = [e for e in doc["entries"] if not (e["config"] == config and e["batch"] == batch and e["rg_version"] == version)] + [entry]
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({k: entry[k] for k in ("config", "batch", "rg_version", "hbm_bytes_per_launch", "hbm_bytes_per_step")}))
    for k, v in dense.items():
        print(k, "HBM bytes per launch", v["hbm_bytes_per_launch"])


main()
